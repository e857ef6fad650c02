for w in 0 3 4 6 8; do
  if [ $w = 0 ]; then unset SMM_HIP_STREAM_WGS_PER_CU; else export SMM_HIP_STREAM_WGS_PER_CU=$w; fi
  echo "wgs per CU: $w"; python tools/spmv_sweep.py --rows 1250000 --configs 3:2 --reps 40 2>&1 | grep family | cut -c1-120
done
