# sub-steps per tile (Q) of the masks march: 2 against 4, fp64 by size; fp32 at 512^3
for n in 256 320 384 448 512; do
  for q in 2 4; do
    echo -n "fp64 ${n}^3 Q=$q: "; SMM_HIP_MASKS_MARCH_Q=$q SMM_HIP_PATTERN_CONST=0 SMM_HIP_MARCH_MIN_ROWS=0 python tools/spmv_sweep.py --matrix poisson3d --n $n --dtype f64 --configs 3:1 --reps 10 2>&1 | grep -E "family" | cut -c17-110
  done
done
for n in 384 512; do
  for q in 2 4; do
    echo -n "fp32 ${n}^3 Q=$q: "; SMM_HIP_MASKS_MARCH_Q=$q SMM_HIP_PATTERN_CONST=0 SMM_HIP_MARCH_MIN_ROWS=0 python tools/spmv_sweep.py --matrix poisson3d --n $n --dtype f32 --configs 3:1 --reps 10 2>&1 | grep -E "family" | cut -c17-110
  done
done
