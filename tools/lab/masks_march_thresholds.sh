# the masks march (values read) against the wave kernel it replaces, by size: where should the march start?
for spec in "f32 200" "f32 256" "f32 320" "f32 384" "f64 160" "f64 200" "f64 232"; do
  set -- $spec
  for mm in 1 0; do
    echo -n "$1 $2^3 march=$mm: "; SMM_HIP_MASKS_MARCH=$mm SMM_HIP_PATTERN_CONST=0 SMM_HIP_MARCH_MIN_ROWS=0 python tools/spmv_sweep.py --matrix poisson3d --n $2 --dtype $1 --configs 3:1 --reps 20 2>&1 | grep -E "family" | cut -c17-100
  done
done
