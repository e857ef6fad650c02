#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w tools/lab/pageable_copy_probe.hip -o /tmp/probe || exit 1
{ for h in 0 1 2; do echo "== how $h (0 direct, 1 register around the call, 2 staged)"; /tmp/probe 1259712 0 $h; done; echo "== 10 M rows"; for h in 0 1 2; do echo "== how $h"; /tmp/probe 10000000 0 $h; done; } 2>&1 | grep -v amdgpu.ids | tee $OUT/pageable_copy_probe.txt
