#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
echo "== f32fast phases (stamps in registers)" | tee $O/f32fast_phases.txt
for n in 1250000 4000000 10000000; do timeout -k 10 120 profiles/ubench/bin/f32fast_phases $n 2>&1; done | tee -a $O/f32fast_phases.txt
echo "== forward kernel alone at small sizes" | tee $O/small_kernel_time.txt
timeout -k 10 200 python profiles/small_kernel_time.py 2>&1 | grep -v amdgpu | tee -a $O/small_kernel_time.txt
