#!/bin/bash
# profiles/gpu_job_r04l.sh -- (1) what a small synchronous call costs piece by piece, (2) phases of the fp32 fast kernel,
# (3) the fp32 fast kernel with the convergence shortcut, tests + bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04l; mkdir -p $O; cd $R
echo "== launch latency" | tee $O/launch_latency.txt
timeout -k 10 120 profiles/ubench/bin/launch_latency 2>&1 | tee -a $O/launch_latency.txt
echo "== f32fast phases" | tee $O/f32fast_phases.txt
for v in 1e-6f 1e-4f; do for n in 1250000 10000000; do echo "F32F_CONV_REL=$v"; timeout -k 10 120 profiles/ubench/bin/f32fast_phases_$v $n 2>&1; done; done | tee -a $O/f32fast_phases.txt
echo "== tests" | tee $O/f32fast.txt
timeout -k 10 300 python -m pytest tests/test_gpu_f32fast.py -m gpu -q 2>&1 | tail -3 | tee -a $O/f32fast.txt
timeout -k 10 400 python tests/soak_f32fast.py 150 2>&1 | grep -v amdgpu | tail -3 | tee -a $O/f32fast.txt
for i in 1 2; do timeout -k 10 200 python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench c5', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done | tee -a $O/f32fast.txt
for n in 1250000 4000000 10000000; do timeout -k 10 200 python profiles/f32fast_bench.py $n 2>&1 | grep "fp32 fast"; done | tee -a $O/f32fast.txt
