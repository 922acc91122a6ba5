#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04r; mkdir -p $O; cd $R
echo "== f32fast (tiles of 1280)" | tee $O/f32fast.txt
timeout -k 10 300 python -m pytest tests/test_gpu_f32fast.py -m gpu -q 2>&1 | tail -3 | tee -a $O/f32fast.txt
timeout -k 10 400 python tests/soak_f32fast.py 150 2>&1 | grep -v amdgpu | tail -2 | tee -a $O/f32fast.txt
for i in 1 2; do timeout -k 10 200 python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench c5', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done | tee -a $O/f32fast.txt
for n in 1250000 4000000 10000000; do timeout -k 10 200 python profiles/f32fast_bench.py $n 2>&1 | grep "fp32 fast"; done | tee -a $O/f32fast.txt
echo "== previous build (4ade32a) on this box" | tee -a $O/f32fast.txt
for n in 1250000 4000000 10000000; do timeout -k 10 200 python profiles/bench_with_lib.py profiles/ab_prev/librtmi_pre.so --workload c5 --no-cpu-baseline --nsrc $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4ade32a bench c5 nsrc', $n, d['value'], d['ms_per_step'])"; timeout -k 10 200 python bench.py --workload c5 --no-cpu-baseline --nsrc $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-tree bench c5 nsrc', $n, d['value'], d['ms_per_step'])"; done | tee -a $O/f32fast.txt
echo "== sweep engine: 1 x 1024 against 2 x 512 lanes per CU, steady state (--steps 300 --warmup 30)" | tee $O/pt_teams.txt
for rep in 1 2; do for t in 1 2; do for rows in 3 4; do
  RTMI_PT_TEAMS=$t RTMI_PT_SPECULATE=$rows timeout -k 10 300 python bench.py --workload pt --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('teams=$t rows=$rows ms_per_sweep %.4f' % d['ms_per_step'], d['config']['engine'][:80])"
done; done; done | tee -a $O/pt_teams.txt
echo "== default bench pt (engine trial before the warm-up)" | tee -a $O/pt_teams.txt
timeout -k 10 300 python bench.py --workload pt --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_pt.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default ms_per_sweep %.4f' % d['ms_per_step'], d['config']['engine'][:80])" | tee -a $O/pt_teams.txt
