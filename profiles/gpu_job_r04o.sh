#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04o; mkdir -p $O; cd $R
echo "== f32fast phases" | tee $O/f32fast_phases.txt
for n in 1250000 10000000; do timeout -k 10 120 profiles/ubench/bin/f32fast_phases $n 2>&1; done | tee -a $O/f32fast_phases.txt
echo "== f32fast tests" | tee $O/f32fast.txt
timeout -k 10 300 python -m pytest tests/test_gpu_f32fast.py -m gpu -q 2>&1 | tail -3 | tee -a $O/f32fast.txt
timeout -k 10 400 python tests/soak_f32fast.py 150 2>&1 | grep -v amdgpu | tail -2 | tee -a $O/f32fast.txt
for i in 1 2; do timeout -k 10 200 python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench c5', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done | tee -a $O/f32fast.txt
for n in 1250000 4000000 10000000; do timeout -k 10 200 python profiles/f32fast_bench.py $n 2>&1 | grep "fp32 fast"; done | tee -a $O/f32fast.txt
echo "== sweep engine: 1 x 1024 against 2 x 512 lanes per CU" | tee $O/pt_teams.txt
bash profiles/pt_teams.sh $O/pt_teams.txt
echo "== ladder adaptation + speculation tests" | tee $O/pytest.txt
timeout -k 10 600 python -m pytest tests/test_tempering.py -m gpu -q -k "ladder or speculative" 2>&1 | tail -3 | tee -a $O/pytest.txt
RTMI_PT_TEAMS=2 timeout -k 10 600 python -m pytest tests/test_tempering.py tests/test_gpu_posterior.py -m gpu -q 2>&1 | tail -3 | tee -a $O/pytest.txt
