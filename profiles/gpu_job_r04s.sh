#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04s; mkdir -p $O; cd $R
echo "== sweep engine: workgroups per CU x rows per chain, steady state (--steps 300 --warmup 30)" | tee $O/pt_teams.txt
for rep in 1 2; do for cfg in "1 3" "2 4" "4 4" "0 0"; do set -- $cfg
  if [ "$1" = 0 ]; then label="defaults"; envs="A=1"; else label="teams=$1 rows=$2"; envs="RTMI_PT_TEAMS=$1 RTMI_PT_SPECULATE=$2"; fi
  env $envs timeout -k 10 300 python bench.py --workload pt --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label ms_per_sweep %.4f' % d['ms_per_step'], 'rows', d['config']['speculation_rows_per_chain'], d['config']['engine'][:80])"
done; done | tee -a $O/pt_teams.txt
echo "== tests under the new defaults" | tee $O/pytest.txt
timeout -k 10 900 python -m pytest tests/test_tempering.py tests/test_gpu_posterior.py tests/test_sampler_io.py -m gpu -q 2>&1 | tail -3 | tee -a $O/pytest.txt
echo "== config 3 at full size" | tee -a $O/pt_teams.txt
timeout -k 10 300 python profiles/config3_run.py > $O/config3_full_run.json 2> $O/config3.err; head -8 $O/config3_full_run.json | tee -a $O/pt_teams.txt
