#!/bin/bash
# profiles/pt_teams.sh <outfile> -- the proportional sweep engine with one workgroup of 1024 lanes per CU against two of 512
# (RTMI_PT_TEAMS): config 3's shape (three speculated rows, the default there) and config 4's per-GPU share (default: no
# speculation, round 3's static engine; RTMI_PT_SPECULATE=1: the proportional engine with one row), interleaved runs on one box.
O=${1:?outfile}
run() {  # label, env..., -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --workload pt "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$label', 'ms_per_sweep %.4f' % d['ms_per_step'], 'rows', c.get('speculation_rows_per_chain'), '|', str(c.get('engine',''))[:60])"
}
for rep in 1 2; do
  run "config-3 shape, teams=1" RTMI_PT_TEAMS=1 --
  run "config-3 shape, teams=2" RTMI_PT_TEAMS=2 --
  run "config-3 shape, teams=2, 4 rows" RTMI_PT_TEAMS=2 RTMI_PT_SPECULATE=4 --
  run "config-4 share, default engine" RTMI_PT_TEAMS=1 -- --kmax 30 --nsrc 100000 --steps 100
  run "config-4 share, proportional 1 row, teams=1" RTMI_PT_TEAMS=1 RTMI_PT_SPECULATE=1 -- --kmax 30 --nsrc 100000 --steps 100
  run "config-4 share, proportional 1 row, teams=2" RTMI_PT_TEAMS=2 RTMI_PT_SPECULATE=1 -- --kmax 30 --nsrc 100000 --steps 100
done | tee -a $O
