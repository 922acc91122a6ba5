#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04p; mkdir -p $O; cd $R
for t in "" 1 2; do
  echo "== RTMI_PT_TEAMS='$t'" | tee -a $O/pt_engine.txt
  if [ -z "$t" ]; then RTMI_VERBOSE=1 timeout -k 10 300 python bench.py --workload pt > $O/b.json 2> $O/b.err
  else RTMI_PT_TEAMS=$t RTMI_VERBOSE=1 timeout -k 10 300 python bench.py --workload pt > $O/b.json 2> $O/b.err; fi
  grep -v amdgpu $O/b.err | tail -5 | tee -a $O/pt_engine.txt
  tail -1 $O/b.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_sweep %.4f' % d['ms_per_step'], d['config'].get('engine'))" | tee -a $O/pt_engine.txt
done
