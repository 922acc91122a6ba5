#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04q; mkdir -p $O; cd $R
one() { # label, lib or "", env...
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then env RTMI_PT_VERBOSE=1 "$@" timeout -k 10 300 python profiles/bench_with_lib.py $lib --workload pt > $O/b.json 2> $O/b.err
  else env RTMI_PT_VERBOSE=1 "$@" timeout -k 10 300 python bench.py --workload pt > $O/b.json 2> $O/b.err; fi
  echo "== $label: $(grep 'engines timed' $O/b.err | tail -1)" | tee -a $O/pt_engine.txt
  tail -1 $O/b.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms_per_sweep %.4f' % d['ms_per_step'], d['config'].get('engine')[:90])" | tee -a $O/pt_engine.txt
}
for rep in 1 2; do
one "4ade32a build" profiles/ab_prev/librtmi_pre.so A=1
one "in-tree build" "" A=1
one "in-tree build, teams=2" "" RTMI_PT_TEAMS=2
one "in-tree build, teams=2, 4 rows" "" RTMI_PT_TEAMS=2 RTMI_PT_SPECULATE=4
done
echo "== f32fast tile size (per-thread sources 4 / 5 / 6)" | tee $O/f32fast_tile.txt
for pt in 4 5 6; do for n in 1250000 10000000; do echo "F32F_PER_THREAD=$pt"; timeout -k 10 120 profiles/ubench/bin/f32fast_phases_pt$pt $n 2>&1 | head -1; done; done | tee -a $O/f32fast_tile.txt
timeout -k 10 120 profiles/ubench/bin/f32fast_phases_pt4 1250000 | tee -a $O/f32fast_tile.txt
