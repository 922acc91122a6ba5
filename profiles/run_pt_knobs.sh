#!/bin/bash
# profiles/run_pt_knobs.sh <tag> -- the fused body's scheduling knobs (tuned on C2's 16-wave workgroups) on the sampler's sweep
# (config 3's shape: 8-wave workgroups, a couple of records per lane): one knob at a time against the defaults
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04ap}; mkdir -p $O; cd $R
run() { echo "$* $(env "$@" python bench.py --workload pt --steps 600 --warmup 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
{
run A=defaults
run RTMI_REFILL_MIN=8
run RTMI_REFILL_MIN=16
run RTMI_REFILL_MIN=48
run RTMI_TAIL_GROUP=0
run RTMI_TAIL_GROUP=64
run RTMI_TAIL_GROUP=256
run RTMI_RETIRE_MAX=0
run RTMI_RETIRE_MAX=6
run RTMI_RETIRE_MAX=24
run RTMI_PROMOTE_AGE=0
run RTMI_PROMOTE_AGE=2
run RTMI_PROMOTE_AGE=6
run A=defaults
} | tee $O/pt_knobs.txt
