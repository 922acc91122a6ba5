#!/bin/bash
# profiles/run_pt_round.sh [tag] -- the sampler's figures: bench pt (default = engine trial + incremental likelihood), its
# variants, config 3 at full size, two ranks on this GPU, the kernel trace.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r02m}; mkdir -p $O; cd $R
export RTMI_PT_VERBOSE=1
python bench.py --workload pt --steps 300 --warmup 30 > $O/bench_pt.json 2> $O/bench_pt.err; grep "rtmi:" $O/bench_pt.err; grep -o '"ms_per_step": [0-9.]*\|"launches_per_step": [0-9.]*' $O/bench_pt.json
for v in "0 0" "0 1" "1 0" "1 1"; do set -- $v; echo "persistent=$1 incremental=$2"; RTMI_PT_PERSISTENT=$1 RTMI_PT_INCREMENTAL=$2 python bench.py --workload pt --steps 300 --warmup 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done | tee $O/pt_variants.txt
python profiles/config3_run.py 10000 > $O/config3_full_run.json 2> $O/config3.err; grep "rtmi:" $O/config3.err; head -c 230 $O/config3_full_run.json | tr '\n' ' '; echo
RTMI_PT_INCREMENTAL=0 python profiles/config3_run.py 10000 > $O/config3_full_run_plain.json 2>/dev/null; head -c 230 $O/config3_full_run_plain.json | tr '\n' ' '; echo
RTMI_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload pt --steps 50 --warmup 10 > $O/bench_pt_2rank_gloo.json 2> $O/bench_pt_2rank_gloo.err; grep -o '"ms_per_step": [0-9.]*\|"launches_per_step": [0-9.]*' $O/bench_pt_2rank_gloo.json
python profiles/ab_libs.py 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_after_inc.txt
cd /tmp && export TMPDIR=/tmp
RTMI_PT_PERSISTENT=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${1:-r02m}/trace_pt3 -- python3 $R/bench.py --workload pt --steps 100 --warmup 10 > $O/trace_pt.log 2>&1 || echo "pt trace failed"
cat $(find $R/gpurun_out/prof_${1:-r02m}/trace_pt3 -name "*kernel_stats.csv") | cut -c1-160 | head -5
