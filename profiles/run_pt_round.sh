R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02m; mkdir -p $O; cd $R
python bench.py --workload pt --steps 300 --warmup 30 > $O/bench_pt.json 2> $O/bench_pt.err; tail -c 500 $O/bench_pt.json
RTMI_PT_PERSISTENT=0 python bench.py --workload pt --steps 300 --warmup 30 > $O/bench_pt_graph_engine.json 2>> $O/bench_pt.err; tail -c 300 $O/bench_pt_graph_engine.json
python profiles/config3_run.py 10000 > $O/config3_full_run.json 2> $O/config3.err; head -c 200 $O/config3_full_run.json
RTMI_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload pt --steps 50 --warmup 5 > $O/bench_pt_2rank_gloo.json 2> $O/bench_pt_2rank_gloo.err; tail -c 200 $O/bench_pt_2rank_gloo.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02m/trace_pt2 -- python3 $R/bench.py --workload pt --steps 100 --warmup 10 > $O/trace_pt.log 2>&1 || echo "pt trace failed"
cat $(find $R/gpurun_out/prof_r02m/trace_pt2 -name "*kernel_stats.csv") | cut -c1-200
