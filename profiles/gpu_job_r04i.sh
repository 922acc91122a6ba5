#!/bin/bash
# profiles/gpu_job_r04i.sh -- the build with the shape-based default (speculate where a pass is latency-bound): tests, defaults, posterior figures.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
export RTMI_PT_VERBOSE=1
rm -f $O/ABORT
step() { [ -e $O/ABORT ] && { echo "skipped after a timeout: $*"; return 99; }; "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; touch $O/ABORT; fi; return $rc; }
echo "== pytest -m gpu"
step timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
echo "== posterior tests, with their figures"
step timeout -k 10 600 python -m pytest tests/test_gpu_posterior.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids | grep "KS\|P(k=2)\|sigma\|passed\|failed" | tee $O/posterior.txt
echo "== defaults" | tee $O/pt_defaults.txt
step timeout -k 10 200 python bench.py --workload pt --no-cpu-baseline --steps 300 --warmup 30 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"speculation_rows_per_chain": [0-9]*\|rtmi: sweep engines.*' | tr '\n' ' ' | sed "s/^/config-3 shape (default) /" | tee -a $O/pt_defaults.txt; echo | tee -a $O/pt_defaults.txt
step timeout -k 10 300 python bench.py --workload pt --no-cpu-baseline --nsrc 100000 --kmax 30 --steps 60 --warmup 50 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"speculation_rows_per_chain": [0-9]*\|rtmi: sweep engines.*' | tr '\n' ' ' | sed "s/^/config-4 share (default) /" | tee -a $O/pt_defaults.txt; echo | tee -a $O/pt_defaults.txt
step env RTMI_PT_PERSISTENT=0 timeout -k 10 200 python bench.py --workload pt --no-cpu-baseline --steps 300 --warmup 30 2>&1 | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' ' | sed "s/^/config-3 shape, per-slot launches (speculative, default rows) /" | tee -a $O/pt_defaults.txt; echo | tee -a $O/pt_defaults.txt
step env RTMI_PT_PERSISTENT=0 RTMI_PT_SPECULATE=0 timeout -k 10 200 python bench.py --workload pt --no-cpu-baseline --steps 300 --warmup 30 2>&1 | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' ' | sed "s/^/config-3 shape, per-slot launches, one at a time /" | tee -a $O/pt_defaults.txt; echo | tee -a $O/pt_defaults.txt
echo "== config 3 at full size"; step timeout -k 10 300 python profiles/config3_run.py 10000 > $O/config3_full_run.json 2> $O/config3.err; head -c 400 $O/config3_full_run.json
echo "== 2 ranks on this one GPU, gloo"
RTMI_BENCH_BACKEND=gloo step timeout -k 10 300 python bench.py --gpus 2 --steps 10 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; tail -c 900 $O/bench_2rank_gloo.json
