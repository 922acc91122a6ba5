#!/bin/bash
# profiles/gpu_job_r04h.sh -- proportional engine with XCD-major layout: where the workgroups' time goes; timings; C2 A/B; bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R
export RTMI_PT_VERBOSE=1
rm -f $O/ABORT
step() { [ -e $O/ABORT ] && { echo "skipped after a timeout: $*"; return 99; }; "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; touch $O/ABORT; fi; return $rc; }
echo "== timings (bench pt, 32 chains x 1e4 sources, k <= 10)" | tee $O/pt_timings.txt
for spec in 0 1 2 3 4; do
  step env RTMI_PT_SPECULATE=$spec RTMI_PT_PERSISTENT=1 timeout -k 10 200 python bench.py --workload pt --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' ' | sed "s/^/spec=$spec /"; echo
done | tee -a $O/pt_timings.txt
for spec in 1 2 3 4; do
  step env RTMI_PT_QSTATS=1 RTMI_PT_SPECULATE=$spec RTMI_PT_PERSISTENT=1 timeout -k 10 200 python bench.py --workload pt --no-cpu-baseline --steps 300 --warmup 30 2>&1 | grep -o '"ms_per_step": [0-9.]*\|rtmi: speculative sweeps (300).*' | tr '\n' ' ' | sed "s/^/spec=$spec (instrumented) /"; echo
done | tee -a $O/pt_timings.txt
echo "== config 4 share (32 chains x 1e5 sources, k <= 30)" | tee -a $O/pt_timings.txt
for spec in 0 1 2; do
  step env RTMI_PT_QSTATS=1 RTMI_PT_SPECULATE=$spec RTMI_PT_PERSISTENT=1 timeout -k 10 300 python bench.py --workload pt --no-cpu-baseline --nsrc 100000 --kmax 30 --steps 60 --warmup 40 2>&1 | grep -o '"ms_per_step": [0-9.]*\|rtmi: speculative sweeps (60).*' | tr '\n' ' ' | sed "s/^/spec=$spec /"; echo
done | tee -a $O/pt_timings.txt
echo "== C2 kernel, round 3 against this build (one process, interleaved)" | tee $O/ab_c2.txt
cp profiles/ab_prev/librtmi_r03.so profiles/ab_prev/librtmi.so
step timeout -k 10 300 python profiles/ab_libs.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_c2.txt
echo "== bench.py (default workload) and --workload pt"
step timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
step timeout -k 10 300 python bench.py --workload pt --steps 300 --warmup 30 > $O/bench_pt.json 2> $O/bench_pt.err; tail -c 1200 $O/bench_pt.json
echo "== pytest -m gpu"
step timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
