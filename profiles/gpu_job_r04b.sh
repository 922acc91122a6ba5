#!/bin/bash
# profiles/gpu_job_r04b.sh -- the speculative sweep over the task queue: bit identity against round 3, timings, tests.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
export RTMI_PT_VERBOSE=1
REF=profiles/ab_prev/librtmi_r03.so; NEW=raytracerfortran_amd/librtmi.so
rm -f $O/ABORT
step() { [ -e $O/ABORT ] && { echo "skipped after a timeout: $*"; return 99; }; "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; touch $O/ABORT; fi; return $rc; }
step env RTMI_PT_PERSISTENT=0 timeout -k 10 300 python profiles/pt_state_dump.py $REF $O/ref.npz 2>&1 | grep -v amdgpu.ids
for spec in 1 2 4; do for ts in 640 2560; do
  echo "== queue engine: RTMI_PT_SPECULATE=$spec RTMI_PT_TASK_SRC=$ts"
  step env RTMI_PT_SPECULATE=$spec RTMI_PT_PERSISTENT=1 RTMI_PT_TASK_SRC=$ts timeout -k 10 300 python profiles/pt_state_dump.py $NEW $O/new_s${spec}_t${ts}.npz 2>&1 | grep -v amdgpu.ids | tail -5
  python profiles/cmp_npz.py $O/ref.npz $O/new_s${spec}_t${ts}.npz launches 2>&1 | cut -c1-200 | head -8
done; done | tee $O/bit_identity.txt
echo "== timings (bench pt, 32 chains x 1e4 sources, k <= 10)" | tee $O/pt_timings.txt
step timeout -k 10 200 python profiles/bench_with_lib.py $REF --workload pt --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/r03 default /' | tee -a $O/pt_timings.txt
for spec in 0 1 2 3 4; do for ts in 1280 2560 3840; do
  [ $spec = 0 ] && [ $ts != 2560 ] && continue
  step env RTMI_PT_SPECULATE=$spec RTMI_PT_PERSISTENT=1 RTMI_PT_TASK_SRC=$ts timeout -k 10 200 python bench.py --workload pt --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"sources_retraced_fraction": [0-9.]*' | tr '\n' ' ' | sed "s/^/spec=$spec task=$ts /"; echo
done; done | tee -a $O/pt_timings.txt
echo "== config 4 share (32 chains x 1e5 sources, k <= 30)" | tee -a $O/pt_timings.txt
for spec in 0 1 2 4; do for ts in 2560 3840; do
  [ $spec = 0 ] && [ $ts != 2560 ] && continue
  step env RTMI_PT_SPECULATE=$spec RTMI_PT_PERSISTENT=1 RTMI_PT_TASK_SRC=$ts timeout -k 10 300 python bench.py --workload pt --no-cpu-baseline --nsrc 100000 --kmax 30 --steps 60 --warmup 40 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"sources_retraced_fraction": [0-9.]*' | tr '\n' ' ' | sed "s/^/spec=$spec task=$ts /"; echo
done; done | tee -a $O/pt_timings.txt
echo "== pytest -m gpu (default build settings)"
step timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
