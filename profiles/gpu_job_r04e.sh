#!/bin/bash
# profiles/gpu_job_r04e.sh -- queue engine v3 (ticket ring + low-priority ring): where the workgroups' time goes, timings.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
export RTMI_PT_VERBOSE=1
rm -f $O/ABORT
step() { [ -e $O/ABORT ] && { echo "skipped after a timeout: $*"; return 99; }; "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; touch $O/ABORT; fi; return $rc; }
echo "== timings (bench pt, 32 chains x 1e4 sources, k <= 10)" | tee $O/pt_timings.txt
step env RTMI_PT_SPECULATE=0 RTMI_PT_PERSISTENT=1 timeout -k 10 200 python bench.py --workload pt --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/spec=0 static persistent /' | tee -a $O/pt_timings.txt
for spec in 1 2 3 4; do for ts in "768 2560" "1280 2560" "2560 2560" "1280 3840"; do set -- $ts
  step env RTMI_PT_QSTATS=1 RTMI_PT_SPECULATE=$spec RTMI_PT_PERSISTENT=1 RTMI_PT_TASK_FINE=$1 RTMI_PT_TASK_SRC=$2 timeout -k 10 200 python bench.py --workload pt --no-cpu-baseline --steps 300 --warmup 30 2>&1 | grep -o '"ms_per_step": [0-9.]*\|rtmi: queue sweeps (300).*' | tr '\n' ' ' | sed "s/^/spec=$spec fine=$1 coarse=$2 /"; echo
done; done | tee -a $O/pt_timings.txt
echo "== config 4 share (32 chains x 1e5 sources, k <= 30)" | tee -a $O/pt_timings.txt
step env RTMI_PT_SPECULATE=0 RTMI_PT_PERSISTENT=1 timeout -k 10 300 python bench.py --workload pt --no-cpu-baseline --nsrc 100000 --kmax 30 --steps 60 --warmup 40 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/spec=0 static persistent /' | tee -a $O/pt_timings.txt
for spec in 1 2 4; do for ts in "3840 3840" "3840 12800" "12800 12800"; do set -- $ts
  step env RTMI_PT_QSTATS=1 RTMI_PT_SPECULATE=$spec RTMI_PT_PERSISTENT=1 RTMI_PT_TASK_FINE=$1 RTMI_PT_TASK_SRC=$2 timeout -k 10 300 python bench.py --workload pt --no-cpu-baseline --nsrc 100000 --kmax 30 --steps 60 --warmup 40 2>&1 | grep -o '"ms_per_step": [0-9.]*\|rtmi: queue sweeps (60).*' | tr '\n' ' ' | sed "s/^/spec=$spec fine=$1 coarse=$2 /"; echo
done; done | tee -a $O/pt_timings.txt
echo "== pytest -m gpu"
step timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.txt
