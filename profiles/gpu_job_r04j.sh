#!/bin/bash
# profiles/gpu_job_r04j.sh -- drop-in small path: spinning on a stream-written completion word against hipStreamSynchronize.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
for spin in 1 0 1 0; do
  RTMI_DROPIN_SPIN=$spin timeout -k 10 200 python bench.py --no-cpu-baseline --no-chain-parallel 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['dropin']
print('spin=$spin', {k: round(v,2) for k,v in d.items() if isinstance(v,float)})"
done | tee $O/dropin_spin.txt
timeout -k 10 200 python profiles/small_call_latency.py 2>&1 | grep -v amdgpu | tail -8 | tee -a $O/dropin_spin.txt
timeout -k 10 300 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_shim.py -m gpu -q 2>&1 | tail -3 | tee -a $O/dropin_spin.txt
# fp32 fast kernel with balanced contiguous slices (this build) -- config 5's per-GPU share and larger
echo "== f32fast" | tee $O/f32fast.txt
timeout -k 10 300 python -m pytest tests/test_gpu_f32fast.py -m gpu -q 2>&1 | tail -3 | tee -a $O/f32fast.txt
for i in 1 2; do timeout -k 10 200 python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done | tee -a $O/f32fast.txt
timeout -k 10 300 python profiles/f32fast_bench.py 2>&1 | grep -v amdgpu | tail -20 | tee -a $O/f32fast.txt
echo "== f32fast, previous build (4ade32a), then this one, at 1.25e6 / 4e6 / 1e7" | tee -a $O/f32fast.txt
for n in 1250000 4000000 10000000; do for lib in profiles/ab_prev/librtmi_pre.so "" profiles/ab_prev/librtmi_pre.so ""; do
  echo "lib=${lib:-in-tree}"; RTMI_AB_LIB=$lib timeout -k 10 200 python profiles/f32fast_bench.py $n 2>&1 | grep "fp32 fast"; done; done | tee -a $O/f32fast.txt
