#!/bin/bash
# profiles/run_pt_cmp.sh [tag] -- the two engines of the device-resident sweep (per-slot launches replayed as a hipGraph;
# one persistent cooperative launch per sweep) must leave every chain bit-identical; then their timings, the automatic
# choice (a timed trial in the first long run), and two ranks sharing this one GPU (where the persistent kernel loses).
O=gpurun_out/${1:-pt_cmp}
mkdir -p $O
export RTMI_PT_VERBOSE=1
RTMI_PT_PERSISTENT=0 timeout -k 10 200 python profiles/pt_state_dump.py raytracerfortran_amd/librtmi.so $O/pt_graph.npz 2>&1 | grep -v amdgpu.ids
RTMI_PT_PERSISTENT=1 timeout -k 10 200 python profiles/pt_state_dump.py raytracerfortran_amd/librtmi.so $O/pt_pers.npz 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python profiles/pt_state_dump.py raytracerfortran_amd/librtmi.so $O/pt_auto.npz 2>&1 | grep -v amdgpu.ids
python profiles/cmp_npz.py $O/pt_graph.npz $O/pt_pers.npz launches | tee $O/engines_equal.txt
python profiles/cmp_npz.py $O/pt_graph.npz $O/pt_auto.npz launches | tee -a $O/engines_equal.txt
for e in 0 1 auto; do echo "RTMI_PT_PERSISTENT=$e"; if [ $e = auto ]; then unset RTMI_PT_PERSISTENT; else export RTMI_PT_PERSISTENT=$e; fi
  timeout -k 10 100 python bench.py --workload pt --no-cpu-baseline --steps 200 2>&1 | grep -v amdgpu.ids | grep -o 'rtmi: sweep engines.*\|"ms_per_step": [0-9.]*\|"launches_per_step": [0-9.]*'; done | tee $O/engines_ms.txt
unset RTMI_PT_PERSISTENT
echo "two ranks on this one GPU (gloo), automatic choice" | tee -a $O/engines_ms.txt
RTMI_BENCH_BACKEND=gloo timeout -k 10 200 python bench.py --gpus 2 --workload pt --steps 50 --warmup 8 2>&1 | grep -o 'rtmi: sweep engines.*\|"ms_per_step": [0-9.]*\|"launches_per_step": [0-9.]*' | tee -a $O/engines_ms.txt
