#!/bin/bash
# profiles/run_pt_cmp.sh -- the two engines of the device-resident sweep (per-slot launches replayed as a hipGraph; one
# persistent cooperative launch per sweep) must leave every chain bit-identical; then their timings.
set -e
O=gpurun_out/${1:-pt_cmp}
mkdir -p $O
export RTMI_PT_VERBOSE=1
RTMI_PT_PERSISTENT=0 timeout -k 10 200 python profiles/pt_state_dump.py raytracerfortran_amd/librtmi.so $O/pt_graph.npz 2>&1 | grep -v amdgpu.ids
RTMI_PT_PERSISTENT=1 timeout -k 10 200 python profiles/pt_state_dump.py raytracerfortran_amd/librtmi.so $O/pt_pers.npz 2>&1 | grep -v amdgpu.ids
python profiles/cmp_npz.py $O/pt_graph.npz $O/pt_pers.npz launches | tee $O/engines_equal.txt
for e in 0 1 0 1; do echo "RTMI_PT_PERSISTENT=$e"; RTMI_PT_PERSISTENT=$e timeout -k 10 100 python bench.py --workload pt --no-cpu-baseline --steps 200 2>&1 | grep -v amdgpu.ids | grep -o '"ms_per_step": [0-9.]*\|"launches_per_step": [0-9.]*'; done | tee $O/engines_ms.txt
