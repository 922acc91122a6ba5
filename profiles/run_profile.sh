#!/bin/bash
# profiles/run_profile.sh <tag> -- rocprofv3 passes over bench.py on the GPU box (called through gpurun).
# Kernel trace + stats in one run; PMC counters in separate runs (never combined with tracing options).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dropin"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || echo "trace failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || echo "pmc fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || echo "pmc write failed"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || echo "pmc sq failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || echo "pmc sq2 failed"
find $OUT -name "*.csv" | head -50
