#!/bin/bash
# profiles/run_pt_spec2.sh <tag> -- rows per window uniform (RTMI_PT_SPEC_PROP=0) against in proportion to the chain's slots (1)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04y}; mkdir -p $O; cd $R
for pr in 0 1; do for s in 4 6 8; do
  echo "prop=$pr spec=$s $(RTMI_PT_SPEC_PROP=$pr RTMI_PT_SPECULATE=$s RTMI_PT_VERBOSE=1 RTMI_PT_QSTATS=1 python bench.py --workload pt --steps 300 --warmup 30 2>$O/err.txt | grep -o '"ms_per_step": [0-9.]*') $(grep -o 'speculative sweeps.*' $O/err.txt | tail -1)"
  echo "prop=$pr spec=$s (no instrumentation) $(RTMI_PT_SPEC_PROP=$pr RTMI_PT_SPECULATE=$s python bench.py --workload pt --steps 300 --warmup 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done; done | tee $O/pt_spec.txt
