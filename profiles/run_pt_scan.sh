#!/bin/bash
# profiles/run_pt_scan.sh <tag> -- multi-sweep launches: rows per window x proportional windows on config 3's shape; config 4's share with 0 / 1 / 2 rows
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04ab}; mkdir -p $O; cd $R
for pr in 0 1; do for s in 4 5 6 7 8; do
  echo "config3 prop=$pr spec=$s $(RTMI_PT_SPEC_PROP=$pr RTMI_PT_SPECULATE=$s python bench.py --workload pt --steps 600 --warmup 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done; done | tee $O/pt_scan.txt
for t in 1 2; do echo "config3 spec=6 teams=$t $(RTMI_PT_TEAMS=$t RTMI_PT_SPECULATE=6 python bench.py --workload pt --steps 600 --warmup 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done | tee -a $O/pt_scan.txt
for s in 0 1 2 3; do
  echo "config4share spec=$s $(RTMI_PT_SPECULATE=$s python bench.py --workload pt --nsrc 100000 --kmax 30 --steps 60 --warmup 50 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done | tee -a $O/pt_scan.txt
