#!/bin/bash
# profiles/run_pt_trace.sh <tag> -- kernel trace of the sampler's sweeps (durations of the sweep kernel against the launch-to-launch period)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04z}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in 4 8; do
RTMI_PT_SPECULATE=$s rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pt_$s -- python3 $R/bench.py --workload pt --steps 100 --warmup 10 > $O/trace_pt_$s.log 2>&1 || echo "pt trace failed"
f=$(find $O/trace_pt_$s -name "*kernel_stats.csv" | head -1); echo "== spec=$s"; cut -c1-200 $f | head -6
t=$(find $O/trace_pt_$s -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
sw=[r for r in rows if 'pt_sweep_prop' in r['Kernel_Name']]
sw.sort(key=lambda r:int(r['Start_Timestamp']))
sw=sw[20:]
dur=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in sw]
per=[(int(b['Start_Timestamp'])-int(a['Start_Timestamp']))/1e3 for a,b in zip(sw,sw[1:])]
import statistics as st
print("sweep kernel: median duration %.1f us, median start-to-start period %.1f us, n=%d"%(st.median(dur),st.median(per),len(dur)))
P
done
rm -rf $O/trace_pt_4 $O/trace_pt_8
