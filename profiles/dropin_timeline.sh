#!/bin/bash
# profiles/dropin_timeline.sh <tag> -- copies and kernels of the LAST of six dff_ calls at 1e6 sources, relative to its first copy
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04ao}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl -- python3 $R/profiles/dropin_timeline.py > $O/timeline_run.txt 2>&1
cat $O/timeline_run.txt | grep call
python3 - $O <<'P'
import csv, glob, sys
O = sys.argv[1]
ev = []
for f in glob.glob(O + "/tl/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Kind", "?")) ))
for f in glob.glob(O + "/tl/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel " + r["Kernel_Name"][:40]))
ev.sort()
# the last burst: events after the last gap > 5 ms
start = 0
for i in range(1, len(ev)):
    if ev[i][0] - ev[i - 1][1] > 5_000_000:
        start = i
t0 = ev[start][0]
with open(O + "/dropin_timeline.txt", "w") as out:
    for s, e, name in ev[start:]:
        line = "%8.1f us -> %8.1f us  (%6.1f us)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, name)
        print(line); out.write(line + "\n")
P
rm -rf $O/tl
