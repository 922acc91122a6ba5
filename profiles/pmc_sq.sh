#!/bin/bash
# profiles/pmc_sq.sh <tag> [lib] -- the SQ instruction / cycle counters of the C2 bench for one build of the library
# (RTMI_AB_LIB = another librtmi.so), in a PMC pass of its own.  Prints the fused kernel's per-launch means.
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/profiles/${PROG:-c2}_loop.py"   # PROG=c5: the fp32 fast kernel at config 5's shape
K=${KERNEL:-fused_kernel}
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || echo "pmc sq failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || echo "pmc sq2 failed"
python3 - <<PY
import csv, glob, collections
for d in ("pmc_sq", "pmc_sq2"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % d):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "$K" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(agg.items()):
            print("%-24s %14.0f  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
