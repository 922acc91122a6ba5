#!/bin/bash
# profiles/pmc_f32fast_lds.sh <outfile> -- whose LDS bank conflicts the fp32 fast kernel's counters show: the kernel as it is against
# the kernel without its layer loops (profiles/ubench/bin/f32fast_plain / f32fast_nosolve, built from profiles/ubench/f32fast_phases.hip
# without / with -DF32F_NO_SOLVE), 1.25e6 sources x 20 interfaces, per-launch means of a PMC pass of their own.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUTF=${1:?outfile}; W=$R/gpurun_out/prof_f32lds; mkdir -p $W
cd /tmp && export TMPDIR=/tmp
for v in plain nosolve; do
  rm -rf $W/$v
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $W/$v -- $R/profiles/ubench/bin/f32fast_$v 1250000 > $W/$v.log 2>&1 || echo "pmc $v failed"
  echo "== f32fast_$v" >> $OUTF
  python3 - >> $OUTF <<PY
import csv, glob, collections
for f in glob.glob("$W/$v/*/*_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "f32fast" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, x in sorted(agg.items()):
        print("%-24s %14.0f  (%d launches)" % (c, sum(x) / len(x), len(x)))
PY
done
