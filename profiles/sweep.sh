#!/bin/bash
# profiles/sweep.sh -- A/B of the two kernels and of the fused kernel's scheduling knobs (one process per
# setting).  Prints one line per setting.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-44s value=%.3e tt/s  kernel_ms=%.4f' % (sys.argv[1], d['value'], d['roofline']['kernel_ms']))" "$LABEL"; }
LABEL="plain" run --path plain
for W in 1 2 3 4 6 8; do for M in 2 4 8 16; do
  export RTMI_SOLVE_WAVES=$W RTMI_REFILL_MIN=$M; LABEL="fused waves/SIMD=$W refill_min=$M" run --path fused
done; done
