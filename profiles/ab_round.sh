#!/bin/bash
# profiles/ab_round.sh <other librtmi.so> <label> -- the round's build against another build on ONE box, every shape that
# DESIGN.md quotes: interleaved kernel timings (ab_multi.py) and the sampler's sweeps through bench.py, alternating.
L=$1; N=${2:-other}
python profiles/ab_multi.py $N=$L cur=x 2>&1 | grep -v amdgpu.ids
j() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],4), 'ms per sweep')"; }
for i in 1 2; do
  python profiles/bench_with_lib.py $L --workload pt --steps 300 --warmup 30 2>/dev/null | j "$N  pt 32 x 1e4, k <= 10      "
  python bench.py --workload pt --steps 300 --warmup 30 2>/dev/null | j "cur  pt 32 x 1e4, k <= 10      "
done
for i in 1 2; do
  python profiles/bench_with_lib.py $L --workload pt --nsrc 100000 --kmax 30 --steps 60 --warmup 50 2>/dev/null | j "$N  pt 32 x 1e5, k <= 30      "
  python bench.py --workload pt --nsrc 100000 --kmax 30 --steps 60 --warmup 50 2>/dev/null | j "cur  pt 32 x 1e5, k <= 30      "
done
