#!/bin/bash
# profiles/run_round.sh <tag> -- every measurement DESIGN.md / README.md quote, on the GPU box (called through gpurun):
#   gpurun --timeout 1100 -- 'bash profiles/run_round.sh r02e'
# then, back in the build container:  python profiles/summarize.py r02e && cp gpurun_out/r02e/* profiles/r02e/
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
echo "== bench (C2 headline, with drop-in block and CPU baseline)"; python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
echo "== bench c4"; python bench.py --workload c4 --steps 20 > $O/bench_c4.json 2> $O/bench_c4.err
echo "== bench pt"; python bench.py --workload pt --steps 300 --warmup 30 > $O/bench_pt.json 2> $O/bench_pt.err; tail -c 300 $O/bench_pt.json
echo "== bench pt at config 4's per-GPU share (warm-up past the engine trial, which opens once the sweeps have kmax node slots)"; python bench.py --workload pt --nsrc 100000 --kmax 30 --steps 60 --warmup 50 > $O/bench_pt_config4_share.json 2> $O/bench_pt_c4.err
echo "== bench c5 (fp32 fast) + the three single-GPU paths at config 5's shape"; python bench.py --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err
python profiles/f32fast_bench.py > $O/f32fast_bench.txt 2>&1; python profiles/f32fast_bench.py 10000000 > $O/f32fast_bench_1e7.txt 2>&1
echo "== 2 ranks on this one GPU, gloo (self-launched by bench.py --gpus 2)"
RTMI_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
RTMI_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload pt --steps 50 --warmup 5 > $O/bench_pt_2rank_gloo.json 2> $O/bench_pt_2rank_gloo.err
echo "== config 3 at full size"; python profiles/config3_run.py 10000 > $O/config3_full_run.json 2> $O/config3.err; head -c 300 $O/config3_full_run.json
echo "== phase clocks"; python profiles/phase_clocks.py 1000000 3 0 > $O/phase_clocks_coarse.txt 2>&1; python profiles/phase_clocks.py 1000000 3 1 > $O/phase_clocks.txt 2>&1
python profiles/phase_clocks.py 320000 3 0 > $O/phase_clocks_320k.txt 2>&1
echo "== secondary"; python profiles/fp32_vs_fp64.py > $O/fp32_vs_fp64.txt 2>&1; python profiles/launch_gap.py > $O/launch_gap.txt 2>&1
python profiles/ab_batch.py > $O/ab_batch.txt 2>&1; python profiles/dropin_modes.py > $O/dropin_modes.txt 2>&1
python profiles/gpu_div_vs_cpu.py > $O/gpu_div_vs_cpu.txt 2>&1; python profiles/near_allones_scan.py > $O/near_allones_scan.txt 2>&1
echo "== rocprofv3: C2 (trace + PMC passes), then the PT sweep (trace)"
bash profiles/run_profile.sh $TAG > $O/run_profile.log 2>&1
PROG=c5 KERNEL=f32fast bash profiles/pmc_sq.sh ${TAG}_c5 > $O/pmc_c5fast.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/trace_pt -- python3 $R/bench.py --workload pt --steps 100 --warmup 10 > $O/trace_pt.log 2>&1 || echo "pt trace failed"
cd $R; ls $O
