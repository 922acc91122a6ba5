mkdir -p gpurun_out/r03b
python -m pytest tests -m gpu -x -q > gpurun_out/r03b/gputest2.log 2>&1; tail -5 gpurun_out/r03b/gputest2.log
python profiles/f32fast_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03b/f32fast_bench.txt
python profiles/ab_multi.py r02=profiles/ab_prev/librtmi_r02.so cur=x 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03b/ab_acc124.txt
for i in 1 2; do
python profiles/bench_with_lib.py profiles/ab_prev/librtmi_r02.so --workload pt --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r02 pt ms/sweep', j['ms_per_step'])"
python bench.py --workload pt --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cur pt ms/sweep', j['ms_per_step'])"
done
python profiles/bench_with_lib.py profiles/ab_prev/librtmi_r02.so --workload pt --nsrc 100000 --kmax 30 --steps 60 --warmup 20 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r02 pt c4share ms/sweep', j['ms_per_step'])"
python bench.py --workload pt --nsrc 100000 --kmax 30 --steps 60 --warmup 20 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cur pt c4share ms/sweep', j['ms_per_step'])"
