mkdir -p gpurun_out/r03c
python -m pytest tests -m gpu -x -q > gpurun_out/r03c/gputest.log 2>&1; tail -3 gpurun_out/r03c/gputest.log
python profiles/ab_multi.py r02=profiles/ab_prev/librtmi_r02.so head=profiles/ab_prev/librtmi_head.so cur=x 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03c/ab_prefetch_static.txt
python profiles/phase_clocks.py 1000000 3 0 2>&1 | egrep "load|solve  |total|phase a"
for i in 1 2; do
python profiles/bench_with_lib.py profiles/ab_prev/librtmi_head.so --workload pt --steps 300 --warmup 30 2>gpurun_out/r03c/bwl.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('head pt ms/sweep', j['ms_per_step'])"
python bench.py --workload pt --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cur pt ms/sweep', j['ms_per_step'])"
done
python profiles/bench_with_lib.py profiles/ab_prev/librtmi_head.so --workload pt --nsrc 100000 --kmax 30 --steps 60 --warmup 20 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('head pt c4share ms/sweep', j['ms_per_step'])"
python bench.py --workload pt --nsrc 100000 --kmax 30 --steps 60 --warmup 20 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cur pt c4share ms/sweep', j['ms_per_step'])"
tail -3 gpurun_out/r03c/bwl.err
