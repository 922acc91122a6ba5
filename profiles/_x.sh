export RTMI_PT_VERBOSE=1
echo "config-3 shape"; python bench.py --workload pt --steps 300 --warmup 30 2>&1 | grep -o 'rtmi: sweep.*\|"ms_per_step": [0-9.]*\|"launches_per_step": [0-9.]*'
echo "config-4 share"; python bench.py --workload pt --nsrc 100000 --kmax 30 --steps 60 --warmup 60 2>&1 | grep -o 'rtmi: sweep.*\|"ms_per_step": [0-9.]*\|"launches_per_step": [0-9.]*'
echo "config 3 full run"; python profiles/config3_run.py 10000 2>&1 >/tmp/c3.json | grep -o 'rtmi: sweep.*'; head -c 200 /tmp/c3.json | tr '\n' ' '; echo
echo "two ranks, one GPU"; RTMI_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload pt --steps 50 --warmup 14 2>&1 | grep -o 'rtmi: sweep.*\|"ms_per_step": [0-9.]*\|"launches_per_step": [0-9.]*'
echo "k30 miniature"; timeout -k 10 200 python profiles/pt_state_dump.py raytracerfortran_amd/librtmi.so /tmp/x.npz 2>&1 | grep -o 'rtmi: sweep.*'
timeout -k 10 300 python -m pytest tests/test_tempering.py -x -q 2>&1 | tail -2
