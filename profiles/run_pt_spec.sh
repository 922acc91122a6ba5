#!/bin/bash
# profiles/run_pt_spec.sh <tag> -- the speculative sweep: exactness tests, then config 3's shape at 1 ... 8 rows per chain
# (steady state, RTMI_PT_QSTATS for where the workgroups' time goes), config 3 at full size.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04v}; mkdir -p $O; cd $R
python -m pytest tests/test_tempering.py tests/test_gpu_posterior.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for s in 0 2 3 4 5 6 8; do
  echo "spec=$s $(RTMI_PT_SPECULATE=$s RTMI_PT_VERBOSE=1 RTMI_PT_QSTATS=1 python bench.py --workload pt --steps 300 --warmup 30 2>$O/err_$s.txt | grep -o '"ms_per_step": [0-9.]*') $(grep -o 'speculative sweeps.*' $O/err_$s.txt | tail -1)"
done | tee $O/pt_spec.txt
for s in 4 6 8; do echo "spec=$s (no instrumentation) $(RTMI_PT_SPECULATE=$s python bench.py --workload pt --steps 300 --warmup 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done | tee -a $O/pt_spec.txt
python profiles/config3_run.py 10000 > $O/config3_full_run.json 2> $O/config3.err; head -c 200 $O/config3_full_run.json | tr '\n' ' '; echo
