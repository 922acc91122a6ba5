#!/bin/bash
# which earlier GPU test of tests/test_tempering.py makes test_speculative_step_is_exact fail when it runs after it?
cd ${GRAFT_REPO_ROOT:-.}
for k in config4_miniature persistent_sweep sums_are_exact first_node nan_likelihood two_handles native_sampler_on_test_1 beta_zero chains_sharded device_swap_round device_run_without tempered_run_on_test_1; do
  r=$(timeout -k 10 200 python -m pytest tests/test_tempering.py -m gpu -q -k "$k or speculative" 2>&1 | tail -1)
  echo "$k: $r"
done
