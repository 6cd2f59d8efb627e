#!/bin/bash
# round 6 run 23: whole GPU suite after the arena test's shapes moved off the partial-row layers; deterministic-mode model tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r06_23_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r06_23_all.log; tail -4 gpurun_out/r06_23_all.log | cut -c1-300
M355_DETERMINISTIC=1 timeout 1700 python -m pytest tests/test_conv_gpu.py tests/test_gan_modules.py tests/test_headline_batch_gpu.py tests/test_exact_mode_gpu.py -m gpu -q > gpurun_out/r06_23_det.log 2>&1; echo "det rc=$?" >> gpurun_out/r06_23_det.log; tail -4 gpurun_out/r06_23_det.log | cut -c1-300
