#!/bin/bash
# round 6 run 20: partial-row class weight gradients -- whole GPU suite (default), the model-level tests in deterministic mode
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r06_20_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r06_20_all.log; tail -3 gpurun_out/r06_20_all.log | cut -c1-300
M355_DETERMINISTIC=1 timeout 1700 python -m pytest tests/test_conv_gpu.py tests/test_gan_modules.py tests/test_headline_batch_gpu.py tests/test_exact_mode_gpu.py -m gpu -q > gpurun_out/r06_20_det.log 2>&1; echo "det rc=$?" >> gpurun_out/r06_20_det.log; tail -3 gpurun_out/r06_20_det.log | cut -c1-300
