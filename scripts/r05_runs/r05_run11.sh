#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_gan_modules.py tests/test_gan_io_gpu.py tests/test_headline_batch_gpu.py -m gpu -q -x > gpurun_out/r05_11_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05_11_tests.log
tail -6 gpurun_out/r05_11_tests.log | cut -c1-300
M355_TOP=60 timeout 300 python scripts/layer_times.py 64 > gpurun_out/r05_11_layers.txt 2>&1
grep -E "8->64|total" gpurun_out/r05_11_layers.txt
