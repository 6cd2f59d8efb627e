#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gan_elem_gpu.py tests/test_gan_modules.py tests/test_gan_io_gpu.py tests/test_exact_mode_gpu.py tests/test_distributed_gpu.py -m gpu -q -x > gpurun_out/r05_14_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05_14_tests.log
tail -15 gpurun_out/r05_14_tests.log | cut -c1-400
M355_TOP=70 timeout 300 python scripts/layer_times.py 64 > gpurun_out/r05_14_layers.txt 2>&1
grep -E "cproj|512->1|total" gpurun_out/r05_14_layers.txt
M355_NO_TAIL_FUSION=1 M355_TOP=70 timeout 300 python scripts/layer_times.py 64 > gpurun_out/r05_14_layers_nofuse.txt 2>&1
grep -E "cproj|512->1|total" gpurun_out/r05_14_layers_nofuse.txt
