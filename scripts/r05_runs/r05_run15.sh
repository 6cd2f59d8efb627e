#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gan_elem_gpu.py -m gpu -q -x -k tail -s > gpurun_out/r05_15_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05_15_tests.log
grep -n "conv_out.weight\|passed\|failed\|Error" gpurun_out/r05_15_tests.log | cut -c1-600 | head
M355_TOP=70 timeout 300 python scripts/layer_times.py 64 > gpurun_out/r05_15_layers.txt 2>&1
grep -E "cproj|512->1|total" gpurun_out/r05_15_layers.txt
