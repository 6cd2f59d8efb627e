#!/bin/bash
# round 5 run 1: sub-pixel upsample conv -- conv tests, whole GPU suite, per-layer table, bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q > gpurun_out/r05_1_conv.log 2>&1; echo "conv rc=$?" >> gpurun_out/r05_1_conv.log
tail -15 gpurun_out/r05_1_conv.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_conv_gpu.py > gpurun_out/r05_1_rest.log 2>&1; echo "rest rc=$?" >> gpurun_out/r05_1_rest.log
tail -15 gpurun_out/r05_1_rest.log
timeout 300 python scripts/layer_times.py 64 > gpurun_out/r05_1_layers.txt 2>&1
M355_NO_SUBPIXEL=1 timeout 300 python scripts/layer_times.py 64 > gpurun_out/r05_1_layers_nosub.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/r05_1_bench.err | tail -1 > gpurun_out/r05_1_bench.json
cut -c1-400 gpurun_out/r05_1_bench.json
