#!/bin/bash
# round 6 run 24: the per-layer table at batch 16 (which launches carry the 9.7 ms cycle of BASELINE configs[2])
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
M355_TOP=120 timeout 300 python scripts/layer_times.py 16 > gpurun_out/r06_layers_b16.txt 2>&1; tail -1 gpurun_out/r06_layers_b16.txt
