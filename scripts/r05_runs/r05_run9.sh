#!/bin/bash
# round 5 run 9: the bench line (with CPU baselines + step parity), rocprofv3 kernel stats + FETCH/WRITE PMC passes of the same command,
# per-layer table
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/make_profile.sh r05_v1 --steps 20 --warmup 5 > gpurun_out/r05_9_profile.log 2>&1
tail -3 gpurun_out/r05_9_profile.log | cut -c1-300
M355_TOP=120 timeout 300 python scripts/layer_times.py 64 > gpurun_out/r05_layers.txt 2>&1
tail -2 gpurun_out/r05_layers.txt
