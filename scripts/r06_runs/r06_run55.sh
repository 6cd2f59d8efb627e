#!/bin/bash
# round 6 run 55: per-launch constants of the main conv kernels on the final tree (time against the batch, intercept of the fit)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python scripts/probes/launch_constants.py 2>/dev/null | tee gpurun_out/r06_55_launch_constants.txt
