#!/bin/bash
# round 6 run 10: which Python lines launch the ATen kernels of a cycle (torch.profiler with stacks), batch 64 and batch 16
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python scripts/small_kernel_sources.py 64 256 > gpurun_out/r06_10_aten_b64.txt 2>&1; head -50 gpurun_out/r06_10_aten_b64.txt | cut -c1-330
