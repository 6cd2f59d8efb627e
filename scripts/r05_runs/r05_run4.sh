#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python scripts/exact_grad_probe.py > gpurun_out/r05_4_probe.txt 2>&1
cat gpurun_out/r05_4_probe.txt | tail -90
