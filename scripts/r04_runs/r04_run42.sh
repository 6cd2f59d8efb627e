#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
M355_TOP=70 timeout 600 python scripts/layer_times.py 64 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_layers.txt; tail -3 gpurun_out/r04_layers.txt
