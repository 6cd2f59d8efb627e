#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
M355_TOP=40 timeout 600 python scripts/layer_times.py 128 2>&1 | grep -v "amdgpu.ids\|Warning\|warnings.warn" > gpurun_out/r04_layers_b128.txt; head -32 gpurun_out/r04_layers_b128.txt | cut -c1-150
