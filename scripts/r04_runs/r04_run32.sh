#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python scripts/chamfer_rate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_chamfer_rate2.txt
