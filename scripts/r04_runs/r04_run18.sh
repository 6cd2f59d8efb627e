#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python scripts/soak_determinism.py 100 8 128 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r04_soak.txt
timeout 900 python scripts/soak_determinism.py 25 32 256 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a gpurun_out/r04_soak.txt
