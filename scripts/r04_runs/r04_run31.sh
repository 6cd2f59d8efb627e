#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python scripts/soak_determinism.py 12 8 512 3 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r04_soak_512.txt | cut -c1-330
timeout 900 python scripts/soak_determinism.py 12 16 512 2 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a gpurun_out/r04_soak_512.txt | cut -c1-330
