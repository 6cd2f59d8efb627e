#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python scripts/r04_poison_other.py 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-300 | tee gpurun_out/r04_poison.txt
