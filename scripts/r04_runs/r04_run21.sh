#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python scripts/r04_det_diag3.py 8 256 3 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -32
timeout 600 python scripts/r04_det_diag3.py 16 256 3 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -32
