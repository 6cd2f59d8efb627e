#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
DIAG_B=8 DIAG_R=256 timeout 600 python scripts/r04_uninit_diag.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 | head -70
