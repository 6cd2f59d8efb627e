#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python scripts/r04_det512.py 2>&1 | grep -v amdgpu.ids | tail -20 | cut -c1-500
