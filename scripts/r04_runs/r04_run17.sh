#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1800 python -m pytest tests -m gpu -q -rx 2>&1 | grep -v amdgpu.ids | tail -25 | cut -c1-3000
