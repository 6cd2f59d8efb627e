#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python scripts/r04_bf16_oracle.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -4 | cut -c1-400
