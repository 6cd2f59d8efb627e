#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python scripts/r04_big_batch.py 256; timeout 900 python scripts/r04_big_batch.py 512 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-400
