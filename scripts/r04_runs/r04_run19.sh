#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for b in 8 4 16; do
timeout 600 python scripts/r04_det_diag.py $b 256 2>&1 | grep -v amdgpu.ids | head -40
done
