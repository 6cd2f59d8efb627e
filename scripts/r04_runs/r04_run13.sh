#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
echo "--- poison, streams off"; timeout 600 python scripts/r04_uninit_diag.py 2>&1 | grep -v amdgpu.ids | grep -v "^== [1-5] \|differing tensors: 0" | tail -30
echo "--- poison, streams on"; DIAG_STREAMS=1 timeout 600 python scripts/r04_uninit_diag.py 2>&1 | grep -v amdgpu.ids | grep -v "^== [1-5] \|differing tensors: 0" | tail -30
echo "--- G step of cycle 2"; timeout 600 python scripts/r04_fork_diag2.py 2>&1 | grep -v amdgpu.ids | tail -70
echo "--- G step of cycle 1"; DIAG_PRE=0 timeout 600 python scripts/r04_fork_diag2.py 2>&1 | grep -v amdgpu.ids | tail -10
