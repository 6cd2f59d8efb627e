#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
echo "--- whole GPU suite with the deterministic mode on from the environment"
M355_DETERMINISTIC=1 timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-400
echo "--- whole GPU suite with the second stream forced on"
M355_STREAMS=1 timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-400
