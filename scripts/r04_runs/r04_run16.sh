#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_distributed_gpu.py -m gpu -q -rx 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-1500
done
