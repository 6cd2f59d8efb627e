#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_distributed_gpu.py tests/test_gan_modules.py -m gpu -q -rx 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-300
