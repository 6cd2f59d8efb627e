#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_proj_gpu.py -m gpu -q -k "repeat" 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300
