#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -k "every_shape_class" 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-500
