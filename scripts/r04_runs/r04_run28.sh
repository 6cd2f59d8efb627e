#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_conv_gpu.py -m gpu -q -k repeat_bit_identically 2>&1 | grep -v amdgpu.ids | tail -30 | cut -c1-300
timeout 900 python -m pytest tests/test_gan_modules.py -m gpu -q -k "deterministic_cycles_at_256" 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-300
timeout 900 python scripts/soak_determinism.py 8 64 256 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r04_soak_b64.txt | cut -c1-330
