#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python scripts/r04_det512.py 2>&1 | grep -v amdgpu.ids | tail -20 | cut -c1-500
echo "--- projection + conv tests with the deterministic mode on from the environment"
M355_DETERMINISTIC=1 timeout 1800 python -m pytest tests/test_proj_gpu.py tests/test_conv_gpu.py tests/test_headline_batch_gpu.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-400
echo "--- projection tests, default"
timeout 1800 python -m pytest tests/test_proj_gpu.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400
python bench.py --workload proj --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('proj default', d['value'], d['ms_per_step'])"
M355_DETERMINISTIC=1 python bench.py --workload proj --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('proj deterministic', d['value'], d['ms_per_step'])"
