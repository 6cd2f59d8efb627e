#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_render.py tests/test_recon_step.py tests/test_mesh.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -15 | cut -c1-600
echo "--- the same with the mode on from the environment"
M355_DETERMINISTIC=1 timeout 900 python -m pytest tests/test_render.py tests/test_recon_step.py tests/test_mesh.py tests/test_reconstruction.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-400
for m in 0 1; do M355_DETERMINISTIC=$m python bench.py --workload recon --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('recon deterministic=$m', round(d['value'],1), d['unit'], round(d['ms_per_step'],3))"; done
