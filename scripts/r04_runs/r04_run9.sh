#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_proj_gpu.py -m gpu -q 2>&1 | tail -12
for m in 0 1; do M355_DETERMINISTIC=$m timeout 300 python bench.py --no-cpu-baseline --workload proj 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); k=j['kernels_ms_per_step']; print('det=$m', round(j['ms_per_step'],4), {a:round(b,4) for a,b in k.items() if 'render' in a}, j.get('parity_ok'))"; done
