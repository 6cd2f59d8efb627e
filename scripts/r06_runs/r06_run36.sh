#!/bin/bash
# round 6 run 36: what the driver runs at round end, in its order: the GPU suite with -x, smoke(), the default bench line (timed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_36_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r06_36_gpu.log; tail -2 gpurun_out/r06_36_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r06_36_smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_36_bench.json 2> gpurun_out/r06_36_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_36_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value'],1), round(d['ms_per_step'],3), d.get('parity_ok'), round(r['frac'],4), r.get('traffic'), d['cpu_baseline']['value'])
PY
