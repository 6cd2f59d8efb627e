#!/bin/bash
# round 6 run 1: round-start tree + ADVICE r5 fixes + the conv_halo A/B-switch cleanup: whole GPU suite, per-layer tables at batch 64 and
# batch 16 (all m355 kernels), the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r06_1_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r06_1_all.log
tail -4 gpurun_out/r06_1_all.log | cut -c1-300
M355_TOP=150 timeout 300 python scripts/layer_times.py 64 > gpurun_out/r06_1_layers_b64.txt 2>&1
M355_TOP=150 timeout 300 python scripts/layer_times.py 16 > gpurun_out/r06_1_layers_b16.txt 2>&1
tail -1 gpurun_out/r06_1_layers_b64.txt; tail -1 gpurun_out/r06_1_layers_b16.txt
timeout 600 python bench.py --no-cpu-baseline --no-step-parity 2> gpurun_out/r06_1_bench.err | tail -1 > gpurun_out/r06_1_bench.json
timeout 600 python bench.py --no-cpu-baseline --no-step-parity --batch 16 --workload gan 2>> gpurun_out/r06_1_bench.err | tail -1 > gpurun_out/r06_1_bench_b16.json
timeout 600 python bench.py --no-cpu-baseline --no-step-parity --batch 16 --workload gan --graph 2>> gpurun_out/r06_1_bench.err | tail -1 > gpurun_out/r06_1_bench_b16_graph.json
python - <<'PY'
import json
for f in ('r06_1_bench','r06_1_bench_b16','r06_1_bench_b16_graph'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); print(f, round(d['value'],1), round(d['ms_per_step'],3), d.get('parity_ok'), round(d['roofline']['frac'],4), round(d['roofline']['all_conv_tflops'],1))
    except Exception as e: print(f, 'FAILED', e)
PY
