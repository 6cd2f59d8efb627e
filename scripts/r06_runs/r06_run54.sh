#!/bin/bash
# round 6 run 54: the final tree -- whole GPU suite, one determinism soak, scripts/make_profile.sh r06_v7 (bench line + rocprofv3 kernel stats +
# FETCH / WRITE passes of the same command), per-layer table, the batch-16 line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r06_54_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r06_54_all.log
tail -3 gpurun_out/r06_54_all.log | cut -c1-300
( timeout 500 python scripts/soak_determinism.py 8 64 256 ) > gpurun_out/r06_54_soak.txt 2>&1; grep -c "SOAK OK" gpurun_out/r06_54_soak.txt
timeout 1500 bash scripts/make_profile.sh r06_v7 --steps 20 --warmup 5
M355_TOP=150 timeout 300 python scripts/layer_times.py 64 > gpurun_out/r06_layers_b64_v7.txt 2>&1; tail -1 gpurun_out/r06_layers_b64_final.txt
# the counters just collected belong to this tree: with them in place the bench line carries roofline.traffic (same hash)
cp gpurun_out/r06_v7_pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 2> /dev/null | tail -1 > gpurun_out/r06_v7_bench.json
timeout 900 python bench.py --batch 16 --workload gan 2> /dev/null | tail -1 > gpurun_out/r06_cfg3.json
python - <<'PY'
import json
for f in ('r06_v7_bench','r06_cfg3'):
    d=json.load(open('gpurun_out/%s.json'%f)); r=d['roofline']
    print(f, round(d['value'],1), round(d['ms_per_step'],3), d.get('parity_ok'), round(r['frac'],4), round(r['all_conv_tflops'],1), r.get('traffic'), round(r['avg_kernel_us'],1), r.get('rocprof_avg_kernel_us'))
PY
