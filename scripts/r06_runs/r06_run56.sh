#!/bin/bash
# round 6 run 56 (called three times = three boxes): the default bench line of the final tree, as the driver runs it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_56_bench_$(date +%s).json
python - <<'PY'
import json,glob
f=sorted(glob.glob('gpurun_out/r06_56_bench_*.json'))[-1]; d=json.load(open(f)); r=d['roofline']
print(f, round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3), d['parity_ok'], round(r['frac'],4), round(r['all_conv_tflops'],1), round(d['sustained']['sclk_mhz']), round(d['sustained']['power_w']))
PY
