#!/bin/bash
# round 6 run 17: the whole GPU suite under global switches (as round 4 did: scripts/r04_runs/r04_run33.sh) on the final tree -- deterministic
# mode everywhere, the second stream forced on at every batch, the spectral-norm prefetch off -- and the default bench line once more
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
M355_DETERMINISTIC=1 timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r06_17_det.log 2>&1; echo "det rc=$?" >> gpurun_out/r06_17_det.log; tail -3 gpurun_out/r06_17_det.log | cut -c1-300
M355_STREAMS=1 timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r06_17_streams.log 2>&1; echo "streams rc=$?" >> gpurun_out/r06_17_streams.log; tail -3 gpurun_out/r06_17_streams.log | cut -c1-300
M355_SN_PREFETCH=0 M355_NO_SUBPIXEL_SMALL=1 timeout 1700 python -m pytest tests/test_gan_modules.py tests/test_conv_gpu.py tests/test_headline_batch_gpu.py -m gpu -q > gpurun_out/r06_17_off.log 2>&1; echo "off rc=$?" >> gpurun_out/r06_17_off.log; tail -3 gpurun_out/r06_17_off.log | cut -c1-300
timeout 900 python bench.py --no-exact-cycle 2> gpurun_out/r06_17_bench.err | tail -1 > gpurun_out/r06_17_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_17_bench.json')); r=d['roofline']
print(round(d['value'],1), round(d['ms_per_step'],3), d.get('parity_ok'), round(r['frac'],4), r.get('traffic'), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if k.startswith('sn_') or k.startswith('weight_prep')})
PY
