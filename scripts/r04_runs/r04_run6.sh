#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_gan_modules.py tests/test_gan_elem_gpu.py tests/test_reconstruction.py tests/test_recon_step.py -m gpu -q -x 2>&1 | tail -15
for v in 0 1 0 1; do
  if [ $v = 0 ]; then export M355_NO_SPLITK=1; else unset M355_NO_SPLITK; fi
  timeout 400 python bench.py --no-cpu-baseline --workload gan 2> $OUT/r04_sk$v.err | tail -1 > $OUT/r04_sk$v.json
  python - <<P
import json
j=json.load(open("$OUT/r04_sk$v.json")); k=j["kernels_ms_per_step"]
print("splitk=$v", round(j["ms_per_step"],3), round(j["roofline"]["all_conv_tflops"],1), {a:round(b,3) for a,b in k.items() if a in ("k_conv_glds","bn_stats_partial","fold2x2","bn_finalize")}, round(sum(k.values()),3))
P
done
unset M355_NO_SPLITK
M355_TIMER_TAGS=1 timeout 300 python bench.py --no-cpu-baseline --workload gan --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/r04_sk_layers.json
python - <<P
import json
j=json.load(open("$OUT/r04_sk_layers.json")); k=j["kernels_ms_per_step"]
for a,b in sorted(k.items(), key=lambda x:-x[1]):
    if "k_conv_glds" in a or "conv2d_fwd_ws" in a or "k_wgrad_dma" in a: print(round(b,4), a)
P
