#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_proj_gpu.py tests/test_distributed_gpu.py -m gpu -q 2>&1 | tail -8
bash scripts/make_profile.sh r04_v4 --steps 10 --warmup 3 2>&1 | tail -3
python - <<P
import json
j=json.load(open("$OUT/r04_v4_bench.json")); k=j["kernels_ms_per_step"]
print("r04_v4", round(j["value"],1), round(j["ms_per_step"],3), round(j["roofline"]["all_conv_tflops"],1), j.get("parity_ok"), round(j["gan_ms_per_cycle"],3), j["roofline"]["frac"])
print({a:round(b,3) for a,b in k.items() if a in ("proj_render_bwd","proj_render_fwd")}, j["roofline_proj"].get("bwd_over_fwd"))
P
