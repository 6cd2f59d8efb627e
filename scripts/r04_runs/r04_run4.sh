#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/r04_run4_pytest.txt
cat $OUT/r04_run4_pytest.txt
python scripts/thin_rate.py 0.6 wgrad > $OUT/r04_thin_rate_b.txt 2>&1; cat $OUT/r04_thin_rate_b.txt
python scripts/chamfer_rate.py > $OUT/r04_chamfer_rate.txt 2>&1; tail -12 $OUT/r04_chamfer_rate.txt
timeout 400 python bench.py 2> $OUT/r04_v3_bench.err | tail -1 > $OUT/r04_v3_bench.json
python - <<P
import json
j=json.load(open("$OUT/r04_v3_bench.json")); k=j["kernels_ms_per_step"]
print("r04_v3", round(j["value"],1), round(j["ms_per_step"],3), round(j["roofline"]["all_conv_tflops"],1), j.get("parity_ok"), round(j["gan_ms_per_cycle"],3))
print({a:round(b,3) for a,b in k.items() if a in ("k_wgrad_c8","k_wgrad_smallco","proj_render_bwd","proj_render_fwd","sn_power_iter","k_conv_c8")})
print(j["roofline_proj"].get("bwd_over_fwd"))
P
