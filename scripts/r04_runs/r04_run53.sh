#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
bash scripts/make_profile.sh r04_v14 2>&1 | tail -2
python - <<P
import json
j=json.load(open("$OUT/r04_v14_bench.json")); k=j["kernels_ms_per_step"]
print("r04_v14", round(j["value"],1), round(j["ms_per_step"],3), round(j["roofline"]["all_conv_tflops"],1), j.get("parity_ok"), round(j["gan_ms_per_cycle"],3), j["roofline"]["frac"], j["roofline"]["avg_kernel_us"], j["roofline"]["rocprof_avg_kernel_us"])
P
