#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gan_elem_gpu.py tests/test_gan_modules.py -m gpu -q -x 2>&1 | tail -4
python scripts/affine_rate.py 2>/dev/null > $OUT/r04_affine_rate_new.txt; cat $OUT/r04_affine_rate_new.txt
(cd build_r03tree && python scripts/affine_rate.py 2>/dev/null > $OUT/r04_affine_rate_r03.txt; cat $OUT/r04_affine_rate_r03.txt)
timeout 400 python bench.py --no-cpu-baseline 2> /dev/null | tail -1 > $OUT/r04_v6_bench.json
python - <<P
import json
j=json.load(open("$OUT/r04_v6_bench.json")); k=j["kernels_ms_per_step"]
print("r04_v6", round(j["value"],1), round(j["ms_per_step"],3), round(j["roofline"]["all_conv_tflops"],1), j.get("parity_ok"), round(j["gan_ms_per_cycle"],3))
print({a:round(b,3) for a,b in k.items() if 'affine' in a})
P
