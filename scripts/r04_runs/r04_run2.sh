#!/bin/bash
# GPU call 2 of round 4: full suite; bench A/B on one box: streams off / on, wgrad_c8 v1 / planar
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/r04_run2_pytest.txt
cat $OUT/r04_run2_pytest.txt
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline 2> $OUT/${tag}_bench.err | tail -1 > $OUT/${tag}_bench.json
  python - <<P
import json
try:
    j=json.load(open("$OUT/${tag}_bench.json"))
    k=j["kernels_ms_per_step"]
    print("$tag", round(j["value"],1), round(j["ms_per_step"],3), round(j["roofline"]["all_conv_tflops"],1), j.get("parity_ok"), round(j.get("gan_ms_per_cycle"),3), "wgrad_c8", round(k.get("k_wgrad_c8",0),3), "sn", round(k.get("sn_power_iter",0),3))
except Exception as e:
    print("$tag", "FAILED", e)
P
}
run r04_v2_nostream M355_STREAMS=0
run r04_v2 M355_STREAMS=1
run r04_v2_c8v1 M355_STREAMS=1 M355_WGC8_V1=1
run r04_v2_nostream_b M355_STREAMS=0
run r04_v2_b M355_STREAMS=1
