#!/bin/bash
# first GPU call of round 4: the whole GPU suite, then the bench line of this tree and of the round-3 tree on the SAME box
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/r04_run1_pytest.txt
cat $OUT/r04_run1_pytest.txt
timeout 400 python bench.py 2> $OUT/r04_v1_bench.err | tail -1 > $OUT/r04_v1_bench.json
(cd build_r03tree && timeout 400 python bench.py --no-cpu-baseline 2> $OUT/r04_base_bench.err | tail -1 > $OUT/r04_base_bench.json)
timeout 400 python bench.py --no-cpu-baseline 2> $OUT/r04_v1b_bench.err | tail -1 > $OUT/r04_v1b_bench.json
M355_DETERMINISTIC=1 timeout 400 python bench.py --no-cpu-baseline 2> $OUT/r04_v1det_bench.err | tail -1 > $OUT/r04_v1det_bench.json
for f in r04_v1 r04_base r04_v1b r04_v1det; do python - <<P
import json
try:
    j=json.load(open("$OUT/${f}_bench.json"))
    print("$f", j["value"], j["ms_per_step"], j["roofline"]["all_conv_tflops"], j.get("parity_ok"), j.get("gan_ms_per_cycle"))
except Exception as e:
    print("$f", "FAILED", e)
P
done
