#!/bin/bash
# round 5 run 10: whole GPU suite on the final tree (plan-based binding), determinism soaks of the sub-pixel path in the four execution
# modes, bench line with the executed-MAC fields
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r05_10_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r05_10_all.log
tail -6 gpurun_out/r05_10_all.log | cut -c1-300
( timeout 400 python scripts/soak_determinism.py 8 64 256; timeout 400 python scripts/soak_determinism.py 25 32 256; timeout 400 python scripts/soak_determinism.py 6 16 512 3 ) > gpurun_out/r05_soak.txt 2>&1
grep -c "SOAK OK" gpurun_out/r05_soak.txt; grep -v "^/\|warn" gpurun_out/r05_soak.txt | cut -c1-200 | tail -16
timeout 900 python bench.py 2> gpurun_out/r05_10_bench.err | tail -1 > gpurun_out/r05_v2_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_v2_bench.json'))
print(d['value'], d['ms_per_step'], d['parity_ok'], d['roofline']['frac'], d['roofline']['all_conv_tflops'], d['roofline']['executed'])
PY
