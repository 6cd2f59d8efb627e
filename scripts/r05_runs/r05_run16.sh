#!/bin/bash
# round 5 run 16: the final tree -- whole GPU suite, determinism soaks, then scripts/make_profile.sh r05_v4 (bench line, rocprofv3 kernel
# stats of the same command, FETCH / WRITE PMC passes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r05_16_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r05_16_all.log
tail -4 gpurun_out/r05_16_all.log | cut -c1-300
( timeout 400 python scripts/soak_determinism.py 8 64 256; timeout 400 python scripts/soak_determinism.py 6 16 512 3 ) > gpurun_out/r05_16_soak.txt 2>&1
grep -c "SOAK OK" gpurun_out/r05_16_soak.txt
timeout 1500 bash scripts/make_profile.sh r05_v4 --steps 20 --warmup 5
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_v4_bench.json'))
print(d['value'], d['ms_per_step'], d['parity_ok'], d['roofline']['frac'], d['roofline']['all_conv_tflops'], d['roofline']['executed'])
PY
