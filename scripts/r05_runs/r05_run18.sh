#!/bin/bash
# round 5 run 18: whole GPU suite, determinism soaks and the bench line after the deterministic weight gradient's cell change
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r05_18_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r05_18_all.log
tail -4 gpurun_out/r05_18_all.log | cut -c1-300
( timeout 400 python scripts/soak_determinism.py 8 64 256; timeout 400 python scripts/soak_determinism.py 25 32 256; timeout 400 python scripts/soak_determinism.py 6 16 512 3 ) > gpurun_out/r05_18_soak.txt 2>&1
grep -c "SOAK OK" gpurun_out/r05_18_soak.txt
timeout 900 python bench.py 2> gpurun_out/r05_18_bench.err | tail -1 > gpurun_out/r05_v5_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_v5_bench.json'))
print(d['value'], d['ms_per_step'], d['parity_ok'], d['roofline']['frac'], d['roofline']['all_conv_tflops'], d['parity_gan_steps']['product']['ok'], d['parity_gan_steps']['exact']['ok'])
PY
