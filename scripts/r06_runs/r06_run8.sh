#!/bin/bash
# round 6 run 8: the two new GPU tests (captured batch-16 cycle vs the CPU-oracle loop; the exchange inside a hipGraph), the batch-16 line
# with everything in it (profiles/r06_cfg3.json)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gan_modules.py tests/test_distributed_gpu.py -m gpu -q -x -k "batch16 or peer_mapped" > gpurun_out/r06_8_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06_8_tests.log
tail -5 gpurun_out/r06_8_tests.log | cut -c1-400
timeout 900 python bench.py --batch 16 --workload gan 2> gpurun_out/r06_8_cfg3.err | tail -1 > gpurun_out/r06_cfg3.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_cfg3.json'))
print('cfg3', round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('gan_launch'), d.get('parity_ok'), round(d['roofline']['frac'],3), round(d['roofline']['all_conv_tflops'],1), d['roofline'].get('traffic'))
PY
