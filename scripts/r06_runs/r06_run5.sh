#!/bin/bash
# round 6 run 5: the SyncBN exchange over peer-mapped memory (two processes, one GPU), the RCCL-in-capture probe after the prefetch fix,
# bench: projection half on its own stream (A/B), the batch-16 configuration with automatic graph replay and its parity blocks
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_distributed_gpu.py -m gpu -q -x -s > gpurun_out/r06_5_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06_5_tests.log
grep -a "IpcAllReduce\|RCCL-in-graph\|passed\|failed\|xfail\|rc=" gpurun_out/r06_5_tests.log | cut -c1-400 | tail -8
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d.get('gan_ms_per_cycle'), d.get('proj_ms_per_step'), d['config'].get('gan_launch'), d.get('parity_ok'))"
}
for rep in 1 2; do
  one serial "X=1" ""
  one overlap_p "M355_BENCH_OVERLAP_P=1" ""
done 2>&1 | tee gpurun_out/r06_5_ab.txt
timeout 900 python bench.py --batch 16 --workload gan 2> gpurun_out/r06_5_cfg3.err | tail -1 > gpurun_out/r06_5_cfg3.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_5_cfg3.json'))
print('cfg3', round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('gan_launch'), d.get('parity_ok'), round(d['roofline']['frac'],3), round(d['roofline']['all_conv_tflops'],1))
print({k: (v.get('ok') if isinstance(v, dict) else v) for k, v in d.items() if k.startswith('parity')})
PY
tail -3 gpurun_out/r06_5_cfg3.err | cut -c1-300
