#!/bin/bash
# round 6 run 4: replay-only / eager-only busy-idle anatomy (windowed by launches), the distributed GPU tests with the generator's
# two-message gradient all-reduce, the bench's exact-build block with fresh Adam state in both builds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_distributed_gpu.py tests/test_gan_modules.py -m gpu -q -x > gpurun_out/r06_4_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06_4_tests.log
tail -3 gpurun_out/r06_4_tests.log | cut -c1-400
cd /tmp; export TMPDIR=/tmp
for cfg in "16 40" "64 20 --eager"; do
  tag=$(echo $cfg | tr ' ' '_' | tr -d '-')
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o t -- python $GRAFT_REPO_ROOT/scripts/graph_trace.py $cfg > $GRAFT_REPO_ROOT/gpurun_out/r06_4_trace_$tag.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/rocpd_gaps.py /tmp/prof_$tag/t_results.db 0.5 > $GRAFT_REPO_ROOT/gpurun_out/r06_4_gaps_$tag.txt 2>&1
  echo $tag; head -1 $GRAFT_REPO_ROOT/gpurun_out/r06_4_gaps_$tag.txt
done
cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py --no-cpu-baseline 2> gpurun_out/r06_4_bench.err | tail -1 > gpurun_out/r06_4_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_4_bench.json'))
print(round(d['value'],1), round(d['ms_per_step'],3), d.get('parity_ok'))
print(json.dumps(d.get('exact'))[:1500])
PY
