#!/bin/bash
# round 6 run 3: the bench line's new blocks (exact-build cycle, timed-batch D parity, CPU baselines at batch 16 / physical cores) -- how
# long does the default run take now?; eager vs hipGraph replay at batch 64 (same box); replay-only / eager-only kernel traces with
# busy / idle anatomy at batch 16 (graph) and batch 64 (eager, graph)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 1200 python bench.py 2> gpurun_out/r06_3_bench.err | tail -1 > gpurun_out/r06_3_bench.json ) 2> gpurun_out/r06_3_bench_time.txt
tail -3 gpurun_out/r06_3_bench_time.txt; tail -5 gpurun_out/r06_3_bench.err | cut -c1-400
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_3_bench.json'))
print(round(d['value'],1), round(d['ms_per_step'],3), d.get('parity_ok'))
for k in ('exact','parity_gan_timed_batch','cpu_baseline','cpu_baseline_gan'):
    print(k, json.dumps(d.get(k))[:900])
PY
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d.get('gan_ms_per_cycle'), d['config'].get('gan_launch'), d.get('parity_ok'))"
}
for rep in 1 2; do
  one b64_eager "X=1" ""
  one b64_graph "X=1" "--graph"
done 2>&1 | tee gpurun_out/r06_3_ab.txt
cd /tmp; export TMPDIR=/tmp
for cfg in "16 40" "64 20" "64 20 --eager"; do
  tag=$(echo $cfg | tr ' ' '_' | tr -d '-')
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o t -- python $GRAFT_REPO_ROOT/scripts/graph_trace.py $cfg > $GRAFT_REPO_ROOT/gpurun_out/r06_3_trace_$tag.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/rocpd_gaps.py /tmp/prof_$tag/t_results.db 0.5 > $GRAFT_REPO_ROOT/gpurun_out/r06_3_gaps_$tag.txt 2>&1
  echo $tag; head -1 $GRAFT_REPO_ROOT/gpurun_out/r06_3_gaps_$tag.txt
done
