#!/bin/bash
# round 5 run 3: EXACT build tests (all cases), bench line with the step-level parity, SQ counters (db copied back)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/exact_report.jsonl
timeout 1500 python -m pytest tests/test_exact_mode_gpu.py -q --timeout 1200 > gpurun_out/r05_3_exact.log 2>&1; echo "exact rc=$?" >> gpurun_out/r05_3_exact.log
tail -40 gpurun_out/r05_3_exact.log | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline 2> gpurun_out/r05_3_bench.err | tail -1 > gpurun_out/r05_3_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_3_bench.json'))
print(d['value'], d['ms_per_step'], d.get('parity_ok'))
print(json.dumps(d.get('parity_gan_steps'), indent=0)[:2500])
PY
tail -5 gpurun_out/r05_3_bench.err
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d /tmp/pmc_sq -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_conv_r05.py > $GRAFT_REPO_ROOT/gpurun_out/r05_3_pmc.log 2>&1
ls -la /tmp/pmc_sq/ >> $GRAFT_REPO_ROOT/gpurun_out/r05_3_pmc.log
python $GRAFT_REPO_ROOT/scripts/pmc_sq_mfma.py /tmp/pmc_sq/pmc_results.db > $GRAFT_REPO_ROOT/gpurun_out/r05_pmc_sq_conv.txt 2>> $GRAFT_REPO_ROOT/gpurun_out/r05_3_pmc.log
sz=$(stat -c %s /tmp/pmc_sq/pmc_results.db); if [ "$sz" -lt 30000000 ]; then cp /tmp/pmc_sq/pmc_results.db $GRAFT_REPO_ROOT/gpurun_out/r05_pmc_sq.db; fi
tail -3 $GRAFT_REPO_ROOT/gpurun_out/r05_3_pmc.log; head -20 $GRAFT_REPO_ROOT/gpurun_out/r05_pmc_sq_conv.txt | cut -c1-200
