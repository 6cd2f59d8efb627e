#!/bin/bash
# round 5 run 2: EXACT build tests, whole GPU suite on the rebuilt product library, SQ counters of the conv families
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/exact_report.jsonl
timeout 1500 python -m pytest tests/test_exact_mode_gpu.py -q -x --timeout 1200 > gpurun_out/r05_2_exact.log 2>&1; echo "exact rc=$?" >> gpurun_out/r05_2_exact.log
tail -25 gpurun_out/r05_2_exact.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_exact_mode_gpu.py > gpurun_out/r05_2_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r05_2_all.log
tail -8 gpurun_out/r05_2_all.log
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d /tmp/pmc_sq -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_conv_r05.py > $GRAFT_REPO_ROOT/gpurun_out/r05_2_pmc.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_sq_mfma.py /tmp/pmc_sq/pmc_results.db > $GRAFT_REPO_ROOT/gpurun_out/r05_pmc_sq_conv.txt 2>> $GRAFT_REPO_ROOT/gpurun_out/r05_2_pmc.log
tail -5 $GRAFT_REPO_ROOT/gpurun_out/r05_2_pmc.log; head -12 $GRAFT_REPO_ROOT/gpurun_out/r05_pmc_sq_conv.txt
