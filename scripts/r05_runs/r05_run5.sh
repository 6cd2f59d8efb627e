#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/exact_report.jsonl
timeout 600 python scripts/exact_grad_probe.py > gpurun_out/r05_5_probe.txt 2>&1
sort -k3 -g -r gpurun_out/r05_5_probe.txt | head -8
timeout 1500 python -m pytest tests/test_exact_mode_gpu.py -q --timeout 1200 > gpurun_out/r05_5_exact.log 2>&1; echo "exact rc=$?" >> gpurun_out/r05_5_exact.log
tail -12 gpurun_out/r05_5_exact.log | cut -c1-250
timeout 900 python -m pytest tests/test_gan_elem_gpu.py tests/test_gan_modules.py tests/test_gan_io_gpu.py -m gpu -x -q > gpurun_out/r05_5_gan.log 2>&1; tail -3 gpurun_out/r05_5_gan.log
