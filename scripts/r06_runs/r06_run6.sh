#!/bin/bash
# round 6 run 6: the per-channel SyncBN exchange (two processes, one GPU), then the whole GPU suite on this tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_distributed_gpu.py -m gpu -q -x -s -k "peer_mapped" > gpurun_out/r06_6_ipc.log 2>&1; echo "rc=$?" >> gpurun_out/r06_6_ipc.log
grep -a "IpcAllReduce\|passed\|failed\|rc=\|Error" gpurun_out/r06_6_ipc.log | cut -c1-400 | tail -8
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r06_6_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r06_6_all.log
tail -4 gpurun_out/r06_6_all.log | cut -c1-300
