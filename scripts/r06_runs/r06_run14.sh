#!/bin/bash
# round 6 run 14: in-kernel stamps (lib/libstamp.so, -DM355_DBG_STAMP) of the class kernels incl. D.conv2's PAIR dgrad, with the tile epilogue
# split into convert + stores | init_acc | next prologue
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
M355_LIB=libstamp.so timeout 600 python scripts/stamp_halo.py 128 > gpurun_out/r06_stamp_halo.txt 2>&1
grep -a -v "amdgpu.ids" gpurun_out/r06_stamp_halo.txt | cut -c1-220 | head -150
