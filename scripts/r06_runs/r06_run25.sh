#!/bin/bash
# round 6 run 25: the mesh discriminator's layers against the batch, with and without the 64 x 64 small-tile choice
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for e in "A=1" "M355_NO_SMALL_TILE=1"; do echo "$e"; env $e timeout 300 python scripts/probes/mesh_d_small.py 2>/dev/null; done 2>&1 | tee gpurun_out/r06_25_mesh_d.txt
