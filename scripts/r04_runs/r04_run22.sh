#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for b in 8 16; do
  for p in 1 2 3; do timeout 300 python scripts/r04_det_hash.py $b 256 6 1 2>&1 | grep "^B\|Error" | cut -c1-700; done
  timeout 300 python scripts/r04_det_hash.py $b 256 6 3 2>&1 | grep "^B\|Error" | cut -c1-700
done
