#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for b in 16 8; do
  timeout 300 python scripts/r04_poison_hash.py $b 256 3 2>&1 | grep "^B\|Error" | cut -c1-700
  POISON=00 timeout 300 python scripts/r04_poison_hash.py $b 256 3 2>&1 | grep "^B\|Error" | cut -c1-700
  POISON=71 timeout 300 python scripts/r04_poison_hash.py $b 256 3 2>&1 | grep "^B\|Error" | cut -c1-700
  POISON=3c timeout 300 python scripts/r04_poison_hash.py $b 256 3 2>&1 | grep "^B\|Error" | cut -c1-700
  POISON=ff timeout 300 python scripts/r04_poison_hash.py $b 256 3 2>&1 | grep "^B\|Error" | cut -c1-700
  timeout 300 python scripts/r04_poison_hash.py $b 256 3 2>&1 | grep "^B\|Error" | cut -c1-700
done
