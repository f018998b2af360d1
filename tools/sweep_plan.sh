#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for R in 1250 10000; do for B in 16 32 64 128; do
  echo -n "reads $R GKLHIP_PLAN_BLOCKS=$B: "; GKLHIP_PLAN_BLOCKS=$B GKLHIP_TIMING=1 python tools/quick_bench.py --reads $R --steps 30 2>&1 | grep phases | tail -1
done; done
