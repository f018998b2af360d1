#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for R in 10000 1250; do for F in 3072 6144 12288 24576 49152; do
  echo -n "reads $R GKLHIP_FB_WANTED_JOBS=$F: "; GKLHIP_FB_WANTED_JOBS=$F GKLHIP_TIMING=1 python tools/quick_bench.py --reads $R --steps 20 2>&1 | grep -o "fallback [0-9.]* ms\|[0-9]* chunks, [0-9]* jobs" | tail -2 | tr '\n' ' '; echo
done; done
