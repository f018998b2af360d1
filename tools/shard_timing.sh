#!/bin/bash
# Dev tool: plan statistics and kernel split of an eighth-shard batch (GKLHIP_TIMING) for a few job-count targets.
cd ${GRAFT_REPO_ROOT:-.}
for F in 6144 12288 24576 49152; do
  echo "== GKLHIP_FB_WANTED_JOBS=$F"
  GKLHIP_FB_WANTED_JOBS=$F GKLHIP_TIMING=1 python tools/quick_bench.py --reads 1250 --steps 40 2>&1 | grep -o "main [0-9.]* ms.*fallback [0-9.]* ms\|[0-9]* affected reads, [0-9]* chunks, [0-9]* jobs" | tail -2
done
