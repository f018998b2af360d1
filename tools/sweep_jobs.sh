#!/bin/bash
# Dev tool: job-granularity sweeps (main-pass groups, fp64-pass job length) on the eighth-shard and the full batch.
cd ${GRAFT_REPO_ROOT:-.}
for R in 1250 10000; do
  for W in 4096 8192 16384 32768; do
    echo -n "reads $R GKLHIP_WANTED_JOBS=$W: "; GKLHIP_WANTED_JOBS=$W python tools/quick_bench.py --reads $R --steps 40 2>&1 | grep -o "main [0-9.]* ms.*groups [0-9]*"
  done
  for F in 1536 3072 6144 12288 24576; do
    echo -n "reads $R GKLHIP_FB_WANTED_JOBS=$F: "; GKLHIP_FB_WANTED_JOBS=$F python tools/quick_bench.py --reads $R --steps 40 2>&1 | grep -o "fallback [0-9.]* ms"
  done
done
