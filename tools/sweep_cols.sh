#!/bin/bash
# Dev tool: fp32 pass of the bench batch against the haplotype group size (columns per job)
for c in 1024 1536 2048 2560 3072 4096; do echo -n "target_cols $c: "; GKLHIP_TARGET_COLS=$c python tools/quick_bench.py --steps 30 | tail -1 | cut -c60-175; done
