#!/bin/bash
# Dev tool: kernel-trace stats of the device-resident step (the plan kernels' durations)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ptr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptr -o t -- python $REPO/tools/quick_bench.py --steps 40 > /dev/null 2>&1
find /tmp/ptr -name "*kernel_stats.csv" -exec head -9 {} \; | cut -d, -f1-4 | cut -c1-110
