#!/bin/bash
# Dev tool: PDHMM cross entry point (fixture x32 / x4), a variant library (GKL_AMD_PDHMM_LIB) against the built one, alternating on one box.
cd ${GRAFT_REPO_ROOT:-.}
V=${1:-gkl_amd/lib/libgklhip_pdhmm_head.so}
for X in 32 4; do for REP in 1 2 3; do
  echo -n "x$X variant: "; GKL_AMD_PDHMM_LIB=$V python tests/perf_pdhmm.py --reps 5 --fixture-x $X 2>&1 | grep "cross entry" | grep -o "kernel [0-9.]* ms = [0-9.]* GCUPS"
  echo -n "x$X built:   "; python tests/perf_pdhmm.py --reps 5 --fixture-x $X 2>&1 | grep "cross entry" | grep -o "kernel [0-9.]* ms = [0-9.]* GCUPS"
done; done
