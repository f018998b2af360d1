cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for v in "" _noasm _novmem _nolds _nodpp _novmemnoldsnodpp; do
  echo -n "variant '$v': "; GKL_AMD_PDHMM_LIB=gkl_amd/lib/libgklhip_pdhmm$v.so python tests/perf_pdhmm.py --reps 5 --fixture-x 32 2>&1 | grep "cross entry" | grep -o "kernel [0-9.]* ms"
done; done
