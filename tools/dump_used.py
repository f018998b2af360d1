import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from gkl_amd import native
from gkl_amd.synth import make_batch
b = make_batch("hc")
with native.PairHmmContext() as c:
    c.compute(b)
    r32, r64, u = c.raw(b.n_pairs)
np.save("gpurun_out/used64_hc.npy", u.reshape(b.n_reads, b.n_haps))
print(u.mean())
