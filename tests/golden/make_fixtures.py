#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors.json from the REFERENCE's own kernel objects.

Run in the build container (where /root/reference exists and `make -C oracle ref`
has produced oracle/_ref/libgkl_ref_pairhmm.so):

    python tests/golden/make_fixtures.py

Each vector is a small flat batch (inputs as hex) plus what the reference produced
for it: raw fp32 / fp64 kernel sums (bit patterns), the per-pair fallback flag and
the final log10 doubles (bit patterns), for both code paths the reference ships:
engine 1 = the -mavx objects (separate mul/add), engine 2 = the AVX-512 objects
(gcc-11 contracts to FMA), in float-policy mode and in useDoublePrecision mode.
The vectors are data; no reference source is stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from gkl_amd.batch import FlatBatch, HaplotypeDataHolder, ReadDataHolder  # noqa: E402
from gkl_amd.synth import make_batch, random_batch  # noqa: E402
from oracle.oracle import Reference  # noqa: E402


def hexs(a):
    return np.ascontiguousarray(a).tobytes().hex()


def bits32(a):
    return [int(x) for x in np.asarray(a, np.float32).view(np.uint32)]


def bits64(a):
    return [int(x) for x in np.asarray(a, np.float64).view(np.uint64)]


def vector(name, b: FlatBatch, ref: Reference):
    v = dict(name=name, n_reads=b.n_reads, n_haps=b.n_haps, read_off=[int(x) for x in b.read_off],
             hap_off=[int(x) for x in b.hap_off], read_bases=hexs(b.read_bases),
             read_quals=hexs(b.read_quals), ins_gop=hexs(b.ins_gop), del_gop=hexs(b.del_gop),
             gcp=hexs(b.gcp), hap_bases=hexs(b.hap_bases), engines={})
    for eng in (1, 2):
        if eng == 2 and not ref.has_avx512():
            continue
        ref.set_engine(eng)
        out, r32, r64, u = ref.batch(b, use_double=False, want_raw=True)
        outd, _, r64d, _ = ref.batch(b, use_double=True, want_raw=True)
        v["engines"][str(eng)] = dict(out=bits64(out), raw32=bits32(r32), used64=[int(x) for x in u],
                                      raw64_fallback=bits64(np.where(u == 1, r64, 0.0)),
                                      out_double=bits64(outd), raw64_all=bits64(r64d))
    return v


def one_pair(read, q, i, d, c, hap):
    return FlatBatch.from_holders([ReadDataHolder(read, q, i, d, c)], [HaplotypeDataHolder(hap)])


def main():
    ref = Reference()
    rng = np.random.RandomState(424242)
    vs = []
    # simpleTest: PairHmmUnitTest.java:55-89 (quals are the raw bytes "++++" = 43)
    vs.append(vector("simpleTest", one_pair(b"ACGT", b"++++", b"++++", b"++++", b"++++", b"ACGT"), ref))
    vs.append(vector("acgt_small", random_batch(rng, 12, 5), ref))
    vs.append(vector("with_N", random_batch(rng, 10, 6, alphabet=b"ACGTN"), ref))
    vs.append(vector("odd_bytes_bit7_quals",
                     random_batch(rng, 10, 6, alphabet=b"ACGTNacgtnXRY*-", qual_range=(0, 255)), ref))
    # stripe-edge read lengths of the reference (8/16/64-row stripes) and of the HIP kernels
    for R in (1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 250, 254, 255, 256, 257):
        b = random_batch(rng, 1, 3, read_len=(R, R), hap_len=(max(1, R // 2), R + 40))
        vs.append(vector(f"R{R}", b, ref))
    for H in (1, 2, 31, 32, 33, 63, 64, 65, 100, 255, 256, 257, 500):
        b = random_batch(rng, 2, 1, read_len=(1, min(60, H + 5)), hap_len=(H, H))
        vs.append(vector(f"H{H}", b, ref))
    # unrelated reads: forces the fp32 -> fp64 fallback
    vs.append(vector("fallback_forced", random_batch(rng, 6, 4, read_len=(80, 160), hap_len=(100, 200),
                                                     related=False, qual_range=(20, 45)), ref))
    vs.append(vector("hc_small", make_batch("hc", 24, 6, seed=7), ref))
    vs.append(vector("region_small", make_batch("region", 12, 4, seed=8), ref))
    vs.append(vector("long_read", random_batch(rng, 2, 2, read_len=(700, 1100), hap_len=(800, 1200),
                                               qual_range=(10, 45)), ref))
    path = os.path.join(HERE, "ref_vectors.json")
    with open(path, "w") as f:
        json.dump(dict(generator="tests/golden/make_fixtures.py",
                       source="oracle/_ref (reference kernel objects, g++ 11.4, flags of PH/CMakeLists.txt)",
                       vectors=vs), f, separators=(",", ":"))
    n_pairs = sum(v["n_reads"] * v["n_haps"] for v in vs)
    print(f"wrote {path}: {len(vs)} vectors, {n_pairs} pairs, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
