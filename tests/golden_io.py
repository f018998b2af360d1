"""Readers for the committed golden fixtures under tests/golden/."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def normalize(qual_str: str, floor: int) -> bytes:
    """ASCII-33 and clamp, as the reference test does
    (src/test/java/com/intel/gkl/pairhmm/PairHmmUnitTest.java:211-214,309-319)."""
    a = np.frombuffer(qual_str.encode("ascii"), dtype=np.uint8).astype(np.int32) - 33
    return np.maximum(a, floor).astype(np.uint8).tobytes()


def load_testdata():
    """The reference's own golden file (copied data, 104 single-pair cases):
    hap, read, readQual, insQual, delQual, gcp, expected log10 likelihood."""
    cases = []
    with open(os.path.join(GOLDEN, "pairhmm-testdata.txt")) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            hap, read, q, i, d, c, exp = line.split()
            cases.append(dict(hap=hap.encode(), read=read.encode(), q=normalize(q, 6), i=normalize(i, 0),
                              d=normalize(d, 0), c=normalize(c, 0), expected=float(exp)))
    return cases


def load_ref_vectors():
    """Vectors generated from the reference's own objects by tests/golden/make_fixtures.py."""
    with open(os.path.join(GOLDEN, "ref_vectors.json")) as f:
        return json.load(f)


def batch_from_vector(v):
    from gkl_amd.batch import FlatBatch
    h = lambda s: np.frombuffer(bytes.fromhex(s), dtype=np.uint8)  # noqa: E731
    return FlatBatch(v["n_reads"], v["n_haps"], np.array(v["read_off"], np.int64),
                     np.array(v["hap_off"], np.int64), h(v["read_bases"]), h(v["read_quals"]),
                     h(v["ins_gop"]), h(v["del_gop"]), h(v["gcp"]), h(v["hap_bases"]))


def load_pdhmm_file(name):
    """A reference PDHMM fixture (tab separated: hap, [pd bytes], read, 4 x fastq quals, expected;
    parsed like IntelPDHMMUnitTest.java:161-257: quals are fastq-33)."""
    from gkl_amd.pdhmm_batch import PdhmmBatch
    pairs, exp = [], []
    import gzip
    opener = gzip.open if name.endswith(".gz") else open   # (the 3.8 MB file of 1412 vectors is committed gzipped)
    with opener(os.path.join(GOLDEN, name), "rt", encoding="utf-8") as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            c = line.rstrip("\n").split("\t")
            pd = np.array([int(x) for x in c[1][1:-1].split(",")], dtype=np.int8)
            q = lambda s: (np.frombuffer(s.encode("utf-8"), dtype=np.uint8).astype(np.int16) - 33).astype(np.int8)  # noqa: E731
            pairs.append((c[0].encode(), pd, c[2].encode(), q(c[3]), q(c[4]), q(c[5]), q(c[6])))
            exp.append(float(c[7]))
    return PdhmmBatch.from_pairs(pairs), np.array(exp)


def load_pdhmm_holders_file(name="pdhmm_new.txt"):
    """The reference's reads-x-haplotypes PDHMM fixture (three '#' sections: reads with fastq-33 quals,
    haplotypes with their PD byte lists, one expected log10 per (read, haplotype), read-major;
    parsed like IntelPDHMMUnitTest.java:446-524).  Returns (reads, haps, expected[n_reads*n_haps])."""
    reads, haps, exp = [], [], []
    section = 0
    q = lambda s: (np.frombuffer(s.encode("utf-8"), dtype=np.uint8).astype(np.int16) - 33).astype(np.int8)  # noqa: E731
    with open(os.path.join(GOLDEN, name), encoding="utf-8") as f:
        for line in f:
            if line.startswith("#"):
                section += 1
                continue
            c = line.rstrip("\n").split("\t")
            if not c[0]:
                continue
            if section == 1:
                reads.append((np.frombuffer(c[0].encode(), dtype=np.int8), q(c[1]), q(c[2]), q(c[3]), q(c[4])))
            elif section == 2:
                haps.append((np.frombuffer(c[0].encode(), dtype=np.int8),
                             np.array([int(x) for x in c[1][1:-1].split(",")], dtype=np.int8)))
            else:
                exp.append(float(c[0]))
    return reads, haps, np.array(exp)


def load_pdhmm_tail_vectors():
    """tests/golden/pdhmm_tail_vectors.json (reference-generated, make_pdhmm_tail_fixtures.py): list of
    (PdhmmBatch, expected AVX-512-engine doubles, expected AVX2-engine doubles)."""
    import json
    from gkl_amd.pdhmm_batch import PdhmmBatch
    doc = json.load(open(os.path.join(GOLDEN, "pdhmm_tail_vectors.json")))
    out = []
    for v in doc["vectors"]:
        arr = lambda k: np.frombuffer(bytes.fromhex(v[k]), dtype=np.int8).copy()  # noqa: E731
        b = PdhmmBatch(v["batch"], v["max_hap_len"], v["max_read_len"], arr("hap_bases"), arr("hap_pdbases"), arr("read_bases"),
                       arr("read_qual"), arr("read_ins_qual"), arr("read_del_qual"), arr("gcp"),
                       np.asarray(v["hap_lengths"], np.int64), np.asarray(v["read_lengths"], np.int64))
        bits = lambda k: np.array([int(x, 16) for x in v[k]], dtype=np.uint64).view(np.float64)  # noqa: E731
        out.append((b, bits("avx512_bits"), bits("avx2_bits")))
    return out
