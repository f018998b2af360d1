"""Host-side data holders for the PairHMM hot path.

``ReadDataHolder`` / ``HaplotypeDataHolder`` / ``PairHMMNativeArguments`` mirror the
classes of org.broadinstitute:gatk-native-bindings:1.1.0 that the reference's JNI
layer reads reflectively by field name (reference
src/main/native/pairhmm/JavaData.h:55-62): ``readBases``, ``readQuals``,
``insertionGOP``, ``deletionGOP``, ``overallGCP`` and ``haplotypeBases``, all raw
``byte[]`` (Phred values, no +33 offset).

``FlatBatch`` is the flat structure-of-arrays form the C ABI takes
(include/gkl_hip_pairhmm.h): what JavaData::getData (JavaData.h:65-111) builds as
a vector of 56-byte ``testcase`` structs becomes two offset arrays plus six byte
arrays, and the r-major output order ``out[r*n_haps + h]`` (JavaData.h:94-105,
IntelPairHmm.cc:167) is kept.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np


def _bytes_to_u8(b) -> np.ndarray:
    if b is None:
        raise TypeError("byte array is None")
    if isinstance(b, str):
        b = b.encode("ascii")
    if isinstance(b, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(b), dtype=np.uint8)
    return np.ascontiguousarray(b).astype(np.uint8, copy=False)


@dataclass
class ReadDataHolder:
    readBases: Optional[bytes] = None
    readQuals: Optional[bytes] = None
    insertionGOP: Optional[bytes] = None
    deletionGOP: Optional[bytes] = None
    overallGCP: Optional[bytes] = None


@dataclass
class HaplotypeDataHolder:
    haplotypeBases: Optional[bytes] = None


@dataclass
class PairHMMNativeArguments:
    useDoublePrecision: bool = False
    maxNumberOfThreads: int = 1


@dataclass
class FlatBatch:
    n_reads: int
    n_haps: int
    read_off: np.ndarray   # int64[n_reads+1]
    hap_off: np.ndarray    # int64[n_haps+1]
    read_bases: np.ndarray  # uint8[read_off[-1]]
    read_quals: np.ndarray
    ins_gop: np.ndarray
    del_gop: np.ndarray
    gcp: np.ndarray
    hap_bases: np.ndarray  # uint8[hap_off[-1]]
    meta: dict = field(default_factory=dict)

    @property
    def n_pairs(self) -> int:
        return self.n_reads * self.n_haps

    @property
    def read_lens(self) -> np.ndarray:
        return np.diff(self.read_off)

    @property
    def hap_lens(self) -> np.ndarray:
        return np.diff(self.hap_off)

    @property
    def cells(self) -> int:
        """Sum over pairs of rslen*haplen (same definition as JavaData.h:108)."""
        return int(self.read_lens.sum()) * int(self.hap_lens.sum())

    def read_slice(self, lo: int, hi: int) -> "FlatBatch":
        """Reads [lo, hi) against all haplotypes (the multi-GPU shard unit)."""
        a, b = int(self.read_off[lo]), int(self.read_off[hi])
        return FlatBatch(hi - lo, self.n_haps, (self.read_off[lo:hi + 1] - a).astype(np.int64),
                         self.hap_off, self.read_bases[a:b], self.read_quals[a:b],
                         self.ins_gop[a:b], self.del_gop[a:b], self.gcp[a:b], self.hap_bases,
                         dict(self.meta))

    @staticmethod
    def from_holders(reads: Sequence[ReadDataHolder],
                     haps: Sequence[HaplotypeDataHolder]) -> "FlatBatch":
        """Flatten holder arrays like JavaData::getData does.

        As in JavaData.h:86-91 the read length is the length of ``readBases``;
        the other four arrays must be at least that long (the reference would read
        past shorter ones; this boundary raises IllegalArgumentException instead).
        """
        from .errors import IllegalArgumentException

        rb, rq, ri, rd, rc, lens = [], [], [], [], [], []
        for k, r in enumerate(reads):
            if r is None:
                raise IllegalArgumentException(f"read {k} is null")
            b = _bytes_to_u8(r.readBases)
            n = b.size
            others = [_bytes_to_u8(x) for x in (r.readQuals, r.insertionGOP, r.deletionGOP, r.overallGCP)]
            if any(o.size < n for o in others):
                raise IllegalArgumentException(f"read {k}: quality arrays shorter than readBases")
            rb.append(b)
            for dst, o in zip((rq, ri, rd, rc), others):
                dst.append(o[:n])
            lens.append(n)
        hb, hl = [], []
        for k, h in enumerate(haps):
            if h is None:
                raise IllegalArgumentException(f"haplotype {k} is null")
            b = _bytes_to_u8(h.haplotypeBases)
            hb.append(b)
            hl.append(b.size)

        def cat(xs):
            return np.concatenate(xs).astype(np.uint8) if xs else np.zeros(0, np.uint8)

        read_off = np.zeros(len(lens) + 1, np.int64)
        read_off[1:] = np.cumsum(lens)
        hap_off = np.zeros(len(hl) + 1, np.int64)
        hap_off[1:] = np.cumsum(hl)
        return FlatBatch(len(lens), len(hl), read_off, hap_off, cat(rb), cat(rq), cat(ri), cat(rd),
                         cat(rc), cat(hb))

    def to_holders(self):
        reads = []
        for r in range(self.n_reads):
            a, b = int(self.read_off[r]), int(self.read_off[r + 1])
            reads.append(ReadDataHolder(self.read_bases[a:b].tobytes(), self.read_quals[a:b].tobytes(),
                                        self.ins_gop[a:b].tobytes(), self.del_gop[a:b].tobytes(),
                                        self.gcp[a:b].tobytes()))
        haps = []
        for h in range(self.n_haps):
            a, b = int(self.hap_off[h]), int(self.hap_off[h + 1])
            haps.append(HaplotypeDataHolder(self.hap_bases[a:b].tobytes()))
        return reads, haps
