"""Padded 1:1 batch of (read, partially determined haplotype) pairs: the layout
``IntelPDHMM.computePDHMM`` takes (reference src/main/java/com/intel/gkl/pdhmm/IntelPDHMM.java:147-186):
``hap_bases`` / ``hap_pdbases`` are ``[batch][max_hap_len]`` bytes, the five read arrays
``[batch][max_read_len]``, plus per-pair lengths.  Quals are Phred bytes (no +33)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class PdhmmBatch:
    batch: int
    max_hap_len: int
    max_read_len: int
    hap_bases: np.ndarray      # int8 [batch*max_hap_len]
    hap_pdbases: np.ndarray
    read_bases: np.ndarray     # int8 [batch*max_read_len]
    read_qual: np.ndarray
    read_ins_qual: np.ndarray
    read_del_qual: np.ndarray
    gcp: np.ndarray
    hap_lengths: np.ndarray    # int64 [batch]
    read_lengths: np.ndarray

    @property
    def cells(self) -> int:
        return int((self.hap_lengths * self.read_lengths).sum())

    def subset(self, idx) -> "PdhmmBatch":
        idx = np.asarray(idx)
        h = lambda a: a.reshape(self.batch, self.max_hap_len)[idx].reshape(-1)  # noqa: E731
        r = lambda a: a.reshape(self.batch, self.max_read_len)[idx].reshape(-1)  # noqa: E731
        return PdhmmBatch(len(idx), self.max_hap_len, self.max_read_len, h(self.hap_bases), h(self.hap_pdbases),
                          r(self.read_bases), r(self.read_qual), r(self.read_ins_qual), r(self.read_del_qual),
                          r(self.gcp), self.hap_lengths[idx], self.read_lengths[idx])

    def pairs(self, idx=None):
        """The pairs (trimmed to their lengths) in from_pairs' form."""
        idx = range(self.batch) if idx is None else idx
        hb, hp = (a.reshape(self.batch, self.max_hap_len) for a in (self.hap_bases, self.hap_pdbases))
        rs = [a.reshape(self.batch, self.max_read_len) for a in (self.read_bases, self.read_qual, self.read_ins_qual,
                                                                  self.read_del_qual, self.gcp)]
        return [(hb[i, :self.hap_lengths[i]], hp[i, :self.hap_lengths[i]], *[a[i, :self.read_lengths[i]] for a in rs]) for i in idx]

    @staticmethod
    def from_pairs(pairs) -> "PdhmmBatch":
        """pairs: iterable of (hap_bases, hap_pdbases, read_bases, read_qual, ins, del, gcp) byte strings/arrays."""
        pairs = [tuple(np.frombuffer(bytes(x), dtype=np.int8) if isinstance(x, (bytes, bytearray))
                       else np.asarray(x, dtype=np.int8) for x in p) for p in pairs]
        n = len(pairs)
        mh = max((p[0].size for p in pairs), default=0)
        mr = max((p[2].size for p in pairs), default=0)
        hb = np.zeros((n, mh), np.int8)
        hp = np.zeros((n, mh), np.int8)
        rs = [np.zeros((n, mr), np.int8) for _ in range(5)]
        for k, p in enumerate(pairs):
            hb[k, :p[0].size] = p[0]
            hp[k, :p[1].size] = p[1]
            for a, src in zip(rs, p[2:]):
                a[k, :src.size] = src
        return PdhmmBatch(n, mh, mr, hb.reshape(-1), hp.reshape(-1), *[a.reshape(-1) for a in rs],
                          np.array([p[0].size for p in pairs], np.int64), np.array([p[2].size for p in pairs], np.int64))
