"""Host-side mirror of the reference's plugin class ``com.intel.gkl.pdhmm.IntelPDHMM``
(reference src/main/java/com/intel/gkl/pdhmm/IntelPDHMM.java:42-206): same method names, argument
checks and exception messages, so tests read like IntelPDHMMUnitTest.java.  It talks to the same C ABI
as the JNI shim (include/gkl_hip_pdhmm.h) and never computes anything itself."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import native
from .batch import ReadDataHolder
from .errors import IllegalArgumentException, NullPointerException, OutOfMemoryError, RuntimeException
from .pdhmm_batch import PdhmmBatch


@dataclass
class PDHaplotypeDataHolder:
    """HaplotypeDataHolder of gatk-native-bindings' pdhmm package: bases + PD flag bytes
    (fields read by name in src/main/native/pdhmm/JavaData.h:172-173)."""
    haplotypeBases: Optional[bytes] = None
    haplotypePDBases: Optional[bytes] = None


@dataclass
class PDHMMNativeArguments:
    maxNumberOfThreads: int = 1
    avxLevel: str = "FASTEST_AVAILABLE"
    openMPSetting: str = "FASTEST_AVAILABLE"
    maxMemoryInMB: int = 512


def _check_array_size(array, expected, name):  # IntelPDHMM.java:133-145
    if array is None:
        raise NullPointerException(f"{name} must not be null.")
    if not hasattr(array, "__len__"):
        raise IllegalArgumentException(f"{name} is not an array.")
    if len(array) != expected:
        raise IllegalArgumentException(f"Array {name} has size {len(array)}, but expected size is {expected}.")


class IntelPDHMM:
    NATIVE_LIBRARY_NAME = "gkl_pdhmm"

    def __init__(self):
        self._ctx: Optional[native.PdhmmContext] = None
        self._max_memory_mb = 512

    def load(self, tempDir=None) -> bool:
        try:
            native.load_pdhmm_library()
        except RuntimeException:
            return False
        import torch
        return torch.cuda.is_available()

    def initialize(self, args: Optional[PDHMMNativeArguments]) -> None:
        if args is None:
            args = PDHMMNativeArguments()
        self._max_memory_mb = native.pdhmm_available_memory_mb(args.maxMemoryInMB)   # capped once, here (pdhmm-implementation.h:204-235)
        if self._ctx is not None:
            self._ctx.close()
        self._ctx = native.PdhmmContext()

    def computeLikelihoods(self, readDataArray: Sequence[ReadDataHolder],
                           haplotypeDataArray: Sequence[PDHaplotypeDataHolder], likelihoodArray) -> None:
        if readDataArray is None or haplotypeDataArray is None or likelihoodArray is None:
            raise NullPointerException("One or more input arrays are null. Please ensure readDataArray, "
                                       "haplotypeDataArray, and likelihoodArray are properly initialized.")
        if len(likelihoodArray) != len(readDataArray) * len(haplotypeDataArray):
            raise IllegalArgumentException("likelihoodArray length must be equal to readDataArray length * "
                                           "haplotypeDataArray length")
        if self._ctx is None:
            raise RuntimeException("computeLikelihoods before initialize")
        try:
            if len(readDataArray) == 0 or len(haplotypeDataArray) == 0:
                raise IllegalArgumentException("no pairs to process")
            one = b"\0"
            reads = PdhmmBatch.from_pairs([(one, one, r.readBases, r.readQuals, r.insertionGOP, r.deletionGOP,
                                            r.overallGCP) for r in readDataArray])
            haps = PdhmmBatch.from_pairs([(h.haplotypeBases, h.haplotypePDBases, one, one, one, one, one)
                                          for h in haplotypeDataArray])
            ref_batch = native.pdhmm_reference_batch_pairs(self._max_memory_mb, reads.max_read_len, haps.max_hap_len,
                                                           reads.batch * haps.batch)   # JavaData.h:86-101
            if ref_batch <= 0:
                raise IllegalArgumentException("Batch size is too small.")
            likelihoodArray[:] = self._ctx.compute_cross(reads, haps, ref_batch)  # read-major (JavaData.h:190)
        except OutOfMemoryError:
            raise OutOfMemoryError("Memory allocation failed")
        except IllegalArgumentException:
            raise IllegalArgumentException("Ran into invalid argument issue")

    def computePDHMM(self, hap_bases, hap_pdbases, read_bases, read_qual, read_ins_qual, read_del_qual, gcp,
                     hap_lengths, read_lengths, batchSize: int, maxHapLength: int, maxReadLength: int) -> np.ndarray:
        hap_n, read_n = maxHapLength * batchSize, maxReadLength * batchSize
        for arr, n, name in ((hap_bases, hap_n, "hap_bases"), (hap_pdbases, hap_n, "hap_pdbases"),
                             (read_bases, read_n, "read_bases"), (read_qual, read_n, "read_qual"),
                             (read_ins_qual, read_n, "read_ins_qual"), (read_del_qual, read_n, "read_del_qual"),
                             (gcp, read_n, "gcp"), (hap_lengths, batchSize, "hap_lengths"),
                             (read_lengths, batchSize, "read_lengths")):
            _check_array_size(arr, n, name)
        if batchSize <= 0:
            raise IllegalArgumentException("batchSize must be greater than 0.")
        if maxHapLength <= 0:
            raise IllegalArgumentException("maxHapLength must be greater than 0. Cannot perform PDHMM on empty sequence")
        if maxReadLength <= 0:
            raise IllegalArgumentException("maxReadLength must be greater than 0. Cannot perform PDHMM on empty sequence")
        if self._ctx is None:
            raise RuntimeException("computePDHMM before initialize")
        i8 = lambda a: np.ascontiguousarray(a, np.int8)  # noqa: E731
        b = PdhmmBatch(batchSize, maxHapLength, maxReadLength, i8(hap_bases), i8(hap_pdbases), i8(read_bases),
                       i8(read_qual), i8(read_ins_qual), i8(read_del_qual), i8(gcp),
                       np.ascontiguousarray(hap_lengths, np.int64), np.ascontiguousarray(read_lengths, np.int64))
        try:
            return self._ctx.compute(b)
        except IllegalArgumentException as e:
            raise IllegalArgumentException(f"IllegalArgument exception thrown from native pdhmm function call {e}")

    def done(self) -> None:
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None
