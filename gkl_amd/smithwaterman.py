"""Host-side mirror of the reference's plugin class ``com.intel.gkl.smithwaterman.IntelSmithWaterman``
(reference src/main/java/com/intel/gkl/smithwaterman/IntelSmithWaterman.java:44-191) and of the
gatk-native-bindings types it uses (SWParameters, SWOverhangStrategy, SWNativeAlignerResult): same method
names, argument checks and exception messages, so tests read like SmithWatermanUnitTest.java.  It talks to
the same C ABI as the JNI shim (include/gkl_hip_sw.h) and never computes anything itself."""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Optional

from . import native
from .errors import IllegalArgumentException, NullPointerException, OutOfMemoryError, RuntimeException


class SWOverhangStrategy(enum.Enum):
    SOFTCLIP = 9        # IntelSmithWaterman.getStrategy, IntelSmithWaterman.java:160-177
    INDEL = 10
    LEADING_INDEL = 11
    IGNORE = 12


@dataclass(frozen=True)
class SWParameters:
    matchValue: int
    mismatchPenalty: int
    gapOpenPenalty: int
    gapExtendPenalty: int

    def getMatchValue(self):
        return self.matchValue

    def getMismatchPenalty(self):
        return self.mismatchPenalty

    def getGapOpenPenalty(self):
        return self.gapOpenPenalty

    def getGapExtendPenalty(self):
        return self.gapExtendPenalty


@dataclass(frozen=True)
class SWNativeAlignerResult:
    cigar: str
    alignment_offset: int


class IntelSmithWaterman:
    NATIVE_LIBRARY_NAME = "gkl_smithwaterman"
    MAX_SW_SEQUENCE_LENGTH = 32 * 1024 - 1   # IntelSmithWaterman.java:53
    MAXIMUM_SW_MATCH_VALUE = 64 * 1024       # :55

    def __init__(self):
        self._ctx: Optional[native.SwContext] = None

    def load(self, tempDir=None) -> bool:
        """True when the library and a gfx950 device are usable (the reference gates on AVX2 here, :77-112);
        initNative is part of load in the reference."""
        try:
            if self._ctx is None:
                self._ctx = native.SwContext()
            return True
        except RuntimeException:
            return False

    def align(self, refArray, altArray, parameters: SWParameters, overhangStrategy: SWOverhangStrategy) -> SWNativeAlignerResult:
        if refArray is None:
            raise NullPointerException("Reference data array is null.")
        if altArray is None:
            raise NullPointerException("Alternate data array is null.")
        if parameters is None:
            raise NullPointerException("Parameter structure is null.")
        if overhangStrategy is None:
            raise NullPointerException("OverhangStrategy is null.")
        if len(refArray) <= 0 or len(altArray) <= 0:
            raise IllegalArgumentException("Cannot align empty sequences")
        strategy = self.getStrategy(overhangStrategy)
        cigar_len = 2 * max(len(refArray), len(altArray))
        if len(refArray) > self.MAX_SW_SEQUENCE_LENGTH or len(altArray) > self.MAX_SW_SEQUENCE_LENGTH:
            raise IllegalArgumentException(f"Sequences exceed maximum length of {self.MAX_SW_SEQUENCE_LENGTH} bytes")
        if parameters.getMatchValue() > self.MAXIMUM_SW_MATCH_VALUE:
            raise IllegalArgumentException(f"Match value parameter exceed maximum value of {self.MAXIMUM_SW_MATCH_VALUE}")
        if cigar_len <= 0 or strategy < 9 or strategy > 12:
            raise IllegalArgumentException("Strategy is invalid.")
        if self._ctx is None:
            raise RuntimeException("align before load")
        try:
            cigar, _, offset = self._ctx.align(bytes(refArray), bytes(altArray),
                                               (parameters.getMatchValue(), parameters.getMismatchPenalty(),
                                                parameters.getGapOpenPenalty(), parameters.getGapExtendPenalty()),
                                               strategy, cigar_len)
        except OutOfMemoryError:
            raise OutOfMemoryError("Memory allocation failed")
        except IllegalArgumentException:
            raise IllegalArgumentException("Ran into invalid argument issue")
        return SWNativeAlignerResult(cigar.decode("utf-8").strip(), offset)

    @staticmethod
    def getStrategy(strategy: SWOverhangStrategy) -> int:
        return int(strategy.value) if isinstance(strategy, SWOverhangStrategy) else 0

    def close(self) -> None:
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None
