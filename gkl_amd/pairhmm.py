"""Host-side mirror of the reference's plugin class ``com.intel.gkl.pairhmm.IntelPairHmm``
(reference src/main/java/com/intel/gkl/pairhmm/IntelPairHmm.java:41-167), so tests of this
path read like the reference's own TestNG tests (PairHmmUnitTest.java).

Same method names, argument meaning and error behaviour:

* ``load(tempDir)``       -> True iff the native library is present and usable
                             (IntelPairHmm.java:65-78; there the gate is AVX support, here
                             it is a visible gfx950 device);
* ``initialize(args)``    -> ``initNative`` (:85-119); ``None`` means
                             ``useDoublePrecision=False, maxNumberOfThreads=1`` (:86-90);
* ``computeLikelihoods``  -> NullPointerException on a null top-level argument (:134-136),
                             OutOfMemoryError / IllegalArgumentException re-raised with the
                             reference's fixed messages (:139-145);
* ``done()``              -> ``doneNative`` (:153-155).

The Java class talks to libgkl_pairhmm.so through JNI; this mirror talks to the same C ABI
through ctypes.  It never computes anything itself.
"""
from __future__ import annotations

import logging
from typing import Optional, Sequence

import numpy as np

from . import native
from .batch import FlatBatch, HaplotypeDataHolder, PairHMMNativeArguments, ReadDataHolder
from .errors import (IllegalArgumentException, NullPointerException, OutOfMemoryError,
                     RuntimeException)

logger = logging.getLogger("gkl_amd.pairhmm")


class IntelPairHmm:
    NATIVE_LIBRARY_NAME = "gkl_pairhmm"

    def __init__(self):
        self.nativeLibraryName = self.NATIVE_LIBRARY_NAME
        self.useOmp = False
        self._ctx: Optional[native.PairHmmContext] = None
        self._loaded = False

    def setNativeLibraryName(self, name: str) -> None:
        self.nativeLibraryName = name

    def load(self, tempDir=None) -> bool:
        """True if the native library is supported here and loaded, False otherwise."""
        if self.nativeLibraryName not in ("gkl_pairhmm", "gkl_pairhmm_omp"):
            # NativeLibraryLoader's whitelist (NativeLibraryLoader.java:61-76): unknown names
            # such as "gkl_pairhmm_shacc" do not load (PairHmmUnitTest.java:91-98).
            return False
        try:
            lib = native.load_library()
        except RuntimeException as e:
            logger.warning("GKL-HIP PairHMM library not loaded: %s", e)
            return False
        if lib.gklhip_device_count() <= 0:
            return False
        self._loaded = True
        return True

    def initialize(self, args: Optional[PairHMMNativeArguments]) -> None:
        if args is None:
            args = PairHMMNativeArguments(useDoublePrecision=False, maxNumberOfThreads=1)
        if self._ctx is not None:
            self._ctx.close()
        self._ctx = native.PairHmmContext(use_double=args.useDoublePrecision,
                                          max_threads=args.maxNumberOfThreads,
                                          finalize=native.FINALIZE_REFERENCE_HOST)
        if not self.useOmp and args.maxNumberOfThreads != 1:
            logger.warning("Ignoring request for %d threads; not using OpenMP implementation",
                           args.maxNumberOfThreads)

    def computeLikelihoods(self, readDataArray: Sequence[ReadDataHolder],
                           haplotypeDataArray: Sequence[HaplotypeDataHolder],
                           likelihoodArray: np.ndarray) -> None:
        if readDataArray is None or haplotypeDataArray is None or likelihoodArray is None:
            raise NullPointerException("Input is null")
        if self._ctx is None:
            raise RuntimeException("computeLikelihoods before initialize")
        try:
            batch = FlatBatch.from_holders(readDataArray, haplotypeDataArray)
            self._ctx.compute(batch, likelihoodArray)
        except OutOfMemoryError as e:
            logger.warning("Exception thrown from native PairHMM computeLikelihoodsNative function call %s", e)
            raise OutOfMemoryError("Memory allocation failed")
        except IllegalArgumentException as e:
            logger.warning("Exception thrown from native PairHMM computeLikelihoodsNative function call %s", e)
            raise IllegalArgumentException("Ran into invalid argument issue")

    def done(self) -> None:
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None


class IntelPairHmmOMP(IntelPairHmm):
    """IntelPairHmmOMP.java:29-35: same natives, library name gkl_pairhmm_omp, honours
    maxNumberOfThreads (here: host threads of the reference-exact finalisation)."""

    def __init__(self):
        super().__init__()
        self.setNativeLibraryName("gkl_pairhmm_omp")
        self.useOmp = True


class IntelPairHmmFpga(IntelPairHmm):
    """IntelPairHmmFpga.java:32-39: a stub whose load() only warns and reports True."""

    def load(self, tempDir=None) -> bool:
        logger.warning("FPGA PairHMM is not supported; using the GPU implementation")
        return True
