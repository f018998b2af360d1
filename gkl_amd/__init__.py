"""gkl_amd -- MI355X-native PairHMM forward hot path behind GKL's IntelPairHmm surface.

Only what the hot path needs lives here: the HIP kernels and C-ABI library
(``csrc/``), the ctypes binding (``native``), the host-side mirror of the
reference's ``IntelPairHmm`` plugin class (``pairhmm``), synthetic workload
generators (``synth``) and the multi-GPU read-range sharding (``shard``).
"""
from .batch import (FlatBatch, HaplotypeDataHolder, PairHMMNativeArguments,  # noqa: F401
                    ReadDataHolder)
from .errors import (IllegalArgumentException, NullPointerException,  # noqa: F401
                     OutOfMemoryError, RuntimeException)

__all__ = ["FlatBatch", "ReadDataHolder", "HaplotypeDataHolder", "PairHMMNativeArguments",
           "IllegalArgumentException", "NullPointerException", "OutOfMemoryError",
           "RuntimeException"]
