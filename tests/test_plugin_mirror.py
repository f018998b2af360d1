"""Tests written like the reference's PairHmmUnitTest.java, against the Python mirror of the
IntelPairHmm plugin class."""
import numpy as np
import pytest

from gkl_amd.batch import HaplotypeDataHolder, PairHMMNativeArguments, ReadDataHolder
from gkl_amd.errors import IllegalArgumentException, NullPointerException
from gkl_amd.pairhmm import IntelPairHmm, IntelPairHmmFpga, IntelPairHmmOMP


def test_invalid_inputs_for_compute_likelihoods():
    # testInvalidInputsForComputeLikelihoods (PairHmmUnitTest.java:29-53): NPE raised host-side
    hmm = IntelPairHmm()
    reads = [ReadDataHolder(b"ACGT", b"++++", b"++++", b"++++", b"++++")]
    haps = [HaplotypeDataHolder(b"ACGT")]
    out = np.zeros(1)
    for args in ((None, haps, out), (reads, None, out), (reads, haps, None)):
        with pytest.raises(NullPointerException):
            hmm.computeLikelihoods(*args)


def test_fpga_and_unknown_library_names():
    # fpgaTest (PairHmmUnitTest.java:91-98)
    hmm = IntelPairHmm()
    hmm.setNativeLibraryName("gkl_pairhmm_shacc")
    assert hmm.load(None) is False
    assert IntelPairHmmFpga().load(None) is True


@pytest.mark.gpu
@pytest.mark.parametrize("cls,threads", [(IntelPairHmm, 1), (IntelPairHmmOMP, 10)])
def test_simple_and_omp(cls, threads):
    # simpleTest / omp_Test (PairHmmUnitTest.java:55-89,100-169)
    hmm = cls()
    assert hmm.load(None)
    hmm.initialize(PairHMMNativeArguments(useDoublePrecision=False, maxNumberOfThreads=threads))
    reads = [ReadDataHolder(b"ACGT", b"++++", b"++++", b"++++", b"++++")]
    haps = [HaplotypeDataHolder(b"ACGT")]
    out = np.zeros(1)
    hmm.computeLikelihoods(reads, haps, out)
    assert abs(out[0] - (-6.022797e-01)) <= 1e-5
    hmm.done()


@pytest.mark.gpu
def test_data_file(golden_cases):
    # dataFileTest (PairHmmUnitTest.java:171-234): both precisions, abs tol 1e-5
    for use_double in (False, True):
        hmm = IntelPairHmm()
        assert hmm.load(None)
        hmm.initialize(PairHMMNativeArguments(useDoublePrecision=use_double, maxNumberOfThreads=1))
        for c in golden_cases:
            out = np.zeros(1)
            hmm.computeLikelihoods([ReadDataHolder(c["read"], c["q"], c["i"], c["d"], c["c"])],
                                   [HaplotypeDataHolder(c["hap"])], out)
            assert abs(out[0] - c["expected"]) <= 1e-5
        hmm.done()


@pytest.mark.gpu
def test_short_quality_array_is_illegal_argument():
    hmm = IntelPairHmm()
    assert hmm.load(None)
    hmm.initialize(None)
    with pytest.raises(IllegalArgumentException) as e:
        hmm.computeLikelihoods([ReadDataHolder(b"ACGT", b"+++", b"++++", b"++++", b"++++")],
                               [HaplotypeDataHolder(b"ACGT")], np.zeros(1))
    assert str(e.value) == "Ran into invalid argument issue"  # IntelPairHmm.java:143-145
    hmm.done()
