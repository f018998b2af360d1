import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference's own kernel objects (oracle/_ref), when built and runnable here."""
    from oracle.oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    try:
        return Reference()
    except RuntimeError as e:  # no AVX on this host
        pytest.skip(str(e))


@pytest.fixture(scope="session")
def golden_cases():
    from tests.golden_io import load_testdata
    return load_testdata()


# `pytest -x` stops at the first failure, so the parity tests of the SURVEY 8 rows run BEFORE the robustness,
# concurrency and fuzz tests: a failure in one of those can then no longer leave a row's parity unexamined.
# Order: oracle / host logic -> PairHMM parity -> JNI parity -> PDHMM parity -> Smith-Waterman parity -> plugin
# mirrors -> shards / multi-device -> robustness -> combiner -> context churn -> fuzz -> bench contract.
_FILE_ORDER = [
    "test_oracle.py", "test_cabi_cpu.py",
    "test_gpu_parity.py", "test_jni_shim.py", "test_pdhmm.py", "test_sw.py", "test_plugin_mirror.py",
    "test_long_reads.py",
    "test_shard.py", "test_multi_device.py", "test_gpu_robustness.py", "test_small_call_combiner.py",
    "test_context_churn.py", "test_fuzz_gpu.py", "test_bench_contract.py",
]


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(_FILE_ORDER)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), len(_FILE_ORDER) - 1.5))  # stable within a file
    # A device-side wait that never ends (the long-read kernel's wavefronts wait for each other's step counters, the
    # combiner's callers for each other's launches) must fail ONE test, not hold the GPU box until its time limit:
    # every GPU test gets a wall-clock limit when pytest-timeout is installed (it is in this image).
    if config.pluginmanager.hasplugin("timeout"):
        for it in items:
            if it.get_closest_marker("gpu") and not it.get_closest_marker("timeout"):
                it.add_marker(pytest.mark.timeout(900, method="thread"))
