import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order: the driver runs `pytest -x -q -m gpu`, so whatever fails first hides everything behind it.  Oracle / golden
# comparisons come first -- one test per BASELINE.json config (test_00_configs), then the parity files --, the bench-contract and
# multi-rank files (subprocesses of bench.py: presence / consistency checks, no clocks) last.  Files not named keep their place
# (alphabetical) between the two groups.
_FIRST = ["test_00_configs", "test_hip_parity", "test_config2_hg38", "test_fuzz_parity", "test_sampler", "test_svb", "test_blow5", "test_dropin",
          "test_long_reads", "test_split_chains", "test_precount", "test_two_contexts", "test_range_sharding", "test_many_reads", "test_full_size"]
_LAST = ["test_abi", "test_release_build", "test_distributed_gloo", "test_bench_multi_gpu"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _FIRST:
            return (0, _FIRST.index(name))
        if name in _LAST:
            return (2, _LAST.index(name))
        return (1, 0)
    items.sort(key=rank)                      # (stable: the order within a file is kept)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: build it on demand (gcc, <1 s)."""
    so = os.path.join(ROOT, "oracle", "libsqg_oracle.so")
    src = os.path.join(ROOT, "oracle", "sqg_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libsqg_oracle.so"])
    yield
