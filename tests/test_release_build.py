"""The release library (libsqg_hip.so, the product) cannot be steered by the host program's environment: the A/B knobs, forced code
paths, fault injection and timing-only ablations of tools/README.md exist only in the -DSQG_DEV build (libsqg_hip_dev.so), which the
Python binding picks for a process that sets one of them.  Both carry the hash of the sources they were built from."""
import os
import re
import subprocess

import numpy as np
import pytest

import orc
from squigulator_amd import api, build, model, profiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strings(path):
    return set(re.findall(rb"SQG_[A-Z0-9_]{3,}", open(path, "rb").read()))


def test_release_library_does_not_contain_the_development_knobs():
    build.build()
    rel, dev = _strings(build.LIB), _strings(build.LIB_DEV)
    knobs = {k.encode() for k in api.DEV_KNOBS}
    assert not (rel & knobs), rel & knobs
    assert not any(s.startswith((b"SQG_ABL", b"SQG_TEST")) for s in rel)
    assert knobs <= dev, knobs - dev                       # ... and the binding's list is what the development build reads
    src = b"".join(open(os.path.join(build.CSRC, f), "rb").read() for f in os.listdir(build.CSRC) if f.endswith(".h"))
    assert set(re.findall(rb'SQG_DEV_ENV\("([A-Z0-9_]+)"\)', src)) == knobs
    # the only environment variables the release library reads: diagnostics on stderr
    assert set(re.findall(rb'[^_A-Z]getenv\("([A-Z0-9_]+)"\)', src)) == {b"SQG_VERBOSE", b"SQG_DEBUG_SYNC", b"SQG_STAGE_TIMING"}


def test_libraries_carry_the_hash_of_their_sources():
    build.build()
    h = build.source_hash()
    assert build.stamped_hash(build.LIB) == h and build.stamped_hash(build.LIB_DEV) == h
    assert not build.needs_build(build.LIB) and not build.needs_build(build.LIB_DEV)
    assert api.build_info(api.load_library(build.LIB)) == {"source_hash": h, "dev": "0"}
    assert api.build_info(api.load_library(build.LIB_DEV)) == {"source_hash": h, "dev": "1"}


def test_binding_picks_the_development_build_only_when_a_knob_is_set(monkeypatch):
    for k in api.DEV_KNOBS + ("SQG_LIB",):
        monkeypatch.delenv(k, raising=False)
    assert api.default_library_path() == build.LIB
    monkeypatch.setenv("SQG_VERBOSE", "1")                 # a diagnostic, read by both
    assert api.default_library_path() == build.LIB
    monkeypatch.setenv("SQG_SPLIT_CHAINS", "5")
    assert api.default_library_path() == build.LIB_DEV


@pytest.mark.gpu
def test_release_library_ignores_result_changing_knobs(monkeypatch):
    """SQG_ABL_NOFIX (skips the FP64 fix-ups: wrong samples), SQG_TEST_DELTA_X (inflated error bound), SQG_TEST_NO_LEAN,
    SQG_TEST_ORDER_FAULT (fails the batch) in the environment of a program that loads the release library: the oracle's output,
    and the fast path still taken"""
    rng = np.random.default_rng(31)
    prof, fl = profiles.get_profile("dna-r10-prom")
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    reads = [bytes(rng.choice(list(b"ACGT"), int(m)).astype(np.uint8)) for m in rng.integers(300, 4000, 300)]
    orac = orc.Oracle(prof, fl, k, mean, stdv, 42, num_workers=1)
    want = orac.run_batch_seqs(reads, want_ss=False)
    orac.close()
    for name, val in (("SQG_ABL_NOFIX", "1"), ("SQG_TEST_DELTA_X", "1"), ("SQG_TEST_NO_LEAN", "1"), ("SQG_TEST_ORDER_FAULT", "1"),
                      ("SQG_SPLIT_CHAINS", "0"), ("SQG_PART_CLAIMS", "1")):
        monkeypatch.setenv(name, val)
    gen = api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED, lib_path=build.LIB)
    assert api.build_info(gen.L)["dev"] == "0"
    b = gen.submit(reads)
    sig = b.signal()
    for i, w in enumerate(want):
        np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig, err_msg=f"read {i}")
    tm = gen.timing()
    assert tm["lean_ms"] > 0                                # SQG_TEST_NO_LEAN was not obeyed
    assert 0 < tm["fallback_samples"] < b.n_samples // 100  # ... nor SQG_TEST_DELTA_X (every sample through FP64) nor SQG_ABL_NOFIX (none)
    assert gen.probe_lds_order(64, 4)[1] is True           # ... nor SQG_PART_CLAIMS
    b.free(); gen.close()
    # the development build, same environment minus the fault: obeys (the generic kernel takes everything)
    monkeypatch.delenv("SQG_TEST_ORDER_FAULT"); monkeypatch.delenv("SQG_ABL_NOFIX")
    gen = api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
    assert api.build_info(gen.L)["dev"] == "1"
    b = gen.submit(reads)
    sig = b.signal()
    for i, w in enumerate(want):
        np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig, err_msg=f"dev, read {i}")
    assert gen.timing()["fallback_samples"] == b.n_samples          # SQG_TEST_DELTA_X=1: every sample through the FP64 fix-ups
    b.free(); gen.close()
