"""SURVEY.md section 8b, last row: a CPU backend with identical symbols.  oracle/libsqg_cpu.so implements include/sqg.h on
the CPU oracle (test infrastructure, never loaded by the product): one host program drives both libraries."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from squigulator_amd import api, model, options, profiles
from refvec_cases import REFVEC_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_LIB = os.path.join(ROOT, "oracle", "libsqg_cpu.so")
REFVEC = os.path.join(ROOT, "tests", "golden", "refvec")


@pytest.fixture(scope="module", autouse=True)
def _built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libsqg_cpu.so"], stdout=subprocess.DEVNULL)


def test_same_symbols_as_the_header():
    hdr = open(os.path.join(ROOT, "include", "sqg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(sqg_[a-z0-9_]+)\s*\(", hdr)))
    L = api.load_library(CPU_LIB)
    for n in names:
        assert hasattr(L, n), n
    assert set(names) == set(api.EXPORTS)


def _fixture(cid):
    d = np.load(os.path.join(REFVEC, cid + ".npz"))
    meta = d["meta"]
    so = np.concatenate(([0], np.cumsum(meta[:, 7])))
    qo = np.concatenate(([0], np.cumsum(meta[:, 4])))
    reads = [bytes(d["seq"][qo[i]:qo[i + 1]]) for i in range(len(meta))]
    return options.parse_args(str(d["cmd"])), int(d["k"]), meta, reads, d["sig"], so, d["offset"], d["median"]


@pytest.mark.parametrize("cid", ["r9_t1", "r9_tk16", "r10_t1", "rna004_prefix", "r9_ideal_time"])
def test_cpu_backend_reproduces_the_compiled_reference_vectors(cid):
    """the reference's own outputs (tests/golden/refvec) through the sqg_* entry points of the CPU backend"""
    o, k, meta, reads, sig, so, offset, median = _fixture(cid)
    mean, stdv = model.synthetic_model(k)
    gen = api.SignalGenerator(o.profile, o.flags, k, mean, stdv, o.seed, num_workers=o.threads, amp_noise=o.amp_noise, lib_path=CPU_LIB)
    done = 0
    while done < len(reads):
        nb = min(o.batch, len(reads) - done)
        b = gen.submit(reads[done:done + nb])
        got = b.signal()
        for i in range(nb):
            np.testing.assert_array_equal(got[b.sig_off[i]:b.sig_off[i + 1]], sig[so[done + i]:so[done + i + 1]])
            assert b.offset[i] == offset[done + i] and b.median_before[i] == median[done + i]
        b.free()
        done += nb
    gen.close()


def test_the_c_host_example_runs_on_the_cpu_backend(tmp_path):
    """examples/process_db_gpu.c, unchanged, linked against the CPU backend: the plain-C caller of include/sqg.h end to end
    (sampler, signals, svb-zd) without a GPU"""
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / "process_db_cpu")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "process_db_gpu.c"),
                           "-L", os.path.dirname(CPU_LIB), "-lsqg_cpu", "-Wl,-rpath," + os.path.dirname(CPU_LIB), "-o", exe])
    r = subprocess.run([exe, "5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("read ") == 5


def _host_program(lib_path, tmp, tag):
    """one host program: sample on a resident genome, run, fetch, compress, write BLOW5 -- against whichever library"""
    prof, fl = profiles.get_profile("dna-r10-prom")
    mean, stdv = model.synthetic_model(9)
    import bench
    contigs = bench.synthetic_genome_host(1.0)
    gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=4, mode=api.MODE_CERTIFIED, lib_path=lib_path)
    gen.load_genome(contigs, 1500, api.SAMPLE_DNA)
    path = os.path.join(tmp, tag + ".blow5")
    w = api.Blow5Writer(path, prof, fl, threads=2, lib_path=lib_path)
    out = []
    for nb in (12, 9):
        b = gen.sample(nb).run().wait()
        enc, eo = b.compress()
        out.append((b.signal().copy(), np.array(b.sig_off), np.array(b.offset), np.array(b.median_before), b.dwell().copy(), b.reads(),
                    enc.copy(), eo.copy(), dict(b.sampled)))
        w.write_batch(b, [b"S1_%d" % i for i in range(nb)])
        b.free()
    w.close()
    gen.close()
    return out, open(path, "rb").read()


@pytest.mark.gpu
def test_one_host_program_two_backends(tmp_path):
    cpu, f_cpu = _host_program(CPU_LIB, str(tmp_path), "cpu")
    hip, f_hip = _host_program(None, str(tmp_path), "hip")
    for a, b in zip(cpu, hip):
        for x, y in zip(a[:5], b[:5]):
            np.testing.assert_array_equal(x, y)
        assert a[5] == b[5]
        np.testing.assert_array_equal(a[6], b[6]); np.testing.assert_array_equal(a[7], b[7])
        for key in ("ref_idx", "ref_pos", "rlen", "ref_len"):
            np.testing.assert_array_equal(a[8][key], b[8][key])
        assert a[8]["strand"] == b[8]["strand"]
    assert f_cpu == f_hip
