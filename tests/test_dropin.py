"""The drop-in, compiled and run: oracle/_ref/ref_host_gpu is the REFERENCE's own host code (genread.c, ref.c, format.c, the
record helpers of gensig.c, slow5lib -- compiled where they lie, oracle/Makefile) with process_db() replaced by the binding of
INTEGRATION.md (oracle/ref_host_gpu.c: process_db_gpu over include/sqg.h), linked against libsqg_hip.so by name.  The files it
writes -- SLOW5 / BLOW5 through the reference's slow5lib, PAF / SAM / FASTA through its format.c -- must be `cmp`-identical to what
oracle/_ref/ref_harness (the reference's own gen_sig, same configuration) writes.  src/sim.c:514-627.

Without a GPU the same binary runs against the CPU backend (LD_LIBRARY_PATH leads the name libsqg_hip.so to oracle/libsqg_cpu.so):
that checks the binding code itself in the build container.  Both binaries are built only where /root/reference is mounted and
travel to the GPU box as files."""
import filecmp
import os
import subprocess

import numpy as np
import pytest

from squigulator_amd import build, model, options

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INPUTS = os.path.join(ROOT, "tests", "golden", "inputs")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
HOSTGPU = os.path.join(ROOT, "oracle", "_ref", "ref_host_gpu")

CASES = [
    ("r9_t1", "nCoV-2019.reference.fasta -x dna-r9-prom -n 8 --seed 42 -r 1000 -t1", "slow5"),
    ("rna004_prefix", "rnasequin_sequences_2.4.fa -x rna004-prom -n 3 --seed 42 -t1 --prefix=yes", "blow5"),
    ("r10_tk8", "nCoV-2019.reference.fasta -x dna-r10-prom -n 16 --seed 42 -r 600 -t 8 -K 8", "blow5"),
    ("r10_t1_k5", "nCoV-2019.reference.fasta -x dna-r10-prom -n 12 --seed 7 -r 800 -t1 -K 5 --paf-ref", "slow5"),   # several batches: streams carry over
    ("rna9_prefix", "rnasequin_sequences_2.4.fa -x rna-r9-prom -n 3 --seed 42 -t1 --prefix=yes", "slow5"),
]


def _need_binaries():
    if not (os.path.exists(HARNESS) and os.path.exists(HOSTGPU)):
        pytest.skip("oracle/_ref/ref_harness / ref_host_gpu are built only where the upstream tree is mounted (make -C oracle ref)")


def _config(cmdline, tmp, tag, ext, extra=None):
    o = options.parse_args(cmdline)
    k = o.kmer_size_default
    mpath = os.path.join(tmp, f"synthetic_{k}.model")
    if not os.path.exists(mpath):
        mean, stdv = model.synthetic_model(k)
        model.write_f5c_model(mpath, k, mean, stdv)
    outs = {"slow5": os.path.join(tmp, f"{tag}.{ext}"), "paf": os.path.join(tmp, f"{tag}.paf"), "sam": os.path.join(tmp, f"{tag}.sam"),
            "fasta_out": os.path.join(tmp, f"{tag}.fa")}
    cfg = {"fasta": os.path.join(INPUTS, o.ref), "model": mpath, "flags": o.flags, "amp_noise": repr(float(np.float32(o.amp_noise))),
           "seed": o.seed, "threads": o.threads, "batch": o.batch, "nreads": o.nreads, "rlen": o.rlen, **outs}
    for name, v in zip(("digitisation", "sample_rate", "bps", "range", "offset_mean", "offset_std", "median_before_mean",
                        "median_before_std", "dwell_mean", "dwell_std"), o.profile.as_tuple()):
        cfg[name] = repr(float(v))
    cfg.update(extra or {})
    path = os.path.join(tmp, f"{tag}.cfg")
    with open(path, "w") as f:
        f.write("".join(f"{a}={b}\n" for a, b in cfg.items()))
    return path, outs


def _compare(cmdline, ext, tmp, env, modes):
    cfg, want = _config(cmdline, tmp, "ref", ext)
    subprocess.run([HARNESS, cfg], check=True, timeout=600)
    for mode in modes:
        cfg, got = _config(cmdline, tmp, "gpu_" + mode, ext, {"mode": mode, "device": 0})
        p = subprocess.run([HOSTGPU, cfg], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        for kind in ("slow5", "paf", "sam", "fasta_out"):
            assert os.path.getsize(want[kind]) > 0
            assert filecmp.cmp(want[kind], got[kind], shallow=False), f"{kind} ({mode}) differs from the reference's: {cmdline}"


@pytest.mark.parametrize("cid,cmdline,ext", CASES, ids=[c[0] for c in CASES])
def test_binding_against_the_cpu_backend(cid, cmdline, ext, tmp_path):
    """no GPU needed: the name libsqg_hip.so resolved to the CPU backend of the same ABI (test infrastructure)"""
    _need_binaries()
    cpu = os.path.join(ROOT, "oracle", "libsqg_cpu.so")
    if not os.path.exists(cpu):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libsqg_cpu.so"])
    d = tmp_path / "lib"
    d.mkdir()
    os.symlink(cpu, d / "libsqg_hip.so")
    env = dict(os.environ, LD_LIBRARY_PATH=str(d) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    _compare(cmdline, ext, str(tmp_path), env, ("exact",))


@pytest.mark.gpu
@pytest.mark.parametrize("cid,cmdline,ext", CASES, ids=[c[0] for c in CASES])
def test_reference_host_with_the_gpu_library(cid, cmdline, ext, tmp_path):
    """the product library behind the reference's own host code, both arithmetic modes"""
    _need_binaries()
    build.build()
    env = {k: v for k, v in os.environ.items() if k != "LD_LIBRARY_PATH"}
    ldd = subprocess.run(["ldd", HOSTGPU], env=env, capture_output=True, text=True).stdout
    assert os.path.realpath(build.LIB) in [os.path.realpath(t.split("=>")[1].split("(")[0].strip()) for t in ldd.splitlines() if "libsqg_hip.so" in t]
    _compare(cmdline, ext, str(tmp_path), env, ("certified", "exact"))


def test_integration_document_quotes_the_compiled_binding():
    """INTEGRATION.md's binding is the text of oracle/ref_host_gpu.c (tools/sync_integration.py), not a copy that can drift"""
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sync_integration.py"), "--check"])
    assert p.returncode == 0, "run tools/sync_integration.py"
