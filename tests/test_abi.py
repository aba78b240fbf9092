"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/sqg.h declares, and
fails loudly (no fallback) without a GPU.  No compute calls."""
import ctypes as C
import os
import re

import pytest

from squigulator_amd import api, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return api.load_library()


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "sqg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sqg_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sqg.h but not exported"
    assert set(names) == set(api.EXPORTS), set(names) ^ set(api.EXPORTS)


def test_error_strings_and_scheduler_rule(lib):
    assert lib.sqg_strerror(0) == b"ok"
    assert b"HIP" in lib.sqg_strerror(-5)
    # src/thread.c:80-99: contiguous blocks of ceil(n/T); -t1 puts everything on worker 0
    assert [lib.sqg_worker_of(i, 10, 4) for i in range(10)] == [0, 0, 0, 1, 1, 1, 2, 2, 2, 3]
    assert [lib.sqg_worker_of(i, 5, 1) for i in range(5)] == [0] * 5
    assert [lib.sqg_worker_of(i, 3, 8) for i in range(3)] == [0, 1, 2]


def test_no_silent_cpu_fallback(lib):
    """Without a usable GPU the product refuses to run (the oracle is never a fallback)."""
    if lib.sqg_device_count() > 0:
        pytest.skip("a GPU is present")
    from squigulator_amd import model, profiles
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    with pytest.raises(api.SqgError) as e:
        api.SignalGenerator(prof, fl, 6, mean, stdv, seed=1)
    assert e.value.code == -5


def test_rejects_bad_config(lib):
    cfg = api.CCfg()
    h = C.c_void_p()
    assert lib.sqg_create(C.byref(cfg), C.byref(h)) == -1      # abi_version 0
    assert lib.sqg_create(None, C.byref(h)) == -1


def test_product_does_not_reference_the_oracle():
    """Nothing under squigulator_amd/ or include/ may import, link or call oracle/."""
    bad = []
    for base in ("squigulator_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                    txt = open(os.path.join(d, f), errors="ignore").read()
                    if re.search(r"sqg_oracle|libsqg_oracle|import orc\b|from orc\b|oracle/", txt.replace("never touches oracle/", "")):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def _build_example(tmp_path):
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    build.build()
    src = os.path.join(ROOT, "examples", "process_db_gpu.c")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           "-fsyntax-only", src])
    libdir = os.path.dirname(build.LIB)
    exe = str(tmp_path / "process_db_gpu")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-L", libdir, "-lsqg_hip",
                           "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_header_is_plain_c_and_the_c_example_compiles(tmp_path):
    """include/sqg.h is consumed by a C host (the reference is C): the example host loop must compile as C99 with
    -pedantic and link against the library without any HIP/C++/torch header.  Without a GPU the program must fail loudly
    at sqg_create (no CPU fallback)."""
    import subprocess
    exe = _build_example(tmp_path)
    r = subprocess.run([exe, "4"], capture_output=True, text=True)
    if r.returncode != 0:
        assert "sqg_create" in r.stderr, r.stderr


@pytest.mark.gpu
def test_the_c_example_runs_on_the_gpu(tmp_path):
    """the plain-C caller of include/sqg.h (examples/process_db_gpu.c) on a real device: every read comes back"""
    import subprocess
    exe = _build_example(tmp_path)
    r = subprocess.run([exe, "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("read ") == 4, r.stdout
