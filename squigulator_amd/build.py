"""Build the HIP shared library in-tree (no JIT cache: the .so must travel with the repo snapshot).

    python -m squigulator_amd.build          # -> squigulator_amd/csrc/libsqg_hip.so      (release: the product)
                                             #    squigulator_amd/csrc/libsqg_hip_dev.so  (-DSQG_DEV: reads the A/B and test knobs
                                             #    of tools/README.md from the environment; tests that force a code path, tools/)

Both carry the sha256 of the sources they were built from (`SQG_SOURCE_HASH=...;` in the file, sqg_build_info() at run time); a
library is rebuilt when that stamp differs from the tree's -- not by file times, which a checkout or a copy to the GPU box changes.

hipcc cross-compiles gfx950 code objects without a GPU, so this also runs in the CPU-only
build container.  -fno-slp-vectorize: the SLP pass pairs the two events a k_events thread handles into v_pk_* ops that need
s_nop padding on gfx950 and measure 6 % slower.  -ffp-contract=off: the FP64 path must round x*s+m twice like the reference
(which is built with gcc -std=c99, no FMA contraction; see DESIGN.md "Exact arithmetic").
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsqg_hip.so")
LIB_DEV = os.path.join(CSRC, "libsqg_hip_dev.so")
SOURCES = [os.path.join(CSRC, "sqg_hip.hip")]
import glob
import hashlib


def headers() -> list:
    """every header the one translation unit includes: include/sqg.h and all of csrc/*.h (globbed: a new header is a dependency
    the moment it exists)"""
    return [os.path.join(os.path.dirname(HERE), "include", "sqg.h")] + sorted(glob.glob(os.path.join(CSRC, "*.h")))


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the sources the library is built from, in a fixed order: stamps the PMC traffic file
    (tools/make_traffic.py) so that bench.py can tell whether the counters it quotes belong to the kernels it runs"""
    h = hashlib.sha256()
    for p in SOURCES + headers():
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def file_hash(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]

ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


_MARK = b"SQG_SOURCE_HASH="


def stamped_hash(lib: str) -> str | None:
    """the source hash a built library carries (read from the file: no dlopen, no HIP runtime), None if it has none"""
    try:
        with open(lib, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(_MARK)
    if i < 0:
        return None
    j = blob.find(b";", i)
    return blob[i + len(_MARK):j].decode("ascii", "replace") if 0 <= j - i <= 64 else None


def needs_build(lib: str = LIB) -> bool:
    return stamped_hash(lib) != source_hash()


def flags(dev: bool = False, extra=()) -> list:
    """the hipcc line, without output and sources (tools/ build their A/B variants from it: `python -m squigulator_amd.build --flags`)"""
    fl = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-slp-vectorize",
          "-Wall", "-Wno-unused-result", f'-DSQG_SOURCE_HASH="{source_hash()}"']
    if dev:
        fl.append("-DSQG_DEV")
    return fl + list(extra)


def build_variant(out: str, dev: bool = True, extra=(), verbose: bool = True) -> str:
    cmd = [hipcc_path()] + flags(dev, extra) + ["-o", out] + SOURCES + ["-lz"]
    if verbose:
        print("[squigulator_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    """both libraries (two independent hipcc runs, side by side); returns the release one"""
    todo = [(LIB, False), (LIB_DEV, True)]
    todo = [(o, d) for o, d in todo if force or needs_build(o)]
    procs = []
    for out, dev in todo:
        cmd = [hipcc_path()] + flags(dev) + ["-o", out] + SOURCES + ["-lz"]
        if verbose:
            print("[squigulator_amd.build]", " ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    for out, _ in todo:
        if stamped_hash(out) != source_hash():
            raise RuntimeError(f"{out}: the built library does not carry the tree's source hash")
    return LIB


if __name__ == "__main__":
    if "--flags" in sys.argv:
        print(" ".join(flags(dev=True)))
    else:
        build(force="--force" in sys.argv)
        print(LIB)
