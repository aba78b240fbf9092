"""Build the HIP shared library in-tree (no JIT cache: the .so must travel with the repo snapshot).

    python -m squigulator_amd.build          # -> squigulator_amd/csrc/libsqg_hip.so

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


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + headers())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-result", "-o", LIB] + SOURCES + ["-lz"]
    if verbose:
        print("[squigulator_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
