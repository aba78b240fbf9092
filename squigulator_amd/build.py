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
HEADERS = [os.path.join(os.path.dirname(HERE), "include", "sqg.h")] + \
          [os.path.join(CSRC, h) for h in ("sqg_kernels.h", "k_common.h", "k_events.h", "k_part.h", "k_samples.h", "k_sampler.h", "k_svb.h",
                                           "h_common.h", "h_context.h", "h_stage.h", "h_sampler.h", "h_run.h", "h_results.h", "h_blow5.h")]
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
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


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
