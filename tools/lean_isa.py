#!/usr/bin/env python3
"""The instruction listing DESIGN.md cites for the dominant kernel: the four-step sample loop of k_samples_lean<false, 4>, disassembled
from the sources as build.py compiles them, every instruction with its class and the issue cost tools/ubench3.hip measured for that
class on gfx950 -> profiles/<round>_lean_isa.md.  Runs without a GPU (hipcc -S for the device side only).

    python tools/lean_isa.py [r04]

What is listed is the loop's MAIN path: the blocks every 64-sample step executes.  The blocks behind the acceptance test's rare branch
(an undecided sample is parked in LDS: ~1 % of the steps take it) are counted separately."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from squigulator_amd import build  # noqa: E402

KERNEL = "_Z14k_samples_leanILb0ELi4EEv9SigParamsi"
# cycles per wave-instruction per SIMD at the nominal clock, tools/ubench3.hip
COST = {"trans": 8.3, "full": 2.5, "half": 4.4}
FULL = ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_mov_b32", "v_min_u32", "v_max_u32",
        "v_min_f32", "v_max_f32", "v_ashrrev_i32")
TRANS = ("v_log_f32", "v_sqrt_f32", "v_cos_f32", "v_sin_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32")


def classify(op):
    if op.startswith(TRANS):
        return "VALU trans"
    if op.startswith("v_"):
        return "VALU full-rate" if op.startswith(FULL) else "VALU half-rate"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "sqg.s")
        fl = [f for f in build.flags(dev=False) if f not in ("-shared", "-fPIC")]
        subprocess.check_call([build.hipcc_path()] + fl + ["--cuda-device-only", "-S", "-o", asm] + build.SOURCES, stderr=subprocess.DEVNULL)
        text = open(asm).read().splitlines()
    a = next(i for i, ln in enumerate(text) if ln.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(text)) if "s_endpgm" in text[i])
    fn = text[a:b + 1]
    meta = "\n".join(text)
    regs = {}
    for blk in meta.split("\n  - "):                       # the code object's metadata: one map per kernel
        if re.search(r"\.name:\s+" + KERNEL + r"\s", blk):
            for key in (".vgpr_count", ".sgpr_count", ".group_segment_fixed_size", ".vgpr_spill_count"):
                mm = re.search(re.escape(key) + r":\s+(\d+)", blk)
                regs[key.lstrip(".")] = int(mm.group(1)) if mm else None
    # inner loops (depth 2) and their extents: from the header label to the backward branch to it
    loops = []
    for i, ln in enumerate(fn):
        if "Inner Loop Header: Depth=2" in ln:
            lab = fn[i - 1].split(":")[0].strip()
            j = next(k for k in range(i, len(fn)) if re.search(r"s_c?branch\w*\s+" + re.escape(lab) + r"\b", fn[k]))
            loops.append((lab, i - 1, j))
    lab, lo, hi = max(loops, key=lambda t: sum("global_store_short" in x for x in fn[t[1]:t[2] + 1]))
    body = fn[lo:hi + 1]
    n_steps = sum("global_store_short" in x for x in body)
    rows, rare, i = [], [], 0
    while i < len(body):
        ln = body[i].strip()
        op = ln.split()[0] if ln and not ln.startswith((";", ".")) and not ln.endswith(":") else None
        mm = re.match(r"s_cbranch_execz\s+(\.LBB\d+_\d+)", ln)
        if mm:                                              # a guarded region: main (the int16 store) or rare (parking)?
            end = next(k for k in range(i + 1, len(body)) if body[k].startswith(mm.group(1) + ":"))
            region = body[i + 1:end]
            rows.append((ln, classify("s_cbranch")))
            if any("global_store_short" in x for x in region) and not any("ds_write" in x for x in region):
                i += 1
                continue
            rare += [x.strip() for x in region if x.strip() and not x.strip().startswith((";", ".")) and not x.strip().endswith(":")]
            i = end
            continue
        if op:
            rows.append((ln, classify(op)))
        i += 1
    cnt = {}
    for _, c in rows:
        cnt[c] = cnt.get(c, 0) + 1
    valu = sum(v for k, v in cnt.items() if k.startswith("VALU"))
    cyc = cnt.get("VALU trans", 0) * COST["trans"] + cnt.get("VALU full-rate", 0) * COST["full"] + cnt.get("VALU half-rate", 0) * COST["half"]
    out = os.path.join(ROOT, "profiles", f"{rnd}_lean_isa.md")
    with open(out, "w") as f:
        f.write(f"# k_samples_lean<false, 4>: the sample loop as compiled ({rnd}; tools/lean_isa.py, source hash {build.source_hash()})\n\n")
        f.write("`hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize` (squigulator_amd/build.py), device side, "
                f"loop `{lab}` of `{KERNEL}`: **{n_steps} steps of 64 samples per iteration** (one wavefront; four steps unrolled, the pipeline "
                "registers alternate).  Listed: the main path -- what every step executes.  Behind the acceptance test's rare branch (an "
                f"undecided sample is parked in LDS, ~1 % of the steps): {len(rare)} more instructions per iteration, not listed.\n\n")
        f.write("Issue cost per wave-instruction and SIMD (tools/ubench3.hip, nominal clock): full-rate VALU (fp32 add/mul/fma, 32-bit add/sub/logic/shift, "
                f"mov, min) {COST['full']} cycles; half-rate VALU (`v_mad_u64_u32`, `v_mul_lo_u32`, conversions, compares, `v_mbcnt`, `v_lshl_add_u32`, SDWA forms) "
                f"{COST['half']}; transcendental (`v_log_f32`, `v_sqrt_f32`, `v_cos_f32`) {COST['trans']}.\n\n")
        f.write("| class | per iteration (4 steps) | per step = per 64 samples |\n|---|---|---|\n")
        for k in ("VALU full-rate", "VALU half-rate", "VALU trans", "SALU", "branch", "LDS", "VMEM", "wait", "other"):
            if cnt.get(k):
                f.write(f"| {k} | {cnt[k]} | {cnt[k] / n_steps:.2f} |\n")
        f.write(f"| **VALU, all** | **{valu}** | **{valu / n_steps:.2f}** |\n")
        f.write(f"| VALU issue cycles by the table | {cyc:.0f} | {cyc / n_steps:.1f} ({cyc / valu:.2f} per instruction) |\n\n")
        f.write(f"Registers / LDS of the kernel (code-object metadata): {regs}\n\n")
        f.write("What the VALU instructions of one step are (step 2 of the iteration; the other three differ in register names and immediate offsets only): "
                "sample -> event (`v_mbcnt_lo`, `v_mbcnt_hi`, `v_lshl_add_u32`: the record's LDS address), jump index (`v_sub_u32_sdwa`), the Lehmer step "
                "`state * 2 a^(2j+1) mod (2^31 - 1)` (`v_mad_u64_u32`, `v_lshrrev`, `v_add`, `v_add`, `v_min`), the two uniforms (`v_cvt_f32_u32`, `v_mul_f32`; "
                "`v_mul_lo_u32`, `v_cvt_f32_i32`, `v_mul_f32`, `v_fmac_f32`), Box-Muller (`v_log_f32`, `v_fmamk_f32`, `v_sqrt_f32`, `v_cos_f32`, `v_mul_f32`), "
                "digitisation and the exactness test (`v_fma_f32`, `v_add_f32` x2, `v_sub_f32`, `v_cmp_nlt_f32`, `v_cmp_lt_u32`), the table pointer for the "
                "next-but-one step's broadcast read (`v_mov_b32`), the int16 value (`v_add_u32`) + `global_store_short`.\n\n")
        f.write("```\n")
        for ln, c in rows:
            f.write(f"{c:15s} {ln}\n")
        f.write("```\n")
    print(out, f"{valu / n_steps:.2f} VALU per step, {cyc / n_steps:.1f} cycles per step by the table")


if __name__ == "__main__":
    main()
