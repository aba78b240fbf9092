#!/usr/bin/env python3
"""Writes tools/ubench6.hip: FOUR steps of k_samples_lean's loop (one iteration of the compiled loop) under different instruction orders.

tools/ubench5 showed that the bare step costs 115 cycles in the compiler's order, 96 with the transcendentals grouped and 101 with two
steps interleaved: on gfx950 two VALU instructions of different wavefronts share a 4-cycle issue slot when they are compatible (a plain
fp32/int op beside another plain op, a conversion, a compare, mbcnt, min, lshl_add ...), never beside a transcendental or an SDWA form.
Here every schedule keeps the step's true dependencies (registers are per step), so what is measured can be written in the kernel.

    python tools/ubench6_gen.py && hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench6 tools/ubench6.hip && /tmp/ubench6"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# one step; registers per step: P0:P1 (a pair), M, T, U, W, A, S, O, X.  shared: v60-62 (lane masks), v57, v70, v82; s40, s41
I = {
    "mad":   "v_mad_u64_u32 v[{P0}:{P1}], vcc, v{P0}, v{M}, 0",
    "lshr":  "v_lshrrev_b32 v{P0}, 1, v{P0}",
    "add1":  "v_add_u32 v{P0}, v{P0}, v{P1}",
    "add2":  "v_add_u32 v{P1}, 0x80000001, v{P0}",
    "min":   "v_min_u32 v{P1}, v{P0}, v{P1}",
    "cvtu":  "v_cvt_f32_u32 v{P0}, v{P1}",
    "mbl":   "v_mbcnt_lo_u32_b32 v{X}, v60, v62",
    "mbh":   "v_mbcnt_hi_u32_b32 v{U}, v61, v{X}",
    "lsa":   "v_lshl_add_u32 v{W}, v{U}, 4, s41",
    "mullo": "v_mul_lo_u32 v{T}, v{P1}, s40",
    "u1":    "v_mul_f32 v{P0}, 0x30000000, v{P0}",
    "log":   "v_log_f32 v{M}, v{P0}",
    "cvti":  "v_cvt_f32_i32 v{T}, v{T}",
    "u2a":   "v_mul_f32 v{P0}, 0x37034e00, v{P0}",
    "fmamk": "v_fmamk_f32 v{M}, v{M}, 0xbfb17218, v82",
    "u2b":   "v_fmac_f32 v{P0}, 0x30000000, v{T}",
    "sqrt":  "v_sqrt_f32 v{M}, v{M}",
    "cos":   "v_cos_f32 v{P0}, v{P0}",
    "mov":   "v_mov_b32 v{T}, s41",
    "cmpu":  "v_cmp_lt_u32 vcc, s40, v{P1}",
    "g":     "v_mul_f32 v{P0}, v{M}, v{P0}",
    "dig":   "v_fmac_f32 v{A}, v{P0}, v{S}",
    "r1":    "v_add_f32 v{P0}, 0x4b400000, v{A}",
    "r2":    "v_add_f32 v{M}, 0xcb400000, v{P0}",
    "r3":    "v_sub_f32 v{M}, v{A}, v{M}",
    "cmpf":  "v_cmp_nlt_f32_e64 s[42:43], |v{M}|, s41",
    "val":   "v_add_u32 v{P0}, v{O}, v{P0}",
    "sdwa":  "v_sub_u32_sdwa v{P0}, v70, v57 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1",
}
COMPILED = ["mad", "lshr", "add1", "add2", "min", "cvtu", "mbl", "mullo", "mbh", "u1", "log", "cvti", "u2a", "lsa", "fmamk", "u2b", "sqrt", "cos",
            "mov", "cmpu", "g", "dig", "r1", "r2", "r3", "cmpf", "val", "sdwa"]
NSTEP = 4
SHARED = {57: 48, 60: 49, 61: 50, 62: 51, 70: 52, 82: 53}


def regs(s):
    b = 8 + 10 * s
    return dict(P0=b, P1=b + 1, M=b + 2, T=b + 3, U=b + 4, W=b + 5, A=b + 6, S=b + 7, O=b + 8, X=b + 9)


def emit(name, s):
    x = I[name].format(**regs(s))
    for k, v in SHARED.items():
        x = x.replace(f"v{k}", f"v{v}")
    return x


def by_step(groups, nstep=NSTEP):
    """for each group: step 0's instructions of the group, then step 1's ..."""
    return [emit(n, s) for g in groups for s in range(nstep) for n in g]


def by_instr(groups, nstep=NSTEP):
    """for each group: each instruction for all steps in turn"""
    return [emit(n, s) for g in groups for n in g for s in range(nstep)]


def schedules():
    S = []
    S.append(("compiled", by_step([COMPILED]), "four steps one after the other, each in the compiler's order"))
    S.append(("interleave4", by_instr([COMPILED]), "the compiler's order, the four steps instruction by instruction"))
    S.append(("interleave2", [emit(n, s) for pair in ((0, 1), (2, 3)) for n in COMPILED for s in pair], "steps 0|1 interleaved, then steps 2|3 interleaved"))
    lcg = ["mad", "lshr", "add1", "add2", "min"]
    uni = ["cvtu", "mullo", "u1"]
    uni2 = ["cvti", "u2a", "u2b"]
    idx = ["mbl", "mbh", "lsa", "mov"]
    tail = ["cmpu", "g", "dig", "r1", "r2", "r3", "cmpf", "val", "sdwa"]
    S.append(("phases", by_instr([lcg, uni, ["log"], uni2, ["cos"], ["fmamk"], ["sqrt"], idx, tail]), "phases over the four steps; log x4, cos x4, fmamk x4, sqrt x4"))
    S.append(("phases_t8", by_instr([lcg, uni, uni2, ["cos"], ["log"], ["fmamk"], ["sqrt"], idx, tail]), "cos x4 + log x4 back to back (u1 kept in a copy: one v_mov more, not counted)"))
    S.append(("phases_idx_mid", by_instr([lcg, uni, ["log"], uni2, idx, ["cos"], ["fmamk"], ["sqrt"], tail]), "the index work between log and cos"))
    # pair the slow non-transcendental instructions with plain ones explicitly: alternate classes inside the phases
    S.append(("phases_zip", by_instr([["mad", "lshr"], ["mbl", "add1"], ["mbh", "add2"], ["min", "mov"], ["cvtu", "lsa"], ["mullo", "u1"], ["log"], ["cvti", "u2a"], ["u2b"], ["cos"],
                                      ["fmamk"], ["sqrt"], ["cmpu", "g"], ["dig", "r1"], ["r2", "r3"], ["cmpf", "val"], ["sdwa"]]), "phases, slow ops next to plain ops"))
    S.append(("trans_last", by_instr([lcg, uni, uni2, idx, ["cmpu", "mov"]]) + by_instr([["log"], ["cos"], ["fmamk"], ["sqrt"]]) + by_instr([["g", "dig", "r1", "r2", "r3", "cmpf", "val", "sdwa"]]),
              "everything before the transcendentals first, then 4 x (log, cos, fmamk, sqrt), then the tails"))
    S.append(("by_step_trans_grouped", by_step([lcg + uni + uni2 + idx, ["log", "cos", "fmamk", "sqrt"], tail]), "per step, but its three transcendentals together"))
    S.append(("no_sdwa", by_instr([lcg, uni, ["log"], uni2, ["cos"], ["fmamk"], ["sqrt"], idx, tail[:-1]]), "phases without the SDWA subtraction (27 instructions per step)"))
    return S


def main():
    lines = ["// generated by tools/ubench6_gen.py -- do not edit", "#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <cstring>", ""]
    clob = ", ".join(f'"v{i}"' for i in range(8, 56)) + ', "vcc", "s40", "s41", "s42", "s43"'
    init = "\\n\\t".join([f"v_mov_b32 v{i}, 1.0" for i in range(8, 56)] + ["s_mov_b32 s40, 0x3f8ccccd", "s_mov_b32 s41, 0x3f8ccccd"])
    meta = []
    yard = ["v_mul_lo_u32 v%d, v%d, v48" % (8 + i % 8, 8 + i % 8) for i in range(4 * 28)]
    for name, seq, note in [("mul_lo", yard, "yardstick: 112 x v_mul_lo_u32")] + schedules():
        b = "\\n\\t".join(seq)
        kn = "k_" + name
        lines.append(f"__global__ __launch_bounds__(64) void {kn}(float* out, int iters) {{")
        lines.append(f'    asm volatile("{init}" ::: {clob});')
        lines.append("    for (int it = 0; it < iters; it++)")
        lines.append(f'        asm volatile("{b}" ::: {clob});')
        lines.append('    float r; asm volatile("v_add_f32 %0, v8, v16" : "=v"(r));')
        lines.append("    out[blockIdx.x * 64 + threadIdx.x] = r;")
        lines.append("}")
        meta.append((kn, name, len(seq), note))
    lines.append("""
typedef void (*kern_t)(float*, int);
static double run(kern_t k, float* out, int w) {
    const int iters = 512;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * 4 * w;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(a); hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best * 1e-3 * 2.4e9 / ((double)iters * w);          // nominal cycles per iteration (four steps) per SIMD
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    struct T { kern_t k; const char* name; int n; const char* note; };
    const T tests[] = {""")
    for kn, name, n, note in meta:
        lines.append(f'        {{{kn}, "{name}", {n}, "{note}"}},')
    lines.append("""    };
    const int ws[] = {2, 4, 6, 7, 8};
    double yard[9] = {0};
    for (int w : ws) yard[w] = run(tests[0].k, out, w) / tests[0].n;       // nominal cycles per v_mul_lo_u32
    printf("cycles per STEP per SIMD (an iteration of four steps / 4) in units of v_mul_lo_u32 / 4, by wavefronts per SIMD\\n");
    printf("%-22s %4s | %7s %7s %7s %7s %7s | %s\\n", "schedule", "n", "w=2", "w=4", "w=6", "w=7", "w=8", "");
    for (const T& t : tests) {
        double c[9];
        for (int w : ws) c[w] = 4 * run(t.k, out, w) / yard[w] / 4;
        printf("%-22s %4d | %7.1f %7.1f %7.1f %7.1f %7.1f | %s\\n", t.name, t.n, c[2], c[4], c[6], c[7], c[8], t.note);
    }
    return 0;
}""")
    with open(os.path.join(ROOT, "tools", "ubench6.hip"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
