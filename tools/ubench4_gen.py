#!/usr/bin/env python3
"""Writes tools/ubench4.hip: which FORMS of the plain fp32 / int32 instructions issue at the fast (~2-cycle) rate on gfx950, and what a mix costs.

tools/ubench3.hip measured one instruction at a time, always with a single VGPR source (`v_add_f32 v, 1.0, v`).  k_samples_lean's
counters say its instructions cost 4.3 cycles on average where that table predicts 3.7: this bench varies what ubench3 held fixed --
the number of distinct VGPR sources, their register banks (index mod 4), SGPR / literal sources, VOP2 against VOP3 encodings,
alternation with half-rate and transcendental instructions, wavefronts per SIMD, and chains per wavefront.  Every body is explicit
registers in one asm block (v8..v59 are the bench's; the compiler is told they are clobbered), so what runs is what is written.

    python tools/ubench4_gen.py && hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench4 tools/ubench4.hip && /tmp/ubench4

Prints, per body and per wavefronts-per-SIMD: wall time x nominal 2.4 GHz / instructions issued per SIMD (as ubench3), and the same
relative to v_mul_lo_u32 (x4: a half-rate instruction is taken to be one full 4-cycle pass)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCH = 8
REPS = 4


def body(tmpl_list, nch=NCH, reps=REPS):
    """tmpl_list: templates applied round-robin per chain; {d} chain register, {a}/{b}/{c} aux registers with chosen bank offsets"""
    out = []
    n = 0
    for r in range(reps):
        for i in range(nch):
            for t in tmpl_list:
                d = 8 + i                                                      # (v8..v59: eight wavefronts fit a SIMD)
                out.append(t.format(d=f"v{d}", e=f"v{16 + i}",
                                    a0=f"v{24 + i}", b0=f"v{32 + i}",          # same bank as d (and as each other)
                                    a1=f"v{25 + i}", b2=f"v{34 + i}",          # banks d+1, d+2
                                    dd=f"v[{44 + 2 * i}:{45 + 2 * i}]",         # a 64-bit chain register
                                    s="s40", s2="s41"))
                n += 1
    return "\\n\\t".join(out), n


TESTS = [
    # name, templates, note
    ("add_lit", ["v_add_f32 {d}, 1.0, {d}"], "ubench3's form: one VGPR source"),
    ("add_vv_samebank", ["v_add_f32 {d}, {a0}, {d}"], "two VGPR sources, same bank"),
    ("add_vv_otherbank", ["v_add_f32 {d}, {a1}, {d}"], "two VGPR sources, banks differ"),
    ("add_sv", ["v_add_f32 {d}, {s}, {d}"], "SGPR + VGPR"),
    ("add_nochain", ["v_add_f32 {d}, {a1}, {b2}"], "destination is not a source: pure issue"),
    ("mul_vv", ["v_mul_f32 {d}, {a1}, {d}"], ""),
    ("fma_ddd", ["v_fma_f32 {d}, {d}, {d}, {d}"], "ubench3's form"),
    ("fma_3v_samebank", ["v_fma_f32 {d}, {a0}, {b0}, {d}"], "three distinct VGPRs in one bank"),
    ("fma_3v_otherbank", ["v_fma_f32 {d}, {a1}, {b2}, {d}"], "three distinct VGPRs in three banks"),
    ("fma_svv", ["v_fma_f32 {d}, {s}, {a1}, {d}"], "SGPR, VGPR, VGPR"),
    ("fma_ssv", ["v_fma_f32 {d}, {s}, {s}, {d}"], ""),
    ("fmac_3v", ["v_fmac_f32 {d}, {a1}, {b2}"], "VOP2, three distinct VGPRs"),
    ("fmac_2v", ["v_fmac_f32 {d}, {a1}, {a1}"], ""),
    ("fmamk", ["v_fmamk_f32 {d}, {a1}, 0x3f000000, {d}"], "literal"),
    ("fmaak", ["v_fmaak_f32 {d}, {a1}, {d}, 0x3f000000"], "literal"),
    ("add_u32_vv", ["v_add_u32 {d}, {a1}, {d}"], ""),
    ("and_vv", ["v_and_b32 {d}, {a1}, {d}"], ""),
    ("lshr", ["v_lshrrev_b32 {d}, 3, {d}"], ""),
    ("mov_vv", ["v_mov_b32 {d}, {a1}"], ""),
    ("min_u32", ["v_min_u32 {d}, {a1}, {d}"], ""),
    ("mul_lo", ["v_mul_lo_u32 {d}, {d}, {a1}"], "the yardstick: 4 cycles"),
    ("cvt_f32_u32", ["v_cvt_f32_u32 {d}, {d}"], ""),
    ("cmp_f32", ["v_cmp_lt_f32 vcc, {a1}, {d}"], ""),
    ("mbcnt", ["v_mbcnt_lo_u32_b32 {d}, -1, {d}"], ""),
    ("lshl_add", ["v_lshl_add_u32 {d}, {d}, 3, {a1}"], ""),
    ("mad_u64", ["v_mad_u64_u32 {dd}, vcc, {d}, {a1}, 0"], ""),
    ("log", ["v_log_f32 {d}, {d}"], ""),
    ("sqrt", ["v_sqrt_f32 {d}, {d}"], ""),
    ("cos", ["v_cos_f32 {d}, {d}"], ""),
    ("pk_fma", ["v_pk_fma_f32 {dd}, {dd}, {dd}, {dd}"], "two fp32 per lane"),
    ("pk_mul", ["v_pk_mul_f32 {dd}, {dd}, {dd}"], ""),
    ("pk_add", ["v_pk_add_f32 {dd}, {dd}, {dd}"], ""),
    # alternations (two chains: {d} and {e}); the per-instruction figure is the mean of the pair
    ("add+add", ["v_add_f32 {d}, 1.0, {d}", "v_add_f32 {e}, 1.0, {e}"], ""),
    ("add+mul_lo", ["v_add_f32 {d}, 1.0, {d}", "v_mul_lo_u32 {e}, {e}, {a1}"], "fast next to half-rate: (2+4)/2 if they add"),
    ("add+cvt", ["v_add_f32 {d}, 1.0, {d}", "v_cvt_f32_u32 {e}, {e}"], ""),
    ("add+cmp", ["v_add_f32 {d}, 1.0, {d}", "v_cmp_lt_f32 vcc, {a1}, {e}"], ""),
    ("add+log", ["v_add_f32 {d}, 1.0, {d}", "v_log_f32 {e}, {e}"], "fast next to a transcendental"),
    ("3add+log", ["v_add_f32 {d}, 1.0, {d}", "v_add_f32 {d}, 1.0, {d}", "v_add_f32 {d}, 1.0, {d}", "v_log_f32 {e}, {e}"], ""),
    ("7add+log", ["v_add_f32 {d}, 1.0, {d}"] * 7 + ["v_log_f32 {e}, {e}"], ""),
    ("15add+log", ["v_add_f32 {d}, 1.0, {d}"] * 15 + ["v_log_f32 {e}, {e}"], ""),
    ("cvt+log", ["v_cvt_f32_u32 {d}, {d}", "v_log_f32 {e}, {e}"], ""),
    ("add+cvt+log", ["v_add_f32 {d}, 1.0, {d}", "v_cvt_f32_u32 {d}, {d}", "v_log_f32 {e}, {e}"], ""),
    ("3add+3cvt", ["v_add_f32 {d}, 1.0, {d}"] * 3 + ["v_cvt_f32_u32 {e}, {e}"] * 3, "grouped instead of alternating"),
    ("2add+cvt", ["v_add_f32 {d}, 1.0, {d}", "v_add_f32 {d}, 1.0, {d}", "v_cvt_f32_u32 {e}, {e}"], ""),
    ("add+mbcnt", ["v_add_f32 {d}, 1.0, {d}", "v_mbcnt_lo_u32_b32 {e}, -1, {e}"], ""),
    ("add+mad_u64", ["v_add_f32 {d}, 1.0, {d}", "v_mad_u64_u32 {dd}, vcc, {e}, {a1}, 0"], ""),
    ("add+min", ["v_add_f32 {d}, 1.0, {d}", "v_min_u32 {e}, {a1}, {e}"], ""),
    ("add+add_sv", ["v_add_f32 {d}, 1.0, {d}", "v_add_f32 {e}, {s}, {e}"], ""),
    ("add+lshl_add", ["v_add_f32 {d}, 1.0, {d}", "v_lshl_add_u32 {e}, {e}, 3, {a1}"], ""),
    ("add+sdwa", ["v_add_f32 {d}, 1.0, {d}", "v_sub_u32_sdwa {e}, {a1}, {e} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"], ""),
    ("sdwa", ["v_sub_u32_sdwa {d}, {a1}, {d} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"], ""),
    ("cndmask", ["v_cndmask_b32 {d}, {a1}, {d}, vcc"], "select by VCC (never written in the loop)"),
    ("cndmask_lit", ["v_cndmask_b32 {d}, 0, {d}, vcc"], "ubench3's form"),
    ("cndmask_sgpr", ["v_cndmask_b32_e64 {d}, {a1}, {d}, s[42:43]"], "select by an SGPR pair"),
    ("cndmask_exec", ["v_cndmask_b32_e64 {d}, {a1}, {d}, exec"], ""),
    ("add+cndmask", ["v_add_f32 {d}, 1.0, {d}", "v_cndmask_b32 {e}, {a1}, {e}, vcc"], ""),
    ("cmp+cndmask", ["v_cmp_lt_f32 vcc, {a1}, {d}", "v_cndmask_b32 {e}, {a1}, {e}, vcc"], "the usual pair"),
    ("cmp+add+cndmask", ["v_cmp_lt_f32 vcc, {a1}, {d}", "v_add_f32 {d}, 1.0, {d}", "v_cndmask_b32 {e}, {a1}, {e}, vcc"], ""),
    ("cmp_e64+cndmask_e64", ["v_cmp_lt_f32_e64 s[42:43], {a1}, {d}", "v_cndmask_b32_e64 {e}, {a1}, {e}, s[42:43]"], ""),
    ("cvt+cmp", ["v_cvt_f32_u32 {d}, {d}", "v_cmp_lt_f32 vcc, {a1}, {e}"], ""),
    ("cvt+mul_lo", ["v_cvt_f32_u32 {d}, {d}", "v_mul_lo_u32 {e}, {e}, {a1}"], ""),
    ("mul_lo+log", ["v_mul_lo_u32 {d}, {d}, {a1}", "v_log_f32 {e}, {e}"], ""),
    ("log+sqrt+cos", ["v_log_f32 {d}, {d}", "v_sqrt_f32 {e}, {e}", "v_cos_f32 {d}, {d}"], ""),
    ("add+fma3", ["v_add_f32 {d}, {a1}, {d}", "v_fma_f32 {e}, {a1}, {b2}, {e}"], ""),
    ("add_dep_e", ["v_add_f32 {d}, 1.0, {e}", "v_add_f32 {e}, 1.0, {d}"], "each instruction reads the previous one's result"),
]


def main():
    lines = ["// generated by tools/ubench4_gen.py -- do not edit", "#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <cstring>", ""]
    clob = ", ".join(f'"v{i}"' for i in range(8, 60)) + ', "vcc", "s40", "s41", "s42", "s43"'
    init = "\\n\\t".join([f"v_mov_b32 v{i}, 1.0" for i in range(8, 60)] + ["s_mov_b32 s40, 0x3f8ccccd", "s_mov_b32 s41, 0x3f8ccccd"])
    meta = []
    for nch in (NCH, 1):
        for name, tl, note in TESTS:
            if nch == 1 and name not in ("add_lit", "fma_3v_otherbank", "mul_lo", "log", "add+mul_lo", "add+log"):
                continue
            b, n = body(tl, nch=nch, reps=REPS * (NCH // nch))
            kn = f"k_{name.replace('+', '_')}_c{nch}".replace("-", "_")
            lines.append(f"__global__ __launch_bounds__(64) void {kn}(float* out, int iters) {{")
            lines.append(f'    asm volatile("{init}" ::: {clob});')
            lines.append("    for (int it = 0; it < iters; it++)")
            lines.append(f'        asm volatile("{b}" ::: {clob});')
            lines.append('    float r; asm volatile("v_add_f32 %0, v8, v16" : "=v"(r));')
            lines.append("    out[blockIdx.x * 64 + threadIdx.x] = r;")
            lines.append("}")
            meta.append((kn, name, nch, n, note))
    lines.append("""
typedef void (*kern_t)(float*, int);
static double run(kern_t k, float* out, int w, int n_per_iter) {
    const int iters = 2048;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * 4 * w;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(a); hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best * 1e-3 * 2.4e9 / ((double)iters * n_per_iter * w);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    struct T { kern_t k; const char* name; int nch; int n; const char* note; };
    const T tests[] = {""")
    for kn, name, nch, n, note in meta:
        lines.append(f'        {{{kn}, "{name}", {nch}, {n}, "{note}"}},')
    lines.append("""    };
    double yard[9] = {0};
    for (const T& t : tests) if (!strcmp(t.name, "mul_lo") && t.nch == 8) for (int w : {1, 2, 4, 8}) yard[w] = run(t.k, out, w, t.n);
    printf("cycles per wave-instruction per SIMD at a nominal 2.4 GHz | the same in units of v_mul_lo_u32 / 4, per wavefronts per SIMD\\n");
    printf("%-20s %2s | %6s %6s %6s %6s | %6s %6s %6s %6s | %s\\n", "body", "ch", "w=1", "w=2", "w=4", "w=8", "w=1", "w=2", "w=4", "w=8", "");
    for (const T& t : tests) {
        double c[9];
        for (int w : {1, 2, 4, 8}) c[w] = run(t.k, out, w, t.n);
        printf("%-20s %2d | %6.2f %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f %6.2f | %s\\n", t.name, t.nch, c[1], c[2], c[4], c[8],
               4 * c[1] / yard[1], 4 * c[2] / yard[2], 4 * c[4] / yard[4], 4 * c[8] / yard[8], t.note);
    }
    return 0;
}""")
    with open(os.path.join(ROOT, "tools", "ubench4.hip"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
