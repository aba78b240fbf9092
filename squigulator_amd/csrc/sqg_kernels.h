// sqg_kernels.h -- gfx950 device code of the per-read signal path (included by sqg_hip.hip).
//
//   k_init_rows   per-(worker,k-mer) stream seeds                       (src/sim.c:238-257)
//   k_dwell       per-event dwell draw from the worker's time stream   (src/gensig.c:254-257)
//   k_scan        read lengths -> output offsets
//   k_signal      one wavefront per worker chain: ranks, in-order stream hand-out, samples
//                                                                      (src/gensig.c:226-356)
//   k_fixup       FP64 recomputation of the samples the certified fp32 path could not decide
//   k_certify     exhaustive error sweep of the fp32 normal-deviate path over all 2^31-2 states
//   k_store_probe int16 streaming-store ceiling
//
// Arithmetic modes.  EXACT: every draw goes through the FP64 restatement of nrng()
// (src/rand.h:87-94).  CERTIFIED: a draw is first evaluated with fp32 hardware transcendentals;
// the result is accepted only if the digitised value provably cannot differ from the FP64 one
// (|frac - 1/2| test against a bound built from the swept error delta_x), and
// is otherwise recomputed in FP64.  Both modes produce identical int16 streams.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define LCG_M 2147483647u
#define LCG_A 16807u

#define POW_N 1024          // entries per jump table
// d_pow layout (uint32 each):
//   [0*POW_N + j] = a^(2j+1)   first draw of sample/event j after a base state
//   [1*POW_N + j] = a^(2j+2)   second draw
//   [2*POW_N + j] = a^(2j)     jump over j draws-pairs
//   [3*POW_N + j] = a^(2*1024*j)
//   [4*POW_N + j] = a^(2*1024*1024*j)
#define POW_TABLES 5

#define NEAR_ONE_BITS 17    // c1 > M - 2^17 (u within 6e-5 of 1): always taken to the FP64 path

// ---- MINSTD in canonical form: c' = a*c mod (2^31-1), c in [1, M-1] -------------------------
__host__ __device__ static inline uint32_t lcg_mul(uint32_t a, uint32_t b) {
    const unsigned long long p = (unsigned long long)a * b;
    uint32_t r = (uint32_t)(p & LCG_M) + (uint32_t)(p >> 31);
    r = (r & LCG_M) + (r >> 31);
    return r;
}
// same product, result only reduced to [0, 2^32) (congruent mod M): enough for the cosine argument
__device__ static inline uint32_t lcg_mul_lazy(uint32_t a, uint32_t b) {
    const unsigned long long p = (unsigned long long)a * b;
    return (uint32_t)(p & LCG_M) + (uint32_t)(p >> 31);
}

// a^(2n) for n < 2^30 from three table levels
__device__ static inline uint32_t lcg_jump2(const uint32_t* __restrict__ pw, uint32_t n) {
    uint32_t r = pw[2 * POW_N + (n & (POW_N - 1))];
    const uint32_t hi = (n >> 10) & (POW_N - 1), hi2 = n >> 20;
    if (hi) r = lcg_mul(r, pw[3 * POW_N + hi]);
    if (hi2) r = lcg_mul(r, pw[4 * POW_N + hi2]);
    return r;
}

// (double)x/2147483647 with the reference's corrected state (src/rand.h:82-84)
__device__ static inline double lcg_uniform(uint32_t c) {
    return (double)(c ? c : LCG_M) / 2147483647.0;
}

// nrng body, src/rand.h:87-94, for two consecutive draws c1, c2 (FP64, no contraction)
__device__ static inline double box_muller_exact(uint32_t c1, uint32_t c2) {
    const double u = lcg_uniform(c1);
    const double t = (2.0 * 3.14159265) * lcg_uniform(c2);
    return sqrt(-2.0 * log(u)) * cos(t);
}

// fp32 evaluation of the same deviate.  c1 canonical; r2 = second draw, any representative in
// [0, 2^32) of its residue.  v_log_f32 is log2, v_cos_f32 takes turns.  The 6.2831853-vs-2*pi
// ratio (1 - 1.1e-9) is below fp32 resolution; the sweep prices it with everything else.
__device__ static inline float box_muller_fast(uint32_t c1, uint32_t r2) {
    const float uf = (float)c1 * 4.656612873077393e-10f;                  // c1 * 2^-31 (exact scaling)
    const float lg = __builtin_amdgcn_logf(uf);
    const float y = __builtin_fmaf(lg, -1.3862943611198906f, -9.313225750491594e-10f);   // -2 ln(c1/M)
    const float r = __builtin_amdgcn_sqrtf(y);
    const float cs = __builtin_amdgcn_cosf((float)r2 * 4.656612873077393e-10f);
    return r * cs;
}

// (int16_t)double as gcc/x86-64 lowers it (cvttsd2si r32, low half): src/gensig.c:270
__device__ static inline int16_t to_i16(double v) {
    int32_t t;
    if (v > -2147483649.0 && v < 2147483648.0) t = (int32_t)v; else t = (int32_t)0x80000000u;
    return (int16_t)(uint16_t)((uint32_t)t & 0xffffu);
}

// one sample, FP64 path: float s = nrng(...); raw = s*dig/range - offset  (src/gensig.c:264-270)
__device__ static inline int16_t sample_exact(uint32_t c1, float m, float sd, double dig, double range, double offset) {
    const double z = box_muller_exact(c1, lcg_mul(c1, LCG_A));
    const float s = (float)((z * (double)sd) + (double)m);
    return to_i16((double)s * dig / range - offset);
}

// base -> 2-bit code, src/seq.h:14-27
__host__ __device__ static inline uint32_t base_code(uint8_t b) {
    switch (b) {
    case 'C': case 'c': case 'Y': case 'B': return 1;
    case 'G': case 'g': case 'S': case 'K': return 2;
    case 'T': case 't': case 'U': return 3;
    default: return 0;   // A a R W M D H V and anything unknown
    }
}

// ---- descriptors ---------------------------------------------------------------------------
struct ReadDesc {
    long long base_off;   // first byte of segment 0 in the batch's base buffer
    long long ev_off;     // first event of this read in the batch's event arrays
    double offset;        // slow5 offset of this read (drawn on the host)
    int len0, len1;       // bytes in segment 0 (read incl. attached prefix) and 1 (RNA stall)
    int ne0, ne1;         // events per segment
    int worker;           // context-local worker index
    uint32_t time_c0;     // worker's time-stream state at the start of this read
};

struct FixEntry {         // one sample handed to the FP64 path
    long long at;         // absolute index into the signal slab
    long long ev;         // batch-wide event index (k-mer recomputed from the bases)
    uint32_t c1;          // first draw of the sample
    int read;
    int shifted;          // inside the RNA adaptor level-shift window
    int pad;
};

struct SigParams {
    const ReadDesc* reads;
    const int* chain_off;        // [n_chains+1]
    const int* chain_reads;      // read indices grouped per worker chain, batch order inside a chain
    const int* chain_order;      // launch order (longest chain first)
    const uint8_t* bases;
    const uint16_t* dwell;       // per event (null when dwell is constant)
    const unsigned long long* seglen;  // [2*n_reads] samples in segment 0 / 1
    const long long* sig_off;    // [n_reads+1]
    const float2* model;         // {level_mean, (float)(level_stdv*amp_noise)}
    const uint32_t* pw;
    uint32_t* rows;              // [n_local_workers][num_kmer]
    int16_t* sig;
    unsigned int* err;
    FixEntry* fix;               // certified mode: undecided samples
    unsigned int* fix_count;
    unsigned int fix_cap;
    double dig, range, kd;       // kd = dig/range
    float delta_x;               // swept bound on |x_fast - x_exact| (incl. margin)
    int k, num_kmer;
    int const_sps;               // (int)dwell_mean, used when dwell == null
    int use_streams;             // 0 in --ideal / --ideal-amp (src/gensig.c:265-269)
    int rna;                     // reverse the signal (src/gensig.c:348-354)
    int shift_len;               // RNA+prefix: 79*(int)dwell_mean samples get -shift (src/genread.c:79-86)
    int shift;                   // (int16)(30*dig/range)
};

// ---- k_init_rows ---------------------------------------------------------------------------
__global__ void k_init_rows(uint32_t* rows, int num_kmer, long long seed, int worker_lo, long long n_total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const long long w = i / num_kmer, j = i % num_kmer;
    long long s = seed + (w + worker_lo) * ((long long)num_kmer + 10) + j;
    s %= (long long)LCG_M;
    if (s < 0) s += LCG_M;
    rows[i] = (uint32_t)s;
}

// ---- k_dwell: one thread per event of the batch --------------------------------------------
// sps = round(nrng(rand_time)); sps = sps<1 ? -sps+1 : sps           (src/gensig.c:255-256)
template <int MODE>
__global__ __launch_bounds__(256) void k_dwell(const ReadDesc* __restrict__ reads, int n_reads,
                                               const int* __restrict__ blk_read, long long n_events,
                                               const uint32_t* __restrict__ pw, double dmean, double dstd,
                                               float delta_x,
                                               uint16_t* __restrict__ dwell,
                                               unsigned long long* __restrict__ seglen,
                                               unsigned int* __restrict__ err) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = gid < n_events;
    int r = blk_read[blockIdx.x];
    int sps = 0, seg = 0;
    if (valid) {
        while (r + 1 < n_reads && gid >= reads[r + 1].ev_off) r++;
        const uint32_t e = (uint32_t)(gid - reads[r].ev_off);
        const uint32_t c = lcg_mul(reads[r].time_c0, lcg_jump2(pw, e));
        const uint32_t c1 = lcg_mul(c, LCG_A);
        bool decided = false;
        if (MODE == 1) {
            // v' = x'*s + m in fp32; round(v) = floor(v+1/2) unless v is within eps of a half-integer
            const float sf = (float)dstd, mf = (float)dmean;
            const float x = box_muller_fast(c1, lcg_mul_lazy(c1, LCG_A));
            const float g = __builtin_fmaf(x, sf, mf) + 0.5f;
            const float fl = floorf(g);
            const float fr = g - fl;
            const float mag = fabsf(mf) + 7.0f * fabsf(sf) + 1.0f;
            // delta_x*s (swept) + float roundings of s, m, the fma and the +1/2 (each <= 2^-24 * mag) + slack
            const float eps = delta_x * fabsf(sf) + 4.0f * 5.9604645e-8f * mag + 1e-6f;
            if (fabsf(fr - 0.5f) < 0.5f - eps && c1 <= LCG_M - (1u << NEAR_ONE_BITS) && fabsf(g) < 1.0e6f) {
                sps = (int)fl;
                decided = true;
            }
        }
        if (!decided) {
            const double z = box_muller_exact(c1, lcg_mul(c1, LCG_A));
            const double v = (z * dstd) + dmean;                 // nrng: (x * s) + m
            sps = (int)round(v);                                 // src/gensig.c:255
        }
        sps = sps < 1 ? -sps + 1 : sps;                          // src/gensig.c:256
        if (sps > 65535) { atomicOr(err, 1u); sps = 65535; }
        dwell[gid] = (uint16_t)sps;
        seg = e >= (uint32_t)reads[r].ne0;
    }
    // per-read totals: one atomic per wavefront when the wave is inside one (read, segment)
    const int key = valid ? (r * 2 + seg) : -1;
    const int key0 = __shfl(key, 0);
    if (__all(key == key0)) {
        int s = sps;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        if ((threadIdx.x & 63) == 0 && key0 >= 0) atomicAdd(&seglen[key0], (unsigned long long)s);
    } else if (valid) {
        atomicAdd(&seglen[key], (unsigned long long)sps);
    }
}

// ---- k_scan: sig_off = exclusive scan of per-read totals (single workgroup) -----------------
__global__ __launch_bounds__(1024) void k_scan(const unsigned long long* __restrict__ seglen, int n_reads,
                                               long long* __restrict__ sig_off, unsigned int* __restrict__ err) {
    __shared__ long long wsum[16];
    __shared__ long long carry;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_reads; base += 1024) {
        const int i = base + tid;
        long long v = 0;
        if (i < n_reads) {
            v = (long long)(seglen[2 * i] + seglen[2 * i + 1]);
            if (v >= 4294967295LL) atomicOr(err, 2u);        // src/sim.c:559-562
        }
        long long x = v;
        for (int o = 1; o < 64; o <<= 1) { long long y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) wsum[wid] = x;
        __syncthreads();
        long long woff = 0;
        for (int w = 0; w < wid; w++) woff += wsum[w];
        const long long c = carry;
        if (i < n_reads) sig_off[i] = c + woff + x - v;
        __syncthreads();
        if (tid == 1023) carry = c + woff + x;
        __syncthreads();
    }
    if (tid == 0) sig_off[n_reads] = carry;
}

// ---- k_signal ------------------------------------------------------------------------------
// One workgroup of NT threads per worker chain (a worker's reads of this batch, in batch order).
// A read is walked in segments of NT consecutive events, one event per thread:
//   event phase  (whole workgroup): k-mer rank, dwell, block scan -> first sample of each event;
//                hand-out of the per-(worker,k-mer) Lehmer streams IN EVENT ORDER: events are
//                binned by k-mer in an LDS hash table, bin members listed via a block scan, and
//                each event sums the dwell of the same-k-mer events before it (bins hold 1-3
//                events).  Stream states live in HBM/L2 (rows[worker][rank]); one load and, for
//                the last event of a bin, one store per event, by an O(1) jump a^(2*samples).
//   sample phase (per wavefront, no block barriers): wave w emits the samples of events
//                [64w, 64w+64) of the segment, 64 consecutive samples per step (contiguous int16
//                stores).  sample -> event through start-marker bytes in LDS + ballot/mbcnt; the
//                two draws of a sample are two modular multiplications of the event's state with
//                per-slot constants a^(2j+1), a^(2j+2) held in LDS.
#ifndef SQG_SIGNAL_WAVES_PER_SIMD
#define SQG_SIGNAL_WAVES_PER_SIMD 4
#endif
#define MK_W 1024          // marker window (samples) per wavefront
#define MULT_N 256         // LDS jump constants cover events of up to 256 samples
#define BIN_EMPTY 0xffffffffu

template <int NT>
struct SigLds {
    uint4 rec_a[NT];            // {c_ev, first sample in segment, F | level_mean, sdk | sd}
    uint2 rec_b[NT];            // {I | constant sample, thr | rank}
    uint32_t keys[2 * NT];      // hash bins: k-mer rank
    uint32_t bins[2 * NT];      // events per bin; after the scan (first member slot << 16) | count
    uint32_t mem[NT];           // bin members: (event index in segment << 16) | dwell
    uint2 mult[MULT_N];         // {a^(2j+1), a^(2j+2)}
    uint32_t jump[MULT_N];      // a^(2j)
    uint8_t mk[NT / 64][MK_W];  // event-start markers, one window per wavefront
    uint8_t lut[256];           // base -> 2-bit code
    int wsum[NT / 64];
    int wsum2[NT / 64];
};

__device__ static inline int wave_incl_scan(int v, int lane) {
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(v, o); if (lane >= o) v += y; }
    return v;
}

template <int NT>
__device__ static inline uint32_t jump2_lds(const SigLds<NT>& L, const uint32_t* __restrict__ pw, uint32_t n) {
    return n < MULT_N ? L.jump[n] : lcg_jump2(pw, n);
}

// generic per-sample emitter: every option, both modes (used for tiles the fast loop excludes)
template <int MODE, int NT>
__device__ static void emit_generic(const SigParams& P, SigLds<NT>& L, int lane, int ev0, uint8_t* mk, bool valid,
                                    int so_w, int wave_total, uint32_t base_pos, uint32_t read_len, int16_t* out,
                                    long long sig_base, int r, long long ev_first, double offset,
                                    long long shift_lo, long long n1, bool shift_tile) {
    const unsigned long long lane_le = (lane == 63) ? ~0ull : ((2ull << lane) - 1);
    for (int w0 = 0; w0 < wave_total; w0 += MK_W) {
        ((uint4*)mk)[lane] = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (valid && so_w >= w0 && so_w < w0 + MK_W) mk[so_w - w0] = 1;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        int base_ev = __popcll(__ballot(valid && so_w < w0));
        const int w_end = min(w0 + MK_W, wave_total);
        for (int c0 = w0; c0 < w_end; c0 += 64) {
            const int idx = c0 + lane;
            const unsigned long long sm = __ballot(mk[idx - w0] != 0);
            const int ev = ev0 + base_ev + __popcll(sm & lane_le) - 1;
            base_ev += __popcll(sm);
            if (idx < w_end) {
                const uint4 ra = L.rec_a[ev];
                const uint2 rb = L.rec_b[ev];
                const uint32_t j = (uint32_t)idx - (ra.y - (uint32_t)(L.rec_a[ev0].y));   // sample within event
                const uint32_t pos = base_pos + (uint32_t)idx;
                const uint32_t at = P.rna ? (read_len - 1 - pos) : pos;
                const bool in_shift = shift_tile && (long long)pos >= shift_lo && (long long)pos < n1;
                int16_t q;
                bool ok = true;
                uint32_t c1 = 1;
                if (!P.use_streams) {
                    q = (int16_t)(uint16_t)rb.x;
                } else {
                    if (j < MULT_N) c1 = lcg_mul(ra.x, L.mult[j].x);
                    else c1 = lcg_mul(lcg_mul(ra.x, lcg_jump2(P.pw, j)), LCG_A);
                    if (MODE == 1) {
                        const float x = box_muller_fast(c1, lcg_mul_lazy(c1, LCG_A));
                        const float v = __builtin_fmaf(x, __uint_as_float(ra.w), __uint_as_float(ra.z));
                        const float fl = floorf(v);
                        const float fr = v - fl;
                        ok = fabsf(fr - 0.5f) < __uint_as_float(rb.y) && c1 <= LCG_M - (1u << NEAR_ONE_BITS);
                        int n = (int)rb.x + (int)fl;
                        n -= n >> 31;                                      // truncation toward zero (value is not an integer)
                        q = (int16_t)(uint16_t)((uint32_t)n & 0xffffu);
                    } else {
                        const double z = box_muller_exact(c1, lcg_mul(c1, LCG_A));
                        const float sv = (float)((z * (double)__uint_as_float(ra.w)) + (double)__uint_as_float(ra.z));   // src/gensig.c:268
                        q = to_i16((double)sv * P.dig / P.range - offset);                                             // src/gensig.c:270
                    }
                }
                if (in_shift) q = (int16_t)(uint16_t)(((int)q - P.shift) & 0xffff);
                if (ok) out[at] = q;
                if (MODE == 1) {
                    const unsigned long long am = __ballot(!ok);
                    if (am) {                                              // hand the undecided samples to k_fixup
                        unsigned int slot0 = 0;
                        const int leader = __ffsll((long long)am) - 1;
                        if (lane == leader) slot0 = atomicAdd(P.fix_count, (unsigned int)__popcll(am));
                        slot0 = __shfl(slot0, leader);
                        if (!ok) {
                            const unsigned int slot = slot0 + (unsigned int)__popcll(am & lane_le) - 1u;
                            if (slot < P.fix_cap) {
                                FixEntry fe; fe.at = sig_base + at; fe.c1 = c1; fe.ev = ev_first + ev; fe.read = r; fe.shifted = in_shift;
                                P.fix[slot] = fe;
                            } else atomicOr(P.err, 8u);
                        }
                    }
                }
            }
        }
    }
}

// the hot loop: certified mode, events of <= MULT_N samples, no level-shift window, positive ADC values
template <int NT>
__device__ static inline void emit_fast(const SigParams& P, SigLds<NT>& L, int lane, int ev0, uint8_t* mk, bool valid,
                                        int so_w, int wave_total, uint32_t base_pos, uint32_t read_len, int16_t* out,
                                        long long sig_base, int r, long long ev_first, float thr, uint32_t so_first) {
    const unsigned long long lane_le = (lane == 63) ? ~0ull : ((2ull << lane) - 1);
    const bool rna = P.rna != 0;
    const uint32_t a_top = read_len - 1 - base_pos;
    for (int w0 = 0; w0 < wave_total; w0 += MK_W) {
        ((uint4*)mk)[lane] = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (valid && so_w >= w0 && so_w < w0 + MK_W) mk[so_w - w0] = 1;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        int base_ev = ev0 - 1 + __popcll(__ballot(valid && so_w < w0));
        const int w_end = min(w0 + MK_W, wave_total);
        for (int c0 = w0; c0 < w_end; c0 += 64) {
            const int idx = c0 + lane;
            const unsigned long long sm = __ballot(mk[idx - w0] != 0);
            const int ev = base_ev + __popcll(sm & lane_le);
            base_ev += __popcll(sm);
            const uint4 ra = L.rec_a[ev];
            const int I = (int)L.rec_b[ev].x;
            const uint32_t j = ((uint32_t)idx + so_first - ra.y) & (MULT_N - 1);
            const uint2 mu = L.mult[j];
            const uint32_t c1 = lcg_mul(ra.x, mu.x);
            const uint32_t r2 = lcg_mul_lazy(ra.x, mu.y);
            const float x = box_muller_fast(c1, r2);
            const float v = __builtin_fmaf(x, __uint_as_float(ra.w), __uint_as_float(ra.z));
            const float fl = floorf(v);
            const float fr = v - fl;
            const bool act = idx < w_end;
            const bool ok = fabsf(fr - 0.5f) < thr && c1 <= LCG_M - (1u << NEAR_ONE_BITS);
            const int n = I + (int)fl;
            const uint32_t at = rna ? (a_top - (uint32_t)idx) : (base_pos + (uint32_t)idx);
            if (act && ok) out[at] = (int16_t)(uint16_t)((uint32_t)n & 0xffffu);
            const unsigned long long am = __ballot(act && !ok);
            if (am) {
                unsigned int slot0 = 0;
                const int leader = __ffsll((long long)am) - 1;
                if (lane == leader) slot0 = atomicAdd(P.fix_count, (unsigned int)__popcll(am));
                slot0 = __shfl(slot0, leader);
                if (act && !ok) {
                    const unsigned int slot = slot0 + (unsigned int)__popcll(am & lane_le) - 1u;
                    if (slot < P.fix_cap) {
                        FixEntry fe; fe.at = sig_base + at; fe.c1 = c1; fe.ev = ev_first + ev; fe.read = r; fe.shifted = 0;
                        P.fix[slot] = fe;
                    } else atomicOr(P.err, 8u);
                }
            }
        }
    }
}

template <int MODE, int NT>
__global__ __launch_bounds__(NT, SQG_SIGNAL_WAVES_PER_SIMD) void k_signal(const SigParams P) {
    __shared__ SigLds<NT> L;
    constexpr int NW = NT / 64, HT = 2 * NT;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < MULT_N; i += NT) { L.mult[i] = make_uint2(P.pw[i], P.pw[POW_N + i]); L.jump[i] = P.pw[2 * POW_N + i]; }
    for (int i = tid; i < 256; i += NT) L.lut[i] = (uint8_t)base_code((uint8_t)i);

    const int chain = P.chain_order[blockIdx.x];
    const int c_lo = P.chain_off[chain], c_hi = P.chain_off[chain + 1];
    uint32_t* row = P.rows ? P.rows + (size_t)P.reads[P.chain_reads[c_lo]].worker * P.num_kmer : nullptr;
    const int k = P.k;
    const double kd = P.kd;
    uint8_t* mk = L.mk[wid];
    __syncthreads();

    for (int ci = c_lo; ci < c_hi; ci++) {
        const int r = P.chain_reads[ci];
        const ReadDesc rd = P.reads[r];
        const long long sig_base = P.sig_off[r];
        const uint32_t read_len = (uint32_t)(P.sig_off[r + 1] - sig_base);
        const long long n1 = (long long)P.seglen[2 * r];             // samples of segment 0
        const long long shift_lo = n1 - P.shift_len;                  // src/genread.c:79
        const int ne = rd.ne0 + rd.ne1;
        const double offset = rd.offset;
        int16_t* out = P.sig + sig_base;
        uint32_t done = 0;                                            // samples emitted so far in this read

        for (int s0 = 0; s0 < ne; s0 += NT) {
            // ================= event phase =================
            const int e = s0 + tid;
            const bool valid = e < ne;
            uint32_t rank = 0;
            int sps = 0;
            if (valid) {
                const uint8_t* bp = P.bases + rd.base_off + (e < rd.ne0 ? (long long)e : (long long)rd.len0 + (e - rd.ne0));
                for (int i = 0; i < k; i++) rank = (rank << 2) | L.lut[bp[i]];                 // src/seq.h:31-42
                sps = P.dwell ? (int)P.dwell[rd.ev_off + e] : P.const_sps;
            }
            const float2 md = valid ? P.model[rank] : make_float2(0.f, 0.f);
            const int incl = wave_incl_scan(sps, lane);
            if (lane == 63) L.wsum[wid] = incl;
            if (P.use_streams) for (int i = tid; i < HT; i += NT) { L.keys[i] = BIN_EMPTY; L.bins[i] = 0; }
            __syncthreads();                                                                  // (1)
            int woff = 0, seg_total = 0;
            for (int w = 0; w < NW; w++) { const int x = L.wsum[w]; if (w < wid) woff += x; seg_total += x; }
            const int so = woff + incl - sps;                 // first sample of my event within the segment
            const int wave_total = __shfl(incl, 63);

            uint32_t c_ev = 0;
            if (P.use_streams) {
                uint32_t h = (rank * 2654435761u) >> (32 - (31 - __builtin_clz(HT)));
                uint32_t ord = 0;
                if (valid) {
                    for (;;) {
                        const uint32_t old = atomicCAS(&L.keys[h], BIN_EMPTY, rank);
                        if (old == BIN_EMPTY || old == rank) break;
                        h = (h + 1) & (HT - 1);
                    }
                    ord = atomicAdd(&L.bins[h], 1u);
                }
                __syncthreads();                                                              // (2)
                const uint32_t b0 = L.bins[2 * tid], b1 = L.bins[2 * tid + 1];
                const int local = (int)(b0 + b1);
                const int incl2 = wave_incl_scan(local, lane);
                if (lane == 63) L.wsum2[wid] = incl2;
                __syncthreads();                                                              // (3)
                int woff2 = 0;
                for (int w = 0; w < wid; w++) woff2 += L.wsum2[w];
                const uint32_t ex = (uint32_t)(woff2 + incl2 - local);
                L.bins[2 * tid] = (ex << 16) | b0;
                L.bins[2 * tid + 1] = ((ex + b0) << 16) | b1;
                __syncthreads();                                                              // (4)
                uint32_t bv = 0;
                if (valid) { bv = L.bins[h]; L.mem[(bv >> 16) + ord] = ((uint32_t)tid << 16) | (uint32_t)sps; }
                __syncthreads();                                                              // (5)
                uint32_t prior = 0, total = (uint32_t)sps;
                bool last = true;
                if (valid && (bv & 0xffffu) > 1u) {
                    total = 0;
                    const uint32_t m0 = bv >> 16, mc = bv & 0xffffu;
                    for (uint32_t m = 0; m < mc; m++) {
                        const uint32_t v = L.mem[m0 + m];
                        const uint32_t t2 = v >> 16, s2 = v & 0xffffu;
                        total += s2;
                        if (t2 < (uint32_t)tid) prior += s2;
                        if (t2 > (uint32_t)tid) last = false;
                    }
                }
                uint32_t c_row = 0;
                if (valid) c_row = __hip_atomic_load(&row[rank], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();                                                              // (6) all states read before any is advanced
                if (valid) {
                    c_ev = prior ? lcg_mul(c_row, jump2_lds(L, P.pw, prior)) : c_row;
                    if (last) __hip_atomic_store(&row[rank], lcg_mul(c_row, jump2_lds(L, P.pw, total)),
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            float thr = 1.0f;
            bool fast_ok = true;
            if (!P.use_streams) {
                // no amplitude noise (--ideal / --ideal-amp): s = level_mean, one digitisation per event (src/gensig.c:266,270)
                const int16_t qc = to_i16((double)md.x * P.dig / P.range - offset);
                L.rec_a[tid] = make_uint4(0u, (uint32_t)so, 0u, 0u);
                L.rec_b[tid] = make_uint2((uint32_t)(uint16_t)qc, 0u);
            } else if (MODE == 1) {
                // v = s_f*dig/range - offset  ~  x*(sd*kd) + (m*kd - offset) = x*sdk + (I + F)
                const double mkd = (double)md.x * kd;
                const double mk = mkd - offset;
                const double fl = floor(mk);
                const float F = (float)(mk - fl);
                const float sdk = (float)((double)md.y * kd);
                const float asdk = fabsf(sdk);
                // error budget (DESIGN.md "Certified fast path"): swept |x'-x| * sdk; float narrowing of s
                // (2^-24 (|m| kd + 6.56 sdk)); roundings of sdk (x6.56), of F (2^-25) and of the fma
                // (2^-24 (6.56 sdk + 1)); FP64 roundings and the fp32 evaluation of eps itself in the slack
                const float eps = P.delta_x * asdk + 5.9604645e-8f * ((float)fabs(mkd) + 21.0f * asdk + 3.0f) + 2.0e-7f;
                thr = 0.5f - eps;
                if (!(fabs(fl) < 1.0e9)) thr = -1.0f;                     // absurd profile: everything goes to FP64
                fast_ok = !valid || (sps <= MULT_N && fl - 7.0 * (double)asdk > 2.0 && fl < 1.0e9);
                L.rec_a[tid] = make_uint4(c_ev, (uint32_t)so, __float_as_uint(F), __float_as_uint(sdk));
                L.rec_b[tid] = make_uint2((uint32_t)(int)fl, __float_as_uint(thr));
            } else {
                L.rec_a[tid] = make_uint4(c_ev, (uint32_t)so, __float_as_uint(md.x), __float_as_uint(md.y));
                L.rec_b[tid] = make_uint2(0u, 0u);
            }
            __syncthreads();                                                                  // (7)

            // ================= sample phase (wave-local) =================
            if (wave_total > 0) {
                const uint32_t base_pos = done + (uint32_t)woff;
                const bool shift_tile = P.shift_len > 0 && (long long)base_pos + wave_total > shift_lo && (long long)base_pos < n1;
                const int so_w = so - woff;
                bool use_fast = false;
                if (MODE == 1 && P.use_streams && !shift_tile) {
                    use_fast = __all(fast_ok);
                    if (use_fast) {
                        float t = valid ? thr : 1.0f;
                        for (int o = 32; o > 0; o >>= 1) t = fminf(t, __shfl_xor(t, o));
                        thr = t;
                    }
                }
                if (use_fast) emit_fast<NT>(P, L, lane, wid * 64, mk, valid, so_w, wave_total, base_pos, read_len, out,
                                            sig_base, r, rd.ev_off + s0, thr, (uint32_t)woff);
                else emit_generic<MODE, NT>(P, L, lane, wid * 64, mk, valid, so_w, wave_total, base_pos, read_len, out,
                                            sig_base, r, rd.ev_off + s0, offset, shift_lo, n1, shift_tile);
            }
            done += (uint32_t)seg_total;
            __syncthreads();                                                                  // (8)
        }
        if (done != read_len && tid == 0) atomicOr(P.err, 4u);
    }
}

// ---- k_fixup: FP64 path for the samples k_signal<CERTIFIED> left undecided -------------------
__global__ __launch_bounds__(256) void k_fixup(const SigParams P) {
    const unsigned int n = min(*P.fix_count, P.fix_cap);
    for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const FixEntry fe = P.fix[i];
        const ReadDesc rd = P.reads[fe.read];
        const int e = (int)(fe.ev - rd.ev_off);
        const uint8_t* bp = P.bases + rd.base_off + (e < rd.ne0 ? (long long)e : (long long)rd.len0 + (e - rd.ne0));
        uint32_t rank = 0;
        for (int q = 0; q < P.k; q++) rank = (rank << 2) | base_code(bp[q]);
        const float2 md = P.model[rank];
        int16_t q = sample_exact(fe.c1, md.x, md.y, P.dig, P.range, rd.offset);
        if (fe.shifted) q = (int16_t)(uint16_t)(((int)q - P.shift) & 0xffff);
        P.sig[fe.at] = q;
    }
}

// ---- k_certify: max |x_fast - x_exact| over every state the fp32 path may accept ------------
// The deviate is a function of c1 alone (c2 = a*c1 mod M), so the sweep is exhaustive.  Both
// representatives of the lazily reduced second draw are tried.
__global__ __launch_bounds__(256) void k_certify(unsigned int* __restrict__ max_bits) {
    float m = 0.f;
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    for (unsigned long long c = 1 + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
         c <= LCG_M - (1u << NEAR_ONE_BITS); c += stride) {
        const uint32_t c1 = (uint32_t)c, c2 = lcg_mul(c1, LCG_A);
        const double xe = box_muller_exact(c1, c2);
        const float e0 = fabsf((float)((double)box_muller_fast(c1, c2) - xe));
        const float e1 = fabsf((float)((double)box_muller_fast(c1, c2 + LCG_M) - xe));
        m = fmaxf(m, fmaxf(e0, e1));
        if (!(e0 == e0) || !(e1 == e1)) m = __builtin_inff();
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(max_bits, __float_as_uint(m));
}

// ---- k_store_probe: pure streaming store, the measured HBM write ceiling --------------------
__global__ __launch_bounds__(256) void k_store_probe(uint4* __restrict__ dst, size_t n16, uint32_t v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        dst[i] = make_uint4(v, v + 1, v + 2, (uint32_t)i);
}
