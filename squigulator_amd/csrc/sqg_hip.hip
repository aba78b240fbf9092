// sqg_hip.hip -- MI355X (gfx950) implementation of include/sqg.h.
//
// Hand-written HIP for the per-read signal path of a nanopore simulator:
//   k_dwell   : per-event dwell draw (Gaussian, folded at <1) from the worker's time stream,
//               addressed by LCG jump-ahead, + per-read sample totals      (src/gensig.c:254-257)
//   k_scan    : exclusive scan of read lengths -> output offsets
//   k_signal  : one wavefront per worker: k-mer ranks, per-(worker,k-mer) stream hand-out with
//               in-order duplicate resolution, per-sample Box-Muller + digitisation, coalesced
//               int16 stores, RNA reversal / adaptor level shift folded into the store
//                                                                          (src/gensig.c:226-356)
// No MFMA anywhere: this is an integer-LCG / transcendental / streaming-store path.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see squigulator_amd/build.py).
//
// The product path never touches oracle/: this file is self-contained.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/sqg.h"

// ------------------------------------------------------------------------------------------
// MINSTD Lehmer generator in canonical form.
// The reference keeps the UNCORRECTED Schrage value (src/rand.h:79-85); that sequence is
// congruent to c_{n+1} = 16807 * c_n mod (2^31-1), and the uniform it returns is c/(2^31-1)
// with c in [1, M-1] (c == 0, reachable only from a seed = 0 mod M, returns 1.0 forever).
// Canonical form makes position n addressable: c_n = a^n c_0 mod M.
// ------------------------------------------------------------------------------------------
#define LCG_M 2147483647u
#define LCG_A 16807u

#define POW_N 1024          // entries per jump table
// table layout in d_pow (uint32 each):
//   [0*POW_N + j] = a^(2j+1)     first draw of sample/event j after the base state
//   [1*POW_N + j] = a^(2j+2)     second draw
//   [2*POW_N + j] = a^(2j)       jump by j samples (2 draws each)
//   [3*POW_N + j] = a^(2*1024*j)
//   [4*POW_N + j] = a^(2*1024*1024*j)
#define POW_TABLES 5

__host__ __device__ static inline uint32_t lcg_mul(uint32_t a, uint32_t b) {
    const unsigned long long p = (unsigned long long)a * b;
    uint32_t r = (uint32_t)(p & LCG_M) + (uint32_t)(p >> 31);
    r = (r & LCG_M) + (r >> 31);
    return r;
}

static uint32_t lcg_pow(uint32_t base, unsigned long long e) {
    uint32_t r = 1, b = base;
    while (e) { if (e & 1) r = lcg_mul(r, b); b = lcg_mul(b, b); e >>= 1; }
    return r;
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

// nrng body, src/rand.h:87-94, for the two consecutive draws c1, c2 (FP64, no contraction)
__device__ static inline double box_muller_exact(uint32_t c1, uint32_t c2) {
    const double u = lcg_uniform(c1);
    const double t = (2.0 * 3.14159265) * lcg_uniform(c2);
    return sqrt(-2.0 * log(u)) * cos(t);
}

// (int16_t)double as gcc/x86-64 lowers it (cvttsd2si r32, low half): src/gensig.c:270
__device__ static inline int16_t to_i16(double v) {
    int32_t t;
    if (v > -2147483649.0 && v < 2147483648.0) t = (int32_t)v; else t = (int32_t)0x80000000u;
    return (int16_t)(uint16_t)((uint32_t)t & 0xffffu);
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

// ------------------------------------------------------------------------------------------
// device-side descriptors
// ------------------------------------------------------------------------------------------
struct ReadDesc {
    long long base_off;   // first byte of segment 0 in the batch's base buffer
    long long ev_off;     // first event of this read in the batch's event arrays
    double offset;        // slow5 offset of this read (drawn on the host)
    int len0, len1;       // bytes in segment 0 (read incl. attached prefix) and 1 (RNA stall)
    int ne0, ne1;         // events per segment
    int worker;           // context-local worker index
    uint32_t time_c0;     // worker's time-stream state at the start of this read
};

struct SigParams {
    const ReadDesc* reads;
    const int* chain_off;        // [n_chains+1]
    const int* chain_reads;      // read indices, grouped per worker, in batch order
    const uint8_t* bases;
    const uint16_t* dwell;       // per event (null when dwell is constant)
    const unsigned long long* seglen;  // [2*n_reads] samples in segment 0 / 1
    const long long* sig_off;    // [n_reads+1]
    const float2* model;         // {level_mean, (float)(level_stdv*amp_noise)}
    const uint32_t* pw;
    uint32_t* rows;              // [n_local_workers][num_kmer]
    int16_t* sig;
    unsigned int* err;
    double dig, range;
    int k, num_kmer;
    int const_sps;               // (int)dwell_mean, used when dwell == null
    int use_streams;             // 0 in --ideal / --ideal-amp (src/gensig.c:265-269)
    int rna;                     // reverse the signal (src/gensig.c:348-354)
    int shift_len;               // RNA+prefix: 79*(int)dwell_mean samples get -shift (src/genread.c:79-86)
    int shift;                   // (int16)(30*dig/range)
};

// ------------------------------------------------------------------------------------------
// k_init_rows: kmer_gen[tid][j] seed = s_tid + j, s_tid = seed + tid*(num_kmer+10)  (src/sim.c:238-257)
// ------------------------------------------------------------------------------------------
__global__ void k_init_rows(uint32_t* rows, int num_kmer, long long seed, int worker_lo, long long n_total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const long long w = i / num_kmer, j = i % num_kmer;
    long long s = seed + (w + worker_lo) * ((long long)num_kmer + 10) + j;
    s %= (long long)LCG_M;
    if (s < 0) s += LCG_M;
    rows[i] = (uint32_t)s;
}

// ------------------------------------------------------------------------------------------
// k_dwell: one thread per event of the batch
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dwell(const ReadDesc* __restrict__ reads, int n_reads,
                                               const int* __restrict__ blk_read, long long n_events,
                                               const uint32_t* __restrict__ pw, double dmean, double dstd,
                                               uint16_t* __restrict__ dwell,
                                               unsigned long long* __restrict__ seglen,
                                               unsigned int* __restrict__ err) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = gid < n_events;
    int r = blk_read[blockIdx.x];
    int sps = 0, seg = 0;
    if (valid) {
        while (r + 1 < n_reads && gid >= reads[r + 1].ev_off) r++;
        const ReadDesc rd = reads[r];
        const uint32_t e = (uint32_t)(gid - rd.ev_off);
        const uint32_t c = lcg_mul(rd.time_c0, lcg_jump2(pw, e));
        const uint32_t c1 = lcg_mul(c, LCG_A), c2 = lcg_mul(c1, LCG_A);
        const double z = box_muller_exact(c1, c2);
        const double v = (z * dstd) + dmean;                 // nrng: (x * s) + m
        sps = (int)round(v);                                 // src/gensig.c:255
        sps = sps < 1 ? -sps + 1 : sps;                      // src/gensig.c:256
        if (sps > 65535) { atomicOr(err, 1u); sps = 65535; }
        dwell[gid] = (uint16_t)sps;
        seg = e >= (uint32_t)rd.ne0;
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

// ------------------------------------------------------------------------------------------
// k_scan: sig_off = exclusive scan of per-read totals (single workgroup; n_reads is small)
// ------------------------------------------------------------------------------------------
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

// ------------------------------------------------------------------------------------------
// k_signal: one wavefront (= one workgroup of 64) per worker that has reads in this batch.
// Everything is wave-synchronous; the worker's reads are walked in batch order, each read in
// tiles of 64 consecutive events, so every k-mer stream is handed out in event order exactly
// as the reference's nested loop does.
// ------------------------------------------------------------------------------------------
#define TAG_N 1024

template <bool LDS_ROW>
__global__ __launch_bounds__(64) void k_signal(const SigParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x;
    uint32_t* ev_c = smem;                 // [64] stream state at event start
    uint32_t* ev_so = smem + 64;           // [64] first sample of event within tile
    uint32_t* ev_rank = smem + 128;        // [64]
    uint32_t* tag = smem + 192;            // [TAG_N/4] hashed "last writer" bytes
    uint32_t* srow = smem + 192 + TAG_N / 4;
    uint8_t* tagb = (uint8_t*)tag;

    const int chain = blockIdx.x;
    const int c_lo = P.chain_off[chain], c_hi = P.chain_off[chain + 1];
    const int worker = P.reads[P.chain_reads[c_lo]].worker;
    uint32_t* grow = P.rows + (size_t)worker * P.num_kmer;
    uint32_t* row = LDS_ROW ? srow : grow;
    if (LDS_ROW && P.use_streams) {
        for (int i = lane; i < P.num_kmer; i += 64) srow[i] = grow[i];
        __syncthreads();
    }
    const int k = P.k;
    const double dig = P.dig, range = P.range;

    for (int ci = c_lo; ci < c_hi; ci++) {
        const int r = P.chain_reads[ci];
        const ReadDesc rd = P.reads[r];
        const long long sig_base = P.sig_off[r];
        const long long read_len = P.sig_off[r + 1] - sig_base;
        const long long n1 = (long long)P.seglen[2 * r];            // samples of segment 0
        const long long shift_lo = n1 - P.shift_len;                 // src/genread.c:79
        const int ne = rd.ne0 + rd.ne1;
        const double offset = rd.offset;
        long long done = 0;                                          // samples emitted so far in this read

        for (int t0 = 0; t0 < ne; t0 += 64) {
            const int e = t0 + lane;
            const bool valid = e < ne;
            uint32_t rank = 0;
            int sps = 0;
            if (valid) {
                const long long bp = rd.base_off + (e < rd.ne0 ? (long long)e : (long long)rd.len0 + (e - rd.ne0));
                for (int i = 0; i < k; i++) rank = (rank << 2) | base_code(P.bases[bp + i]);   // src/seq.h:31-42
                sps = P.dwell ? (int)P.dwell[rd.ev_off + e] : P.const_sps;
            }
            // exclusive scan of sps over the tile
            int incl = sps;
            for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
            const int tile_total = __shfl(incl, 63);
            const int so = incl - sps;

            uint32_t c_ev = 0;
            if (P.use_streams) {
                // --- in-order hand-out of each k-mer stream within the tile ---
                const uint32_t h = rank & (TAG_N - 1);
                if (valid) tagb[h] = (uint8_t)lane;
                __syncthreads();
                const bool loser = valid && tagb[h] != (uint8_t)lane;
                unsigned long long lm = __ballot(loser);
                int prior = 0;            // samples earlier lanes of this tile drew from my stream
                bool last = true;         // am I the last event of my k-mer in this tile?
                while (lm) {
                    const int l = __ffsll((long long)lm) - 1;
                    const uint32_t rl = __shfl(rank, l);
                    const bool in_g = valid && rank == rl;
                    const unsigned long long g = __ballot(in_g);
                    unsigned long long gg = g;
                    while (gg) {
                        const int j = __ffsll((long long)gg) - 1;
                        gg &= gg - 1;
                        const int sj = __shfl(sps, j);
                        if (in_g && lane > j) prior += sj;
                    }
                    if (in_g) last = (lane == 63 - __clzll((long long)g));
                    lm &= ~g;
                }
                __syncthreads();
                uint32_t c_row = 0;
                if (valid) c_row = row[rank];
                __syncthreads();
                if (valid) {
                    c_ev = prior ? lcg_mul(c_row, lcg_jump2(P.pw, (uint32_t)prior)) : c_row;
                    if (last) row[rank] = lcg_mul(c_ev, lcg_jump2(P.pw, (uint32_t)sps));
                }
            }
            ev_c[lane] = c_ev;
            ev_so[lane] = valid ? (uint32_t)so : 0xffffffffu;
            ev_rank[lane] = rank;
            __syncthreads();

            // --- samples of this tile, 64 per step ---
            const int nev = min(64, ne - t0);
            for (int s0 = 0; s0 < tile_total; s0 += 64) {
                const int idx = s0 + lane;
                if (idx < tile_total) {
                    int lo = 0, hi = nev - 1;                 // largest event with ev_so <= idx
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (ev_so[mid] <= (uint32_t)idx) lo = mid; else hi = mid - 1;
                    }
                    const uint32_t j = (uint32_t)idx - ev_so[lo];
                    const float2 md = P.model[ev_rank[lo]];
                    float s;
                    if (P.use_streams) {
                        const uint32_t c0 = ev_c[lo];
                        uint32_t c1, c2;
                        if (j < POW_N) { c1 = lcg_mul(c0, P.pw[j]); c2 = lcg_mul(c0, P.pw[POW_N + j]); }
                        else { const uint32_t cj = lcg_mul(c0, lcg_jump2(P.pw, j)); c1 = lcg_mul(cj, LCG_A); c2 = lcg_mul(c1, LCG_A); }
                        const double z = box_muller_exact(c1, c2);
                        s = (float)((z * (double)md.y) + (double)md.x);        // float s = nrng(...), src/gensig.c:268
                    } else {
                        s = md.x;                                             // src/gensig.c:266
                    }
                    int16_t q = to_i16((double)s * dig / range - offset);     // src/gensig.c:270
                    const long long pos = done + idx;                         // index within the read, generation order
                    if (pos >= shift_lo && pos < n1) q = (int16_t)(uint16_t)(((int)q - P.shift) & 0xffff);
                    const long long at = P.rna ? (read_len - 1 - pos) : pos;
                    P.sig[sig_base + at] = q;
                }
            }
            done += tile_total;
            __syncthreads();
        }
        if (done != read_len && lane == 0) atomicOr(P.err, 4u);
    }
    if (LDS_ROW && P.use_streams) {
        __syncthreads();
        for (int i = lane; i < P.num_kmer; i += 64) grow[i] = srow[i];
    }
}

// pure int16 streaming store: the measured HBM write ceiling the roofline is quoted against
__global__ __launch_bounds__(256) void k_store_probe(uint4* __restrict__ dst, size_t n16, uint32_t v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        dst[i] = make_uint4(v, v + 1, v + 2, (uint32_t)i);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct sqg_ctx {
    sqg_cfg_t cfg;
    int k = 0, num_kmer = 0, T = 0, wlo = 0, whi = 0, nw = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t* d_rows = nullptr;
    float2* d_model = nullptr;
    uint32_t* d_pow = nullptr;
    unsigned int* d_err = nullptr;
    int16_t* d_sig = nullptr; size_t sig_cap = 0;
    uint16_t* d_dwell = nullptr; size_t dwell_cap = 0;
    unsigned long long* d_seglen = nullptr; long long* d_sigoff = nullptr; size_t reads_cap = 0;
    std::vector<uint32_t> time_c;          // canonical time-stream state per local worker
    std::vector<long long> off_x, med_x;   // raw Schrage states (as the reference keeps them)
    unsigned long long next_stage = 0, next_run = 0;
    sqg_timing_t timing = {0, 0, 0, 0};
    bool use_dwell_stream = true, use_kmer_streams = true;
    std::string err;
};

struct sqg_batch {
    unsigned long long seq = 0;
    int n = 0;
    long long n_events = 0, n_bases = 0, n_samples = 0;
    int n_chains = 0;
    std::vector<long long> ev_off, sig_off;
    std::vector<double> offset, median;
    std::vector<unsigned long long> seglen_host;   // only when dwell is constant
    uint8_t* d_bases = nullptr;
    ReadDesc* d_reads = nullptr;
    int* d_blk_read = nullptr;
    int* d_chain_off = nullptr;
    int* d_chain_reads = nullptr;
    long long* h_sigoff = nullptr;   // pinned
    bool ran = false, waited = false;
};

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return e_ == hipErrorOutOfMemory ? SQG_ENOMEM : SQG_EDEVICE;                       \
        }                                                                                      \
    } while (0)

// the reference's rng()/nrng() on the host, for the two per-read scalar draws that are
// RETURNED as doubles (offset, median_before; src/gensig.c:311-317): made with the host's libm
// so they are the very doubles the CPU reference produces on this machine.
static double host_rng(long long* xp) {            // src/rand.h:79-85
    const long long x = *xp;
    const long long nx = 16807LL * (x % 127773LL) - 2836LL * (x / 127773LL);
    *xp = nx;
    return (double)(nx > 0 ? nx : nx + 2147483647LL) / 2147483647;
}
static double host_nrng(double m, double s, long long* xp) {   // src/rand.h:87-94
    double u = 0.0, t = 0.0;
    while (u == 0.0) u = host_rng(xp);
    while (t == 0.0) t = 2.0 * 3.14159265 * host_rng(xp);
    const double z = std::sqrt(-2.0 * std::log(u)) * std::cos(t);
    return (z * s) + m;
}

static uint32_t canon(long long s) {
    s %= (long long)LCG_M;
    if (s < 0) s += LCG_M;
    return (uint32_t)s;
}

extern "C" const char* sqg_strerror(int code) {
    switch (code) {
    case SQG_OK: return "ok";
    case SQG_EINVAL: return "invalid argument or unsupported configuration";
    case SQG_ENOMEM: return "out of memory";
    case SQG_EDEVICE: return "HIP runtime error";
    case SQG_ESEQUENCE: return "batches must be run in staging order";
    case SQG_ENODEVICE: return "no usable HIP device";
    case SQG_EOVERFLOW: return "read too long (>= UINT32_MAX samples) or dwell > 65535";
    default: return "unknown error";
    }
}

extern "C" const char* sqg_last_error(const sqg_ctx_t* ctx) { return ctx ? ctx->err.c_str() : ""; }

extern "C" int sqg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return SQG_ENODEVICE;
    return n;
}

extern "C" int32_t sqg_worker_of(int32_t i, int32_t n_rec, int32_t T) {
    if (T <= 1) return 0;                                  // src/thread.c:122-125
    const int32_t step = (n_rec + T - 1) / T;              // src/thread.c:80
    return i / step;
}

extern "C" void sqg_destroy(sqg_ctx_t* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->d_rows); (void)hipFree(ctx->d_model); (void)hipFree(ctx->d_pow); (void)hipFree(ctx->d_err);
    (void)hipFree(ctx->d_sig); (void)hipFree(ctx->d_dwell); (void)hipFree(ctx->d_seglen); (void)hipFree(ctx->d_sigoff);
    for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int sqg_create(const sqg_cfg_t* cfg, sqg_ctx_t** out) {
    if (!cfg || !out) return SQG_EINVAL;
    *out = nullptr;
    if (cfg->abi_version != SQG_ABI_VERSION) return SQG_EINVAL;
    if (cfg->kmer_size < 1 || cfg->kmer_size > 9 || !cfg->model) return SQG_EINVAL;
    if (cfg->num_workers < 1 || cfg->worker_lo < 0 || cfg->worker_hi > cfg->num_workers || cfg->worker_lo >= cfg->worker_hi) return SQG_EINVAL;
    if (!(cfg->profile.range != 0.0) || !(cfg->profile.dwell_mean >= 1.0)) return SQG_EINVAL;
    if (cfg->profile.dwell_mean + 8.0 * std::fabs(cfg->profile.dwell_std) > 60000.0) return SQG_EINVAL;
    if (cfg->mode != SQG_MODE_EXACT && cfg->mode != SQG_MODE_CERTIFIED) return SQG_EINVAL;
    const long long nk = 1LL << (2 * cfg->kmer_size);
    // canonical-form validity: |seed| + T*(nk+10) must stay where Schrage's uncorrected state is
    // within (-M, M) after one step (see DESIGN.md "LCG")
    const double span = std::fabs((double)cfg->seed) + (double)cfg->num_workers * (double)(nk + 10);
    if (span > 9.0e10) return SQG_EINVAL;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SQG_ENODEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return SQG_EINVAL;

    sqg_ctx* c = new (std::nothrow) sqg_ctx();
    if (!c) return SQG_ENOMEM;
    c->cfg = *cfg;
    c->cfg.model = nullptr;
    c->k = (int)cfg->kmer_size; c->num_kmer = (int)nk; c->T = cfg->num_workers;
    c->wlo = cfg->worker_lo; c->whi = cfg->worker_hi; c->nw = c->whi - c->wlo;
    c->use_dwell_stream = !(cfg->flags & (SQG_IDEAL | SQG_IDEAL_TIME));
    c->use_kmer_streams = !(cfg->flags & (SQG_IDEAL | SQG_IDEAL_AMP));
    int rc = SQG_OK;
    auto fail = [&](int code) { sqg_destroy(c); return code; };
#define CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = (e_ == hipErrorOutOfMemory) ? SQG_ENOMEM : SQG_EDEVICE; fprintf(stderr, "[sqg] %s: %s\n", #call, hipGetErrorString(e_)); return fail(rc); } } while (0)
    CHK(hipSetDevice(cfg->device));
    CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto& e : c->ev) CHK(hipEventCreate(&e));

    // pore model: {level_mean, (float)(level_stdv*amp_noise)}  (src/sim.c:249)
    std::vector<float2> hm((size_t)nk);
    for (long long j = 0; j < nk; j++) {
        const float sd = cfg->model[j].level_stdv * cfg->amp_noise;
        hm[(size_t)j] = make_float2(cfg->model[j].level_mean, sd);
    }
    CHK(hipMalloc(&c->d_model, (size_t)nk * sizeof(float2)));
    CHK(hipMemcpy(c->d_model, hm.data(), (size_t)nk * sizeof(float2), hipMemcpyHostToDevice));

    // jump tables
    std::vector<uint32_t> pw((size_t)POW_TABLES * POW_N);
    {
        const uint32_t a2 = lcg_mul(LCG_A, LCG_A);
        uint32_t p = 1;                                     // a^(2j)
        for (int j = 0; j < POW_N; j++) {
            pw[2 * POW_N + j] = p;
            pw[0 * POW_N + j] = lcg_mul(p, LCG_A);
            pw[1 * POW_N + j] = lcg_mul(p, a2);
            p = lcg_mul(p, a2);
        }
        const uint32_t step1 = p;                           // a^(2*1024)
        p = 1;
        for (int j = 0; j < POW_N; j++) { pw[3 * POW_N + j] = p; p = lcg_mul(p, step1); }
        const uint32_t step2 = p;                           // a^(2*1024*1024)
        p = 1;
        for (int j = 0; j < POW_N; j++) { pw[4 * POW_N + j] = p; p = lcg_mul(p, step2); }
    }
    CHK(hipMalloc(&c->d_pow, pw.size() * sizeof(uint32_t)));
    CHK(hipMemcpy(c->d_pow, pw.data(), pw.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    CHK(hipMalloc(&c->d_err, sizeof(unsigned int)));
    CHK(hipMemset(c->d_err, 0, sizeof(unsigned int)));

    // per-(worker,k-mer) stream states
    if (c->use_kmer_streams) {
        const long long total = (long long)c->nw * nk;
        CHK(hipMalloc(&c->d_rows, (size_t)total * sizeof(uint32_t)));
        const int blocks = (int)((total + 255) / 256);
        hipLaunchKernelGGL(k_init_rows, dim3(blocks), dim3(256), 0, c->stream, c->d_rows, (int)nk, (long long)cfg->seed, c->wlo, total);
        CHK(hipGetLastError());
    }
    // scalar streams (src/sim.c:241-247): time = s+2, offset = s+4, median = s+5
    c->time_c.resize((size_t)c->nw); c->off_x.resize((size_t)c->nw); c->med_x.resize((size_t)c->nw);
    for (int w = 0; w < c->nw; w++) {
        const long long s = (long long)cfg->seed + (long long)(w + c->wlo) * (nk + 10);
        c->time_c[(size_t)w] = canon(s + 2);
        c->off_x[(size_t)w] = s + 4;
        c->med_x[(size_t)w] = s + 5;
    }
    CHK(hipStreamSynchronize(c->stream));
#undef CHK
    *out = c;
    return SQG_OK;
}

static int ensure(sqg_ctx* c, void** p, size_t* cap, size_t need, size_t elem) {
    if (need <= *cap) return SQG_OK;
    size_t ncap = std::max(need, *cap + *cap / 2);
    if (*p) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(*p)); *p = nullptr; *cap = 0; }
    HIPCHK(c, hipMalloc(p, ncap * elem));
    *cap = ncap;
    return SQG_OK;
}

static const char kStallRna[] = "AAAAAGAAAAAACCCCCCCCCCCCCCCCCC";                  // src/genread.c:87
static const char kStallDna[] = "TTTTTTTTTTTTTTTTTTAATCAA";                       // src/genread.c:110
static const char kAdaptorDna[] = "GGCGTCTGCTTGGGTGTTTAACCTTTTTTTTTTAATGTACTTCGTTCAGTTACGTATTGCT";  // src/genread.c:38
static const char kAdaptorRna[] = "TGATGATGAGGGATAGACGATGGTTGTTTCTGTTGGTGCTGATATTGCTTTTTTTTTTTTTATGATGCAAGATACGCAC";  // src/genread.c:39
static const int kPolyA = 158;                                                   // src/genread.c:37
static const char kShortHack[] = "ACGTACGTACGTA";   // src/gensig.c:242-245: "ACGTACGTACGT" + its NUL (rank 0)

extern "C" void sqg_batch_free(sqg_ctx_t* ctx, sqg_batch_t* b) {
    if (!b) return;
    if (ctx) { (void)hipSetDevice(ctx->cfg.device); if (ctx->stream) (void)hipStreamSynchronize(ctx->stream); }
    (void)hipFree(b->d_bases); (void)hipFree(b->d_reads); (void)hipFree(b->d_blk_read);
    (void)hipFree(b->d_chain_off); (void)hipFree(b->d_chain_reads);
    if (b->h_sigoff) (void)hipHostFree(b->h_sigoff);
    delete b;
}

extern "C" int sqg_batch_stage(sqg_ctx_t* c, int32_t n, const char* seqs, const int64_t* seq_off,
                               const int32_t* worker, sqg_batch_t** out) {
    if (!c || !out || n < 0 || (n > 0 && (!seqs || !seq_off))) return SQG_EINVAL;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const sqg_profile_t& p = c->cfg.profile;
    const bool rna = c->cfg.flags & SQG_RNA, prefix = c->cfg.flags & SQG_PREFIX;
    const int k = c->k;

    sqg_batch* b = new (std::nothrow) sqg_batch();
    if (!b) return SQG_ENOMEM;
    b->n = n; b->seq = c->next_stage;
    b->ev_off.assign((size_t)n + 1, 0); b->sig_off.assign((size_t)n + 1, 0);
    b->offset.resize((size_t)n); b->median.resize((size_t)n);
    std::vector<ReadDesc> rd((size_t)n);
    std::vector<int> wk((size_t)n);

    // pass 1: worker ids, segment geometry
    long long nb = 0, nev = 0;
    for (int i = 0; i < n; i++) {
        const int w = worker ? worker[i] : sqg_worker_of(i, n, c->T);
        if (w < c->wlo || w >= c->whi) { delete b; c->err = "read assigned to a worker this context does not own"; return SQG_EINVAL; }
        wk[(size_t)i] = w - c->wlo;
        const long long len = seq_off[i + 1] - seq_off[i];
        if (len < 0 || len > 2000000000LL) { delete b; return SQG_EINVAL; }
        long long len0 = len;
        if (prefix) len0 += rna ? (kPolyA + (long long)strlen(kAdaptorRna)) : ((long long)strlen(kStallDna) + (long long)strlen(kAdaptorDna));
        int ne0, l0;
        if (len0 < k) { ne0 = 5; l0 = 5 + k - 1; }                  // src/gensig.c:242-245
        else { ne0 = (int)(len0 - k + 1); l0 = (int)len0; }
        int ne1 = 0, l1 = 0;
        if (prefix && rna) { l1 = (int)strlen(kStallRna); ne1 = l1 - k + 1; }   // src/genread.c:87-88
        ReadDesc& d = rd[(size_t)i];
        d.base_off = nb; d.ev_off = nev; d.len0 = l0; d.len1 = l1; d.ne0 = ne0; d.ne1 = ne1; d.worker = wk[(size_t)i];
        b->ev_off[(size_t)i] = nev;
        nb += l0 + l1; nev += ne0 + ne1;
    }
    b->ev_off[(size_t)n] = nev; b->n_events = nev; b->n_bases = nb;

    // pass 2: base buffer (prefix/stall attached as src/genread.c:95-123 does)
    std::vector<uint8_t> hb((size_t)nb + 16, (uint8_t)'A');
    for (int i = 0; i < n; i++) {
        const ReadDesc& d = rd[(size_t)i];
        uint8_t* dst = hb.data() + d.base_off;
        const char* src = seqs + seq_off[i];
        const long long len = seq_off[i + 1] - seq_off[i];
        long long len0 = len;
        if (prefix) len0 += rna ? (kPolyA + (long long)strlen(kAdaptorRna)) : ((long long)strlen(kStallDna) + (long long)strlen(kAdaptorDna));
        if (len0 < k) {
            memcpy(dst, kShortHack, (size_t)d.len0);
        } else if (!prefix) {
            memcpy(dst, src, (size_t)len);
        } else if (rna) {
            memcpy(dst, src, (size_t)len);
            memset(dst + len, 'A', (size_t)kPolyA);
            memcpy(dst + len + kPolyA, kAdaptorRna, strlen(kAdaptorRna));
        } else {
            const size_t st = strlen(kStallDna), ad = strlen(kAdaptorDna);
            memcpy(dst, kStallDna, st);
            memcpy(dst + st, kAdaptorDna, ad);
            memcpy(dst + st + ad, src, (size_t)len);
        }
        if (d.len1) memcpy(dst + d.len0, kStallRna, (size_t)d.len1);
    }

    // pass 3: per-worker chains in batch order; host-side scalar streams advance in that order
    std::vector<int> count((size_t)c->nw, 0);
    for (int i = 0; i < n; i++) count[(size_t)wk[(size_t)i]]++;
    std::vector<int> chain_of((size_t)c->nw, -1), chain_off;
    chain_off.push_back(0);
    for (int w = 0; w < c->nw; w++) if (count[(size_t)w]) { chain_of[(size_t)w] = (int)chain_off.size() - 1; chain_off.push_back(chain_off.back() + count[(size_t)w]); }
    b->n_chains = (int)chain_off.size() - 1;
    std::vector<int> fill(chain_off.begin(), chain_off.end() - 1), chain_reads((size_t)n);
    for (int i = 0; i < n; i++) chain_reads[(size_t)fill[(size_t)chain_of[(size_t)wk[(size_t)i]]]++] = i;

    const uint32_t a2 = lcg_mul(LCG_A, LCG_A);
    for (int i = 0; i < n; i++) {                         // index order == per-worker order within a worker
        ReadDesc& d = rd[(size_t)i];
        const size_t w = (size_t)d.worker;
        if (c->cfg.flags & SQG_IDEAL) {                   // src/gensig.c:311-313
            d.offset = p.offset_mean; b->median[(size_t)i] = p.median_before_mean;
        } else {                                          // src/gensig.c:315-316
            d.offset = host_nrng(p.offset_mean, p.offset_std, &c->off_x[w]);
            b->median[(size_t)i] = host_nrng(p.median_before_mean, p.median_before_std, &c->med_x[w]);
        }
        b->offset[(size_t)i] = d.offset;
        d.time_c0 = c->time_c[w];
        if (c->use_dwell_stream)                          // two draws per event (src/gensig.c:255)
            c->time_c[w] = lcg_mul(c->time_c[w], lcg_pow(a2, (unsigned long long)(d.ne0 + d.ne1)));
    }
    if (!c->use_dwell_stream) {                           // constant dwell: lengths are known now
        const unsigned long long sps = (unsigned long long)(int)p.dwell_mean;
        b->seglen_host.resize((size_t)2 * n);
        for (int i = 0; i < n; i++) { b->seglen_host[(size_t)2 * i] = sps * rd[(size_t)i].ne0; b->seglen_host[(size_t)2 * i + 1] = sps * rd[(size_t)i].ne1; }
    }

    // dwell kernel launch geometry: first read of every 256-event block
    const long long nblk = (nev + 255) / 256;
    std::vector<int> blk_read((size_t)std::max<long long>(nblk, 1), 0);
    {
        int r = 0;
        for (long long bi = 0; bi < nblk; bi++) {
            const long long g = bi * 256;
            while (r + 1 < n && g >= rd[(size_t)r + 1].ev_off) r++;
            blk_read[(size_t)bi] = r;
        }
    }

    auto bail = [&](int code) { sqg_batch_free(c, b); return code; };
#define CHKB(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { c->err = std::string(#call) + ": " + hipGetErrorString(e_); return bail(e_ == hipErrorOutOfMemory ? SQG_ENOMEM : SQG_EDEVICE); } } while (0)
    CHKB(hipMalloc(&b->d_bases, hb.size()));
    CHKB(hipMemcpyAsync(b->d_bases, hb.data(), hb.size(), hipMemcpyHostToDevice, c->stream));
    CHKB(hipMalloc(&b->d_reads, std::max<size_t>(1, rd.size()) * sizeof(ReadDesc)));
    if (n) CHKB(hipMemcpyAsync(b->d_reads, rd.data(), rd.size() * sizeof(ReadDesc), hipMemcpyHostToDevice, c->stream));
    CHKB(hipMalloc(&b->d_blk_read, blk_read.size() * sizeof(int)));
    CHKB(hipMemcpyAsync(b->d_blk_read, blk_read.data(), blk_read.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    CHKB(hipMalloc(&b->d_chain_off, chain_off.size() * sizeof(int)));
    CHKB(hipMemcpyAsync(b->d_chain_off, chain_off.data(), chain_off.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    CHKB(hipMalloc(&b->d_chain_reads, std::max<size_t>(1, chain_reads.size()) * sizeof(int)));
    if (n) CHKB(hipMemcpyAsync(b->d_chain_reads, chain_reads.data(), chain_reads.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    CHKB(hipHostMalloc(&b->h_sigoff, ((size_t)n + 1) * sizeof(long long), hipHostMallocDefault));
    CHKB(hipStreamSynchronize(c->stream));     // staging buffers above are stack-owned
#undef CHKB
    c->next_stage++;
    *out = b;
    return SQG_OK;
}

extern "C" int sqg_batch_run(sqg_ctx_t* c, sqg_batch_t* b) {
    if (!c || !b) return SQG_EINVAL;
    if (b->ran || b->seq != c->next_run) return SQG_ESEQUENCE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const sqg_profile_t& p = c->cfg.profile;
    const int n = b->n;
    int rc;
    if ((size_t)n + 1 > c->reads_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_seglen); (void)hipFree(c->d_sigoff); c->d_seglen = nullptr; c->d_sigoff = nullptr;
        const size_t cap = (size_t)n + 1 + (size_t)n / 2;
        HIPCHK(c, hipMalloc(&c->d_seglen, 2 * cap * sizeof(unsigned long long)));
        HIPCHK(c, hipMalloc(&c->d_sigoff, cap * sizeof(long long)));
        c->reads_cap = cap;
    }
    if ((rc = ensure(c, (void**)&c->d_dwell, &c->dwell_cap, (size_t)b->n_events + 64, sizeof(uint16_t)))) return rc;

    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    if (n > 0) {
        if (c->use_dwell_stream) {
            HIPCHK(c, hipMemsetAsync(c->d_seglen, 0, (size_t)2 * n * sizeof(unsigned long long), c->stream));
            const long long nblk = (b->n_events + 255) / 256;
            if (nblk > 0)
                hipLaunchKernelGGL(k_dwell, dim3((unsigned)nblk), dim3(256), 0, c->stream, b->d_reads, n, b->d_blk_read,
                                   b->n_events, c->d_pow, p.dwell_mean, p.dwell_std, c->d_dwell, c->d_seglen, c->d_err);
        } else {
            HIPCHK(c, hipMemcpyAsync(c->d_seglen, b->seglen_host.data(), (size_t)2 * n * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
        }
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, c->stream, c->d_seglen, n, c->d_sigoff, c->d_err);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    // output size is data-dependent: read the scan back (tiny), size the slab, then emit
    if (n > 0) {
        HIPCHK(c, hipMemcpyAsync(b->h_sigoff, c->d_sigoff, ((size_t)n + 1) * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else {
        b->h_sigoff[0] = 0;
    }
    b->n_samples = b->h_sigoff[n];
    for (int i = 0; i <= n; i++) b->sig_off[(size_t)i] = b->h_sigoff[i];
    if ((rc = ensure(c, (void**)&c->d_sig, &c->sig_cap, (size_t)b->n_samples + 64, sizeof(int16_t)))) return rc;

    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    if (n > 0 && b->n_chains > 0) {
        SigParams P;
        P.reads = b->d_reads; P.chain_off = b->d_chain_off; P.chain_reads = b->d_chain_reads; P.bases = b->d_bases;
        P.dwell = c->use_dwell_stream ? c->d_dwell : nullptr;
        P.seglen = c->d_seglen; P.sig_off = c->d_sigoff; P.model = c->d_model; P.pw = c->d_pow; P.rows = c->d_rows;
        P.sig = c->d_sig; P.err = c->d_err; P.dig = p.digitisation; P.range = p.range;
        P.k = c->k; P.num_kmer = c->num_kmer; P.const_sps = (int)p.dwell_mean;
        P.use_streams = c->use_kmer_streams ? 1 : 0;
        P.rna = (c->cfg.flags & SQG_RNA) ? 1 : 0;
        const bool rna_prefix = (c->cfg.flags & SQG_RNA) && (c->cfg.flags & SQG_PREFIX);
        P.shift_len = rna_prefix ? (int)strlen(kAdaptorRna) * (int)p.dwell_mean : 0;
        {   // int16_t off = 30*dig/range (src/genread.c:82): double -> int16 as the CPU does it
            const double v = 30 * p.digitisation / p.range;
            int32_t t = (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : (int32_t)0x80000000u;
            P.shift = (int)(int16_t)(uint16_t)((uint32_t)t & 0xffffu);
        }
        const bool lds_row = c->k <= 6;
        const size_t smem = (192 + TAG_N / 4 + (lds_row ? (size_t)c->num_kmer : 0)) * sizeof(uint32_t);
        if (lds_row) hipLaunchKernelGGL(k_signal<true>, dim3((unsigned)b->n_chains), dim3(64), smem, c->stream, P);
        else hipLaunchKernelGGL(k_signal<false>, dim3((unsigned)b->n_chains), dim3(64), smem, c->stream, P);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    b->ran = true;
    c->next_run++;
    return SQG_OK;
}

extern "C" int sqg_batch_wait(sqg_ctx_t* c, sqg_batch_t* b, sqg_result_t* res) {
    if (!c || !b || !b->ran) return SQG_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!b->waited) {
        unsigned int e = 0;
        HIPCHK(c, hipMemcpy(&e, c->d_err, sizeof e, hipMemcpyDeviceToHost));
        if (e) {
            HIPCHK(c, hipMemset(c->d_err, 0, sizeof e));
            c->err = "device reported: " + std::string((e & 1) ? "dwell>65535 " : "") + ((e & 2) ? "read>=UINT32_MAX samples " : "") + ((e & 4) ? "internal length mismatch" : "");
            return (e & 4) ? SQG_EDEVICE : SQG_EOVERFLOW;
        }
        float d = 0, s = 0, t = 0;
        HIPCHK(c, hipEventElapsedTime(&d, c->ev[0], c->ev[1]));
        HIPCHK(c, hipEventElapsedTime(&s, c->ev[2], c->ev[3]));
        HIPCHK(c, hipEventElapsedTime(&t, c->ev[0], c->ev[3]));
        c->timing.dwell_ms = d; c->timing.signal_ms = s; c->timing.total_ms = t; c->timing.fallback_samples = 0;
        b->waited = true;
    }
    if (res) {
        res->n_reads = b->n; res->n_events = b->n_events; res->n_samples = b->n_samples; res->n_bases = b->n_bases;
        res->sig_off = (const int64_t*)b->sig_off.data(); res->ev_off = (const int64_t*)b->ev_off.data();
        res->offset = b->offset.data(); res->median_before = b->median.data();
        res->d_signal = c->d_sig; res->d_dwell = c->use_dwell_stream ? c->d_dwell : nullptr;
    }
    return SQG_OK;
}

extern "C" int sqg_fetch_signal(sqg_ctx_t* c, sqg_batch_t* b, int16_t* dst) {
    if (!c || !b || !b->ran || !dst) return SQG_EINVAL;
    if (b->seq + 1 != c->next_run) return SQG_ESEQUENCE;      // slab already reused
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (b->n_samples) HIPCHK(c, hipMemcpy(dst, c->d_sig, (size_t)b->n_samples * sizeof(int16_t), hipMemcpyDeviceToHost));
    return SQG_OK;
}

extern "C" int sqg_fetch_dwell(sqg_ctx_t* c, sqg_batch_t* b, int32_t* dst) {
    if (!c || !b || !b->ran || !dst) return SQG_EINVAL;
    if (b->seq + 1 != c->next_run) return SQG_ESEQUENCE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!c->use_dwell_stream) {
        for (long long i = 0; i < b->n_events; i++) dst[i] = (int)c->cfg.profile.dwell_mean;
        return SQG_OK;
    }
    std::vector<uint16_t> tmp((size_t)b->n_events);
    if (b->n_events) HIPCHK(c, hipMemcpy(tmp.data(), c->d_dwell, tmp.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); i++) dst[i] = tmp[i];
    return SQG_OK;
}

extern "C" int sqg_get_timing(sqg_ctx_t* c, sqg_timing_t* t) {
    if (!c || !t) return SQG_EINVAL;
    *t = c->timing;
    return SQG_OK;
}

extern "C" int sqg_submit(sqg_ctx_t* c, int32_t n, const char* seqs, const int64_t* seq_off,
                          const int32_t* worker, sqg_batch_t** out, sqg_result_t* res) {
    if (!out) return SQG_EINVAL;
    int rc = sqg_batch_stage(c, n, seqs, seq_off, worker, out);
    if (rc) return rc;
    if ((rc = sqg_batch_run(c, *out)) || (rc = sqg_batch_wait(c, *out, res))) { sqg_batch_free(c, *out); *out = nullptr; }
    return rc;
}

extern "C" int sqg_probe_store_bandwidth(sqg_ctx_t* c, size_t bytes, int iters, float* ms_per_pass) {
    if (!c || !ms_per_pass || iters < 1 || bytes < 4096) return SQG_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    void* buf = nullptr;
    HIPCHK(c, hipMalloc(&buf, bytes));
    const size_t n16 = bytes / 16;
    hipEvent_t a, z;
    HIPCHK(c, hipEventCreate(&a)); HIPCHK(c, hipEventCreate(&z));
    hipLaunchKernelGGL(k_store_probe, dim3(256 * 8), dim3(256), 0, c->stream, (uint4*)buf, n16, 1u);   // warm-up
    HIPCHK(c, hipEventRecord(a, c->stream));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_store_probe, dim3(256 * 8), dim3(256), 0, c->stream, (uint4*)buf, n16, (uint32_t)i);
    HIPCHK(c, hipEventRecord(z, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, a, z));
    *ms_per_pass = ms / iters;
    (void)hipEventDestroy(a); (void)hipEventDestroy(z); (void)hipFree(buf);
    return SQG_OK;
}
