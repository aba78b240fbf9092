// sqg_kernels.h -- gfx950 device code of the per-read signal path (included by sqg_hip.hip).
//
//   k_init_rows   per-(worker,k-mer) stream seeds                       (src/sim.c:238-257)
//   k_dwell       per-event dwell draw from the worker's time stream   (src/gensig.c:254-257)
//   k_scan        read lengths -> output offsets
//   k_events      per worker chain: ranks, in-order hand-out of the k-mer streams
//   k_samples     per 64-event tile: the samples                       (src/gensig.c:226-356)
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
#include <type_traits>
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

// fp32 evaluation of the same deviate from the canonical first draw c1 (the second one is a function of it).
// v_log_f32 is log2, v_cos_f32 takes turns.  The 6.2831853-vs-2*pi ratio (1 - 1.1e-9) is below fp32
// resolution; the sweep (k_certify) prices it with everything else.
__device__ static inline float box_muller_fast(uint32_t c1) {
    const float uf = (float)c1 * 4.656612873077393e-10f;                  // c1 * 2^-31 (exact scaling)
    const float lg = __builtin_amdgcn_logf(uf);
    const float y = __builtin_fmaf(lg, -1.3862943611198906f, -9.313225750491594e-10f);   // -2 ln(c1/M)
    const float r = __builtin_amdgcn_sqrtf(y);
    // second uniform c2/M = frac(a*c1/M): four full-rate FP64/convert instructions instead of a modular
    // multiplication plus an int->float conversion (the product is exact to 2^-39, far below fp32 resolution)
    const double t2 = (double)c1 * (16807.0 / 2147483647.0);                // a / M
    const float cs = __builtin_amdgcn_cosf((float)__builtin_amdgcn_fract(t2));
    return r * cs;
}

// dwell draw in FP64 (src/gensig.c:255), kept out of line: it is taken for ~4e-5 of the events and must not
// set the register budget of the kernels that call it
__device__ __attribute__((noinline)) static int dwell_exact(uint32_t c1, double dstd, double dmean) {
    const double z = box_muller_exact(c1, lcg_mul(c1, LCG_A));
    return (int)round((z * dstd) + dmean);
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
    int tile_off;         // first 64-event tile of this read in the batch's tile arrays
    int fast;             // certified mode: every ADC value of this read is provably in (2, 65000) -> lean kernel
    int stile_off;        // first 256-event super tile of this read
    int pad;
};

struct FixEntry {         // one sample handed to the FP64 path
    long long at;         // absolute index into the signal slab
    long long ev;         // batch-wide event index (k-mer recomputed from the bases)
    uint32_t c1;          // first draw of the sample
    int read;
    int shifted;          // inside the RNA adaptor level-shift window
    int pad;
};

// work item of k_samples_lean (256 consecutive events of one read), filled by k_items
struct ItemDesc {
    long long ev_first;          // index (in evrec / dwell) of the item's first event
    long long sig_base;          // index (in sig) of the read's first sample
    double offset;               // the read's slow5 offset
    int n_ev;                    // events in the item (1..256); 0: not taken (queued for the generic kernel, or empty)
    int n_samples;               // samples in the item
    uint32_t at0;                // position within the read of the item's first sample (RNA: counted from the read's end)
    int ev_read0;                // index within the read of the item's first event
    int read;                    // read index (fix-up overflow path)
    int pad;
};

struct SigParams {
    const ReadDesc* reads;
    const int* chain_off;        // [n_chains+1]
    const int* chain_reads;      // read indices grouped per worker chain, batch order inside a chain
    const int* chain_order;      // launch order (longest chain first)
    const uint8_t* bases;
    const uint16_t* dwell;       // per event (null when dwell is constant)
    uint16_t* dwell_out;         // k_events with inline dwell draws: the same array, written
    unsigned long long* seglen_out;    // ... and the per-read segment totals
    double dmean, dstd;          // dwell_mean, dwell_std
    const unsigned long long* seglen;  // [2*n_reads] samples in segment 0 / 1
    const long long* sig_off;    // [n_reads+1]
    const float2* model;         // {level_mean, (float)(level_stdv*amp_noise)}
    const uint32_t* pw;
    uint32_t* rows;              // [n_local_workers][num_kmer]: k <= 6 the stream states; k > 6 the samples each stream has produced
    uint32_t seed_base, seed_step;   // (seed + worker_lo*(4^k+10)) mod M and (4^k+10) mod M: the initial state of local worker w,
                                     // k-mer j is (seed_base + w*seed_step + j) mod M (src/sim.c:238-256)
    int16_t* sig;
    unsigned int* err;
    FixEntry* fix;               // certified mode: undecided samples
    unsigned int* fix_count;
    unsigned int fix_cap;
    uint2* evrec;                // per event {stream state at its first draw, k-mer rank}
    uint32_t* tile_so;           // per 64-event tile: its first sample within the read
    const int* tile_read;        // per tile: read index
    const int* stile_read;       // per 256-event super tile (lean kernel work item): read index
    ItemDesc* items;             // per super tile: what k_samples_lean needs, in one 48-B record
    int lean_epl;                // events per lane of the lean kernel (4, 2 or 1): a super tile is 64*lean_epl events
    int* slow_tiles;             // tiles the lean sample kernel left to the generic one
    unsigned int* slow_count;
    uint4* tfix;                 // lean kernel: FIX_SLOTS undecided samples per tile {index in read, c1, event in read, 0}
    unsigned char* tfix_n;       // lean kernel: entries used per tile
    double dig, range, kd;       // kd = dig/range
    float delta_x;               // swept bound on |x_fast - x_exact| (incl. margin)
    float thr_all;               // 1/2 - (largest eps over all k-mers): acceptance threshold of the lean kernel
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
// Event e of a read uses draws 2e+1, 2e+2 after the worker's time-stream state at the start of
// the read: position addressed by the jump a^(2e) (two LDS table levels, a third in memory).
#define DW_RD 16           // read descriptors cached per block (reads are >= ~190 events)
#define DW_IT 8            // events per thread: the block's fixed latencies (tables, descriptors) are paid once per 2048 events
#define DW_EPB (256 * DW_IT)
template <int MODE>
__global__ __launch_bounds__(256) void k_dwell(const ReadDesc* __restrict__ reads, int n_reads,
                                               const int* __restrict__ blk_read, long long n_events,
                                               const uint32_t* __restrict__ pw, double dmean, double dstd,
                                               float delta_x,
                                               uint16_t* __restrict__ dwell,
                                               unsigned long long* __restrict__ seglen,
                                               unsigned int* __restrict__ err) {
    __shared__ uint32_t j0[POW_N], j1[POW_N];          // a^(2j), a^(2*1024*j)
    __shared__ long long r_ev[DW_RD + 1];
    __shared__ uint32_t r_c0[DW_RD];
    __shared__ int r_ne0[DW_RD];
    const int tid = threadIdx.x;
    for (int i = tid; i < POW_N; i += 256) { j0[i] = pw[2 * POW_N + i]; j1[i] = pw[3 * POW_N + i]; }
    const int rb = blk_read[blockIdx.x];
    if (tid <= DW_RD) {
        const int q = rb + tid;
        r_ev[tid] = q < n_reads ? reads[q].ev_off : 0x7fffffffffffffffLL;
        if (tid < DW_RD && q < n_reads) { r_c0[tid] = reads[q].time_c0; r_ne0[tid] = reads[q].ne0; }
    }
    __syncthreads();
    const float sf = (float)dstd, mf = (float)dmean;
    const float mag = fabsf(mf) + 7.0f * fabsf(sf) + 1.0f;
    // delta_x*s (swept) + float roundings of s, m, the fma and the +1/2 (each <= 2^-24 * mag) + slack
    const float eps = delta_x * fabsf(sf) + 4.0f * 5.9604645e-8f * mag + 1e-6f;
    int q = 0;                                          // cached descriptor index (monotone over the iterations)
    for (int it = 0; it < DW_IT; it++) {
        const long long gid = (long long)blockIdx.x * DW_EPB + it * 256 + tid;
        const bool valid = gid < n_events;
        int r = rb, sps = 0, seg = 0;
        if (valid) {
            while (q + 1 < DW_RD && gid >= r_ev[q + 1]) q++;
            uint32_t e, c0; int ne0;
            if (gid < r_ev[q + 1]) { e = (uint32_t)(gid - r_ev[q]); c0 = r_c0[q]; ne0 = r_ne0[q]; r = rb + q; }
            else {                                     // more than DW_RD reads in one block: walk the table
                r = rb + q;
                while (r + 1 < n_reads && gid >= reads[r + 1].ev_off) r++;
                e = (uint32_t)(gid - reads[r].ev_off); c0 = reads[r].time_c0; ne0 = reads[r].ne0;
            }
            uint32_t jp = j0[e & (POW_N - 1)];
            const uint32_t hi = (e >> 10) & (POW_N - 1), hi2 = e >> 20;
            if (hi) jp = lcg_mul(jp, j1[hi]);
            if (hi2) jp = lcg_mul(jp, pw[4 * POW_N + hi2]);
            const uint32_t c1 = lcg_mul(lcg_mul(c0, jp), LCG_A);
            bool decided = false;
            if (MODE == 1) {
                // v' = x'*s + m in fp32; round(v) = floor(v+1/2) unless v is within eps of a half-integer
                const float x = box_muller_fast(c1);
                const float g = __builtin_fmaf(x, sf, mf) + 0.5f;
                const float fl = floorf(g);
                const float fr = g - fl;
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
            seg = e >= (uint32_t)ne0;
        }
        // per-read totals: one atomic per wavefront when the wave is inside one (read, segment)
        const int key = valid ? (r * 2 + seg) : -1;
        const int key0 = __shfl(key, 0);
        if (__all(key == key0)) {
            int sum = sps;
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
            if ((tid & 63) == 0 && key0 >= 0) atomicAdd(&seglen[key0], (unsigned long long)sum);
        } else if (valid) {
            atomicAdd(&seglen[key], (unsigned long long)sps);
        }
    }
}

// ---- k_scan: sig_off = exclusive scan of per-read totals (single workgroup) -----------------
// sig_off goes to HBM for the kernels and, through the pinned host mapping, straight to the host (no D2H copy
// between kernels): host_off is visible once the stream has been synchronised.
#define SCAN_PER 8
__global__ __launch_bounds__(1024) void k_scan(const unsigned long long* __restrict__ seglen, int n_reads,
                                               long long* __restrict__ sig_off, long long* __restrict__ host_off,
                                               unsigned int* __restrict__ err, unsigned int* __restrict__ counters) {
    __shared__ long long wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid < 4) counters[tid] = 0;                      // fix-up list / slow-tile list counters of this batch
    // one pass: thread t owns reads [t*per, (t+1)*per)
    const int per = (n_reads + 1023) / 1024;
    const int lo = min(tid * per, n_reads), hi = min(lo + per, n_reads);
    const ulonglong2* sl = reinterpret_cast<const ulonglong2*>(seglen);
    long long len[SCAN_PER];
    long long v = 0;
    bool big = false;
    if (per <= SCAN_PER) {                               // the usual case: all loads in flight together
#pragma unroll
        for (int j = 0; j < SCAN_PER; j++) {
            ulonglong2 q = make_ulonglong2(0, 0);
            if (lo + j < hi) q = sl[lo + j];
            len[j] = (long long)(q.x + q.y);
            big |= len[j] >= 4294967295LL;
            v += len[j];
        }
    } else {
        for (int i = lo; i < hi; i++) {
            const ulonglong2 q = sl[i];
            const long long l = (long long)(q.x + q.y);
            big |= l >= 4294967295LL;
            v += l;
        }
    }
    if (big) atomicOr(err, 2u);                          // src/sim.c:559-562
    long long x = v;
    for (int o = 1; o < 64; o <<= 1) { long long y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    long long run = x - v;
    for (int w = 0; w < wid; w++) run += wsum[w];
    if (per <= SCAN_PER) {
#pragma unroll
        for (int j = 0; j < SCAN_PER; j++) {
            if (lo + j < hi) { sig_off[lo + j] = run; if (host_off) host_off[lo + j] = run; }
            run += len[j];
        }
    } else {
        for (int i = lo; i < hi; i++) {
            sig_off[i] = run; if (host_off) host_off[i] = run;
            const ulonglong2 q = sl[i];
            run += (long long)(q.x + q.y);
        }
    }
    if (tid == 1023) { sig_off[n_reads] = run; if (host_off) host_off[n_reads] = run; }   // the last thread's running total is the grand total
}

// ---- k_events + k_samples ------------------------------------------------------------------
// The per-read loop nest of src/gensig.c:249-282 is split at its only sequential dependency:
//
// k_events   one workgroup of NT threads per worker chain (a worker's reads of this batch, in
//            batch order); a read is walked in segments of NT consecutive events, one event per
//            thread: k-mer rank, dwell, block scan -> first sample of each 64-event tile, and the
//            hand-out of the per-(worker,k-mer) Lehmer streams IN EVENT ORDER: events are binned
//            by k-mer in an LDS hash table, bin members listed through a block scan, and each
//            event sums the dwell of the same-k-mer events before it (bins hold 1-3 events).
//            Stream states live in HBM/L2 (rows[worker][rank]): one load per event and one store
//            per bin, advanced by an O(1) jump a^(2*samples).  Output: 8 B per event
//            {state at the event's first draw, rank}.
// k_samples  one wavefront per 64-event tile, no inter-wave dependency and no block barrier:
//            64 consecutive samples per step (contiguous int16 stores).  sample -> event through
//            start-marker bytes in LDS + ballot/mbcnt; the two draws of a sample are two modular
//            multiplications of the event's state with per-slot constants a^(2j+1), a^(2j+2).
#ifndef SQG_EVENT_THREADS
#define SQG_EVENT_THREADS 256
#endif
#ifndef SQG_EVENT_EPT
#define SQG_EVENT_EPT 2     // consecutive events per thread of k_events (segment = SQG_EVENT_THREADS * SQG_EVENT_EPT events)
#endif
#ifndef SQG_EVENT_WAVES
#define SQG_EVENT_WAVES 6   // waves per SIMD the register allocation of k_events aims at (LDS allows 7 workgroups per CU)
#endif
#define MK_W 1024          // marker window (samples) per wavefront
#define MULT_N 512         // LDS jump constants cover events of up to 512 samples
#define BIN_EMPTY 0xffffffffu

// inclusive wave scan with DPP row shifts/broadcasts (6 VALU, no LDS)
__device__ static inline int wave_incl_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

__device__ static inline int wave_incl_scan(int v, int lane) {
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(v, o); if (lane >= o) v += y; }
    return v;
}

#define LEAN_MAGIC 12582912.0f             // 1.5 * 2^23: t = v + MAGIC rounds v to the nearest integer, in the low bits of t
#define EV_NIL 0xffffu
#define ROW_BUSY 0x80000000u
#define EV_HALO 20          // 2*(k_max-1)+2 extra base codes per segment (segment-0/1 boundary)

// DIRECT (k <= 6): one bin per k-mer rank, no keys, no probing; otherwise an open-addressing hash of 2*SEG bins.
// A segment is SEG = NT*EPT consecutive events of a read, EPT consecutive events per thread.
template <int NT, bool DIRECT, int EPT>
struct EvLds {
    static constexpr int SEG = NT * EPT;
    uint32_t keys[DIRECT ? 1 : 2 * SEG];     // hash bins: k-mer rank
    uint32_t head[DIRECT ? 1 : 2 * SEG];     // hash bin -> most recently inserted event of the segment (EV_NIL: none)
    uint32_t row[DIRECT ? 4096 : 1];         // DIRECT: the worker's stream states, resident for the whole chain; while a segment
                                             // is being handed out, ROW_BUSY | (most recently inserted event of the bin)
    uint32_t st[SEG];                        // DIRECT: the state the bin's first exchanger swapped out of row[]; else: the
                                             // bin's state at the start of the segment, published by its first event
    uint32_t nxt[SEG];          // per event: (dwell << 16) | next event in the same bin
    uint32_t jump[(MULT_N > SEG ? MULT_N : SEG)];    // a^(2j)
    uint8_t codes[SEG + EV_HALO + 4];  // 2-bit base codes of the segment
    uint8_t lut[256];           // base -> 2-bit code (src/seq.h:14-27)
    int wsum[NT / 64];
};

// LDS-only workgroup barrier: does not wait for outstanding global loads/stores
__device__ static inline void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DW: 0 = dwell comes from memory (k_dwell ran) or is constant; 1 = drawn here, certified fp32 path with
// out-of-line FP64 fallback; 2 = drawn here in FP64 (src/gensig.c:254-257)
template <int NT, bool DIRECT, int DW, int EPT>
__global__ __launch_bounds__(NT, SQG_EVENT_WAVES) void k_events(const SigParams P) {
    typedef EvLds<NT, DIRECT, EPT> Lds;
    __shared__ Lds L;
    __shared__ long long n1_sh;
    constexpr int NW = NT / 64, SEG = NT * EPT, HT = 2 * SEG, TL = 64 / EPT;   // TL: lanes per 64-event tile
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < (MULT_N > SEG ? MULT_N : SEG); i += NT) L.jump[i] = P.pw[2 * POW_N + i];
    for (int i = tid; i < 256; i += NT) L.lut[i] = (uint8_t)base_code((uint8_t)i);

    const int chain = P.chain_order[blockIdx.x];
    const int c_lo = P.chain_off[chain], c_hi = P.chain_off[chain + 1];
    uint32_t* row = P.rows ? P.rows + (size_t)P.reads[P.chain_reads[c_lo]].worker * P.num_kmer : nullptr;
    const int k = P.k;
    const uint32_t kmask = (k >= 16) ? 0xffffffffu : ((1u << (2 * k)) - 1u);
    // k > 6: initial state of this worker's k-mer j is (seed_w + j) mod M (src/sim.c:249)
    const uint32_t seed_w = (uint32_t)(((unsigned long long)P.seed_base +
                                        (unsigned long long)(P.rows ? P.reads[P.chain_reads[c_lo]].worker : 0) * P.seed_step) % LCG_M);
    if (DIRECT && P.use_streams) for (int i = tid; i < P.num_kmer; i += NT) L.row[i] = row[i];
    const uint32_t a2nt = DW ? lcg_jump2(P.pw, (uint32_t)SEG) : 0u;      // time-stream jump over one segment
    const float dw_sf = (float)P.dstd, dw_mf = (float)P.dmean;
    // delta_x*s (swept) + float roundings of s, m and the fma (each <= 2^-24 * mag) + slack
    const float dw_eps = P.delta_x * fabsf(dw_sf) + 4.0f * 5.9604645e-8f * (fabsf(dw_mf) + 7.0f * fabsf(dw_sf) + 1.0f) + 1e-6f;
    __syncthreads();

    for (int ci = c_lo; ci < c_hi; ci++) {
        const int r = P.chain_reads[ci];
        const ReadDesc rd = P.reads[r];
        const int ne = rd.ne0 + rd.ne1;
        const uint8_t* rbases = P.bases + rd.base_off;
        const int nbytes = rd.len0 + rd.len1;                           // <= 2^31 (checked at staging)
        // base index of event e: e in segment 0, e + (k-1) in segment 1 (the stall's k-mers do not
        // straddle the boundary, src/genread.c:87-88)
        #define EV_BASE(e_) ((int)(e_) + ((e_) >= rd.ne0 ? rd.len0 - rd.ne0 : 0))
        uint32_t done = 0;                                            // samples before this segment
        uint32_t c_seg = DW ? __builtin_amdgcn_readfirstlane(lcg_mul(rd.time_c0, LCG_A)) : 0u;   // a * (time-stream state at the segment's first event)
        if (DW && tid == 0) n1_sh = -1;
        // prefetch of segment 0: EPT base bytes per thread (+ halo), EPT dwells per thread
        uint8_t b_cur[EPT], b_halo = 'A';
        uint16_t d_cur[EPT];
        {
            const int b0 = EV_BASE(0);
#pragma unroll
            for (int q = 0; q < EPT; q++) {
                const int bi = b0 + tid * EPT + q;
                b_cur[q] = bi < nbytes ? rbases[bi] : (uint8_t)'A';
                d_cur[q] = (!DW && tid * EPT + q < ne && P.dwell) ? P.dwell[rd.ev_off + tid * EPT + q] : (uint16_t)0;
            }
            if (tid < EV_HALO && b0 + SEG + tid < nbytes) b_halo = rbases[b0 + SEG + tid];
        }
        // One segment.  FULL: every event of the segment exists (all but a read's last segment) -- the per-lane
        // validity tests, and the exec-mask juggling they cost on the scalar unit, are compiled out.
        #define EV_IN(e_) (FULL || (e_) < ne)
        auto segment = [&](auto full_tag, const int s0) {
            constexpr bool FULL = decltype(full_tag)::value;
            const int e0 = s0 + tid * EPT;                            // my first event
            const int bseg = EV_BASE(s0);
            uint8_t code_cur[EPT];
#pragma unroll
            for (int q = 0; q < EPT; q++) code_cur[q] = L.lut[b_cur[q]];                     // consumed after the dwell draw
            const uint8_t code_halo = L.lut[tid < EV_HALO ? b_halo : (uint8_t)'A'];
            int sps[EPT];
#pragma unroll
            for (int q = 0; q < EPT; q++) {
                const int e = e0 + q;
                const bool valid = EV_IN(e);
                sps[q] = 0;
                if (DW == 0) {
                    sps[q] = valid ? (P.dwell ? (int)d_cur[q] : P.const_sps) : 0;
                } else if (valid) {
                    // event e uses draws 2e+1, 2e+2 of the worker's time stream after the read's first state
                    const uint32_t c1 = lcg_mul(c_seg, L.jump[tid * EPT + q]);
                    bool decided = false;
                    int v = 0;
                    if (DW == 1) {
                        // round(v) is the integer nearest to v' unless v' is within eps of a half-integer
                        const float x = box_muller_fast(c1);
                        const float g = __builtin_fmaf(x, dw_sf, dw_mf);
                        const float t = g + LEAN_MAGIC;              // |g| < 2^22: the host takes the FP64 variant (DW 2) when dwell_hi >= 1e6
                        const float fl = t - LEAN_MAGIC;
                        if (fabsf(g - fl) < 0.5f - dw_eps && c1 <= LCG_M - (1u << NEAR_ONE_BITS)) { v = (int)__float_as_uint(t) - 0x4b400000; decided = true; }
                    }
                    if (!decided) v = dwell_exact(c1, P.dstd, P.dmean);      // src/gensig.c:255
                    v = v < 1 ? -v + 1 : v;                                  // src/gensig.c:256
                    if (v > 65535) { atomicOr(P.err, 1u); v = 65535; }
                    sps[q] = v;
                    P.dwell_out[rd.ev_off + e] = (uint16_t)v;
                }
            }
            if (DW) c_seg = __builtin_amdgcn_readfirstlane(lcg_mul(c_seg, a2nt));     // wave-uniform: scalar unit
#pragma unroll
            for (int q = 0; q < EPT; q++) L.codes[tid * EPT + q] = code_cur[q];
            if (tid < EV_HALO) L.codes[SEG + tid] = code_halo;
            int lane_total = 0;
#pragma unroll
            for (int q = 0; q < EPT; q++) lane_total += sps[q];
            const int incl = wave_incl_scan_dpp(lane_total);
            if (lane == 63) L.wsum[wid] = incl;
            if (!DIRECT && P.use_streams) for (int i = tid; i < HT; i += NT) { L.keys[i] = BIN_EMPTY; L.head[i] = EV_NIL; }
            lds_barrier();                                                                    // (1)
            int woff = 0, seg_total = 0;
            for (int w = 0; w < NW; w++) { const int x = L.wsum[w]; if (w < wid) woff += x; seg_total += x; }
            const int lane_excl = woff + incl - lane_total;           // samples of this segment before my first event
            uint32_t rank[EPT], h[EPT], swapped[EPT], my_prev[EPT];   // my_prev: the event inserted into my bin just before me
#pragma unroll
            for (int q = 0; q < EPT; q++) {
                const int e = e0 + q;
                rank[q] = 0; swapped[q] = 0; my_prev[q] = EV_NIL;
                if (EV_IN(e)) {
                    const int cb = EV_BASE(e) - bseg;
                    if (q > 0 && e != rd.ne0) {
                        rank[q] = ((rank[q - 1] << 2) | L.codes[cb + k - 1]) & kmask;          // my previous event's k-mer, shifted by one base
                    } else {
                        // src/seq.h:31-42; the usual k are unrolled so that the byte reads are in flight together
                        uint32_t rk = 0;
                        #define EV_RANK(K_) { _Pragma("unroll") for (int i = 0; i < K_; i++) rk = (rk << 2) | L.codes[cb + i]; }
                        switch (k) {
                        case 6: EV_RANK(6) break;
                        case 9: EV_RANK(9) break;
                        case 5: EV_RANK(5) break;
                        default: for (int i = 0; i < k; i++) rk = (rk << 2) | L.codes[cb + i];
                        }
                        #undef EV_RANK
                        rank[q] = rk;
                    }
                }
                h[q] = DIRECT ? rank[q] : (rank[q] * 2654435761u) >> (32 - (31 - __builtin_clz(HT)));
                if (P.use_streams && EV_IN(e)) {
                    const uint32_t id = (uint32_t)(tid * EPT + q);    // event within the segment, in event order
                    if (DIRECT) {
                        // the bin's members chain through row[rank]; the first one of the segment takes the state out
                        swapped[q] = atomicExch(&L.row[rank[q]], ROW_BUSY | id);
                        if (swapped[q] & ROW_BUSY) my_prev[q] = swapped[q] & 0xffffu; else L.st[id] = swapped[q];
                        L.nxt[id] = ((uint32_t)sps[q] << 16) | my_prev[q];
                    } else {
                        for (;;) {
                            const uint32_t old = atomicCAS(&L.keys[h[q]], BIN_EMPTY, rank[q]);
                            if (old == BIN_EMPTY || old == rank[q]) break;
                            h[q] = (h[q] + 1) & (HT - 1);
                        }
                        my_prev[q] = atomicExch(&L.head[h[q]], id);
                        L.nxt[id] = ((uint32_t)sps[q] << 16) | my_prev[q];
                    }
                }
            }
            if (DIRECT) lds_barrier(); else __syncthreads();                                  // (2) global rows: + earlier row stores have landed
            // prefetch the next segment's inputs; they land while this segment waits for its states
            {
                const int s1 = s0 + SEG;
                if (s1 < ne) {
                    const int b1 = EV_BASE(s1);
#pragma unroll
                    for (int q = 0; q < EPT; q++) {
                        const int bi = b1 + tid * EPT + q;
                        b_cur[q] = bi < nbytes ? rbases[bi] : (uint8_t)'A';
                        if (!DW && s1 + tid * EPT + q < ne && P.dwell) d_cur[q] = P.dwell[rd.ev_off + s1 + tid * EPT + q];
                    }
                    if (tid < EV_HALO) b_halo = (b1 + SEG + tid < nbytes) ? rbases[b1 + SEG + tid] : (uint8_t)'A';
                }
            }
            // first sample of every 64-event tile (TL lanes) within the read
            if ((lane & (TL - 1)) == 0 && EV_IN(e0)) P.tile_so[rd.tile_off + (e0 >> 6)] = done + (uint32_t)lane_excl;
            uint32_t c_ev[EPT];
            {
                int run = lane_excl;
#pragma unroll
                for (int q = 0; q < EPT; q++) {
                    if (DW && EV_IN(e0 + q) && e0 + q == rd.ne0) n1_sh = (long long)done + run;   // samples of segment 0
                    run += sps[q];
                    c_ev[q] = 0;
                }
            }
            if (P.use_streams) {
                // dwell drawn from my k-mer's stream by earlier events of this segment, by all of them,
                // and whether I am the last one (who stores the advanced state)
                // the bin's FIRST event (prior == 0) stores the advanced state, so that every event has exactly one
                // modular multiplication: a^(2*prior) for its own state, or a^(2*total) for the bin's next state
                uint32_t prior[EPT], total[EPT], c_row[EPT], fid[EPT];   // fid: the bin's first event (in event order)
                bool first[EPT];
#pragma unroll
                for (int q = 0; q < EPT; q++) {
                    prior[q] = 0; total[q] = (uint32_t)sps[q]; c_row[q] = 0; first[q] = true; fid[q] = 0;
                    if (EV_IN(e0 + q)) {
                        const uint32_t id = (uint32_t)(tid * EPT + q);
                        // walk the bin's other members (bins hold 1-3 events; alone: no iteration)
                        uint32_t t;
                        if (DIRECT) {
                            c_row[q] = swapped[q];                                           // the state itself if I was first to exchange
                            t = L.row[rank[q]] & 0xffffu;                                    // most recently inserted event
                        } else t = L.head[h[q]];
                        if (t == id) t = my_prev[q];
                        fid[q] = id;
                        while (t != EV_NIL) {
                            const uint32_t v = L.nxt[t];
                            const uint32_t s2 = v >> 16, nx = v & 0xffffu;
                            total[q] += s2;
                            if (t < id) { prior[q] += s2; first[q] = false; fid[q] = min(fid[q], t); }
                            if (DIRECT && nx == EV_NIL) c_row[q] = L.st[t];                   // the first to exchange holds the state
                            t = (nx == id) ? my_prev[q] : nx;
                        }
                        if (!DIRECT && first[q]) {
                            // one returning atomic per bin: the samples this stream had produced before the segment; its
                            // state is the seed advanced by two draws per sample
                            const uint32_t n_old = atomicAdd(&row[rank[q]], total[q]);
                            const unsigned long long sv = (unsigned long long)seed_w + rank[q];
                            uint32_t cb = (uint32_t)(sv >= LCG_M ? sv - LCG_M : sv);
                            if (n_old) cb = lcg_mul(cb, n_old < MULT_N ? L.jump[n_old] : lcg_jump2(P.pw, n_old));
                            c_row[q] = cb;
                            L.st[id] = cb;
                        }
                    }
                }
                if (DIRECT) lds_barrier(); else __syncthreads();                                // (3) every state read before any is advanced
#pragma unroll
                for (int q = 0; q < EPT; q++) {
                    if (EV_IN(e0 + q)) {
                        if (DIRECT) {
                            const uint32_t n = first[q] ? total[q] : prior[q];                  // > 0: every event has >= 1 sample
                            const uint32_t m = lcg_mul(c_row[q], n < MULT_N ? L.jump[n] : lcg_jump2(P.pw, n));
                            if (first[q]) { c_ev[q] = c_row[q]; L.row[rank[q]] = m; }
                            else c_ev[q] = m;
                        } else if (first[q]) c_ev[q] = c_row[q];
                        else c_ev[q] = lcg_mul(L.st[fid[q]], prior[q] < MULT_N ? L.jump[prior[q]] : lcg_jump2(P.pw, prior[q]));
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < EPT; q++)
                if (EV_IN(e0 + q)) P.evrec[rd.ev_off + e0 + q] = make_uint2(c_ev[q], rank[q]);
            done += (uint32_t)seg_total;
            // no barrier here: every LDS structure rewritten at the top of the next segment (codes, wsum, bins) was last
            // read before barrier (2)/(3) of this one, which every thread has passed
        };
        #undef EV_IN
        for (int s0 = 0; s0 < ne; s0 += SEG) {
            if (s0 + SEG <= ne) segment(std::true_type{}, s0); else segment(std::false_type{}, s0);
        }
        #undef EV_BASE
        __syncthreads();                                // the chain's next read starts with this read's stores landed
        if (tid == 0) {
            if (DW) {
                const long long n1 = n1_sh >= 0 ? n1_sh : (long long)done;
                P.seglen_out[2 * r] = (unsigned long long)n1;
                P.seglen_out[2 * r + 1] = (unsigned long long)((long long)done - n1);
            } else if ((long long)done != (long long)(P.seglen[2 * r] + P.seglen[2 * r + 1])) atomicOr(P.err, 4u);
        }
        __syncthreads();
    }
    if (DIRECT && P.use_streams) for (int i = tid; i < P.num_kmer; i += NT) row[i] = L.row[i];
}

struct SmpWaveLds {
    uint4 rec_a[64];            // {c_ev, first sample in tile, F | level_mean, sdk | sd}
    uint2 rec_b[64];            // {I | constant sample, thr}
    uint8_t mk[MK_W];           // event-start markers of the current sample window
};
struct SmpLds {
    uint2 mult[MULT_N];         // {a^(2j+1), a^(2j+2)}
    SmpWaveLds w[4];
};

__device__ static inline void push_fix(const SigParams& P, bool bad, int lane, unsigned long long lane_le,
                                       long long at, uint32_t c1, long long ev, int r, int shifted) {
    const unsigned long long am = __ballot(bad);
    if (am) {                                                  // hand the undecided samples to k_fixup
        unsigned int slot0 = 0;
        const int leader = __ffsll((long long)am) - 1;
        if (lane == leader) slot0 = atomicAdd(P.fix_count, (unsigned int)__popcll(am));
        slot0 = __shfl(slot0, leader);
        if (bad) {
            const unsigned int slot = slot0 + (unsigned int)__popcll(am & lane_le) - 1u;
            if (slot < P.fix_cap) {
                FixEntry fe; fe.at = at; fe.c1 = c1; fe.ev = ev; fe.read = r; fe.shifted = shifted; fe.pad = 0;
                P.fix[slot] = fe;
            } else atomicOr(P.err, 8u);
        }
    }
}

// one undecided sample from a divergent region (rare overflow path of the lean kernel)
__device__ static inline void push_fix_one(const SigParams& P, long long at, uint32_t c1, long long ev, int r, int shifted) {
    const unsigned int slot = atomicAdd(P.fix_count, 1u);
    if (slot < P.fix_cap) {
        FixEntry fe; fe.at = at; fe.c1 = c1; fe.ev = ev; fe.read = r; fe.shifted = shifted; fe.pad = 0;
        P.fix[slot] = fe;
    } else atomicOr(P.err, 8u);
}

#define LEAN_EPL_MAX 4                     // events per lane of the lean kernel: 4, 2 or 1 (SigParams.lean_epl, chosen per profile so
                                           // that a work item -- 64*epl consecutive events of a read -- stays below LEAN_MAX_SAMPLES)
#define FIX_SLOTS 8                        // parked undecided samples per super tile (expected ~0.5); overflow -> global list
#define LEAN_MAX_SAMPLES 4096              // samples per work item the 64x64-bit start map covers

// k_items: one thread per 256-event super tile.  Collapses the dependent look-ups of the lean kernel's set-up
// (tile -> read -> tile_so / sig_off / seglen) into one record per item and decides which items the lean
// kernel takes; the others are queued (as 64-event tiles) for k_samples<MODE, GENERIC>.
__global__ __launch_bounds__(256) void k_items(const SigParams P, const int n_stiles, const int n_reads, long long* __restrict__ host_off) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g <= n_reads) host_off[g] = P.sig_off[g];                      // read offsets to the host through the pinned mapping
    if (g >= n_stiles) return;
    const int r = P.stile_read[g];
    const ReadDesc rd = P.reads[r];
    const int LEAN_EPL = P.lean_epl, LEAN_EV = 64 * LEAN_EPL;
    const int lt = g - rd.stile_off;                                   // super tile within the read
    const int ne = rd.ne0 + rd.ne1;
    const int n_ev = min(LEAN_EV, ne - lt * LEAN_EV);
    const long long sig_base = P.sig_off[r];
    const uint32_t read_len = (uint32_t)(P.sig_off[r + 1] - sig_base);
    const uint32_t base_pos = P.tile_so[rd.tile_off + lt * LEAN_EPL];
    const uint32_t next_pos = (lt + 1) * LEAN_EV < ne ? P.tile_so[rd.tile_off + (lt + 1) * LEAN_EPL] : read_len;
    const int n_samples = (int)(next_pos - base_pos);
    bool take = rd.fast != 0 && n_samples <= LEAN_MAX_SAMPLES;
    if (P.shift_len > 0) {                                             // RNA adaptor level-shift window (src/genread.c:79-86)
        const long long n1 = (long long)P.seglen[2 * r];
        if ((long long)base_pos + n_samples > n1 - P.shift_len && (long long)base_pos < n1) take = false;
    }
    if (!take) {                                                       // leave these (up to 4) 64-event tiles to the generic kernel
        const int nt = (n_ev + 63) >> 6;
        const unsigned int q = atomicAdd(P.slow_count, (unsigned int)nt);
        for (int i = 0; i < nt; i++) P.slow_tiles[q + i] = rd.tile_off + lt * LEAN_EPL + i;
    }
    ItemDesc d;
    d.ev_first = rd.ev_off + (long long)lt * LEAN_EV;
    d.sig_base = sig_base;
    d.offset = rd.offset;
    d.n_ev = (take && n_samples > 0) ? n_ev : 0;
    d.n_samples = n_samples;
    d.at0 = P.rna ? read_len - 1u - base_pos : base_pos;
    d.ev_read0 = lt * LEAN_EV;
    d.read = r;
    d.pad = 0;
    P.items[g] = d;
    P.tfix_n[g] = 0;
}

template <int EPL>
struct LeanWaveLds {
    uint4 rec[64 * EPL];                // {c_ev, ((8*first sample) & 0xfff) << 16 | I (16 bits), F - 1/2, sdk}
    unsigned long long bm[64];          // bit s-1 set: an event (other than the item's first) starts at sample s
    int nfix;                           // undecided samples of the item so far
    int pad[3];
};
template <int EPL>
struct LeanLds {
    uint32_t mult[MULT_N];              // a^(2j+1): the first draw of an event's sample j is state * mult[j]
    LeanWaveLds<EPL> w[4];
};

// k_samples_lean: the hot kernel.  Certified fp32 path only, for reads whose ADC values are provably in
// (2, 65000) (ReadDesc.fast), events of <= MULT_N samples, outside the RNA level-shift window; everything
// else is queued (as 64-event tiles) for k_samples<MODE, GENERIC>.
// One wavefront per 256 consecutive events of a read (4 per lane: the dependent global round trips of the
// set-up are paid once per ~2300 samples; the item's descriptors are wave-uniform and live in SGPRs).
// Per step 64 consecutive samples:
//   64-bit slice of the event-start map (v_readlane) -> mbcnt -> event -> {state, first|I, F-1/2, sdk}
//   (one ds_read_b128) -> jump constants (ds_read_b64) -> 2 modular multiplications -> v_log/v_sqrt/v_cos ->
//   v' = fma(x, sdk, F-1/2) -> t = v' + 1.5*2^23 (round to nearest: floor of the ADC value unless it is within
//   eps of an integer) -> acceptance test on v' - (t - 1.5*2^23) -> int16 store of the low half of bits(t) + I.
// The loads of step i+1 are issued before the arithmetic of step i (software pipelining, two steps unrolled
// so that the pipeline registers do not have to be copied).
template <bool RNA, int LEAN_EPL>
__global__ __launch_bounds__(256) void k_samples_lean(const SigParams P, const int n_stiles) {
    __shared__ LeanLds<LEAN_EPL> L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < MULT_N; i += 256) L.mult[i] = P.pw[i];
    __syncthreads();
    LeanWaveLds<LEAN_EPL>& W = L.w[wid];
    const float thr = P.thr_all;
    const char* mult_b = reinterpret_cast<const char*>(L.mult);

    for (int g = blockIdx.x * 4 + wid; g < n_stiles; g += gridDim.x * 4) {
        // the item's descriptor is wave-uniform: scalar load (the constant address space forces s_load; k_items wrote it
        // before this kernel started)
        ItemDesc it;
        {
            const __attribute__((address_space(4))) uint32_t* src =
                reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(P.items + g));
            uint32_t w[sizeof(ItemDesc) / 4];
#pragma unroll
            for (int q = 0; q < (int)(sizeof(ItemDesc) / 4); q++) w[q] = src[q];
            __builtin_memcpy(&it, w, sizeof it);
        }
        const int ne = it.n_ev;                                         // events of this item
        if (ne == 0) continue;                                         // not taken, or empty
        const int wave_total = it.n_samples;
        const int e0 = lane * LEAN_EPL;                                // my first event (within the item)
        const long long gev = it.ev_first + e0;
        // ---- set-up: LEAN_EPL consecutive events per lane ----
        uint2 er[LEAN_EPL];
        int sps[LEAN_EPL];
        if (e0 + LEAN_EPL <= ne) {
            uint32_t ew[2 * LEAN_EPL];
            __builtin_memcpy(ew, P.evrec + gev, 8 * LEAN_EPL);                        // 8-B aligned wide loads
#pragma unroll
            for (int q = 0; q < LEAN_EPL; q++) er[q] = make_uint2(ew[2 * q], ew[2 * q + 1]);
            if (P.dwell) {
                uint16_t dw[LEAN_EPL];
                __builtin_memcpy(dw, P.dwell + gev, 2 * LEAN_EPL);                    // 2-B aligned wide load
#pragma unroll
                for (int q = 0; q < LEAN_EPL; q++) sps[q] = (int)dw[q];
            } else {
#pragma unroll
                for (int q = 0; q < LEAN_EPL; q++) sps[q] = P.const_sps;
            }
        } else {
#pragma unroll
            for (int q = 0; q < LEAN_EPL; q++) {
                const bool v = e0 + q < ne;
                er[q] = v ? P.evrec[gev + q] : make_uint2(0u, 0u);
                sps[q] = v ? (P.dwell ? (int)P.dwell[gev + q] : P.const_sps) : 0;
            }
        }
        float2 md[LEAN_EPL];
#pragma unroll
        for (int q = 0; q < LEAN_EPL; q++) md[q] = (e0 + q < ne) ? P.model[er[q].y] : make_float2(0.f, 0.f);
        int lane_total = 0;
#pragma unroll
        for (int q = 0; q < LEAN_EPL; q++) lane_total += sps[q];
        const int incl = wave_incl_scan_dpp(lane_total);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");         // previous item's LDS reads are done
        W.bm[lane] = 0ull;
        if (lane == 0) W.nfix = 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        {
            int run = incl - lane_total;
#pragma unroll
            for (int q = 0; q < LEAN_EPL; q++) {
                const int so = run; run += sps[q];
                // v = s_f*dig/range - offset  ~  x*(sd*kd) + (m*kd - offset) = x*sdk + (I + F), I = floor(.) in (2, 65000)
                const double mk = (double)md[q].x * P.kd - it.offset;
                const double fl0 = floor(mk);
                const float Fh = (float)(mk - fl0 - 0.5);
                const float sdk = (float)((double)md[q].y * P.kd);
                W.rec[lane * LEAN_EPL + q] = make_uint4(er[q].x, ((((uint32_t)so << 3) & 0xfffu) << 16) | ((uint32_t)(int)fl0 & 0xffffu),
                                                        __float_as_uint(Fh), __float_as_uint(sdk));
                if ((e0 + q < ne) && (lane | q) != 0)                  // so >= 1: every earlier event has >= 1 sample
                    atomicOr(reinterpret_cast<unsigned int*>(W.bm) + ((so - 1) >> 5), 1u << ((so - 1) & 31));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const unsigned long long my_bm = W.bm[lane];
        const uint32_t bm_lo = (uint32_t)my_bm, bm_hi = (uint32_t)(my_bm >> 32);
        char* const out_b = reinterpret_cast<char*>(P.sig + it.sig_base);               // wave-uniform: global_store saddr + 32-bit lane offset
        // byte offset of my sample of step 0 within the read: generation index i is stored at at0 + i (RNA: at0 - i)
        uint32_t voff = RNA ? 2u * (it.at0 - (uint32_t)lane) : 2u * (it.at0 + (uint32_t)lane);
        uint32_t idx8 = (uint32_t)lane << 3;                           // 8 * (my sample index within the item)
        const int ev_read0 = it.ev_read0;                               // event index (within the read) of rec[0]
#if defined(SQG_ABL_NOLOOP)
        const int nfull = 0, rem = wave_total & 1;
#else
        const int nfull = wave_total >> 6, rem = wave_total & 63;
#endif
        int base_ev;

        // event of my sample in step c: events begun in earlier steps + start bits below my lane
        #define LEAN_MAP(c_, ev_) {                                                                              \
            const uint32_t lo_ = __builtin_amdgcn_readlane(bm_lo, (c_)), hi_ = __builtin_amdgcn_readlane(bm_hi, (c_)); \
            ev_ = (int)__builtin_amdgcn_mbcnt_hi(hi_, __builtin_amdgcn_mbcnt_lo(lo_, (uint32_t)base_ev));          \
            base_ev += __builtin_popcount(lo_) + __builtin_popcount(hi_); }
        // one step: issue the loads of step c_+1 into (RN, MN, EN), then the arithmetic of step c_ from (RA, MU, EV)
        #define LEAN_STEP(TAIL, c_, RA, MU, EV, RN, MN, EN) {                                                    \
            LEAN_MAP(min((c_) + 1, 63), EN)                                                                       \
            RN = W.rec[EN];                                                                                       \
            LEAN_ARITH(RA, MU)                                                                                    \
            const float vh = __builtin_fmaf(x, __uint_as_float(RA.w), __uint_as_float(RA.z));                     \
            const float t = vh + LEAN_MAGIC;                                                                      \
            const float d = vh - (t - LEAN_MAGIC);                                                                \
            const bool act = !(TAIL) || (int)(idx8 >> 3) < wave_total;                                            \
            const bool ok = fabsf(d) < thr && c1 <= LCG_M - (1u << NEAR_ONE_BITS);                                \
            if (act && ok LEAN_STORE_COND) *reinterpret_cast<uint16_t*>(out_b + voff) = (uint16_t)((__float_as_uint(t) + RA.y) & 0xffffu); \
            else if (act) {                                        /* ~1 % of steps: park the undecided samples (no round trip) */ \
                const unsigned long long am = __builtin_amdgcn_ballot_w64(true);                                  \
                const int n0 = W.nfix;                                                                            \
                const int slot = n0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u)); \
                if (slot < FIX_SLOTS) P.tfix[(size_t)g * FIX_SLOTS + slot] = make_uint4(voff >> 1, c1, (uint32_t)(ev_read0 + EV), 0u); \
                else push_fix_one(P, it.sig_base + (voff >> 1), c1, it.ev_first + EV, it.read, 0);   /* overflow (never in practice): global list */ \
                if (slot + 1 == n0 + __popcll(am)) W.nfix = slot + 1;          /* the last of them publishes the new count */ \
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                            \
            }                                                                                                     \
            idx8 += 512u;                                                                                         \
            voff = RNA ? voff - 128u : voff + 128u;                                                               \
            MN = *reinterpret_cast<const uint32_t*>(mult_b + (((idx8 - (RN.y >> 16)) & 0xff8u) >> 1)); }

        /* ablation builds (tools/ab_variants.sh; results are wrong): -DSQG_ABL_NOARITH, -DSQG_ABL_NOSTORE, -DSQG_ABL_NOLOOP */
#if defined(SQG_ABL_NOSTORE)
        #define LEAN_STORE_COND && (__float_as_uint(t) == 0x12345u)
#else
        #define LEAN_STORE_COND
#endif
#if defined(SQG_ABL_NOARITH)
        #define LEAN_ARITH(RA, MU) const uint32_t c1 = (RA.x ^ MU) & 0x3fffffffu; const float x = __uint_as_float((RA.x + MU) & 0x3fffffffu);
#else
        #define LEAN_ARITH(RA, MU) const uint32_t c1 = lcg_mul(RA.x, MU); const float x = box_muller_fast(c1);
#endif
        uint4 ra, rb; uint32_t ma, mb; int eva, evb;
        base_ev = 0;
        LEAN_MAP(0, eva)
        ra = W.rec[eva];
        ma = *reinterpret_cast<const uint32_t*>(mult_b + (((idx8 - (ra.y >> 16)) & 0xff8u) >> 1));
        int c = 0;
        for (; c + 2 <= nfull; c += 2) {
            LEAN_STEP(false, c, ra, ma, eva, rb, mb, evb)
            LEAN_STEP(false, c + 1, rb, mb, evb, ra, ma, eva)
        }
        if (c < nfull) {
            LEAN_STEP(false, c, ra, ma, eva, rb, mb, evb)
            ra = rb; ma = mb; eva = evb; c++;
        }
        if (rem) LEAN_STEP(true, c, ra, ma, eva, rb, mb, evb)
        #undef LEAN_STEP
        #undef LEAN_MAP
        #undef LEAN_ARITH
        #undef LEAN_STORE_COND
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int nfix = W.nfix;
        if (nfix && lane == 0) P.tfix_n[g] = (unsigned char)min(nfix, FIX_SLOTS);
    }
}

// MODE 0: FP64 everywhere.  MODE 1: certified fp32 path.
// GENERIC false: the lean kernel; tiles it cannot take (long events, level-shift window, possibly
//                negative ADC values, no-noise modes) are queued for the GENERIC instantiation.
template <int MODE, bool GENERIC>
__global__ __launch_bounds__(256) void k_samples(const SigParams P, const int n_tiles) {
    __shared__ SmpLds L;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < MULT_N; i += 256) L.mult[i] = make_uint2(P.pw[i], P.pw[POW_N + i]);
    __syncthreads();
    SmpWaveLds& W = L.w[wid];
    const unsigned long long lane_le = (lane == 63) ? ~0ull : ((2ull << lane) - 1);   // lanes <= me
    const int n_work = GENERIC && P.slow_tiles ? (int)min(*P.slow_count, (unsigned int)n_tiles) : n_tiles;

    for (int wi = blockIdx.x * 4 + wid; wi < n_work; wi += gridDim.x * 4) {
        const int g = GENERIC && P.slow_tiles ? P.slow_tiles[wi] : wi;
        const int r = P.tile_read[g];
        const ReadDesc rd = P.reads[r];
        const int ne = rd.ne0 + rd.ne1;
        const int e = (g - rd.tile_off) * 64 + lane;
        const bool valid = e < ne;
        const long long ev_first = rd.ev_off + (long long)(g - rd.tile_off) * 64;
        uint2 er = make_uint2(0u, 0u);
        int sps = 0;
        if (valid) {
            er = P.evrec[rd.ev_off + e];
            sps = P.dwell ? (int)P.dwell[rd.ev_off + e] : P.const_sps;
        }
        const float2 md = valid ? P.model[er.y] : make_float2(0.f, 0.f);
        const int incl = wave_incl_scan(sps, lane);
        const int wave_total = __shfl(incl, 63);
        const int so = incl - sps;                                     // first sample of my event within the tile
        const long long sig_base = P.sig_off[r];
        const uint32_t read_len = (uint32_t)(P.sig_off[r + 1] - sig_base);
        const long long n1 = (long long)P.seglen[2 * r];               // samples of segment 0
        const long long shift_lo = n1 - P.shift_len;                    // src/genread.c:79
        const uint32_t base_pos = P.tile_so[g];
        const bool shift_tile = P.shift_len > 0 && (long long)base_pos + wave_total > shift_lo && (long long)base_pos < n1;
        const double offset = rd.offset;
        int16_t* out = P.sig + sig_base;

        float thr = 1.0f;
        bool fast_ok = false;
        uint4 ra; uint2 rb;
        if (!P.use_streams) {
            // no amplitude noise (--ideal / --ideal-amp): s = level_mean, one digitisation per event (src/gensig.c:266,270)
            const int16_t qc = to_i16((double)md.x * P.dig / P.range - offset);
            ra = make_uint4(0u, (uint32_t)so, 0u, 0u);
            rb = make_uint2((uint32_t)(uint16_t)qc, 0u);
        } else if (MODE == 1) {
            // v = s_f*dig/range - offset  ~  x*(sd*kd) + (m*kd - offset) = x*sdk + (I + F)
            const double mkd = (double)md.x * P.kd;
            const double mk = mkd - offset;
            const double fl = floor(mk);
            const float F = (float)(mk - fl);
            const float sdk = (float)((double)md.y * P.kd);
            const float asdk = fabsf(sdk);
            // error budget (DESIGN.md "Certified fast path"): swept |x'-x| * sdk; float narrowing of s
            // (2^-24 (|m| kd + 6.56 sdk)); roundings of sdk (x6.56), of F (2^-25) and of the fma
            // (2^-24 (6.56 sdk + 1)); FP64 roundings and the fp32 evaluation of eps itself in the slack
            const float eps = P.delta_x * asdk + 5.9604645e-8f * ((float)fabs(mkd) + 21.0f * asdk + 3.0f) + 2.0e-7f;
            thr = 0.5f - eps;
            if (!(fabs(fl) < 1.0e9)) thr = -1.0f;                     // absurd profile: everything goes to FP64
            fast_ok = !valid || (sps <= MULT_N && fl - 7.0 * (double)asdk > 2.0 && fl < 1.0e9);
            ra = make_uint4(er.x, (uint32_t)so, __float_as_uint(F), __float_as_uint(sdk));
            rb = make_uint2((uint32_t)(int)fl, __float_as_uint(thr));
        } else {
            ra = make_uint4(er.x, (uint32_t)so, __float_as_uint(md.x), __float_as_uint(md.y));
            rb = make_uint2(0u, 0u);
        }
        const bool take_fast = MODE == 1 && P.use_streams && !shift_tile && __all(fast_ok);
        if (!GENERIC) {
            if (!take_fast) {                                          // leave this tile to the generic kernel
                if (lane == 0) { const unsigned int q = atomicAdd(P.slow_count, 1u); P.slow_tiles[q] = g; }
                continue;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");         // previous tile's LDS reads are done
        W.rec_a[lane] = ra;
        W.rec_b[lane] = rb;
        if (wave_total <= 0) continue;

        if (!GENERIC) {
            // ---------------- the hot loop ----------------
            float t = valid ? thr : 1.0f;
            for (int o = 32; o > 0; o >>= 1) t = fminf(t, __shfl_xor(t, o));
            const bool rna = P.rna != 0;
            const uint32_t a_top = read_len - 1 - base_pos;
            for (int w0 = 0; w0 < wave_total; w0 += MK_W) {
                ((uint4*)W.mk)[lane] = make_uint4(0, 0, 0, 0);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (valid && so >= w0 && so < w0 + MK_W) W.mk[so - w0] = 1;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                int base_ev = __popcll(__ballot(valid && so < w0)) - 1;
                const int w_end = min(w0 + MK_W, wave_total);
                for (int c0 = w0; c0 < w_end; c0 += 64) {
                    const int idx = c0 + lane;
                    const unsigned long long sm = __ballot(W.mk[idx - w0] != 0);
                    const int ev = base_ev + __popcll(sm & lane_le);
                    base_ev += __popcll(sm);
                    const uint4 qa = W.rec_a[ev];
                    const int I = (int)W.rec_b[ev].x;
                    const uint32_t j = ((uint32_t)idx - qa.y) & (MULT_N - 1);
                    const uint2 mu = L.mult[j];
                    const uint32_t c1 = lcg_mul(qa.x, mu.x);
                    const float x = box_muller_fast(c1);
                    const float v = __builtin_fmaf(x, __uint_as_float(qa.w), __uint_as_float(qa.z));
                    const float fl = floorf(v);
                    const float fr = v - fl;
                    const bool act = idx < w_end;
                    const bool ok = fabsf(fr - 0.5f) < t && c1 <= LCG_M - (1u << NEAR_ONE_BITS);
                    const int n = I + (int)fl;
                    const uint32_t at = rna ? (a_top - (uint32_t)idx) : (base_pos + (uint32_t)idx);
                    if (act && ok) out[at] = (int16_t)(uint16_t)((uint32_t)n & 0xffffu);
                    push_fix(P, act && !ok, lane, lane_le, sig_base + at, c1, ev_first + ev, r, 0);
                }
            }
        } else {
            // ---------------- every option, both modes ----------------
            for (int w0 = 0; w0 < wave_total; w0 += MK_W) {
                ((uint4*)W.mk)[lane] = make_uint4(0, 0, 0, 0);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (valid && so >= w0 && so < w0 + MK_W) W.mk[so - w0] = 1;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                int base_ev = __popcll(__ballot(valid && so < w0)) - 1;
                const int w_end = min(w0 + MK_W, wave_total);
                for (int c0 = w0; c0 < w_end; c0 += 64) {
                    const int idx = c0 + lane;
                    const unsigned long long sm = __ballot(W.mk[idx - w0] != 0);
                    const int ev = base_ev + __popcll(sm & lane_le);
                    base_ev += __popcll(sm);
                    const bool act = idx < w_end;
                    const uint4 qa = W.rec_a[act ? ev : 0];
                    const uint2 qb = W.rec_b[act ? ev : 0];
                    const uint32_t j = (uint32_t)idx - qa.y;
                    const uint32_t pos = base_pos + (uint32_t)idx;
                    const uint32_t at = P.rna ? (read_len - 1 - pos) : pos;
                    const bool in_shift = shift_tile && (long long)pos >= shift_lo && (long long)pos < n1;
                    int16_t q = 0;
                    bool ok = true;
                    uint32_t c1 = 1;
                    if (!P.use_streams) {
                        q = (int16_t)(uint16_t)qb.x;
                    } else if (act) {
                        if (j < MULT_N) c1 = lcg_mul(qa.x, L.mult[j].x);
                        else c1 = lcg_mul(lcg_mul(qa.x, lcg_jump2(P.pw, j)), LCG_A);
                        if (MODE == 1) {
                            const float x = box_muller_fast(c1);
                            const float v = __builtin_fmaf(x, __uint_as_float(qa.w), __uint_as_float(qa.z));
                            const float fl = floorf(v);
                            const float fr = v - fl;
                            ok = fabsf(fr - 0.5f) < __uint_as_float(qb.y) && c1 <= LCG_M - (1u << NEAR_ONE_BITS);
                            int n = (int)qb.x + (int)fl;
                            n -= n >> 31;                                  // truncation toward zero (value is not an integer)
                            q = (int16_t)(uint16_t)((uint32_t)n & 0xffffu);
                        } else {
                            const double z = box_muller_exact(c1, lcg_mul(c1, LCG_A));
                            const float sv = (float)((z * (double)__uint_as_float(qa.w)) + (double)__uint_as_float(qa.z));   // src/gensig.c:268
                            q = to_i16((double)sv * P.dig / P.range - offset);                                             // src/gensig.c:270
                        }
                    }
                    if (in_shift) q = (int16_t)(uint16_t)(((int)q - P.shift) & 0xffff);
                    if (act && ok) out[at] = q;
                    if (MODE == 1) push_fix(P, act && !ok, lane, lane_le, sig_base + at, c1, ev_first + ev, r, in_shift ? 1 : 0);
                }
            }
        }
    }
}

// ---- k_fixup: FP64 path for the samples k_signal<CERTIFIED> left undecided -------------------
// per-tile slots of the lean kernel: one thread per super tile walks its (0-8, typically 0-1) parked samples.
// No atomics: a returning atomic per wavefront on one counter costs ~10 ns each and serialises.
__global__ __launch_bounds__(256) void k_fixup_tiles(const SigParams P, const int n_stiles) {
    __shared__ uint16_t work[4][64 * FIX_SLOTS];         // per wavefront: (lane of the item << 4) | slot
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int n = g < n_stiles ? (int)P.tfix_n[g] : 0;
    // spread the wavefront's parked samples (0-8 per item, ~0.5 on average) evenly over its lanes
    const int incl = wave_incl_scan_dpp(n);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) return;
    for (int q = 0; q < n; q++) work[wid][incl - n + q] = (uint16_t)((lane << 4) | q);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (int w = lane; w < total; w += 64) {
        const int code = work[wid][w];
        const int gi = g - lane + (code >> 4), slot = code & 15;
        const int r = P.stile_read[gi];
        const ReadDesc rd = P.reads[r];
        const uint4 fe = P.tfix[(size_t)gi * FIX_SLOTS + slot];
        const int e = (int)fe.z;
        const uint8_t* bp = P.bases + rd.base_off + (e < rd.ne0 ? (long long)e : (long long)rd.len0 + (e - rd.ne0));
        uint32_t rank = 0;
        for (int q = 0; q < P.k; q++) rank = (rank << 2) | base_code(bp[q]);
        const float2 md = P.model[rank];
        P.sig[P.sig_off[r] + fe.x] = sample_exact(fe.y, md.x, md.y, P.dig, P.range, rd.offset);
    }
}

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

// ---- read sampler on the device-resident genome ("next" row) ----------
// gen_read, src/genread.c:125-370: per worker the streams ref_pos (seed s), rand_strand (s+1) and rand_rlen
// (s+3; Erlang-2 with scale rlen/2, the INTEGER quotient) of src/sim.c:238-247, consumed in read order.
#define SQG_SAMPLE_RNA   1      // whole transcripts, '+' strand (src/genread.c:311-355)
#define SQG_SAMPLE_CDNA  2      // transcripts with a strand draw
#define SQG_SAMPLE_TRUNC 4      // --trans-trunc (src/genread.c:303-309)

struct GenomeParams {
    const uint8_t* seq;          // contigs back to back (no terminators)
    const long long* contig_off; // [n_contigs+1]
    const long long* cum;        // [n_contigs] inclusive prefix sums of the contig lengths (src/genread.c:181-191)
    const float* trans_csum;     // --trans-count: cumulative abundances (float, src/ref.c:206-273), or null
    const int* trans_idx;        // ... and the contig of each entry
    long long sum;               // ref->sum
    double grng_b;               // (double)(rlen / 2)
    int n_contigs, n_trans, rlen, flags;
};

struct SampleRec {               // what gen_read returns, per read
    long long src;               // offset of the read's first base in GenomeParams.seq (forward strand)
    int ref_idx, ref_pos, rlen;  // contig, 0-based start, bases copied
    int strand;                  // '+' or '-'
    int n_N;                     // 'N's substituted (src/genread.c:132-138)
    int ref_len;                 // *ref_len of gen_read: the contig's length (DNA) / the transcript part used (RNA)
};

__global__ void k_init_sampler(uint32_t* __restrict__ st, long long seed, int worker_lo, int nw, int num_kmer) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    const long long s = seed + (long long)(w + worker_lo) * ((long long)num_kmer + 10);
    const long long add[3] = {0, 1, 3};                              // ref_pos, rand_strand, rand_rlen
    for (int j = 0; j < 3; j++) {
        long long v = (s + add[j]) % (long long)LCG_M;
        if (v < 0) v += LCG_M;
        st[3 * w + j] = (uint32_t)v;
    }
}

// rng() of src/rand.h:79-85 on a canonical state
__device__ static inline double samp_rng(uint32_t& c) { c = lcg_mul(c, LCG_A); return lcg_uniform(c); }

// exact count of bytes equal to 'N' in p[0..n), by the 64 lanes of a wavefront together (8 bytes per lane per step)
__device__ static inline int count_N(const uint8_t* __restrict__ p, int n, int lane) {
    int cnt = 0;
    const int n8 = n & ~7;
    for (int i = lane * 8; i < n8; i += 512) {
        unsigned long long v;
        __builtin_memcpy(&v, p + i, 8);
        const unsigned long long x = v ^ 0x4e4e4e4e4e4e4e4eull;        // zero byte <=> 'N'
        const unsigned long long t = ~(((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x | 0x7f7f7f7f7f7f7f7full);
        cnt += __popcll(t);
    }
    if (lane < n - n8) cnt += p[n8 + lane] == 'N';
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    return cnt;
}

// one wavefront per worker chain (that worker's reads of the batch, in order): every lane makes the same draws, the
// lanes share the scan of the candidate for 'N's
__global__ __launch_bounds__(64) void k_sample(const GenomeParams G, uint32_t* __restrict__ st, const int* __restrict__ chain_off,
                                               const int* __restrict__ chain_reads, const int* __restrict__ chain_worker,
                                               int n_chains, SampleRec* __restrict__ out, unsigned int* __restrict__ err) {
    const int ch = blockIdx.x, lane = threadIdx.x;
    if (ch >= n_chains) return;
    const int w = chain_worker[ch];
    uint32_t c_pos = st[3 * w], c_strand = st[3 * w + 1], c_len = st[3 * w + 2];
    for (int ci = chain_off[ch]; ci < chain_off[ch + 1]; ci++) {
        SampleRec rec;
        for (int attempt = 0;; attempt++) {
            if (attempt > 100000) { atomicOr(err, 16u); rec.src = 0; rec.ref_idx = 0; rec.ref_pos = 0; rec.rlen = 0; rec.strand = '+'; rec.n_N = 0; rec.ref_len = 0; break; }
            int idx, pos, len, strand = '+';
            if (G.flags & (SQG_SAMPLE_RNA | SQG_SAMPLE_CDNA)) {
                // src/genread.c:283-300: uniform over transcripts, or by the abundance CDF (uniform narrowed to float)
                if (G.n_trans == 0) idx = (int)round(samp_rng(c_pos) * (G.n_contigs - 1));
                else {
                    const float r = (float)samp_rng(c_pos);
                    idx = 0;
                    for (int i = 0; i < G.n_trans; i++) if (r <= G.trans_csum[i]) { idx = G.trans_idx[i]; break; }
                }
                const int clen = (int)(G.contig_off[idx + 1] - G.contig_off[idx]);
                len = clen; pos = 0;
                if (G.flags & SQG_SAMPLE_TRUNC) {                     // src/genread.c:303-309
                    double acc = 0.0;
                    acc += -log(1 - samp_rng(c_len));
                    acc += -log(1 - samp_rng(c_len));
                    const double frac = (acc * G.grng_b) / (double)G.rlen;
                    int tl = (int)(frac * clen);
                    tl = tl > clen ? clen : tl;
                    pos = clen - tl; len = tl;
                }
                if (G.flags & SQG_SAMPLE_CDNA) strand = ((long long)round(samp_rng(c_strand))) ? '+' : '-';
            } else {
                // src/genread.c:243-281
                double acc = 0.0;                                     // grng, src/rand.h:96-102 (Erlang-2)
                acc += -log(1 - samp_rng(c_len));
                acc += -log(1 - samp_rng(c_len));
                len = (int)(acc * G.grng_b);
                const long long at = (long long)round(samp_rng(c_pos) * (double)G.sum);   // src/genread.c:181
                idx = 0;
                while (idx < G.n_contigs - 1 && G.cum[idx] < at) idx++;
                pos = (int)(at - G.cum[idx]) + (int)(G.contig_off[idx + 1] - G.contig_off[idx]);
                strand = ((long long)round(samp_rng(c_strand))) ? '+' : '-';            // src/genread.c:196-200
            }
            if (len < 0) len = 0;
            const int clen = (int)(G.contig_off[idx + 1] - G.contig_off[idx]);
            const int n = min(len, clen - pos);                       // src/genread.c:149-177: clipped at the contig's end
            if (n < 200) continue;                                    // src/genread.c:126
            const long long src = G.contig_off[idx] + pos;
            const int nN = count_N(G.seq + src, n, lane);
            if ((double)nN > 0.1 * (double)n) continue;               // src/genread.c:139-142
            rec.src = src; rec.ref_idx = idx; rec.ref_pos = pos; rec.rlen = n; rec.strand = strand; rec.n_N = nN;
            rec.ref_len = (G.flags & (SQG_SAMPLE_RNA | SQG_SAMPLE_CDNA)) ? len : clen;
            break;
        }
        if (lane == 0) out[chain_reads[ci]] = rec;
    }
    if (lane == 0) { st[3 * w] = c_pos; st[3 * w + 1] = c_strand; st[3 * w + 2] = c_len; }
}

__device__ static const char kd_stall_dna[] = "TTTTTTTTTTTTTTTTTTAATCAA";                       // src/genread.c:110
__device__ static const char kd_adaptor_dna[] = "GGCGTCTGCTTGGGTGTTTAACCTTTTTTTTTTAATGTACTTCGTTCAGTTACGTATTGCT";  // src/genread.c:38
__device__ static const char kd_adaptor_rna[] = "TGATGATGAGGGATAGACGATGGTTGTTTCTGTTGGTGCTGATATTGCTTTTTTTTTTTTTATGATGCAAGATACGCAC";  // src/genread.c:39
__device__ static const char kd_stall_rna[] = "AAAAAGAAAAAACCCCCCCCCCCCCCCCCC";                  // src/genread.c:87

// one workgroup per read: the sampled slice of the genome -> the batch's base buffer, exactly as gen_read returns
// it ('N' -> a base from a FRESH state-100 stream per read, src/genread.c:132-138; '-' -> revcomp, src/seq.h:78-112)
// with the prefix / stall attached as src/genread.c:95-123 does.  read_at: where the read starts in segment 0.
__global__ __launch_bounds__(256) void k_copy_reads(const GenomeParams G, const SampleRec* __restrict__ recs, const ReadDesc* __restrict__ reads,
                                                    uint8_t* __restrict__ bases, int n_reads, int rna, int prefix) {
    __shared__ int wcnt[4];
    __shared__ int carry;
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const SampleRec rec = recs[r];
    const ReadDesc rd = reads[r];
    uint8_t* dst = bases + rd.base_off;
    const int n = rec.rlen;
    int read_at = 0;
    if (prefix) {
        if (rna) {                                                    // read + polyA(158) + adaptor
            for (int i = tid; i < 158; i += 256) dst[n + i] = 'A';
            for (int i = tid; i < (int)sizeof(kd_adaptor_rna) - 1; i += 256) dst[n + 158 + i] = (uint8_t)kd_adaptor_rna[i];
        } else {                                                      // stall + adaptor + read
            const int st = (int)sizeof(kd_stall_dna) - 1, ad = (int)sizeof(kd_adaptor_dna) - 1;
            for (int i = tid; i < st; i += 256) dst[i] = (uint8_t)kd_stall_dna[i];
            for (int i = tid; i < ad; i += 256) dst[st + i] = (uint8_t)kd_adaptor_dna[i];
            read_at = st + ad;
        }
    }
    for (int i = tid; i < rd.len1; i += 256) dst[rd.len0 + i] = (uint8_t)kd_stall_rna[i];
    const uint8_t* src = G.seq + rec.src;
    const bool rev = rec.strand == '-';
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        uint8_t c = i < n ? src[i] : (uint8_t)'A';
        if (rec.n_N) {                                                // ordinal of every 'N' in forward order
            const bool isN = i < n && c == 'N';
            const unsigned long long m = __ballot(isN);
            if (lane == 0) wcnt[wid] = __popcll(m);
            __syncthreads();
            int before = carry;
            for (int w2 = 0; w2 < wid; w2++) before += wcnt[w2];
            const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            if (isN) {
                const int j = before + __popcll(m & ((1ull << lane) - 1)); // 0-based; draw j+1 of the state-100 stream
                uint32_t cst = 100u;
                for (int q = 0; q <= j; q++) cst = lcg_mul(cst, LCG_A);    // <= 10 % of the read: a short walk
                const int v = (int)round(lcg_uniform(cst) * 3);
                c = v == 0 ? 'A' : v == 1 ? 'C' : v == 2 ? 'G' : 'T';
            }
            __syncthreads();
            if (tid == 0) carry += tot;
            __syncthreads();
        }
        if (i < n) {
            if (rev) {
                uint8_t o;
                switch (c) {
                case 'A': case 'a': o = 'T'; break;
                case 'C': case 'c': o = 'G'; break;
                case 'G': case 'g': o = 'C'; break;
                case 'T': case 't': o = 'A'; break;
                default: o = 'T'; break;
                }
                dst[read_at + n - 1 - i] = o;
            } else dst[read_at + i] = c;
        }
    }
}

// ---- svb-zd: slow5lib's signal compression, per read ("next" row) -----
// slow5lib/src/slow5_press.c:1055-1087: int16 -> zig-zag of the delta to the previous sample (first: to 0) ->
// StreamVByte: uint32 count | ceil(count/4) key bytes (2 bits per value = bytes-1, first value in the low
// bits) | the values' 1-4 little-endian bytes.  One quad of samples (= one key byte) per thread.
// One quad: samples 4q..4q+3 arrive in one 8-byte load; the predecessor of the quad's first sample is the last
// sample of the lane to the left (DPP), the wavefront's first lane fetches it.  `full`: the quad has 4 samples.
__device__ static inline void svb_quad(const int16_t* __restrict__ sig, long long n, long long q, int lane,
                                       uint32_t z[4], uint32_t& key, uint32_t& nbytes) {
    int32_t v[5];
    const long long i0 = 4 * q;
    if (i0 + 4 <= n) {
        unsigned long long w;
        __builtin_memcpy(&w, sig + i0, 8);                                       // 2-byte aligned 8-byte load
        v[1] = (int16_t)(w & 0xffff); v[2] = (int16_t)((w >> 16) & 0xffff); v[3] = (int16_t)((w >> 32) & 0xffff); v[4] = (int16_t)(w >> 48);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) v[j + 1] = (i0 + j < n) ? (int32_t)sig[i0 + j] : 0;
    }
    // all lanes of the wavefront call this together with consecutive q (inactive quads carry zeros)
    v[0] = __builtin_amdgcn_update_dpp(0, v[4], 0x138, 0xf, 0xf, false);        // wave_shr:1 -> lane-1's last sample
    if (lane == 0) v[0] = (i0 > 0 && i0 - 1 < n) ? (int32_t)sig[i0 - 1] : 0;
    key = 0; nbytes = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int32_t d = v[j + 1] - v[j];
        z[j] = ((uint32_t)d + (uint32_t)d) ^ (uint32_t)(d >> 31);               // streamvbyte_zigzag.c
        const uint32_t code = z[j] < (1u << 8) ? 0u : z[j] < (1u << 16) ? 1u : z[j] < (1u << 24) ? 2u : 3u;
        if (i0 + j < n) { key |= code << (2 * j); nbytes += code + 1; }
    }
}

// bytes each read's encoding takes: 4 + ceil(n/4) + data bytes
__global__ __launch_bounds__(256) void k_svb_size(const int16_t* __restrict__ sig, const long long* __restrict__ sig_off,
                                                  int n_reads, long long* __restrict__ size) {
    __shared__ unsigned long long wsum[4];
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const long long n = sig_off[r + 1] - sig_off[r], nq = (n + 3) / 4;
    const int16_t* s = sig + sig_off[r];
    const int lane = threadIdx.x & 63;
    unsigned long long sum = 0;
    for (long long q0 = 0; q0 < nq; q0 += 256) {                                // whole wavefronts stay together (DPP)
        const long long q = q0 + threadIdx.x;
        uint32_t z[4], key, nb;
        svb_quad(s, n, q, lane, z, key, nb);
        sum += nb;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    if (lane == 0) wsum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) size[r] = 4 + nq + (long long)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

// exclusive scan of the per-read sizes (single workgroup), also through the pinned host mapping
__global__ __launch_bounds__(1024) void k_svb_scan(const long long* __restrict__ size, int n, long long* __restrict__ off, long long* __restrict__ host_off) {
    __shared__ long long wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int per = (n + 1023) / 1024;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    long long v = 0;
    for (int i = lo; i < hi; i++) v += size[i];
    long long x = v;
    for (int o = 1; o < 64; o <<= 1) { long long y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    long long run = x - v;
    for (int w = 0; w < wid; w++) run += wsum[w];
    for (int i = lo; i < hi; i++) { off[i] = run; host_off[i] = run; run += size[i]; }
    if (tid == 1023) { off[n] = run; host_off[n] = run; }
}

__global__ __launch_bounds__(256) void k_svb_encode(const int16_t* __restrict__ sig, const long long* __restrict__ sig_off, int n_reads,
                                                    const long long* __restrict__ svb_off, uint8_t* __restrict__ out) {
    __shared__ uint32_t wsum[2][4];
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long n = sig_off[r + 1] - sig_off[r], nq = (n + 3) / 4;
    const int16_t* s = sig + sig_off[r];
    uint8_t* o = out + svb_off[r];
    if (tid < 4) o[tid] = (uint8_t)((uint32_t)n >> (8 * tid));          // slow5_press.c:1047: the count word
    uint8_t* keys = o + 4;
    uint8_t* data = keys + nq;
    long long base = 0;                                                  // data bytes of the chunks before this one
    int buf = 0;
    for (long long q0 = 0; q0 < nq; q0 += 256, buf ^= 1) {
        const long long q = q0 + tid;
        uint32_t z[4], key, nb;
        svb_quad(s, n, q, lane, z, key, nb);                             // quads past the end carry zeros
        const int incl = wave_incl_scan_dpp((int)nb);
        if (lane == 63) wsum[buf][wid] = (uint32_t)incl;
        __syncthreads();                                                 // one barrier per chunk: the sums are double-buffered
        uint32_t woff = 0, tot = 0;
        for (int w = 0; w < 4; w++) { const uint32_t x = wsum[buf][w]; if (w < wid) woff += x; tot += x; }
        if (q < nq) {
            keys[q] = (uint8_t)key;
            uint8_t* d = data + base + woff + (uint32_t)incl - nb;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (4 * q + j < n) {
                    const uint32_t code = (key >> (2 * j)) & 3u;
                    if (code == 0) d[0] = (uint8_t)z[j];
                    else {                                                // little endian, one or two stores
                        const uint16_t lo = (uint16_t)z[j];
                        __builtin_memcpy(d, &lo, 2);
                        if (code >= 2) d[2] = (uint8_t)(z[j] >> 16);
                        if (code >= 3) d[3] = (uint8_t)(z[j] >> 24);
                    }
                    d += code + 1;
                }
            }
        }
        base += tot;
    }
}

// ---- k_certify: max |x_fast - x_exact| over every state the fp32 path may accept ------------
// The deviate is a function of c1 alone (c2 = a*c1 mod M), so the sweep is exhaustive.
__global__ __launch_bounds__(256) void k_certify(unsigned int* __restrict__ max_bits) {
    float m = 0.f;
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    for (unsigned long long c = 1 + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
         c <= LCG_M - (1u << NEAR_ONE_BITS); c += stride) {
        const uint32_t c1 = (uint32_t)c, c2 = lcg_mul(c1, LCG_A);
        const double xe = box_muller_exact(c1, c2);
        const float e0 = fabsf((float)((double)box_muller_fast(c1) - xe));
        m = fmaxf(m, e0);
        if (!(e0 == e0)) m = __builtin_inff();
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
