// k_common.h -- constants, the Lehmer generator in canonical form, Box-Muller (FP64 and certified fp32), descriptors
// Part of the device code of the per-read signal path; included through sqg_kernels.h (see there for the overview).
#pragma once

#define LCG_M 2147483647u
#define LCG_A 16807u

#define POW_N 1024          // entries per jump table
// d_pow layout (uint32 each):
//   [0*POW_N + j] = a^(2j+1)   first draw of sample/event j after a base state
//   [1*POW_N + j] = a^(2j+2)   second draw
//   [2*POW_N + j] = a^(2j)     jump over j draws-pairs
//   [3*POW_N + j] = a^(2*1024*j)
//   [4*POW_N + j] = a^(2*1024*1024*j)   j < POW_TOP
#define POW_TABLES 5
#define POW_TOP 4096        // entries of the last table: any 32-bit count
#define POW_WORDS (4 * POW_N + POW_TOP)
#define LCG_ORD2 1073741823u   // (M-1)/2: a^(2n) depends on n mod this

#ifndef SQG_U2
#define SQG_U2 1            // second uniform of the fp32 deviate: 0 = FP64 fract, 1 = int -> float conversion, 2 = mantissa bits (A/B)
#endif

#ifndef SQG_NEARONE
#define SQG_NEARONE 0       // A/B: the near-one states rejected through a NaN instead of a comparison (k_samples_lean)
#endif
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

// the same product with the multiplier handed over DOUBLED (b2 = 2*b < 2^32): the halves of the 64-bit product then are
// p >> 31 and 2*(p & M) themselves -- no funnel shift, no mask -- and their sum r = (p >> 31) + (p & M) < 2M is reduced by one
// conditional subtraction, min(r, r - M) in unsigned arithmetic.  5 VALU instead of 7; the sample loop's form.
__device__ static inline uint32_t lcg_mul_dbl(uint32_t a, uint32_t b2) {
    const unsigned long long p2 = (unsigned long long)a * b2;
    const uint32_t r = (uint32_t)(p2 >> 32) + ((uint32_t)p2 >> 1);
    return min(r, r - LCG_M);
}

// a^(2n) for any 32-bit n from three table levels
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
#if SQG_NEARONE
    // c1 * 2^-31 -- and a NaN for the states next to M (c1 > M - 2^17, where -2 ln u cancels: they always take the FP64 path): with
    // 2^17 added the integer turns negative there, v_log_f32 of a negative number is NaN, and a NaN fails every acceptance test
    // (|d| < thr is false) without a comparison of its own
    const float uf = __builtin_fmaf((float)(int)(c1 + (1u << NEAR_ONE_BITS)), 4.656612873077393e-10f, -6.103515625e-05f);
#else
    const float uf = (float)c1 * 4.656612873077393e-10f;                  // c1 * 2^-31 (exact scaling)
#endif
    const float lg = __builtin_amdgcn_logf(uf);                           // (log2 of the UNscaled c1 would save this multiplication, but
                                                                          // v_log_f32's error grows with |log2|: near 2^30 the swept bound
                                                                          // becomes 10x larger and 2e-3 of the samples fall back: measured)
    const float y = __builtin_fmaf(lg, -1.3862943611198906f, -9.313225750491594e-10f);   // -2 ln(c1/M)
    const float r = __builtin_amdgcn_sqrtf(y);
    // second uniform c2/M = frac(a*c1/M): four full-rate FP64/convert instructions instead of a modular
    // multiplication plus an int->float conversion (the product is exact to 2^-39, far below fp32 resolution)
#if SQG_U2 == 0
    const double t2 = (double)c1 * (16807.0 / 2147483647.0);                // a / M
    const float cs = __builtin_amdgcn_cosf((float)__builtin_amdgcn_fract(t2));
#else
    // ... without FP64: with p = a*c1 (46 bits), frac(p / M) = frac(p 2^-31 + p 2^-62) to 2^-60.  The first term's fraction is
    // the low 31 bits of p: v_mul_lo_u32 keeps 32 of them, and v_cos_f32 is periodic -- an integer more or less does not matter.
    // The second term, < 2^-16, is uf * a 2^-31.  fp32 resolves the sum to 2^-24 turns; k_certify prices that with the rest.
#if SQG_U2 == 1
    const uint32_t plo = c1 * 16807u;
    const float t2 = __builtin_fmaf((float)(int)plo, 4.656612873077393e-10f, uf * 7.826369259425611e-06f);   // a 2^-31
#else
    const uint32_t plo = c1 * 33614u;                                        // low 32 bits of 2p: bits 30..0 of p on top
    // the top 23 bits of the 32 as the mantissa of a float in [1, 2) (truncated: up to 2^-23 turns too low; part of what k_certify measures)
    const float t2 = __builtin_fmaf(uf, 7.826369259425611e-06f, __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, plo, 9))) ;
#endif
    const float cs = __builtin_amdgcn_cosf(t2);
#endif
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

// base -> digit of the 5-letter methylation alphabet A C G M T, src/seq.h:45-60 (upper case only; anything else -> 0)
__host__ __device__ static inline uint32_t meth_code(uint8_t b) {
    switch (b) {
    case 'C': return 1;
    case 'G': return 2;
    case 'M': return 3;
    case 'T': return 4;
    default: return 0;
    }
}

// k-mer rank of k bases: 2 bits per base (src/seq.h:31-42), or base-5 digits for the methylation tables (src/seq.h:62-74)
__device__ static inline uint32_t kmer_rank_of(const uint8_t* bp, int k, int meth) {
    uint32_t rank = 0;
    if (meth) { for (int q = 0; q < k; q++) rank = rank * 5u + meth_code(bp[q]); }
    else { for (int q = 0; q < k; q++) rank = (rank << 2) | base_code(bp[q]); }
    return rank;
}

// the same from ONE pair of loads (9 bases at most; the base buffer ends with slack): the fix-up kernels walk a chain of dependent
// look-ups per sample, and nine byte loads in a loop of unknown length are nine round trips
__device__ static inline uint32_t kmer_rank_wide(const uint8_t* bp, int k, int meth) {
    unsigned long long w0; uint32_t w1;
    __builtin_memcpy(&w0, bp, 8);
    __builtin_memcpy(&w1, bp + 8, 4);
    uint32_t rank = 0;
#pragma unroll
    for (int q = 0; q < 9; q++) {
        const uint8_t b = q < 8 ? (uint8_t)(w0 >> (8 * q)) : (uint8_t)w1;
        if (q < k) rank = meth ? rank * 5u + meth_code(b) : (rank << 2) | base_code(b);
    }
    return rank;
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
    int slot0;            // one-partition hand-out (k <= 6, few workers; k_part.h): slot in part[] of the read's first event;
                          // several partitions with SigParams.evrec32: the read's link, -1 if it is cut into pieces of several links
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
    int shift_lo, shift_hi;      // RNA adaptor level-shift window (src/genread.c:79-86) as sample indices within the item
                                 // (generation order): samples lo <= i < hi get -shift; hi <= lo: none
    int slot_first;              // one-partition hand-out: slot in part[] of the item's first event; bucketed hand-out with 16-bit event
                                 // records (SigParams.evrec32): the item's LINK; else 0
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
    int seglen_zero;                   // k_fixup: words of seglen_out to zero behind the batch (0: none)
    double dmean, dstd;          // dwell_mean, dwell_std
    const unsigned long long* seglen;  // [2*n_reads] samples in segment 0 / 1
    const long long* sig_off;    // [n_reads+1]
    const float2* model;         // {level_mean, (float)(level_stdv*amp_noise)}
    const uint32_t* pw;
    uint32_t* rows;              // [n_local_workers][num_kmer]: k <= 6 the stream states; k > 6 the samples each stream has produced
    uint32_t* link_rows;         // split chains (few workers, many reads): [n_chains][num_kmer], the same per LINK of a worker's chain,
                                 // prepared by k_link_hist / k_link_prefix; k_events then takes its row from here
    uint32_t seed_base, seed_step;   // (seed + worker_lo*(4^k+10)) mod M and (4^k+10) mod M: the initial state of local worker w,
                                     // k-mer j is (seed_base + w*seed_step + j) mod M (src/sim.c:238-256)
    int16_t* sig;
    unsigned int* err;
    FixEntry* fix;               // certified mode: undecided samples
    unsigned int* fix_count;
    unsigned int fix_cap;
    FixEntry* fix_sh;            // ... of the lean kernel: FIX_SHARDS lists of fix_sh_cap entries (a workgroup appends to list blockIdx % FIX_SHARDS:
    unsigned int* fix_sh_count;  // one returning atomic per item with undecided samples, on one of FIX_SHARDS counters 128 B apart)
    unsigned int fix_sh_cap;     // entries per list (sized by the batch: >= FIX_SHARD_CAP_MIN, four times the fix-ups a batch of its size expects)
    unsigned int* fix_sh_stat;   // [FIX_SHARDS] entries k_fixup took from each list
    unsigned int* host_res;      // device-visible pinned host memory of the batch: k_fixup leaves [0] the batch's error word, [1] the global list's
                                 // count, [4 + s] the entries it took from list s -- sqg_batch_wait reads them without a copy of its own
    int fix_tag;                 // FixEntry.pad of this batch's entries (a list's count may include entries that went to the global list instead)
    uint2* evrec;                // per event {stream state at its first draw, k-mer rank}
    uint32_t* evrec32;           // bucketed hand-out, wavefront-per-link passes (k_part_events.h): INSTEAD of evrec, 4 B per event between the
                                 // scatter pass and the sample kernels: rank << EVR_REL_BITS | the event's slot relative to the first slot
                                 // of its (link, partition) -- the absolute slot is lbase[link][rank >> 12] + that
    uint32_t* lbase;             // [n_links][PART_MAX] first slot in part[] of every (link, partition) (written by the scatter pass)
    int* tile_link;              // [n_tiles] the link every 64-event tile belongs to (written by the scatter pass)
    uint32_t* tile_so;           // per 64-event tile: its first sample within the read
    const int* tile_read;        // per tile: read index
    const int* stile_read;       // per 256-event super tile (lean kernel work item): read index
    ItemDesc* items;             // per super tile: what k_samples_lean needs, in one 48-B record
    int lean_epl;                // events per lane of the lean kernel (4, 2 or 1): a super tile is 64*lean_epl events
    int* slow_tiles;             // tiles the lean sample kernel left to the generic one
    unsigned int* slow_count;
    double dig, range, kd;       // kd = dig/range
    float delta_x;               // swept bound on |x_fast - x_exact| (incl. margin)
    float thr_all;               // 1/2 - (largest eps over all k-mers): acceptance threshold of the lean kernel
    int k, num_kmer;
    int meth;                    // 5-letter (A C G M T) table of 5^k rows: ranks are base-5 numbers (src/seq.h:45-74, src/gensig.c:250-253)
    uint32_t meth_top;           // 5^(k-1)
    int num_kmer_pad;            // num_kmer rounded up to whole partitions (k_part.h)
    int const_sps;               // (int)dwell_mean, used when dwell == null
    int dwell_pack;              // no dwell exceeds 1023: 64 of them sum to less than 2^16 (k_part_events: two tile sums per scan)
    int dwell_unbounded;         // the hard bound of a dwell draw (|z| <= 6.5556) exceeds 65535: k_events checks every draw
    int use_streams;             // 0 in --ideal / --ideal-amp (src/gensig.c:265-269)
    int rna;                     // reverse the signal (src/gensig.c:348-354)
    int shift_len;               // RNA+prefix: 79*(int)dwell_mean samples get -shift (src/genread.c:79-86)
    int shift;                   // (int16)(30*dig/range)
    // k > 6, few workers (k_part.h): a worker chain's events bucketed by the top bits of the k-mer rank
    uint32_t* part;              // [n_events], a chain's region partition-major, chain order inside a partition:
                                 // (dwell << 16) | low PART_SUB_BITS of the rank
    uint32_t* part_state;        // [n_events] same slots (k_part_hand): the stream's state at the event's first draw; non-null tells
                                 // the sample kernels that evrec.x is a slot, not a state
    uint32_t* pcnt;              // [n_part][n_links] events per (link, partition)
    const uint32_t* poff;        // [n_part][n_links] first slot in part[] of the link's events of the partition, relative to pstart (k_part_offsets)
    const uint32_t* pstart;      // [n_wchains][n_part] first slot of the (worker chain, partition) (k_part_slices): poff is relative to it
    const int* link_q;           // [n_links] the worker chain a link belongs to
    int one;                     // one-partition hand-out (k <= 6): part[slot] = rank | dwell << 16 and part_state[slot], slots in event order from
                                 // ReadDesc.slot0 on, are what the sample kernels read -- no evrec
    const int4* pieces;          // k_part_events: the links are runs of pieces {read, first event, end event, -}; chain_off indexes them
    uint32_t* piece_total;       // [n_pieces] samples of each piece (first event pass; k_part_tile_bases)
    int n_part;                  // partitions = num_kmer >> PART_SUB_BITS
    int n_links;                 // pcnt / poff are partition-major: [partition][link] (k_part_offsets sweeps a partition's links)
};

#define EVR_REL_BITS 14          // SigParams.evrec32: an event's slot within its (link, partition) -- staging keeps a link below 2^14 events --
                                 // under the 18 bits of a 9-mer rank
#define FIX_SHARDS 1024          // (a single counter: 90 000 returning device-scope atomics per batch on one address run at ~40 ns each and
#define FIX_SHARD_CAP_MIN 2048   // stretch the sample kernel from 2.4 to 5.8 ms: measured; a list that is full sends its items to that one list)
#define FIX_SHARD_STRIDE 32      // words between two counters
#define PART_SUB_BITS 12         // a partition's sub-row: 4096 streams, 16 KiB of LDS (the size of a whole 6-mer row).  (2048-stream
                                 // partitions were measured: k_part_hand runs 16 wavefronts per CU instead of 8, but the scatter's runs
                                 // shrink to 16 B and the sample kernels' state gather spreads: no gain on the whole step)
#define PART_SUB (1 << PART_SUB_BITS)
#define PART_MAX 64              // partitions of a 9-mer table: one per lane of a wavefront (k_events<PART> relies on it)

