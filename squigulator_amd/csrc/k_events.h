// k_events.h -- k_init_rows, k_dwell, k_scan and k_events: dwell draws, k-mer ranks, in-order hand-out of the k-mer streams
// Part of the device code of the per-read signal path; included through sqg_kernels.h (see there for the overview).
#pragma once

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

// ---- k_fill_tiles: tile -> read and super tile -> read maps, one workgroup per read -----------
__global__ __launch_bounds__(64) void k_fill_tiles(const ReadDesc* __restrict__ reads, int n_reads, int lean_ev,
                                                   int* __restrict__ tile_read, int* __restrict__ stile_read) {
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const ReadDesc rd = reads[r];
    const int ne = rd.ne0 + rd.ne1;
    const int nt = (ne + 63) >> 6, nst = (ne + lean_ev - 1) / lean_ev;
    for (int t = threadIdx.x; t < nt; t += 64) tile_read[rd.tile_off + t] = r;
    for (int t = threadIdx.x; t < nst; t += 64) stile_read[rd.stile_off + t] = r;
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

// ---- k_scan: sig_off = exclusive scan of per-read totals ------------------------------------
// One workgroup per 1024 reads, one read per thread (coalesced 16-B loads).  A workgroup publishes the total of its
// reads as {ticket, total} and adds up the totals of the workgroups before it, waiting for their tickets (they were
// dispatched earlier and wait for nobody behind them); `ticket` differs from launch to launch, so the array is never
// cleared.  sig_off goes to HBM for the kernels and, when host_off is given, through the pinned host mapping straight
// to the host (no D2H copy between kernels): visible once the stream has been synchronised.
#define SCAN_WG 1024
struct ScanArgs {                                        // (k_scan's arguments: the scan also runs as extra workgroups of k_part_mid)
    const unsigned long long* seglen; int n_reads;
    long long* sig_off; long long* host_off;
    unsigned int* err; unsigned int* counters;
    unsigned long long* part; unsigned long long ticket;
    unsigned int* shard_counters;
};
// workgroup g of ng (SCAN_WG threads; the workgroups before g were dispatched earlier)
__device__ static inline void scan_body(const ScanArgs& A, const int g, const int ng) {
    const unsigned long long* __restrict__ seglen = A.seglen; const int n_reads = A.n_reads;
    long long* __restrict__ sig_off = A.sig_off; long long* __restrict__ host_off = A.host_off;
    unsigned int* __restrict__ err = A.err; unsigned int* __restrict__ counters = A.counters;
    unsigned long long* __restrict__ part = A.part; const unsigned long long ticket = A.ticket;
    unsigned int* __restrict__ shard_counters = A.shard_counters;
    __shared__ long long wsum[SCAN_WG / 64];
    __shared__ long long before_sh;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (g == 0 && tid < 4) counters[tid] = 0;            // fix-up list / slow-tile list counters of this batch
    if (shard_counters)                                  // ... and the FIX_SHARDS list counters of the lean kernel
        for (int i = g * SCAN_WG + tid; i < FIX_SHARDS; i += ng * SCAN_WG) shard_counters[i * FIX_SHARD_STRIDE] = 0;
    const int i = g * SCAN_WG + tid;
    long long v = 0;
    if (i < n_reads) {
        const ulonglong2 q = reinterpret_cast<const ulonglong2*>(seglen)[i];
        v = (long long)(q.x + q.y);
        if (v >= 4294967295LL) atomicOr(err, 2u);        // src/sim.c:559-562
    }
    long long x = v;
    for (int o = 1; o < 64; o <<= 1) { long long y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    long long run = x - v, total = 0;
    for (int w = 0; w < SCAN_WG / 64; w++) { if (w < wid) run += wsum[w]; total += wsum[w]; }
    if (tid == 0) {
        __hip_atomic_store(&part[2 * g + 1], (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&part[2 * g], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    // totals of the workgroups before mine
    long long mine = 0;
    for (int h = tid; h < g; h += SCAN_WG) {
        while (__hip_atomic_load(&part[2 * h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != ticket) __builtin_amdgcn_s_sleep(1);
        mine += (long long)__hip_atomic_load(&part[2 * h + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    __syncthreads();                                     // wsum has been read by everyone
    if (lane == 0) wsum[wid] = mine;
    __syncthreads();
    if (tid == 0) { long long t = 0; for (int w = 0; w < SCAN_WG / 64; w++) t += wsum[w]; before_sh = t; }
    __syncthreads();
    run += before_sh;
    if (i < n_reads) { sig_off[i] = run; if (host_off) host_off[i] = run; }
    if (i == n_reads - 1) { sig_off[n_reads] = run + v; if (host_off) host_off[n_reads] = run + v; }
}
__global__ __launch_bounds__(SCAN_WG) void k_scan(const ScanArgs A) { scan_body(A, (int)blockIdx.x, (int)gridDim.x); }

// ---- split chains --------------------------------------------------------------------------
// With few workers and many reads (the reference's `-t 1`, or `-t 8 -K 1000`) a worker's chain of reads is cut into
// LINKS of whole reads that k_events walks concurrently.  A link needs the worker's k-mer streams as they stand at its
// first read: the state (k <= 6) or sample count (k > 6) of the worker's row, advanced by the samples the chain's
// earlier links draw from each stream.
//   k_events<HIST> per link: dwell draws, samples per k-mer -> link_rows[link][rank]   (the front half of k_events)
//   k_link_prefix  per (worker chain, k-mer): exclusive scan over the chain's links, applied to the worker's row
//                  -> link_rows[link][rank] = the row as the link finds it; the worker's row is advanced past the batch
//   k_events       as usual, one workgroup per link, dwell from memory
// grid (num_kmer/64, worker chains), 1024 threads: 64 k-mers x 16 groups of consecutive links.
// wlink_off[q]..wlink_off[q+1]: the links of worker chain q, in chain order
// the row `base` after n more samples (k > 6: base + n < 2^32, checked at staging)
template <bool DIRECT>
__device__ static inline uint32_t row_after(const uint32_t* __restrict__ pw, uint32_t base, unsigned long long n) {
    if (!DIRECT) return base + (uint32_t)n;
    if (!n) return base;
    return lcg_mul(base, lcg_jump2(pw, n < 4294967296ull ? (uint32_t)n : (uint32_t)(n % LCG_ORD2)));
}

// before (optional, [n_local_workers][num_kmer]): samples other GPUs draw from each stream ahead of this batch's local
// reads (range sharding, sqg_batch_run_end); the worker's own row is then left to k_rows_advance
template <bool DIRECT>
__global__ __launch_bounds__(1024) void k_link_prefix(const SigParams P, const int* __restrict__ wlink_off, const int* __restrict__ wlink_worker,
                                                      const uint32_t* __restrict__ before) {
    __shared__ unsigned long long sums[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, q = blockIdx.y;
    const int j = blockIdx.x * 64 + lane;
    const bool live = j < P.num_kmer;
    const int l0 = wlink_off[q], l1 = wlink_off[q + 1];
    const int per = (l1 - l0 + 15) / 16, la = min(l0 + g * per, l1), lb = min(la + per, l1);
    const size_t wj = (size_t)wlink_worker[q] * P.num_kmer + j;
    uint32_t* lr = P.link_rows + j;
    unsigned long long sum = 0;
    if (live) for (int l = la; l < lb; l++) sum += lr[(size_t)l * P.num_kmer];
    sums[g][lane] = sum;
    const uint32_t base = live ? P.rows[wj] : 0u;                  // read by every group BEFORE the barrier: group 0 rewrites it behind it (read behind the barrier, a
                                                                   // group that fell behind -- another context's kernels on the same CU -- saw the advanced row: round 4)
    __syncthreads();
    unsigned long long excl = 0, total = 0;
    for (int w = 0; w < 16; w++) { const unsigned long long x = sums[w][lane]; if (w < g) excl += x; total += x; }
    if (!live) return;
    if (before) excl += before[wj];
    uint32_t st = row_after<DIRECT>(P.pw, base, excl);
    for (int l = la; l < lb; l++) {
        const uint32_t cnt = lr[(size_t)l * P.num_kmer];
        lr[(size_t)l * P.num_kmer] = st;
        if (DIRECT) { if (cnt) st = lcg_mul(st, cnt < POW_N ? P.pw[2 * POW_N + cnt] : lcg_jump2(P.pw, cnt)); }
        else st += cnt;
    }
    if (g == 0 && !before) P.rows[wj] = row_after<DIRECT>(P.pw, base, total);
}

// range sharding: samples this batch's local reads draw from each (worker, k-mer) stream, from the link histograms
// (counts zeroed beforehand; workers without local reads stay 0).  Same grid as k_link_prefix.
__global__ __launch_bounds__(1024) void k_link_totals(const SigParams P, const int* __restrict__ wlink_off, const int* __restrict__ wlink_worker,
                                                      uint32_t* __restrict__ counts) {
    __shared__ unsigned long long sums[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, q = blockIdx.y;
    const int j = blockIdx.x * 64 + lane;
    const bool live = j < P.num_kmer;
    const int l0 = wlink_off[q], l1 = wlink_off[q + 1];
    const int per = (l1 - l0 + 15) / 16, la = min(l0 + g * per, l1), lb = min(la + per, l1);
    unsigned long long sum = 0;
    if (live) for (int l = la; l < lb; l++) sum += P.link_rows[(size_t)l * P.num_kmer + j];
    sums[g][lane] = sum;
    __syncthreads();
    if (g == 0 && live) {
        unsigned long long total = 0;
        for (int w = 0; w < 16; w++) total += sums[w][lane];
        counts[(size_t)wlink_worker[q] * P.num_kmer + j] = (uint32_t)total;
    }
}

// range sharding: every local worker's row moves past the whole batch -- the samples drawn here and on the other GPUs
template <bool DIRECT>
__global__ __launch_bounds__(256) void k_rows_advance(uint32_t* __restrict__ rows, const uint32_t* __restrict__ pw, size_t n,
                                                      const uint32_t* __restrict__ before, const uint32_t* __restrict__ mine,
                                                      const uint32_t* __restrict__ after) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long t = (unsigned long long)before[i] + mine[i] + after[i];
    if (t) rows[i] = row_after<DIRECT>(pw, rows[i], t);
}

// k > 6: the rows count samples; a stream's state only depends on the count mod (M-1)/2
__global__ __launch_bounds__(256) void k_rows_normalize(uint32_t* __restrict__ rows, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) rows[i] %= LCG_ORD2;
}

// ---- k_events + k_samples ------------------------------------------------------------------
// The per-read loop nest of src/gensig.c:249-282 is split at its only sequential dependency:
//
// k_events   one workgroup of NT threads per worker chain (a worker's reads of this batch, in
//            batch order); a read is walked in segments of NT*EPT consecutive events, EPT consecutive
//            events per thread: dwell draw, k-mer rank, block scan -> first sample of each 64-event
//            tile, and the hand-out of the per-(worker,k-mer) Lehmer streams IN EVENT ORDER.
//            k <= 6: the worker's 4^k stream states stay in LDS for the whole chain and the events of
//            a segment that share a k-mer chain through the state table itself (atomicExch).
//            k > 6: an LDS open-addressing hash; HBM holds the number of samples each stream has
//            produced, one returning atomicAdd per k-mer bin, state = seed * a^(2*count).
//            Either way an event sums the dwell of the same-k-mer events before it (bins hold 1-3
//            events) and needs one modular multiplication.  Output: 8 B per event {state at the
//            event's first draw, rank}.
// k_samples  (k_samples.h) one wavefront per work item, no inter-wave dependency and no block
//            barrier: 64 consecutive samples per step (contiguous int16 stores).
#ifndef SQG_EVENT_THREADS
#define SQG_EVENT_THREADS 256
#endif
#ifndef SQG_EVENT_EPT
#define SQG_EVENT_EPT 2     // consecutive events per thread of k_events (segment = SQG_EVENT_THREADS * SQG_EVENT_EPT events)
#endif
#ifndef SQG_EVENT_WAVES
#define SQG_EVENT_WAVES 7   // waves per SIMD the register allocation of k_events aims at (LDS allows 7 workgroups per CU)
#endif
#define MK_W 1024          // marker window (samples) per wavefront
#define MULT_N 512         // LDS jump constants cover events of up to 512 samples
#define EV_JUMP_N 256      // k_events' own jump table: bins of >= 256 samples go through the three-level global tables
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
template <int NT, bool DIRECT, int EPT, bool PART = false>
struct EvLds {
    static constexpr int SEG = NT * EPT;
    uint32_t keys[(DIRECT || PART) ? 1 : 2 * SEG];     // hash bins: k-mer rank
    uint32_t head[(DIRECT || PART) ? 1 : 2 * SEG];     // hash bin -> most recently inserted event of the segment (EV_NIL: none)
    uint32_t row[DIRECT ? 4096 : 1];         // DIRECT: the worker's stream states, resident for the whole chain; while a segment
                                             // is being handed out, ROW_BUSY | (most recently inserted event of the bin)
    uint32_t st[PART ? 1 : SEG];             // DIRECT: the state the bin's first exchanger swapped out of row[]; else: the
                                             // bin's state at the start of the segment, published by its first event
    uint32_t nxt[PART ? 1 : SEG];            // per event: (dwell << 16) | next event in the same bin
    // PART (k > 6, split chains; k_part.h): the events are bucketed by the top bits of their rank, stably
    uint4 pmask[PART ? (NT / 64) * PART_MAX : 1];   // per (wavefront, partition): which lanes hold an event of the partition,
                                                    // {first event of the lane: lanes 0-31, 32-63; second event: 0-31, 32-63}
    uint32_t prun[PART ? PART_MAX : 1];      // per partition: events of the link so far (counting pass), or the slot in
                                             // SigParams.part of the link's next event of the partition (scatter pass)
    uint2 pbase[PART ? (NT / 64) * PART_MAX : 1];   // per (wavefront, partition), scatter pass: {slot, position in the segment's
                                                    // partition-sorted order} of the wavefront's first event of the partition
    uint2 sorted[PART ? SEG : 1];            // the segment's events in partition-sorted order: {slot, record}
    uint32_t jump[EV_JUMP_N];   // 2 * a^(2j), j < EV_JUMP_N
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
// HIST: only the front half -- dwell draws, ranks, and the samples each k-mer stream is asked for, accumulated into the
// workgroup's row (split chains, see above)
// PART (k > 6, with HIST or after it): the stream hand-out is left to k_part_hist / k_part_scan / k_part_hand, which walk the
// events bucketed by the top bits of their rank (k_part.h).  With HIST: events per (link, partition) -> P.pcnt.  Without:
// every event is written to its slot in P.part -- (link, partition)'s first slot from P.poff, then stable in event order
// -- and its evrec holds {slot, rank} for k_part_home.
template <int NT, bool DIRECT, int DW, int EPT, bool HIST = false, bool PART = false>
__global__ __launch_bounds__(NT, (NT > 256 ? 4 : SQG_EVENT_WAVES)) void k_events(const SigParams P) {
    static_assert(!PART || (!DIRECT && EPT == 2), "PART: k > 6, two events per thread");
    typedef EvLds<NT, DIRECT, EPT, PART> Lds;
    __shared__ Lds L;
    __shared__ long long n1_sh;
    constexpr int NW = NT / 64, SEG = NT * EPT, HT = 2 * SEG, TL = 64 / EPT;   // TL: lanes per 64-event tile
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < EV_JUMP_N; i += NT) L.jump[i] = P.pw[2 * POW_N + i] << 1;   // (doubled: lcg_mul_dbl, two instructions less per product)
    for (int i = tid; i < 256; i += NT) L.lut[i] = (uint8_t)(P.meth ? meth_code((uint8_t)i) : base_code((uint8_t)i));

    const int chain = P.chain_order[blockIdx.x];
    const int c_lo = P.chain_off[chain], c_hi = P.chain_off[chain + 1];
    uint32_t* row = P.link_rows ? P.link_rows + (size_t)chain * P.num_kmer
                  : P.rows ? P.rows + (size_t)P.reads[P.chain_reads[c_lo]].worker * P.num_kmer : nullptr;
    const int k = P.k;
    const uint32_t kmask = (k >= 16) ? 0xffffffffu : ((1u << (2 * k)) - 1u);
    // k > 6: initial state of this worker's k-mer j is (seed_w + j) mod M (src/sim.c:249)
    const uint32_t seed_w = (uint32_t)(((unsigned long long)P.seed_base +
                                        (unsigned long long)(P.rows ? P.reads[P.chain_reads[c_lo]].worker : 0) * P.seed_step) % LCG_M);
    if (DIRECT && P.use_streams) for (int i = tid; i < P.num_kmer; i += NT) L.row[i] = HIST ? 0u : row[i];
    if (PART) {
        if (tid < PART_MAX) L.prun[tid] = (HIST || tid >= P.n_part) ? 0u : P.poff[(size_t)tid * P.n_links + chain] + P.pstart[(size_t)P.link_q[chain] * P.n_part + tid];
        for (int i = tid; i < (NT / 64) * PART_MAX; i += NT) L.pmask[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    const uint32_t a2nt = DW ? lcg_jump2(P.pw, (uint32_t)SEG) : 0u;      // time-stream jump over one segment
    const uint32_t a2jn = DW ? lcg_jump2(P.pw, (uint32_t)EV_JUMP_N) : 0u; // ... and over EV_JUMP_N events
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
        unsigned long long done = 0;                                  // samples before this segment (a read of >= 2^32 - 1 samples is
                                                                      // rejected by k_scan, src/sim.c:559-562: its tile offsets may wrap)
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
        // the next segment's inputs: EPT base bytes (+ halo) and EPT dwells per thread
        auto prefetch_next = [&](const int s0) {
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
        };
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
            // c_seg * a^(512*h): the segment state advanced to event 256*h (scalar unit)
            uint32_t c_seg_hi[(NT * EPT + EV_JUMP_N - 1) / EV_JUMP_N];
            if (DW) {
                c_seg_hi[0] = c_seg;
#pragma unroll
                for (int h2 = 1; h2 < (NT * EPT + EV_JUMP_N - 1) / EV_JUMP_N; h2++)
                    c_seg_hi[h2] = __builtin_amdgcn_readfirstlane(lcg_mul(c_seg_hi[h2 - 1], a2jn));
            }
#pragma unroll
            for (int q = 0; q < EPT; q++) {
                const int e = e0 + q;
                const bool valid = EV_IN(e);
                sps[q] = 0;
                if (DW == 0) {
                    sps[q] = valid ? (P.dwell ? (int)d_cur[q] : P.const_sps) : 0;
                } else if (valid) {
                    // event e uses draws 2e+1, 2e+2 of the worker's time stream after the read's first state
                    // draw of event id = tid*EPT+q: a^(2*id) = a^(2*(id % 256)) * a^(512*(id / 256)); the second factor is
                    // folded into the (scalar) segment state
                    constexpr int ID_HI = (NT * EPT + EV_JUMP_N - 1) / EV_JUMP_N;
                    const int id_ = tid * EPT + q;
                    uint32_t cs = c_seg_hi[0];
#pragma unroll
                    for (int h2 = 1; h2 < ID_HI; h2++) cs = (id_ / EV_JUMP_N == h2) ? c_seg_hi[h2] : cs;
                    const uint32_t c1 = lcg_mul_dbl(cs, L.jump[id_ % EV_JUMP_N]);
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
                    v = max(v, 1 - v);                                       // src/gensig.c:256: sps < 1 ? -sps + 1 : sps
                    if (P.dwell_unbounded && v > 65535) { atomicOr(P.err, 1u); v = 65535; }   // else: no draw can get there
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
            if (!DIRECT && !HIST && !PART && P.use_streams) for (int i = tid; i < HT; i += NT) { L.keys[i] = BIN_EMPTY; L.head[i] = EV_NIL; }
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
                        // my previous event's k-mer, shifted by one base
                        rank[q] = P.meth ? (rank[q - 1] % P.meth_top) * 5u + L.codes[cb + k - 1] : ((rank[q - 1] << 2) | L.codes[cb + k - 1]) & kmask;
                    } else if (P.meth) {                              // base-5 digits, src/seq.h:62-74
                        uint32_t rk = 0;
                        for (int i = 0; i < k; i++) rk = rk * 5u + L.codes[cb + i];
                        rank[q] = rk;
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
                if (HIST) {
                    if (PART) { if (EV_IN(e)) atomicAdd(&L.prun[rank[q] >> PART_SUB_BITS], 1u); }
                    else if (P.use_streams && EV_IN(e)) { if (DIRECT) atomicAdd(&L.row[rank[q]], (uint32_t)sps[q]); else atomicAdd(&row[rank[q]], (uint32_t)sps[q]); }
                } else if (PART) {
                    if (EV_IN(e)) atomicOr(reinterpret_cast<unsigned int*>(&L.pmask[wid * PART_MAX + (rank[q] >> PART_SUB_BITS)]) + 2 * q + (lane >> 5), 1u << (lane & 31));
                } else if (P.use_streams && EV_IN(e)) {
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
            if (HIST) {
                prefetch_next(s0);
                int run = lane_excl;
#pragma unroll
                for (int q = 0; q < EPT; q++) {
                    if (DW && EV_IN(e0 + q) && e0 + q == rd.ne0) n1_sh = (long long)done + run;   // samples of segment 0
                    run += sps[q];
                }
                done += (unsigned long long)seg_total;
                lds_barrier();                                        // codes / wsum are rewritten by the next segment
                return;
            }
            if (DIRECT || PART) lds_barrier(); else __syncthreads();                          // (2) global rows: + earlier row stores have landed
            prefetch_next(s0);                                        // lands while this segment waits for its states
            // first sample of every 64-event tile (TL lanes) within the read
            if ((lane & (TL - 1)) == 0 && EV_IN(e0)) P.tile_so[rd.tile_off + (e0 >> 6)] = (uint32_t)done + (uint32_t)lane_excl;
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
            if (PART) {
                // every wavefront, lane p: the partition's events of the segment per wavefront -> where this wavefront's events of
                // the partition go: slots (link-wide, L.prun) and positions in the segment's partition-sorted order
                {
                    uint32_t mine_before = 0, all = 0;
#pragma unroll
                    for (int w = 0; w < NW; w++) {
                        const uint4 o = L.pmask[w * PART_MAX + lane];
                        const uint32_t n = (uint32_t)(__builtin_popcount(o.x) + __builtin_popcount(o.y) + __builtin_popcount(o.z) + __builtin_popcount(o.w));
                        if (w < wid) mine_before += n;
                        all += n;
                    }
                    const uint32_t excl = (uint32_t)wave_incl_scan_dpp((int)all) - all;
                    L.pbase[wid * PART_MAX + lane] = make_uint2(L.prun[lane] + mine_before, excl + mine_before);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
#pragma unroll
                for (int q = 0; q < EPT; q++) {
                    if (EV_IN(e0 + q)) {
                        const uint32_t p = rank[q] >> PART_SUB_BITS;
                        const uint4 m = L.pmask[wid * PART_MAX + p];
                        uint32_t before = __builtin_amdgcn_mbcnt_hi(m.y, __builtin_amdgcn_mbcnt_lo(m.x, 0u));   // lower lanes
                        before = __builtin_amdgcn_mbcnt_hi(m.w, __builtin_amdgcn_mbcnt_lo(m.z, before));
                        if (q == 1 && (rank[0] >> PART_SUB_BITS) == p) before++;          // (event e0 exists when e0 + 1 does)
                        const uint2 pb = L.pbase[wid * PART_MAX + p];
                        const uint32_t slot = pb.x + before;
                        L.sorted[pb.y + before] = make_uint2(slot, (rank[q] & (PART_SUB - 1)) | ((uint32_t)sps[q] << 16));
                        c_ev[q] = slot;
                    }
                }
                lds_barrier();                                                                  // (3) every slot taken before the counts move
                // the records go out in sorted order: a partition's run of the segment is contiguous in memory
                {
                    const int n_seg = FULL ? SEG : min(SEG, ne - s0);
#pragma unroll
                    for (int q = 0; q < EPT; q++) {
                        const int j = q * NT + tid;
                        if (j < n_seg) { const uint2 v = L.sorted[j]; P.part[v.x] = v.y; }
                    }
                }
                {   // thread (wavefront w, lane p): the partition's counts move past the segment, the masks are cleared
                    const uint4 m = L.pmask[tid];
                    const uint32_t n = (uint32_t)(__builtin_popcount(m.x) + __builtin_popcount(m.y) + __builtin_popcount(m.z) + __builtin_popcount(m.w));
                    if (n) { atomicAdd(&L.prun[lane], n); L.pmask[tid] = make_uint4(0u, 0u, 0u, 0u); }
                }
            } else if (P.use_streams) {
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
                            if (n_old) cb = lcg_mul_dbl(cb, n_old < EV_JUMP_N ? L.jump[n_old] : lcg_jump2(P.pw, n_old) << 1);
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
                            const uint32_t m = lcg_mul_dbl(c_row[q], n < EV_JUMP_N ? L.jump[n] : lcg_jump2(P.pw, n) << 1);
                            if (first[q]) { c_ev[q] = c_row[q]; L.row[rank[q]] = m; }
                            else c_ev[q] = m;
                        } else if (first[q]) c_ev[q] = c_row[q];
                        else c_ev[q] = lcg_mul_dbl(L.st[fid[q]], prior[q] < EV_JUMP_N ? L.jump[prior[q]] : lcg_jump2(P.pw, prior[q]) << 1);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < EPT; q++)
                if (EV_IN(e0 + q)) P.evrec[rd.ev_off + e0 + q] = make_uint2(c_ev[q], rank[q]);
            done += (unsigned long long)seg_total;
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
    if (PART && HIST && tid < P.n_part) P.pcnt[(size_t)tid * P.n_links + chain] = L.prun[tid];
}

