// k_part.h -- k > 6 (7- to 9-mer tables), few workers with many reads each: the stream hand-out over bucketed events
// Part of the device code of the per-read signal path; included through sqg_kernels.h (see there for the overview).
//
// A 9-mer worker owns 4^9 streams (a 1 MiB row of sample counts).  Walking a read against that row in event order is one
// random 4-byte update per event: the rate of such updates, not bytes, bounds the T = K regime.  With few
// workers the row is shared by many reads, and the dependency "stream (worker, rank) is handed out in event order" only
// couples events of the SAME rank.  So a worker chain's events are bucketed by the top bits of the rank (the partition:
// rank >> 12, 64 of them for 9-mers), stably, and each bucket is walked against its own 4096-entry sub-row in LDS:
//
//   k_events<HIST, PART>  per link: dwell draws, ranks, events per (link, partition)                  -> pcnt
//   k_part_offsets        per (worker chain, partition): pcnt -> first slot of every (link, partition) in part[]
//   k_part_slices         the run of a (worker chain, partition) in part[] is cut into slices of equal length
//   k_events<PART>        per link: every event to its slot, {dwell, low 12 bits of the rank}; evrec = {slot, rank}
//   k_part_hist           per slice: samples per stream over the slice                               -> phist
//   k_part_scan           per (worker chain, rank): exclusive scan over the slices of the rank's partition on top of the
//                         worker's row: phist = the streams' STATES as the slice finds them; the row moves past the batch
//   k_part_hand           per slice: the slice in order against the sub-row in LDS: state[slot] = the stream's
//                         state at the event's first draw; the sub-row advances by a^(2 * dwell)
//   k_samples*            evrec.x is the slot: the sample kernels fetch state[slot]
//
// Everything between the two k_events passes and the sample kernels streams through memory sequentially (4 B per event and pass);
// the scatter and the gather move runs of a (512-event segment, partition), which consecutive segments of a link extend.
#pragma once

// grid (partitions, worker chains), 1024 threads, each with a run of consecutive links of the chain.
//   wlink_off[q]..wlink_off[q+1]   the links of worker chain q, in chain order
// poff[l][p] <- the chain's events of partition p before link l; ptotal[q][p] <- all of them
__device__ static inline void part_offsets_body(const int p, const int q, const uint32_t* __restrict__ pcnt, uint32_t* __restrict__ poff, const int n_part,
                                                const int n_links, const int* __restrict__ wlink_off, uint32_t* __restrict__ ptotal) {
    // An exclusive scan over the chain's links, 64 consecutive links per wavefront and step (whole lines in, whole lines out: with a
    // run of consecutive links per THREAD -- 22 of them at 22000 links per batch -- every load and store instruction touched 64 lines:
    // 25 us for 1.4 M words), a wavefront's steps four at a time (their loads in flight together).
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l0 = wlink_off[q], l1 = wlink_off[q + 1];
    const int chunk = (((l1 - l0 + 15) >> 4) + 63) & ~63;         // links per wavefront
    const int wa = min(l0 + wid * chunk, l1), wb = min(wa + chunk, l1);
    pcnt += (size_t)p * n_links; poff += (size_t)p * n_links;     // (partition-major)
    uint32_t own = 0;
    for (int base = wa; base < wb; base += 256) {
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = pcnt[min(base + 64 * j + lane, n_links - 1)];   // (unconditional: a conditional load is waited for before the next is issued)
#pragma unroll
        for (int j = 0; j < 4; j++) own += base + 64 * j + lane < wb ? v[j] : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) own += (uint32_t)__shfl_xor((int)own, o);
    if (lane == 0) wsum[wid] = own;
    __syncthreads();
    uint32_t carry = 0, total = 0;
    for (int w = 0; w < 16; w++) { if (w < wid) carry += wsum[w]; total += wsum[w]; }
    for (int base = wa; base < wb; base += 256) {
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = pcnt[min(base + 64 * j + lane, n_links - 1)];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int l = base + 64 * j + lane;
            const uint32_t cnt = l < wb ? v[j] : 0u;
            const uint32_t incl = (uint32_t)wave_incl_scan_dpp((int)cnt);
            if (l < wb) poff[l] = carry + incl - cnt;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    if (tid == 0) ptotal[(size_t)q * n_part + p] = total;
}
__global__ __launch_bounds__(1024) void k_part_offsets(const uint32_t* __restrict__ pcnt, uint32_t* __restrict__ poff, const int n_part, const int n_links,
                                                       const int* __restrict__ wlink_off, uint32_t* __restrict__ ptotal) {
    part_offsets_body(blockIdx.x, blockIdx.y, pcnt, poff, n_part, n_links, wlink_off, ptotal);
}

// The events of a (worker chain, partition) lie in part[] in the order the chain produces them; the hand-out walks them in that
// order, and any cut of the run is as good as any other: it is cut into SLICES of slice_len events (the last one shorter), so
// that a partition with many events (k-mers of poly-A tails, adaptors, satellite repeats) is simply more slices, not a longer one.
// One workgroup: pstart[pair] <- the pair's first slot in part[] (pair = chain * n_part + partition: part[] is chain-major, then
// partition-major), pfirst[pair] <- its first slice (pfirst[n_pairs] = number of slices); k_part_slice_bounds: the slots of slice s.  The host sized the tables for n_events / slice_len + n_pairs slices.
// (ptotal is read with agent-scope loads: in k_part_mid other workgroups of the same launch have written it)
__device__ static inline uint32_t ptotal_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ static inline void part_slices_body(uint32_t* __restrict__ pstart, const uint32_t* ptotal, const int n_pairs,
                                               const uint32_t slice_len, uint32_t* __restrict__ pfirst) {
    __shared__ uint32_t wsum[16], wtot[16];
    __shared__ uint32_t carry, carry_ev;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) { carry = 0; carry_ev = 0; }
    __syncthreads();
    for (int base = 0; base < n_pairs; base += 1024) {
        const int i = base + tid;
        const uint32_t tot = i < n_pairs ? ptotal_load(ptotal + i) : 0u;
        const uint32_t ns = (tot + slice_len - 1) / slice_len;
        const uint32_t incl = (uint32_t)wave_incl_scan_dpp((int)ns), incl_ev = (uint32_t)wave_incl_scan_dpp((int)tot);
        if (lane == 63) { wsum[wid] = incl; wtot[wid] = incl_ev; }
        __syncthreads();
        uint32_t first = carry, st = carry_ev;                     // the pair's first slice / first slot: the pairs lie in part[] in pair order
        for (int w = 0; w < wid; w++) { first += wsum[w]; st += wtot[w]; }
        first += incl - ns; st += incl_ev - tot;
        if (i < n_pairs) { pfirst[i] = first; pstart[i] = st; }
        __syncthreads();
        if (tid == 1023) { carry = first + ns; carry_ev = st + tot; }
        __syncthreads();
    }
    if (tid == 0) pfirst[n_pairs] = carry;
}
// slice s: the pair it belongs to by bisection of pfirst, then its slots
__device__ static inline void part_slice_bounds_one(const uint32_t s, const uint32_t* pstart, const uint32_t* ptotal, const int n_pairs,
                                                    const uint32_t slice_len, const uint32_t* pfirst, uint32_t* __restrict__ slice_lo, uint32_t* __restrict__ slice_hi) {
    int lo = 0, hi = n_pairs;                                      // the last pair with pfirst <= s (pairs without events share their successor's)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pfirst[mid] <= s) lo = mid; else hi = mid; }
    const uint32_t k = s - pfirst[lo], st = pstart[lo];
    slice_lo[s] = st + k * slice_len;
    slice_hi[s] = st + min((k + 1) * slice_len, ptotal_load(ptotal + lo));
}
// the bounds of all slices by the one workgroup that has just written pfirst / pstart (part_slices_body): few pairs (one worker chain
// of 9-mers: 64): the tables in LDS, the bisection there
#define PART_BOUNDS_LDS 1024
__device__ static inline void part_slice_bounds_wg(const uint32_t* pstart, const uint32_t* ptotal, const int n_pairs, const uint32_t slice_len,
                                                   const uint32_t* pfirst, uint32_t* __restrict__ slice_lo, uint32_t* __restrict__ slice_hi) {
    __shared__ uint32_t s_first[PART_BOUNDS_LDS + 1], s_start[PART_BOUNDS_LDS], s_total[PART_BOUNDS_LDS];
    __syncthreads();                                               // (the workgroup's own stores: visible to it behind the barrier)
    const uint32_t ns = pfirst[n_pairs];
    if (n_pairs > PART_BOUNDS_LDS) {
        for (uint32_t s = threadIdx.x; s < ns; s += 1024) part_slice_bounds_one(s, pstart, ptotal, n_pairs, slice_len, pfirst, slice_lo, slice_hi);
        return;
    }
    for (int i = threadIdx.x; i < n_pairs; i += 1024) { s_first[i] = pfirst[i]; s_start[i] = pstart[i]; s_total[i] = ptotal_load(ptotal + i); }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < ns; s += 1024) {
        int lo = 0, hi = n_pairs;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_first[mid] <= s) lo = mid; else hi = mid; }
        const uint32_t k = s - s_first[lo], st = s_start[lo];
        slice_lo[s] = st + k * slice_len;
        slice_hi[s] = st + min((k + 1) * slice_len, s_total[lo]);
    }
}
// One workgroup; with slice_lo given it goes on to the slices' bounds (what k_part_slice_bounds does with a thread per slice)
__global__ __launch_bounds__(1024) void k_part_slices(uint32_t* pstart, const uint32_t* ptotal, const int n_pairs,
                                                      const uint32_t slice_len, uint32_t* pfirst, uint32_t* slice_lo, uint32_t* slice_hi) {
    part_slices_body(pstart, ptotal, n_pairs, slice_len, pfirst);
    if (!slice_lo) return;
    part_slice_bounds_wg(pstart, ptotal, n_pairs, slice_len, pfirst, slice_lo, slice_hi);
}

// one thread per slice (the host's bound)
__global__ __launch_bounds__(256) void k_part_slice_bounds(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ ptotal, const int n_pairs,
                                                           const uint32_t slice_len, const uint32_t* __restrict__ pfirst,
                                                           uint32_t* __restrict__ slice_lo, uint32_t* __restrict__ slice_hi) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= pfirst[n_pairs]) return;
    part_slice_bounds_one(s, pstart, ptotal, n_pairs, slice_len, pfirst, slice_lo, slice_hi);
}

// grid: slices (the host's bound; the live ones are pfirst[n_pairs]), 256 threads.  phist[s][sub] <- samples the slice's events draw from the stream
// (sums commute: a thread takes four consecutive records with one 16-B load -- 4-B loads stream at 4.0 TB/s on this machine, 16-B
// loads at 6.3, tools/pmc_calib.hip -- 2048 events of the slice per step, the next step's loads in flight during this one's atomics)
#ifndef PART_HIST_PAD
#define PART_HIST_PAD 0          // A/B: words of LDS a workgroup reserves beyond its table (fewer workgroups per CU: whole rounds of slices)
#endif
#ifndef PART_HIST_V4
#define PART_HIST_V4 1
#endif
#ifndef PART_NT
#define PART_NT 0                // A/B (round 5): part[] read with non-temporal loads in k_part_hist / the hand-out (each record is read once per kernel)
#endif
typedef uint32_t part_u32x4 __attribute__((ext_vector_type(4)));
__device__ static inline uint32_t part_ld(const uint32_t* p) {
#if PART_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ static inline uint4 part_ld(const uint4* p) {
#if PART_NT
    const part_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const part_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
#define PART_LD(p_) part_ld(p_)
// Workgroups from `first_items` on (round 5) prepare the lean sample kernel's work items instead (items_body, k_samples.h: they need the
// scan's offsets and the scatter pass' tile links, both complete before this launch, and nothing of this kernel): k_items as a launch of
// its own was 15 us of work behind a launch gap.
__device__ static inline void items_body(const SigParams& P, const int block, const int n_stiles);
__global__ __launch_bounds__(256) void k_part_hist(const uint32_t* __restrict__ part, const uint32_t* __restrict__ slice_lo,
                                                   const uint32_t* __restrict__ slice_hi, const uint32_t* __restrict__ n_slices,
                                                   uint32_t* __restrict__ phist, const SigParams P, const int n_stiles, const unsigned first_items) {
    __shared__ uint32_t row[PART_SUB + PART_HIST_PAD];
    const int tid = threadIdx.x;
    if (blockIdx.x >= first_items) { items_body(P, (int)(blockIdx.x - first_items), n_stiles); return; }
    if (blockIdx.x >= *n_slices) return;
    for (int i = tid; i < PART_SUB; i += 256) row[i] = 0u;
    __syncthreads();
    const uint32_t lo = slice_lo[blockIdx.x], hi = slice_hi[blockIdx.x];
#if PART_HIST_V4
    // the slice from its first 16-B aligned record on in uint4s; the (up to three) records before that one by one
    const uint32_t lo4 = min((lo + 3u) & ~3u, hi);
    if (lo + (uint32_t)tid < lo4) { const uint32_t rec = part[lo + tid]; atomicAdd(&row[rec & (PART_SUB - 1)], rec >> 16); }
    const uint4* in = reinterpret_cast<const uint4*>(part + lo4) + tid;
    uint4 rec[2], nxt[2];
#pragma unroll
    for (int q = 0; q < 2; q++) rec[q] = PART_LD(in + 256 * q);              // (unconditional: PART_SLACK entries behind the last slice)
    for (uint32_t b = lo4; b < hi; b += 2048) {
#pragma unroll
        for (int q = 0; q < 2; q++) nxt[q] = PART_LD(in + 512 + 256 * q);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t w[4] = {rec[q].x, rec[q].y, rec[q].z, rec[q].w};
            const uint32_t at = b + 1024 * q + 4 * tid;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool live = at + i < hi;
                const uint32_t sub = w[i] & (PART_SUB - 1);
                // a wavefront whose events all fall on one stream (poly-A tails ...): one add of the wavefront's sum instead of 64
                // adds to one address
                if (__builtin_amdgcn_ballot_w64(live && sub == (uint32_t)__builtin_amdgcn_readfirstlane((int)sub)) == ~0ull) {
                    int sum = (int)(w[i] >> 16);
                    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                    if ((tid & 63) == 0) atomicAdd(&row[sub], (uint32_t)sum);
                } else if (live) atomicAdd(&row[sub], w[i] >> 16);
            }
        }
#pragma unroll
        for (int q = 0; q < 2; q++) rec[q] = nxt[q];
        in += 512;
    }
#else
    const uint32_t* in = part + lo + tid;
    uint32_t rec[4], nxt[4];
#pragma unroll
    for (int q = 0; q < 4; q++) rec[q] = in[256 * q];                        // (unconditional: PART_SLACK entries behind the last slice)
    for (uint32_t b = lo; b < hi; b += 1024) {
#pragma unroll
        for (int q = 0; q < 4; q++) nxt[q] = in[1024 + 256 * q];             // the next step's records are in flight during this one's atomics
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bool live = b + 256 * q + tid < hi;
            const uint32_t sub = rec[q] & (PART_SUB - 1);
            // a wavefront whose events all fall on one stream (poly-A tails ...): one add of the wavefront's sum instead of 64
            // adds to one address
            if (__builtin_amdgcn_ballot_w64(live && sub == (uint32_t)__builtin_amdgcn_readfirstlane((int)sub)) == ~0ull) {
                int sum = (int)(rec[q] >> 16);
                for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                if ((tid & 63) == 0) atomicAdd(&row[sub], (uint32_t)sum);
            } else if (live) atomicAdd(&row[sub], rec[q] >> 16);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) rec[q] = nxt[q];
        in += 1024;
    }
#endif
    __syncthreads();
    uint32_t* dst = phist + (size_t)blockIdx.x * PART_SUB;
    for (int i = tid; i < PART_SUB; i += 256) dst[i] = row[i];
}

// Per (worker chain, rank): exclusive scan of phist over the slices of the rank's partition, in chain order, starting
// from the worker's row (sample counts, reduced mod (M-1)/2: only that matters for a^(2n)) plus, with range sharding, what the
// ranges before this one draw (`before`; the row itself is then left to k_rows_advance).  What a slice gets is the stream's STATE
// as the slice finds it: seed(worker, rank) * a^(2 * samples before), the seed being (seed_base + worker*seed_step + rank) mod M
// (src/sim.c:249).  states != 0 (k <= 6): the worker's row holds the streams' states themselves, as every other path of k <= 6
// keeps them, and moves on by a^(2 * samples of the batch).
// grid (num_kmer / R, worker chains), R x G threads: R ranks x G runs of consecutive slices.  64 x 16: 256-B rows of cells per
// wavefront; 16 x 64 when that would leave CUs without a workgroup (one partition with thousands of slices, k <= 6); 256 x 1 when
// the (chain, partition) pairs are many and have a slice or two each (dozens of workers: a workgroup per 64 ranks and pair would be
// millions of nearly idle workgroups).
template <int R, int G>
__global__ __launch_bounds__(R * G) void k_part_scan(uint32_t* __restrict__ phist, uint32_t* __restrict__ rows, const int num_kmer, const int n_part,
                                                    const uint32_t* __restrict__ pfirst, const int* __restrict__ wlink_worker,
                                                    const uint32_t* __restrict__ before, const uint32_t* __restrict__ pw,
                                                    const uint32_t seed_base, const uint32_t seed_step, const int states, unsigned int* __restrict__ err) {
    __shared__ unsigned long long sums[G][R];
    const int lane = threadIdx.x % R, g = threadIdx.x / R, q = blockIdx.y;
    const int j = blockIdx.x * R + lane;
    const bool live = j < num_kmer;
    const int pair = q * n_part + ((blockIdx.x * R) >> PART_SUB_BITS);          // (R consecutive ranks: one partition)
    const uint32_t s0 = pfirst[pair], s1 = pfirst[pair + 1];
    const uint32_t per = (s1 - s0 + G - 1) / G, sa = min(s0 + (uint32_t)g * per, s1), sb = min(sa + per, s1);
    uint32_t* col = phist + (j & (PART_SUB - 1));
    unsigned long long sum = 0;
    if (live) for (uint32_t s = sa; s < sb; s++) sum += col[(size_t)s * PART_SUB];
    sums[g][lane] = sum;
    const int w = wlink_worker[q];
    const size_t wj = (size_t)w * num_kmer + j;
    const uint32_t row = live ? rows[wj] : 0u;                     // (read by every run before the barrier, rewritten by run 0 behind it)
    __syncthreads();
    unsigned long long excl = 0, total = 0;
    for (int w2 = 0; w2 < G; w2++) { const unsigned long long x = sums[w2][lane]; if (w2 < g) excl += x; total += x; }
    if (!live) return;
    auto jump = [&](const unsigned long long n) -> uint32_t { return lcg_jump2(pw, n < 4294967296ull ? (uint32_t)n : (uint32_t)(n % LCG_ORD2)); };
    uint32_t origin;                                               // the state `run` samples are counted from
    unsigned long long run = excl;
    if (states) origin = row;
    else {
        const unsigned long long sv = ((unsigned long long)seed_base + (unsigned long long)w * seed_step) % LCG_M + (unsigned long long)j;
        origin = (uint32_t)(sv >= LCG_M ? sv - LCG_M : sv);
        run += row % LCG_ORD2;
    }
    if (before) run += before[wj];
    const unsigned long long run0 = run - excl;
    for (uint32_t s = sa; s < sb; s++) {
        uint32_t* cell = col + (size_t)s * PART_SUB;
        const uint32_t cnt = *cell;
        *cell = run ? lcg_mul(origin, jump(run)) : origin;
        run += cnt;
    }
    if (g == 0) {
        if (!states && run0 + total > 0xffffffffull) atomicOr(err, 32u);       // one stream asked for >= 2^32 samples by one batch
        if (!before) rows[wj] = states ? (total ? lcg_mul(origin, jump(total)) : origin) : (uint32_t)(run0 + total);
    }
}

// range sharding: samples this batch's local reads draw from each (worker, rank) stream (counts zeroed beforehand)
__global__ __launch_bounds__(256) void k_part_totals(const uint32_t* __restrict__ phist, const int num_kmer, const int n_part, const uint32_t* __restrict__ pfirst,
                                                     const int* __restrict__ wlink_worker, uint32_t* __restrict__ counts) {
    const int q = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= num_kmer) return;
    uint32_t sum = 0;
    const int pair = q * n_part + (j >> PART_SUB_BITS);
    for (uint32_t s = pfirst[pair]; s < pfirst[pair + 1]; s++) sum += phist[(size_t)s * PART_SUB + (j & (PART_SUB - 1))];
    counts[(size_t)wlink_worker[q] * num_kmer + j] = sum;
}

// grid: slices, ONE wavefront each (the sub-row of stream states, 16 KiB, is the workgroup's LDS: no sharing,
// every lane busy).  The slice is walked in order in steps of 1024 events: the step's 16 loads per lane are in flight while the
// step before is handed out, its 16 stores go out together at the end (on gfx950 a wavefront that waits for a load also waits
// for its own earlier stores: they are kept few and far from the loads).  A step is handed out in four phases of 256 events
// (event 64q + lane of the phase is the lane's q-th).
// In-order hand-out of a phase.  First try, plain LDS traffic only: every event writes its position in the phase to its
// stream's tag and reads it back; the events that do not find themselves write "contended" over the tag; an event that still
// finds itself after that is alone on its stream in this phase: it takes the stream's state and advances it by its dwell
// (a^(2 * dwell) from LDS).  The events of contended streams (6 % of them: 256 events over 4096 streams) queue on a small
// hashed tag array with an atomic min of their position, so that the earliest wins the round; what is still waiting after two
// such rounds belongs to streams with many events in the phase (poly-A tails, adaptors): those are taken a stream at a time,
// the dwells of the stream's events scanned in order.
// LDS operations of one wavefront execute in program order, which is all the ordering the protocol needs.
// BIGD: dwells of PART_JT samples and more exist (their multiplier comes from the global jump tables)
#define PART_ATAGS 256
#define PART_JT 256
#define PART_STEP 1024
#ifndef PART_ROW_CLAIMS
#define PART_ROW_CLAIMS 1        // the first-try claims are written into the sub-row itself (no tag array: 18 instead of 26 KiB per wavefront)
#endif
#define PART_CLAIM 0x80000000u
#define PART_CONTENDED 0xc0000000u
#ifndef PART_ATOMIC_ROUNDS
#define PART_ATOMIC_ROUNDS 2     // rounds of the earliest-wins protocol before the contended streams are taken one stream at a time
#endif
#ifndef PART_MANY
#define PART_MANY 24             // waiting events of a phase from which on the streams are taken one at a time straight away
#endif
#define PART_SLACK (4 * PART_STEP + 64)   // entries behind the bucketed events: read ahead by the last step of a slice (k_part_hist: two steps of 2048), and the dump of idle lanes' stores
template <bool BIGD>
__global__ __launch_bounds__(64) void k_part_hand(const uint32_t* __restrict__ part, uint32_t* __restrict__ state_out,
                                                  const uint32_t* __restrict__ slice_lo, const uint32_t* __restrict__ slice_hi,
                                                  const uint32_t* __restrict__ n_slices,
                                                  const uint32_t* __restrict__ phist, const uint32_t* __restrict__ pw, const uint32_t dump) {
    __shared__ uint32_t row[PART_SUB];                            // stream states (< 2^31); during a phase also claims: PART_CLAIM | position
#if !PART_ROW_CLAIMS
    __shared__ uint16_t tg[PART_SUB];                             // position in the phase of the (last) event that took the tag; 0x100: contended
#endif
    __shared__ uint32_t atg[PART_ATAGS];
    __shared__ uint32_t jt[PART_JT];                              // a^(2j)
    __shared__ uint32_t jt1[PART_JT];                             // a^(2*256*j): with jt, any jump below 65536 samples without leaving LDS
    constexpr int NR = PART_STEP / 64;                            // records per lane and step
    const int lane = threadIdx.x;
    if (blockIdx.x >= *n_slices) return;
    const uint4* src = reinterpret_cast<const uint4*>(phist + (size_t)blockIdx.x * PART_SUB);
    for (int i = lane; i < PART_SUB / 4; i += 64) reinterpret_cast<uint4*>(row)[i] = src[i];
    for (int i = lane; i < PART_ATAGS; i += 64) atg[i] = 0xffffffffu;
    for (int i = lane; i < PART_JT; i += 64) jt[i] = pw[2 * POW_N + i];
    for (int i = lane; i < PART_JT; i += 64) jt1[i] = (i & 3) ? lcg_mul(pw[3 * POW_N + (i >> 2)], pw[2 * POW_N + 256 * (i & 3)]) : pw[3 * POW_N + (i >> 2)];
    const uint32_t lo = slice_lo[blockIdx.x], hi = slice_hi[blockIdx.x];
    uint32_t cur[NR], nxt[NR];
    // a^(2n): two LDS look-ups and a multiplication below 65536 samples (a stream with hundreds of events in a phase), else the global tables
    auto jump = [&](uint32_t n) -> uint32_t {
        if (n < PART_JT) return jt[n];
        if (n < PART_JT * PART_JT) return lcg_mul(jt[n & (PART_JT - 1)], jt1[n >> 8]);
        return lcg_jump2(pw, n);
    };
#pragma unroll
    for (int r = 0; r < NR; r++) cur[r] = part[lo + 64 * r + lane];   // (unconditional: PART_SLACK entries behind the last slice)
    __syncthreads();
    for (uint32_t b = lo; b < hi; b += PART_STEP) {
#pragma unroll
        for (int r = 0; r < NR; r++) nxt[r] = part[b + PART_STEP + 64 * r + lane];
        uint32_t out[NR];
#pragma unroll
        for (int ph = 0; ph < NR / 4; ph++) {
            uint32_t sub[4], mul[4], st[4], dw[4];
            uint16_t pri[4];
            bool pend[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t rec = cur[4 * ph + q], d = rec >> 16;
                sub[q] = rec & (PART_SUB - 1);
                dw[q] = d;
                pri[q] = (uint16_t)(64 * q + lane);
                pend[q] = b + 256 * ph + 64 * q + lane < hi;
                mul[q] = jt[d & (PART_JT - 1)];
                if (BIGD && d >= PART_JT) mul[q] = lcg_jump2(pw, d);
            }
#if PART_ROW_CLAIMS
            // the claims live in the sub-row itself (a state never has bit 31): every event reads its stream's state, THEN writes
            // its claim over it; whoever still finds its own claim after the others marked theirs "contended" is alone
#pragma unroll
            for (int q = 0; q < 4; q++) st[q] = __hip_atomic_load(&row[sub[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
            for (int q = 0; q < 4; q++) if (pend[q]) __hip_atomic_store(&row[sub[q]], PART_CLAIM | pri[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            uint32_t tc[4];
#pragma unroll
            for (int q = 0; q < 4; q++) tc[q] = __hip_atomic_load(&row[sub[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
            for (int q = 0; q < 4; q++) if (pend[q] && tc[q] != (PART_CLAIM | pri[q])) __hip_atomic_store(&row[sub[q]], PART_CONTENDED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
            for (int q = 0; q < 4; q++) tc[q] = __hip_atomic_load(&row[sub[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (pend[q] && tc[q] == (PART_CLAIM | pri[q])) {     // alone on the stream in this phase
                    __hip_atomic_store(&row[sub[q]], lcg_mul(st[q], mul[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    pend[q] = false;
                }
            }
#else
#pragma unroll
            for (int q = 0; q < 4; q++) if (pend[q]) __hip_atomic_store(&tg[sub[q]], pri[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            uint16_t t[4];
#pragma unroll
            for (int q = 0; q < 4; q++) t[q] = __hip_atomic_load(&tg[sub[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
            for (int q = 0; q < 4; q++) if (pend[q] && t[q] != pri[q]) __hip_atomic_store(&tg[sub[q]], (uint16_t)0x100, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                t[q] = __hip_atomic_load(&tg[sub[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                st[q] = __hip_atomic_load(&row[sub[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (pend[q] && t[q] == pri[q]) {                   // alone on the stream in this phase
                    __hip_atomic_store(&row[sub[q]], lcg_mul(st[q], mul[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    pend[q] = false;
                }
            }
#endif
            int round = 0;
            for (;;) {                                               // contended streams, in order
                const int n_pend = __builtin_popcountll(__builtin_amdgcn_ballot_w64(pend[0])) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(pend[1])) +
                                   __builtin_popcountll(__builtin_amdgcn_ballot_w64(pend[2])) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(pend[3]));
                if (n_pend == 0) break;
                // two events on a stream (the usual case): the earliest wins a round.  Many waiting events mean a stream with many
                // events (dozens of atomics on one LDS word serialise): such a phase goes stream by stream at once
                if (n_pend < PART_MANY && round++ < PART_ATOMIC_ROUNDS) {
#pragma unroll
                    for (int q = 0; q < 4; q++) if (pend[q]) atomicMin(&atg[sub[q] & (PART_ATAGS - 1)], (uint32_t)pri[q]);
                    uint32_t ta[4], s0[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        ta[q] = __hip_atomic_load(&atg[sub[q] & (PART_ATAGS - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        s0[q] = __hip_atomic_load(&row[sub[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (pend[q] && ta[q] == (uint32_t)pri[q]) {
#if PART_ROW_CLAIMS
                            if (s0[q] & PART_CLAIM) s0[q] = st[q];  // the stream's first event of the phase: the row still holds a claim, the state is the one read before
#endif
                            __hip_atomic_store(&row[sub[q]], lcg_mul(s0[q], mul[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            __hip_atomic_store(&atg[sub[q] & (PART_ATAGS - 1)], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            st[q] = s0[q];
                            pend[q] = false;
                        }
                    }
                    continue;
                }
                // a stream with many events in the phase (homopolymers, an adaptor every read carries): the whole stream in one
                // round -- its events' dwells are scanned in order, event i starts a^(2 * samples before it) in
                uint32_t key = 0xffffffffu;
#pragma unroll
                for (int q = 0; q < 4; q++) if (pend[q]) key = min(key, (uint32_t)pri[q]);
                for (int o = 32; o > 0; o >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, o));
                const int ql = (int)(key >> 6), ll = (int)(key & 63u);
                const uint32_t sub_sel = ql == 0 ? sub[0] : ql == 1 ? sub[1] : ql == 2 ? sub[2] : sub[3];
                const uint32_t sstar = (uint32_t)__shfl((int)sub_sel, ll);
                uint32_t sv = __hip_atomic_load(&row[sstar], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#if PART_ROW_CLAIMS
                {   // (the row may still hold a claim: the state is then the one the stream's events read before claiming)
                    const uint32_t st_sel = ql == 0 ? st[0] : ql == 1 ? st[1] : ql == 2 ? st[2] : st[3];
                    const uint32_t s_lead = (uint32_t)__shfl((int)st_sel, ll);
                    if (sv & PART_CLAIM) sv = s_lead;
                }
#endif
                uint32_t run = 0;                                    // samples of the stream's earlier events of the phase
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool mem = pend[q] && sub[q] == sstar;
                    const int dq = mem ? (int)dw[q] : 0;
                    const int incl = wave_incl_scan_dpp(dq);
                    const uint32_t before = run + (uint32_t)(incl - dq);
                    if (mem) {
                        st[q] = before ? lcg_mul(sv, jump(before)) : sv;
                        pend[q] = false;
                    }
                    run += (uint32_t)__shfl(incl, 63);
                }
                if (lane == 0) __hip_atomic_store(&row[sstar], lcg_mul(sv, jump(run)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) out[4 * ph + q] = st[q];
        }
#pragma unroll
        for (int r = 0; r < NR; r++) { const uint32_t i = b + 64 * r + lane; state_out[i < hi ? i : dump + lane] = out[r]; }
#pragma unroll
        for (int r = 0; r < NR; r++) cur[r] = nxt[r];
    }
}

// ---- the hand-out by ordered LDS atomics ----------------------------------------------------------------------------
// What an event needs is the number of samples the slice's EARLIER events draw from its stream: with n of them the event starts
// at base[stream] * a^(2n), base being the state the slice finds (k_part_scan).  `n = atomicAdd(&cnt[stream], dwell)` is that
// number if the additions happen in event order -- and in one wavefront they do: LDS instructions of a wavefront execute in
// program order, and the lanes of one LDS atomic that meet on an address are served in ascending lane order.  The second half
// is a property of the LDS pipeline, not of the ISA manual: k_lds_order_check measures it on the device when the context is
// created (random address patterns from all-distinct to all-equal, idle lanes), and a context whose device does not pass
// uses k_part_hand, the claim protocol above, which assumes program order only.
// No protocol, no divergence, and a stream with thousands of events in a slice (poly-A tails, adaptors) costs what its
// atomics cost the LDS: one address, one lane after the other.
// a^(2n) comes from three 256-entry tables in LDS (n < 2^24: a stream's samples in one slice; else the global tables).
#define PART_LT 256
#ifndef HAND_ABL
#define HAND_ABL 0
#endif
#ifndef HAND_STEP
#define HAND_STEP 1024          // events per step of k_part_hand_ord (A/B: 2048 -- twice the loads in flight; a slice is a multiple of PART_STEP)
#endif
// Every slice of every batch also SAMPLES the property it stands on, in production shape (this table, these dwells, this
// occupancy): the counts its first PART_CHECK events received are compared with order-free prefix sums over the same records (a
// loop over the earlier events in LDS, ~1 % of the slice's work); a mismatch sets bit 64 of the batch's error word and the batch
// fails (sqg_batch_wait: SQG_EDEVICE).  fault (tests: SQG_TEST_ORDER_FAULT=1): the first two rows' atomics are issued in the
// wrong order, which the check has to notice.
#define PART_CHECK 128
struct HandLds {
    uint32_t base[PART_SUB];                                      // the streams' states as the slice finds them
    uint32_t cnt[PART_SUB];                                       // samples handed out so far
    uint4 chk[PART_CHECK / 4];                                    // the slice's first events, for the order check
    uint32_t lt0[PART_LT], lt1[PART_LT], lt2[PART_LT];            // a^(2j), a^(2 * 256 j), a^(2 * 65536 j), DOUBLED (lcg_mul_dbl)
};
__device__ static inline void hand_tables(HandLds& H, const uint32_t* __restrict__ pw, const int tid, const int nthreads) {
    for (int i = tid; i < PART_LT; i += nthreads) {
        H.lt0[i] = pw[2 * POW_N + i] << 1;
        H.lt1[i] = ((i & 3) ? lcg_mul(pw[3 * POW_N + (i >> 2)], pw[2 * POW_N + 256 * (i & 3)]) : pw[3 * POW_N + (i >> 2)]) << 1;
        H.lt2[i] = ((i & 15) ? lcg_mul(pw[4 * POW_N + (i >> 4)], pw[3 * POW_N + 64 * (i & 15)]) : pw[4 * POW_N + (i >> 4)]) << 1;
    }
}
// One slice, one wavefront (H.lt* filled and visible).  No workgroup barrier inside: LDS instructions of a wavefront execute in order,
// the fences keep the compiler from moving them.
__device__ static __forceinline__ void hand_slice(HandLds& H, const uint32_t* __restrict__ part, uint32_t* __restrict__ state_out, const uint32_t lo, const uint32_t hi,
                                                  const uint32_t* __restrict__ phist_row, const uint32_t* __restrict__ pw, unsigned int* __restrict__ err, const int fault, const int lane) {
    constexpr int NR = HAND_STEP / 64;                            // records per lane and step
    uint32_t* const base = H.base; uint32_t* const cnt = H.cnt; uint4* const chk = H.chk;
    uint32_t* const lt0 = H.lt0; uint32_t* const lt1 = H.lt1; uint32_t* const lt2 = H.lt2;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");         // (the slice before this one has read its tables)
    const uint4* src = reinterpret_cast<const uint4*>(phist_row);
    for (int i = lane; i < PART_SUB / 4; i += 64) { reinterpret_cast<uint4*>(base)[i] = src[i]; reinterpret_cast<uint4*>(cnt)[i] = make_uint4(0u, 0u, 0u, 0u); }
    const uint32_t* in = part + lo + lane;                          // (one address register; the records of a step sit at constant offsets)
    uint32_t* out_p = state_out + lo + lane;
    uint32_t cur[NR], nxt[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) cur[r] = PART_LD(in + 64 * r);     // (unconditional: PART_SLACK entries behind the last slice)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // one step: event 64 r + lane is the lane's r-th -- instruction order, then lane order
    auto step = [&](auto full_tag, auto check_tag, const uint32_t left) {
        constexpr bool FULL = decltype(full_tag)::value, CHECK = decltype(check_tag)::value;
#pragma unroll
        for (int r = 0; r < NR; r++) nxt[r] = PART_LD(in + HAND_STEP + 64 * r);
        uint32_t n[NR], out[NR];
        if (CHECK && fault) {                                       // (test hook) rows 0 and 1 in the wrong order
            const uint32_t t = cur[0]; cur[0] = cur[1]; cur[1] = t;
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            n[r] = 0;
#if HAND_ABL == 1      /* timing-only ablation: no atomics */
            n[r] = cur[r] & 7u;
#else
            if (FULL || (uint32_t)(64 * r + lane) < left) n[r] = __hip_atomic_fetch_add(&cnt[cur[r] & (PART_SUB - 1)], cur[r] >> 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");                          // (the compiler keeps the atomics in this order)
#endif
        }
        if (CHECK && fault) {
            const uint32_t t = cur[0]; cur[0] = cur[1]; cur[1] = t;
            const uint32_t u = n[0]; n[0] = n[1]; n[1] = u;
        }
        if (CHECK) {
            // the slice's first PART_CHECK events (cnt started at zero): what each received against the samples of the earlier ones
            // on its stream, found without any ordering assumption
            static_assert(PART_CHECK == 128, "two rows");
            const bool v0 = FULL || (uint32_t)lane < left, v1 = FULL || (uint32_t)(64 + lane) < left;
            reinterpret_cast<uint32_t*>(chk)[lane] = v0 ? cur[0] : 0u;            // (an event that does not exist: dwell 0)
            reinterpret_cast<uint32_t*>(chk)[64 + lane] = v1 ? cur[1] : 0u;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const uint32_t s0 = cur[0] & (PART_SUB - 1), s1 = cur[1] & (PART_SUB - 1);
            uint32_t w0 = 0, w1 = 0;
#pragma unroll 4
            for (int j4 = 0; j4 < PART_CHECK / 4; j4++) {
                const uint4 q = chk[j4];
                const uint32_t rec[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int j = 4 * j4 + i;
                    const uint32_t d = rec[i] >> 16, sb = rec[i] & (PART_SUB - 1);
                    if (j < lane && sb == s0) w0 += d;
                    if (j < 64 + lane && sb == s1) w1 += d;
                }
            }
            if ((v0 && n[0] != w0) || (v1 && n[1] != w1)) atomicOr(err, 64u);
        }
        // (the step's 2 x NR table reads first, back to back, then the multiplications, then ONE test for the rare second level: a test
        // and a branch per record made every record wait for its own two LDS round trips -- 4100 cycles per 1024-event step and
        // wavefront, of which the atomics are 660)
        {
            uint32_t bs[NR], m0[NR];
#pragma unroll
            for (int r = 0; r < NR; r++) { bs[r] = base[cur[r] & (PART_SUB - 1)]; m0[r] = lt0[n[r] & (PART_LT - 1)]; }
            bool big = false;
#pragma unroll
            for (int r = 0; r < NR; r++) { out[r] = lcg_mul_dbl(bs[r], m0[r]); big |= n[r] >= (uint32_t)PART_LT; }
            if (__builtin_amdgcn_ballot_w64(big)) {                                             // (seldom: a dozen events on one stream before this one)
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    if (__builtin_amdgcn_ballot_w64(n[r] >= (uint32_t)PART_LT)) {
                        out[r] = lcg_mul_dbl(out[r], lt1[(n[r] >> 8) & (PART_LT - 1)]);
                        const uint32_t h = n[r] >> 16;
                        if (h) out[r] = h < PART_LT ? lcg_mul_dbl(out[r], lt2[h]) : lcg_mul(out[r], lcg_jump2(pw, h << 16));
                    }
                }
            }
        }
#if HAND_ABL == 2      /* timing-only ablation: no stores */
        if (out[0] == 0x12345u) out_p[0] = out[1] ^ out[NR - 1];
#elif defined(SQG_ABL_HANDOVER)   /* timing-only ablation (round 5, results wrong): 16 B per slot -- the state with the stream's pore-table row -- as the
                                     hand-over of round 4's review would write them (the row itself is NOT looked up: what is priced is the traffic) */
#pragma unroll
        for (int r = 0; r < NR; r++) if (FULL || (uint32_t)(64 * r + lane) < left) reinterpret_cast<uint4*>(state_out)[(size_t)(out_p - state_out) + 64 * r] = make_uint4(out[r], cur[r], n[r], out[r] ^ cur[r]);
#else
#pragma unroll
        for (int r = 0; r < NR; r++) if (FULL || (uint32_t)(64 * r + lane) < left) out_p[64 * r] = out[r];
#endif
#pragma unroll
        for (int r = 0; r < NR; r++) cur[r] = nxt[r];
        in += HAND_STEP; out_p += HAND_STEP;
    };
    uint32_t b = lo;
    if (b + HAND_STEP <= hi) { step(std::true_type{}, std::true_type{}, 0u); b += HAND_STEP; }
    else if (b < hi) { step(std::false_type{}, std::true_type{}, hi - b); b = hi; }
    for (; b + HAND_STEP <= hi; b += HAND_STEP) step(std::true_type{}, std::false_type{}, 0u);
    if (b < hi) step(std::false_type{}, std::false_type{}, hi - b);
}

__global__ __launch_bounds__(64) void k_part_hand_ord(const uint32_t* __restrict__ part, uint32_t* __restrict__ state_out,
                                                      const uint32_t* __restrict__ slice_lo, const uint32_t* __restrict__ slice_hi,
                                                      const uint32_t* __restrict__ n_slices,
                                                      const uint32_t* __restrict__ phist, const uint32_t* __restrict__ pw,
                                                      unsigned int* __restrict__ err, const int fault) {
    __shared__ HandLds H;
    const int lane = threadIdx.x;
    if (blockIdx.x >= *n_slices) return;
    hand_tables(H, pw, lane, 64);
    hand_slice(H, part, state_out, slice_lo[blockIdx.x], slice_hi[blockIdx.x], phist + (size_t)blockIdx.x * PART_SUB, pw, err, fault, lane);
}

// Are the lanes of one LDS atomic that meet on an address served in ascending lane order, and successive instructions in
// program order?  Every wavefront plays rounds of 16 fetch-adds per lane against a table of the production size (4096 entries:
// the sub-row of k_part_hand_ord) with 16-bit addends (a bucketed event's dwell), addresses from one per lane to one for all
// (1 .. 4096 distinct ones) and some lanes idle, and compares what each returns with the sum over the earlier events
// (instruction, lane) of the round on the same address.  bad <- number of mismatches.  grid: any, 64 threads.
__global__ __launch_bounds__(64) void k_lds_order_check(const int rounds, unsigned int* __restrict__ bad) {
    __shared__ uint32_t tab[PART_SUB];
    __shared__ uint4 ev[16 * 64 / 4];                             // the round's events in order: address << 16 | value (0: idle lane)
    const int lane = threadIdx.x;
    uint32_t s = (uint32_t)(blockIdx.x * 64 + lane) * 2654435761u + 12345u;
    unsigned int wrong = 0;
    for (int it = 0; it < rounds; it++) {
        for (int i = lane; i < PART_SUB; i += 64) tab[i] = 0u;
        const uint32_t spread = 1u << ((it + blockIdx.x) % 13);     // 1 .. 4096 distinct addresses
        uint32_t a[16], v[16], got[16];
        bool on[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            s = s * 1664525u + 1013904223u;
            a[r] = (s >> 18) & (spread - 1);
            v[r] = ((s >> 2) & 0xffffu) | 1u;                       // 1 .. 65535
            on[r] = ((s >> 3) & 15u) != 0u || (it & 1);             // odd rounds: every lane; even: one in 16 idle
            reinterpret_cast<uint32_t*>(ev)[64 * r + lane] = a[r] << 16 | (on[r] ? v[r] : 0u);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) {
            got[r] = 0;
            if (on[r]) got[r] = __hip_atomic_fetch_add(&tab[a[r]], v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
        }
        __syncthreads();
#pragma unroll 1
        for (int r = 0; r < 16; r++) {
            uint32_t want = 0;
            const int me = 64 * r + lane;
            for (int e0 = 0; e0 < me; e0 += 16) {
                uint4 x[4];
#pragma unroll
                for (int i = 0; i < 4; i++) x[i] = ev[e0 / 4 + i];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t w[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
#pragma unroll
                    for (int j = 0; j < 4; j++) if (e0 + 4 * i + j < me && (w[j] >> 16) == a[r]) want += w[j] & 0xffffu;
                }
            }
            if (on[r] && got[r] != want) wrong++;
        }
        __syncthreads();
    }
    if (wrong) atomicAdd(bad, wrong);
}
