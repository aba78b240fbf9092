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
//   k_part_offsets        per worker chain: pcnt -> first slot of every (link, partition) in part[], and the slices
//                         [lo, hi) of part[] that belong to each (group of links, partition)
//   k_events<PART>        per link: every event to its slot, {dwell, low 12 bits of the rank}; evrec = {slot, rank}
//   k_part_hist           per (group, partition): samples per stream over the slice                  -> phist
//   k_part_scan           per (worker chain, rank): exclusive scan over the chain's groups on top of the worker's row:
//                         phist = the sub-row as the group finds it; the worker's row moves past the batch
//   k_part_hand           per (group, partition): the slice in order against the sub-row in LDS: prior[slot] = samples the
//                         event's stream has produced before it
//   k_part_home           per 64-event tile: evrec.x = seed(worker, rank) * a^(2 * prior[slot])
//
// Everything between the two k_events passes and k_part_home streams through memory sequentially (4 B per event and pass);
// the scatter and the gather move runs of a (512-event segment, partition), which consecutive segments of a link extend.
#pragma once

// grid: worker chains; 1024 threads = n_part partitions x (1024 / n_part) runs of consecutive links.
//   wlink_off[q]..wlink_off[q+1]   the links of worker chain q, in chain order
//   link_group[l]                  the group a link belongs to (consecutive links of a chain, ascending)
//   cbase[q]                       first slot of the chain's region in part[]
__global__ __launch_bounds__(1024) void k_part_offsets(uint32_t* __restrict__ pcnt, const int n_part, const int* __restrict__ wlink_off,
                                                       const int* __restrict__ link_group, const uint32_t* __restrict__ cbase,
                                                       uint32_t* __restrict__ slice_lo, uint32_t* __restrict__ slice_hi) {
    __shared__ uint32_t sums[1024];
    __shared__ uint32_t pstart[PART_MAX + 1], ptot[PART_MAX];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int p = tid % n_part, sg = tid / n_part, nsg = 1024 / n_part;
    const int l0 = wlink_off[q], l1 = wlink_off[q + 1];
    const int per = (l1 - l0 + nsg - 1) / nsg, la = min(l0 + sg * per, l1), lb = min(la + per, l1);
    uint32_t sum = 0;
    for (int l = la; l < lb; l++) sum += pcnt[(size_t)l * n_part + p];
    sums[tid] = sum;
    __syncthreads();
    if (tid < n_part) { uint32_t t = 0; for (int s = 0; s < nsg; s++) t += sums[s * n_part + tid]; ptot[tid] = t; }
    __syncthreads();
    if (tid == 0) {                                               // the chain's region: partition-major
        uint32_t run = cbase[q];
        for (int pp = 0; pp < n_part; pp++) { pstart[pp] = run; run += ptot[pp]; }
        pstart[n_part] = run;
    }
    __syncthreads();
    uint32_t at = pstart[p];
    for (int s = 0; s < sg; s++) at += sums[s * n_part + p];
    for (int l = la; l < lb; l++) {
        const uint32_t cnt = pcnt[(size_t)l * n_part + p];
        pcnt[(size_t)l * n_part + p] = at;
        const int g = link_group[l];
        if (l == l0 || link_group[l - 1] != g) {                  // the link opens a group
            slice_lo[(size_t)g * n_part + p] = at;
            if (l != l0) slice_hi[(size_t)(g - 1) * n_part + p] = at;
        }
        if (l == l1 - 1) slice_hi[(size_t)g * n_part + p] = pstart[p + 1];
        at += cnt;
    }
}

// grid: groups x partitions (partition fastest), 256 threads.  phist[g][rank] <- samples the group's events draw from the stream
__global__ __launch_bounds__(256) void k_part_hist(const uint32_t* __restrict__ part, const uint32_t* __restrict__ slice_lo,
                                                   const uint32_t* __restrict__ slice_hi, uint32_t* __restrict__ phist) {
    __shared__ uint32_t row[PART_SUB];
    const int tid = threadIdx.x;
    for (int i = tid; i < PART_SUB; i += 256) row[i] = 0u;
    __syncthreads();
    const uint32_t lo = slice_lo[blockIdx.x], hi = slice_hi[blockIdx.x];
    for (uint32_t i = lo + tid; i < hi; i += 256) {
        const uint32_t rec = part[i];
        atomicAdd(&row[rec & (PART_SUB - 1)], rec >> 16);
    }
    __syncthreads();
    uint32_t* dst = phist + (size_t)blockIdx.x * PART_SUB;       // [g][p][sub] = [g][rank]
    for (int i = tid; i < PART_SUB; i += 256) dst[i] = row[i];
}

// One thread per (worker chain, rank): exclusive scan of phist over the chain's groups, starting from the worker's row
// (sample counts, reduced mod (M-1)/2: only that matters for a^(2n)) plus, with range sharding, what the ranges before this
// one draw (`before`; the row itself is then left to k_rows_advance).  grid (num_kmer / 256, worker chains).
__global__ __launch_bounds__(256) void k_part_scan(uint32_t* __restrict__ phist, uint32_t* __restrict__ rows, const int num_kmer,
                                                   const int* __restrict__ wgroup_off, const int* __restrict__ wlink_worker,
                                                   const uint32_t* __restrict__ before, unsigned int* __restrict__ err) {
    const int q = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= num_kmer) return;
    const size_t wj = (size_t)wlink_worker[q] * num_kmer + j;
    unsigned long long run = rows[wj] % LCG_ORD2;
    if (before) run += before[wj];
    const int g0 = wgroup_off[q], g1 = wgroup_off[q + 1];
    for (int g = g0; g < g1; g++) {
        uint32_t* cell = phist + (size_t)g * num_kmer + j;
        const uint32_t cnt = *cell;
        *cell = (uint32_t)run;
        run += cnt;
    }
    if (run > 0xffffffffull) atomicOr(err, 32u);                  // one stream asked for >= 2^32 samples by one batch
    if (!before) rows[wj] = (uint32_t)run;
}

// range sharding: samples this batch's local reads draw from each (worker, rank) stream (counts zeroed beforehand)
__global__ __launch_bounds__(256) void k_part_totals(const uint32_t* __restrict__ phist, const int num_kmer, const int* __restrict__ wgroup_off,
                                                     const int* __restrict__ wlink_worker, uint32_t* __restrict__ counts) {
    const int q = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= num_kmer) return;
    uint32_t sum = 0;
    for (int g = wgroup_off[q]; g < wgroup_off[q + 1]; g++) sum += phist[(size_t)g * num_kmer + j];
    counts[(size_t)wlink_worker[q] * num_kmer + j] = sum;
}

// grid: groups x partitions, 256 threads.  The slice is walked in order, 64 events per step and wavefront; wavefront v hands
// out the streams whose sub-rank has v in its top two bits (every wavefront reads the whole slice: 4 B per event from L2).
// In-order hand-out inside a step: the lanes of a stream queue on a tag (atomic min of the lane number, so the earliest event
// wins the round), the winner reads and advances the sub-row, the others go another round (rarely: 16 events over 1024 streams).
#define PART_TAGS 256
__global__ __launch_bounds__(256) void k_part_hand(const uint32_t* __restrict__ part, uint32_t* __restrict__ prior_out,
                                                   const uint32_t* __restrict__ slice_lo, const uint32_t* __restrict__ slice_hi,
                                                   const uint32_t* __restrict__ phist) {
    __shared__ uint32_t row[PART_SUB];
    __shared__ uint32_t tags[4][PART_TAGS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t* src = phist + (size_t)blockIdx.x * PART_SUB;
    for (int i = tid; i < PART_SUB; i += 256) row[i] = src[i];
    for (int i = tid; i < 4 * PART_TAGS; i += 256) (&tags[0][0])[i] = 0xffffffffu;
    __syncthreads();
    const uint32_t lo = slice_lo[blockIdx.x], hi = slice_hi[blockIdx.x];
    uint32_t* tg = tags[wid];
    for (uint32_t b = lo; b < hi; b += 64) {
        const uint32_t i = b + lane;
        const uint32_t rec = i < hi ? part[i] : 0u;
        const uint32_t sub = rec & (PART_SUB - 1), d = rec >> 16;
        bool pending = i < hi && (int)(sub >> (PART_SUB_BITS - 2)) == wid;
        const uint32_t h = sub & (PART_TAGS - 1);
        while (__builtin_amdgcn_ballot_w64(pending)) {
            if (pending) atomicMin(&tg[h], (uint32_t)lane);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (pending && __hip_atomic_load(&tg[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == (uint32_t)lane) {
                const uint32_t prior = row[sub];
                row[sub] = prior + d;
                prior_out[i] = prior;                               // (not in place: the other wavefronts still read part[i])
                __hip_atomic_store(&tg[h], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                pending = false;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
    }
}

// one wavefront per 64-event tile: evrec.x (the event's slot, left by k_events<PART>) -> the stream state at the event's first
// draw, seed(worker, rank) * a^(2 * samples before), the seed being (seed_w + rank) mod M (src/sim.c:249)
__global__ __launch_bounds__(256) void k_part_home(const SigParams P, const int n_tiles) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_tiles) return;
    const int r = P.tile_read[g];
    const ReadDesc rd = P.reads[r];
    const int e = (g - rd.tile_off) * 64 + lane;
    if (e >= rd.ne0 + rd.ne1) return;
    uint2* er = P.evrec + rd.ev_off + e;
    const uint2 v = *er;
    const uint32_t prior = P.part_prior[v.x];
    const uint32_t seed_w = (uint32_t)(((unsigned long long)P.seed_base + (unsigned long long)rd.worker * P.seed_step) % LCG_M);
    const unsigned long long sv = (unsigned long long)seed_w + v.y;
    uint32_t c = (uint32_t)(sv >= LCG_M ? sv - LCG_M : sv);
    if (prior) c = lcg_mul(c, lcg_jump2(P.pw, prior));
    er->x = c;
}
