// k_samples.h -- k_items, k_samples_lean (the hot kernel), the generic k_samples, the FP64 fix-up kernels, k_certify, k_store_probe
// Part of the device code of the per-read signal path; included through sqg_kernels.h (see there for the overview).
#pragma once

#ifndef SQG_ABL_FIXK
#define SQG_ABL_FIXK 0                     // timing-only ablations of k_fixup (1: counts only, 2: entries read, not processed)
#endif
#ifndef SQG_LEAN_ITEMS4
#define SQG_LEAN_ITEMS4 1                  // k_items prepares the descriptors of 256-event items as well: one scalar look-up per item instead of a dependent chain of three
                                           // (round 4, A/B in one call: k_samples_lean 2.43-2.46 -> 2.38-2.39 ms, step -0.08 ms with k_items' own 15 us included; 0: the chain)
#endif
#ifndef SQG_LEAN_NOBAR
#define SQG_LEAN_NOBAR 0                   // A/B (round 5): 1 -- the jump table reaches LDS without a workgroup barrier.  The prologue shrinks from 5600 to 3400
                                           // cycles per wavefront, more wavefronts of a SIMD are in their sample loops at once, a step of the loop
                                           // takes 543 instead of 490 cycles -- the loop is bound by VALU issue -- and the kernel 2.47 instead of 2.43 ms
#endif
#ifndef SQG_LEAN_NT
#define SQG_LEAN_NT 0                      // A/B: non-temporal sample stores
#endif
#ifndef SQG_LB_BPERM
#define SQG_LB_BPERM 1                     // evrec32: a partition's first slot through the lane crossbar (ds_bpermute) instead of LDS memory
                                           // (A/B: the LDS copy costs the sample kernel 4 %: a write, a fence, 1 KiB less LDS per workgroup)
#endif

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

#ifndef SQG_LEAN_XCD
#define SQG_LEAN_XCD 1                     // A/B: XCD-contiguous item map of k_samples_lean
#endif
#ifndef SQG_LEAN_PAIR
#define SQG_LEAN_PAIR 0                    // round 6: k_samples_lean with TWO consecutive samples per lane -- one aligned dword of the output per lane and step, the
                                           // arithmetic in v_pk_* fp32 (see "two consecutive samples per lane" in the kernel)
#endif
#ifndef SQG_PAIR_ABL
#define SQG_PAIR_ABL 0                     // timing-only ablations of the pair loop (results are wrong): 1 no drain in front of the last phase, 2 no last phase, 3 no parking
#endif
#define PAIR_TB 36                         // steps of 64 pairs an item of <= LEAN_MAX_SAMPLES samples has (33) + the entries read ahead
typedef float f2_t __attribute__((ext_vector_type(2)));
#define LEAN_EPL_MAX 4                     // events per lane of the lean kernel: 4, 2 or 1 (SigParams.lean_epl, chosen per profile so
                                           // that a work item -- 64*epl consecutive events of a read -- stays below LEAN_MAX_SAMPLES)
#define FIX_SLOTS 8                        // parked undecided samples per super tile (expected ~0.5); overflow -> global list
#define LEAN_MAX_SAMPLES 4096              // samples per work item the 64x64-bit start map covers

// k_items: one thread per super tile, for profiles whose items are short (LEAN_EPL < 4; with 4 events per lane the lean
// kernel does this itself on the scalar unit).  Collapses the dependent look-ups of the lean kernel's set-up
// (tile -> read -> tile_so / sig_off / seglen) into one record per item and decides which items the lean
// kernel takes; the others are queued (as 64-event tiles) for k_samples<MODE, GENERIC>.
// (also run as extra workgroups of k_part_hist: block = index among the items' workgroups)
__device__ static inline void items_body(const SigParams& P, const int block, const int n_stiles) {
    const int g = block * 256 + threadIdx.x;
    if (g >= n_stiles) return;
    const int r = P.stile_read[g];
    const ReadDesc rd = P.reads[r];
    const int LEAN_EPL = P.lean_epl, LEAN_EV = 64 * LEAN_EPL;
    const int lt = g - rd.stile_off;                                   // super tile within the read
    const int ne = rd.ne0 + rd.ne1;
    const int n_ev = min(LEAN_EV, ne - lt * LEAN_EV);
    const long long sig_base = P.sig_off[r];
    const long long read_len64 = P.sig_off[r + 1] - sig_base;
    const uint32_t read_len = (uint32_t)read_len64;
    const uint32_t base_pos = P.tile_so[rd.tile_off + lt * LEAN_EPL];
    const uint32_t next_pos = (lt + 1) * LEAN_EV < ne ? P.tile_so[rd.tile_off + (lt + 1) * LEAN_EPL] : read_len;
    const int n_samples = (int)(next_pos - base_pos);
    bool take = rd.fast != 0 && n_samples <= LEAN_MAX_SAMPLES && n_samples >= 0 && read_len64 < 4294967295LL;   // (a read of >= UINT32_MAX samples fails the batch, src/sim.c:559-562: k_scan; nobody writes it)
    int shift_lo = 0, shift_hi = 0;
    if (P.shift_len > 0) {                                             // RNA adaptor level-shift window (src/genread.c:79-86)
        const long long n1 = (long long)P.seglen[2 * r];
        if ((long long)base_pos + n_samples > n1 - P.shift_len && (long long)base_pos < n1) {
            shift_lo = (int)max(n1 - P.shift_len - (long long)base_pos, 0LL);
            shift_hi = (int)min(n1 - (long long)base_pos, (long long)n_samples);
        }
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
    d.shift_lo = shift_lo; d.shift_hi = shift_hi;
    d.slot_first = P.one ? rd.slot0 + lt * LEAN_EV : P.evrec32 ? (rd.slot0 >= 0 ? rd.slot0 : P.tile_link[rd.tile_off + lt * LEAN_EPL]) : 0;
    P.items[g] = d;
}
__global__ __launch_bounds__(256) void k_items(const SigParams P, const int n_stiles) { items_body(P, (int)blockIdx.x, n_stiles); }

// load through the constant address space: for a wave-uniform address this is a scalar load (the arrays read this way
// were written by earlier kernels of the stream)
template <typename T>
__device__ static inline T sload(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "dword granularity");
    const __attribute__((address_space(4))) uint32_t* src =
        reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(p));
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (int q = 0; q < (int)(sizeof(T) / 4); q++) w[q] = src[q];
    T v;
    __builtin_memcpy(&v, w, sizeof v);
    return v;
}

// (timing-only build, -DSQG_LEAN_TRACE=1) where an item's time goes inside k_samples_lean: shader-clock stamps per wavefront at the
// item's phases, summed per phase into a device array the context prints when it is destroyed (tools/runs/r5k.sh):
//   [0] items  [1] steps  [2] draining the previous item's stores  [3] first-level loads (event records, dwells, partition bases)
//   [4] second-level gathers (pore-table rows, stream states)  [5] tables (FP64 per-event constants, start map)  [6] the sample loop
//   [7] the item's end (parked samples)  [8] between items (the descriptor's scalar loads)  [9] the workgroup's prologue (jump table -> LDS, barrier)
#if defined(SQG_LEAN_TRACE)
#define LEAN_TRACE_SHARDS 4096
__device__ unsigned long long g_lean_trace[LEAN_TRACE_SHARDS * 16];      // (a row of 16 words per shard, 128 B: atomics on ONE line serialise the kernel)
#define LEAN_T(var_) const unsigned long long var_ = __builtin_amdgcn_s_memtime()
#define LEAN_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#if SQG_LEAN_TRACE == 1
#define LEAN_DRAIN0() LEAN_DRAIN()                                    /* 1: the stores are drained BEFORE the item's first loads are issued (what they cost alone) */
#else
#define LEAN_DRAIN0()                                                 /* 2: as in production, the first wait for a load is what waits for them */
#endif
#else
#define LEAN_T(var_)
#define LEAN_DRAIN()
#define LEAN_DRAIN0()
#endif
template <int EPL>
struct LeanWaveLds {
    uint4 rec[64 * EPL];                // {c_ev, (4*first sample) << 16 | I (16 bits), F - 1/2, sdk}
#if SQG_LEAN_PAIR
    uint4 tb[PAIR_TB];                  // per step of 64 PAIRS: {x,y: bit p-1-64c set: the event index goes up at pair p; z: the index before the step; w: tb2 != 0}
#else
    uint4 tb[66];                       // per 64-sample step c: {x,y: bit s-1-64c set: an event (other than the item's first) starts at
                                        // sample s; z: events begun before the step; w: 0}; entries 64, 65 are read ahead, never used
#endif
    int nfix;                           // undecided samples of the item so far
    int pad[3];
    uint4 park[FIX_SLOTS];              // ... parked here {index in read, c1, event in read, in the level-shift window} until the item is done
#if SQG_LEAN_PAIR
    uint2 tb2[PAIR_TB];                 // per step of 64 pairs: the pairs at which the event index advances by TWO (a one-sample event in between)
#endif
#if !SQG_LB_BPERM
    uint32_t lb[PART_MAX];              // evrec32: first slot of every partition of the item's link
#endif
};
template <int EPL>
struct LeanLds {
    uint32_t mult[MULT_N];              // 2 * a^(2j+1): the first draw of an event's sample j is lcg_mul_dbl(state, mult[j])
    LeanWaveLds<EPL> w[4];
};

// k_samples_lean: the hot kernel.  Certified fp32 path only, for reads whose ADC values are provably in
// (2, 65000) (ReadDesc.fast), events of <= MULT_N samples (the RNA level-shift window included); everything
// else is queued (as 64-event tiles) for k_samples<MODE, GENERIC>.
// One wavefront per 256 consecutive events of a read (4 per lane: the dependent global round trips of the
// set-up are paid once per ~2300 samples; the item's descriptors are wave-uniform and live in SGPRs).
// Per step 64 consecutive samples:
//   the step's entry of the event-start table (broadcast LDS read, a step ahead) -> mbcnt -> event -> {state, first|I, F-1/2, sdk}
//   (one ds_read_b128) -> jump constant a^(2j+1) (ds_read_b32) -> one modular multiplication; second uniform =
//   fract(c1 * a/M) in FP64 -> v_log/v_sqrt/v_cos ->
//   v' = fma(x, sdk, F-1/2) -> t = v' + 1.5*2^23 (round to nearest: floor of the ADC value unless it is within
//   eps of an integer) -> acceptance test on v' - (t - 1.5*2^23) -> int16 store of the low half of bits(t) + I.
// The loads of step i+1 are issued before the arithmetic of step i (software pipelining; four steps unrolled: the
// pipeline registers alternate instead of being copied, and the position registers advance once per four steps).
template <bool RNA, int LEAN_EPL>
__global__ __launch_bounds__(256) void k_samples_lean(const SigParams P, const int n_stiles) {
    LEAN_T(tr_k0);
    __shared__ LeanLds<LEAN_EPL> L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#if SQG_LEAN_NOBAR
    // The jump table (2 KiB, doubled: lcg_mul_dbl) without a workgroup barrier (round 5): EVERY wavefront requests the whole table here, in
    // front of its item's own loads, and writes it to LDS once those have come back (a wavefront's loads return in order) -- the four copies
    // are the same bytes at the same addresses, and a wavefront's own LDS writes are visible to it in program order.  The load + barrier
    // in front of everything is 12 % of a wavefront's life (profiles/r05_lean_trace.md): 5600 cycles before its item's first load is issued.
    uint32_t mt_[MULT_N / 64];
#pragma unroll
    for (int q = 0; q < MULT_N / 64; q++) mt_[q] = P.pw[64 * q + lane];
    bool mult_filled = false;
#else
    for (int i = tid; i < MULT_N; i += 256) L.mult[i] = P.pw[i] << 1;        // doubled: lcg_mul_dbl
    __syncthreads();
#endif
    LeanWaveLds<LEAN_EPL>& W = L.w[wid];
    const float thr = P.thr_all;
    const char* mult_b = reinterpret_cast<const char*>(L.mult);

    // workgroup -> items: consecutive workgroup ids are dispatched to different XCDs (id % 8), each with an L2 of its own, while
    // consecutive ITEMS share lines of state[] / part[] (a (link, partition)'s run of slots spans ~4 items of a read): XCD x takes
    // the x-th eighth of the items, so that the neighbours' lines are hits in ITS L2 instead of a second fetch by another one
    unsigned wg0 = blockIdx.x;
#if SQG_LEAN_XCD
    {
        const unsigned q = gridDim.x >> 3, rem = gridDim.x & 7u, x = blockIdx.x & 7u;
        wg0 = x * q + min(x, rem) + (blockIdx.x >> 3);
    }
#endif
#if defined(SQG_LEAN_TRACE)
    unsigned long long tr_prev = 0;
#endif
    for (int g = (int)wg0 * 4 + wid; g < n_stiles; g += gridDim.x * 4) {
        // the item's descriptor -- what the read, its 64-event tiles and the scanned read offsets say about this item -- is
        // wave-uniform: the whole look-up chain runs on the scalar unit (sload: constant-address-space loads)
        ItemDesc it;
        if constexpr (LEAN_EPL < 4 || SQG_LEAN_ITEMS4) {               // short items (1-2 events per lane): k_items prepared the descriptors
            it = sload(P.items + g);
            if (it.n_ev == 0) continue;                                // not taken, or empty
        } else {
            const int r = sload(P.stile_read + g);
            const ReadDesc rd = sload(P.reads + r);
            constexpr int LEAN_EV = 64 * LEAN_EPL;
            const int lt = g - rd.stile_off;                           // item within the read
            const int ne_read = rd.ne0 + rd.ne1;
            const int n_ev = min(LEAN_EV, ne_read - lt * LEAN_EV);
            const long long sig_base = sload(P.sig_off + r);
            const long long read_len64 = sload(P.sig_off + r + 1) - sig_base;
            const uint32_t read_len = (uint32_t)read_len64;
            const uint32_t base_pos = sload(P.tile_so + rd.tile_off + lt * LEAN_EPL);
            const uint32_t next_pos = (lt + 1) * LEAN_EV < ne_read ? sload(P.tile_so + rd.tile_off + (lt + 1) * LEAN_EPL) : read_len;
            const int n_samples = (int)(next_pos - base_pos);
            const bool take = rd.fast != 0 && n_samples <= LEAN_MAX_SAMPLES && n_samples > 0 && read_len64 < 4294967295LL;
            if (!take) {                                               // leave these (up to LEAN_EPL) 64-event tiles to the generic kernel
                if (lane == 0) {
                    const int nt = (n_ev + 63) >> 6;
                    const unsigned int q = atomicAdd(P.slow_count, (unsigned int)nt);
                    for (int i = 0; i < nt; i++) P.slow_tiles[q + i] = rd.tile_off + lt * LEAN_EPL + i;
                }
                continue;
            }
            int shift_lo = 0, shift_hi = 0;
            if (RNA && P.shift_len > 0) {                              // RNA adaptor level-shift window (src/genread.c:79-86)
                const long long n1 = (long long)sload(P.seglen + 2 * r);
                if ((long long)base_pos + n_samples > n1 - P.shift_len && (long long)base_pos < n1) {
                    shift_lo = (int)max(n1 - P.shift_len - (long long)base_pos, 0LL);
                    shift_hi = (int)min(n1 - (long long)base_pos, (long long)n_samples);
                }
            }
            it.ev_first = rd.ev_off + (long long)lt * LEAN_EV;
            it.sig_base = sig_base;
            it.offset = rd.offset;
            it.n_ev = n_ev;
            it.n_samples = n_samples;
            it.at0 = RNA ? read_len - 1u - base_pos : base_pos;
            it.ev_read0 = lt * LEAN_EV;
            it.read = r;
            it.shift_lo = shift_lo; it.shift_hi = shift_hi;
            // (the item's link: the read's, or -- a read cut into pieces -- what the scatter pass noted for the tile)
            it.slot_first = P.one ? rd.slot0 + lt * LEAN_EV : P.evrec32 ? (rd.slot0 >= 0 ? rd.slot0 : sload(P.tile_link + rd.tile_off + lt * LEAN_EPL)) : 0;
        }
        const int ne = it.n_ev;                                         // events of this item
        if (ne == 0) continue;                                         // not taken, or empty
        const int wave_total = it.n_samples;
        LEAN_T(tr_a); LEAN_DRAIN0(); LEAN_T(tr_b);                     // (trace build 1: the previous item's stores have retired)
        const int e0 = lane * LEAN_EPL;                                // my first event (within the item)
        const long long gev = it.ev_first + e0;
        // ---- set-up: LEAN_EPL consecutive events per lane ----
        uint2 er[LEAN_EPL];
        int sps[LEAN_EPL];
        if (P.one) {
            // (wave-uniform) one-partition hand-out: the item's events are consecutive slots: rank and dwell from part[], states below
            const uint32_t gslot = (uint32_t)it.slot_first + (uint32_t)e0;
            if (e0 + LEAN_EPL <= ne) {
                uint32_t pw_[LEAN_EPL];
                __builtin_memcpy(pw_, P.part + gslot, 4 * LEAN_EPL);
#pragma unroll
                for (int q = 0; q < LEAN_EPL; q++) { er[q] = make_uint2(gslot + (uint32_t)q, pw_[q] & 0xffffu); sps[q] = (int)(pw_[q] >> 16); }
            } else {
#pragma unroll
                for (int q = 0; q < LEAN_EPL; q++) {
                    const uint32_t w_ = e0 + q < ne ? P.part[gslot + q] : 0u;
                    er[q] = make_uint2(gslot + (uint32_t)q, w_ & 0xffffu); sps[q] = (int)(w_ >> 16);
                }
            }
        } else if (P.evrec32) {
            // (wave-uniform) bucketed hand-out, 4-B event records {rank, slot within the (link, partition)}: the slot is the first slot of the
            // (link, partition) -- 64 words per link, staged through LDS -- plus the record's low bits
#if SQG_LB_BPERM
            const uint32_t lbv = P.lbase[(size_t)it.slot_first * PART_MAX + lane];      // lane p: partition p's first slot
#else
            W.lb[lane] = P.lbase[(size_t)it.slot_first * PART_MAX + lane];
#endif
            uint32_t ew[LEAN_EPL];
            uint16_t dw[LEAN_EPL];
            __builtin_memcpy(ew, P.evrec32 + gev, 4 * LEAN_EPL);       // (4-B / 2-B aligned wide loads; both arrays end with slack)
            if (P.dwell) __builtin_memcpy(dw, P.dwell + gev, 2 * LEAN_EPL);
            else {
#pragma unroll
                for (int q = 0; q < LEAN_EPL; q++) dw[q] = (uint16_t)P.const_sps;
            }
#if !SQG_LB_BPERM
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // W.lb is written
#endif
#pragma unroll
            for (int q = 0; q < LEAN_EPL; q++) {
                const bool v = e0 + q < ne;
                const uint32_t rank = v ? ew[q] >> EVR_REL_BITS : 0u;
#if SQG_LB_BPERM
                const uint32_t lb_p = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rank >> PART_SUB_BITS) << 2), (int)lbv);   // (lane crossbar: no LDS memory)
#else
                const uint32_t lb_p = W.lb[rank >> PART_SUB_BITS];
#endif
                er[q] = make_uint2(v ? lb_p + (ew[q] & ((1u << EVR_REL_BITS) - 1u)) : 0u, rank);
                sps[q] = v ? (int)dw[q] : 0;
            }
        } else if (e0 + LEAN_EPL <= ne) {
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
        LEAN_DRAIN(); LEAN_T(tr_c);
#if SQG_LEAN_NOBAR
        if (!mult_filled) {                                            // (wave-uniform; the first-level loads above were requested behind the table's)
#pragma unroll
            for (int q = 0; q < MULT_N / 64; q++) L.mult[64 * q + lane] = mt_[q] << 1;
            mult_filled = true;
        }
#endif
        float2 md[LEAN_EPL];
#if defined(SQG_ABL_NODEP)       /* timing-only ablation (results are wrong): the set-up's second-level look-ups do not depend on the first */
#pragma unroll
        for (int q = 0; q < LEAN_EPL; q++) md[q] = (e0 + q < ne) ? P.model[(gev + q) & 0x3ffff] : make_float2(0.f, 0.f);
        if (P.part_state) {
            uint32_t stv[LEAN_EPL];
#pragma unroll
            for (int q = 0; q < LEAN_EPL; q++) stv[q] = P.part_state[gev + q];
#pragma unroll
            for (int q = 0; q < LEAN_EPL; q++) if (e0 + q < ne) er[q].x = (stv[q] + er[q].x) | 1u;
        }
#elif defined(SQG_ABL_HANDOVER)  /* timing-only ablation (round 5, results wrong): ONE 16-B gather per event -- state and pore-table row handed over by the
                                    hand-out kernel -- instead of the row's look-up and the 4-B state gather */
        if (P.part_state && !P.one) {
#pragma unroll
            for (int q = 0; q < LEAN_EPL; q++) {
                uint4 sv = make_uint4(1u, 0u, 0u, 0u);
                if (e0 + q < ne) sv = reinterpret_cast<const uint4*>(P.part_state)[er[q].x];
                er[q].x = sv.x;
                md[q] = (e0 + q < ne) ? make_float2(90.0f + (float)(sv.y & 31u), 1.5f + 0.125f * (float)(sv.z & 7u)) : make_float2(0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int q = 0; q < LEAN_EPL; q++) md[q] = (e0 + q < ne) ? P.model[er[q].y] : make_float2(0.f, 0.f);
            if (P.part_state) {
#pragma unroll
                for (int q = 0; q < LEAN_EPL; q++) if (e0 + q < ne) er[q].x = P.part_state[er[q].x];
            }
        }
#else
#pragma unroll
        for (int q = 0; q < LEAN_EPL; q++) md[q] = (e0 + q < ne) ? P.model[er[q].y] : make_float2(0.f, 0.f);
        if (P.part_state) {                                            // (wave-uniform) k > 6, bucketed hand-out (k_part.h): slot -> state
#pragma unroll
            for (int q = 0; q < LEAN_EPL; q++) if (e0 + q < ne) er[q].x = P.part_state[er[q].x];
        }
#endif
        LEAN_DRAIN(); LEAN_T(tr_d);
        int lane_total = 0;
#pragma unroll
        for (int q = 0; q < LEAN_EPL; q++) lane_total += sps[q];
        const int incl = wave_incl_scan_dpp(lane_total);
#if SQG_LEAN_PAIR
        // ---- two consecutive samples per lane (round 6) ----
        // A lane owns one ALIGNED dword of the output per step: pair p = 64 c + lane holds the samples g0 = 2 p - sh and g1 = g0 + 1 of the item
        // (generation order; sh: whether the item's first sample sits in the second half of its dword -- then pair 0's first half is not
        // the item's).  Both samples are computed from the record of the event g0 belongs to: one sample -> event look-up, one record, one
        // ds_read2 of the jump table, the fp32 arithmetic in v_pk_* (two values per 4-cycle pass), one global_store_dword.  Where an event
        // STARTS at g1 the second half is wrong (it is the "sample sps" of the event before): the item's last phase recomputes the first sample
        // of every event that starts on a second half and writes it over, behind an s_waitcnt vmcnt(0) (a wavefront's stores to one
        // address are then in order).  rec[] here is {F - 1/2, state, sdk, (4 * first sample) << 16 | I}: F and sdk each in the low
        // half of a 64-bit register pair (v_pk_fma's broadcast form), no copy.
        static_assert(SQG_NEARONE == 0 && SQG_U2 == 1, "the pair loop restates box_muller_fast's default form");
        unsigned long long dbl_steps;
        const uint32_t H_ = (uint32_t)it.sig_base + it.at0;            // (low bits of) the index in sig[] of the item's first sample
        const int sh = RNA ? (int)(~H_ & 1u) : (int)(H_ & 1u);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");         // previous item's LDS reads are done
        if (lane < PAIR_TB) { W.tb[lane] = make_uint4(0u, 0u, 0u, 0u); W.tb2[lane] = make_uint2(0u, 0u); }
        if (lane == 0) W.nfix = 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        {
            int run = incl - lane_total;
#pragma unroll
            for (int q = 0; q < LEAN_EPL; q++) {
                const int so = run; run += sps[q];
                const double mk = (double)md[q].x * P.kd - it.offset;
                const double fl0 = floor(mk);
                const float Fh = (float)(mk - fl0 - 0.5);
                const float sdk = (float)((double)md[q].y * P.kd);
                W.rec[lane * LEAN_EPL + q] = make_uint4(__float_as_uint(Fh), er[q].x, __float_as_uint(sdk),
                                                        ((uint32_t)so << 18) | ((uint32_t)(int)fl0 & 0xffffu));     // so < 4096
                if ((e0 + q < ne) && (lane | q) != 0) {
                    // the first pair whose g0 is in this event or behind it: there the event index of the pairs goes up by one -- by two
                    // when a one-sample event sits on the second half in front (it is nobody's g0): the second mask
                    const uint32_t bit = (uint32_t)((so + sh + 1) >> 1) - 1u, m_ = 1u << (bit & 31u);
                    const uint32_t old_ = atomicOr(reinterpret_cast<unsigned int*>(W.tb) + ((bit >> 6) << 2) + ((bit >> 5) & 1u), m_);
                    if (old_ & m_) atomicOr(reinterpret_cast<unsigned int*>(W.tb2) + ((bit >> 6) << 1) + ((bit >> 5) & 1u), m_);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        {   // the event index before each step
            const uint4 mine = lane < PAIR_TB ? W.tb[lane] : make_uint4(0u, 0u, 0u, 0u);
            const uint2 m2 = lane < PAIR_TB ? W.tb2[lane] : make_uint2(0u, 0u);
            const int pc = __builtin_popcount(mine.x) + __builtin_popcount(mine.y) + __builtin_popcount(m2.x) + __builtin_popcount(m2.y);
            const uint32_t z_ = (uint32_t)(wave_incl_scan_dpp(pc) - pc);
            if (lane < PAIR_TB) W.tb[lane].z = z_;
            dbl_steps = __builtin_amdgcn_ballot_w64((m2.x | m2.y) != 0u);       // (bit c: step c has a pair at which the index goes up by two -- an SGPR pair: the loop's test is scalar)
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        char* const out_b = reinterpret_cast<char*>(P.sig + it.sig_base);               // wave-uniform: global_store saddr + 32-bit lane offset
        const int n_s = wave_total;
        const int np_ = (n_s + sh + 1) >> 1;                           // pairs of the item
        // byte offset (from out_b) of my dword of step 0: DNA 2 (at0 - sh) + 4 p, g0 in the low half; RNA 2 (at0 + sh - 1) - 4 p, g0 in the high half
        uint32_t voff = RNA ? 2u * (it.at0 + (uint32_t)sh - 1u) - 4u * (uint32_t)lane : 2u * (it.at0 - (uint32_t)sh) + 4u * (uint32_t)lane;
        uint32_t idx4 = 4u * (2u * (uint32_t)lane - (uint32_t)sh);    // 4 * g0 of the current group's first step
        const uint32_t near_lim = LCG_M - (1u << NEAR_ONE_BITS);
        #define PAIR_EVOF(tq_, t2p_, step_) ({ int e__ = (int)__builtin_amdgcn_mbcnt_hi((tq_).y, __builtin_amdgcn_mbcnt_lo((tq_).x, (tq_).z));       \
            if ((dbl_steps >> (step_)) & 1ull) { const uint2 m2__ = *(t2p_);                                                                      \
                e__ = (int)__builtin_amdgcn_mbcnt_hi(m2__.y, __builtin_amdgcn_mbcnt_lo(m2__.x, (uint32_t)e__)); }                                \
            e__; })
        // an undecided sample (g_, c1_) of event EV_ joins the item's parked samples (as in the one-sample loop)
        #define PAIR_PARK(cond_, g_, c1_, shf_, EV_) if (cond_) {                                                                                 \
                const unsigned long long am = __builtin_amdgcn_ballot_w64(true);                                                                  \
                const int n0 = W.nfix;                                                                                                            \
                const int slot = n0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));          \
                const uint32_t at_ = RNA ? it.at0 - (uint32_t)(g_) : it.at0 + (uint32_t)(g_);                                                     \
                if (slot < FIX_SLOTS) W.park[slot] = make_uint4(at_, c1_, (uint32_t)(EV_), (shf_) ? 1u : 0u);                                      \
                else push_fix_one(P, it.sig_base + at_, c1_, it.ev_first + (EV_), it.read, (shf_) ? 1 : 0);                                       \
                if (slot + 1 == n0 + __popcll(am)) W.nfix = slot + 1;                                                                             \
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                                                            \
            }
        // one step of 64 pairs: the loads of the next step into (RN, N0, N1, EN) and the table entry behind it into TQN, then the arithmetic of
        // this one from (RA, M0, M1, EV).  DI: the step's place in its group of two (idx4, voff and the table pointers advance once per group).
        // EDGE: the item's first step when sh (pair 0's first half is not the item's) and its last, partial one.
        #define PAIR_STEP(SH, EDGE, DI, RA, M0, M1, EV, RN, N0, N1, EN, TQ, TQN) {                                                                 \
            EN = PAIR_EVOF(TQ, tb2p + 1 + (DI), c + 1 + (DI));                                                                                                \
            RN = W.rec[EN];                                                                                                                       \
            TQN = tbp[2 + (DI)];                                                                                                                  \
            const uint32_t c1a = lcg_mul_dbl(RA.y, M0), c1b = lcg_mul_dbl(RA.y, M1);                                                              \
            const f2_t uf0 = {(float)c1a, (float)c1b};                                                                                            \
            const f2_t uf = uf0 * 4.656612873077393e-10f;                                                                                         \
            const f2_t lg = {__builtin_amdgcn_logf(uf.x), __builtin_amdgcn_logf(uf.y)};                                                           \
            const f2_t yy = __builtin_elementwise_fma(lg, (f2_t)(-1.3862943611198906f), (f2_t)(-9.313225750491594e-10f));                          \
            const f2_t rr = {__builtin_amdgcn_sqrtf(yy.x), __builtin_amdgcn_sqrtf(yy.y)};                                                         \
            const f2_t pf = {(float)(int)(c1a * 16807u), (float)(int)(c1b * 16807u)};                                                             \
            const f2_t t2 = __builtin_elementwise_fma(pf, (f2_t)(4.656612873077393e-10f), uf * 7.826369259425611e-06f);                            \
            const f2_t cs = {__builtin_amdgcn_cosf(t2.x), __builtin_amdgcn_cosf(t2.y)};                                                           \
            const f2_t xx = rr * cs;                                                                                                              \
            const f2_t vh = __builtin_elementwise_fma(xx, (f2_t)(__uint_as_float(RA.z)), (f2_t)(__uint_as_float(RA.x)));                          \
            const f2_t tt = vh + LEAN_MAGIC;                                                                                                      \
            const f2_t dd = vh - (tt - LEAN_MAGIC);                                                                                               \
            const bool okp = fmaxf(fabsf(dd.x), fabsf(dd.y)) < thr && max(c1a, c1b) <= near_lim;                                                  \
            const int g0 = ((EDGE) || (SH)) ? ((int)idx4 >> 2) + 128 * (DI) : 2 * (64 * (c + (DI)) + lane) - sh;   /* (the second form: scalars + lane, only the rare branch uses it) */ \
            const bool shfa = (SH) && (uint32_t)(g0 - it.shift_lo) < (uint32_t)(it.shift_hi - it.shift_lo);                                       \
            const bool shfb = (SH) && (uint32_t)(g0 + 1 - it.shift_lo) < (uint32_t)(it.shift_hi - it.shift_lo);                                   \
            const uint32_t va = __float_as_uint(tt.x) + RA.w - (shfa ? (uint32_t)P.shift : 0u);                                                   \
            const uint32_t vb = __float_as_uint(tt.y) + RA.w - (shfb ? (uint32_t)P.shift : 0u);                                                   \
            const uint32_t pk = RNA ? __builtin_amdgcn_perm(va, vb, 0x05040100u) : __builtin_amdgcn_perm(vb, va, 0x05040100u);                    \
            char* const dst_b = out_b + (RNA ? -256 * (DI) : 256 * (DI));                                                                         \
            bool real_a = true, real_b = true;                                                                                                    \
            if (EDGE) {                                                                                                                           \
                real_a = g0 >= 0 && g0 < n_s; real_b = g0 >= 0 && g0 + 1 < n_s;                                                                   \
                if (real_b) *reinterpret_cast<uint32_t*>(dst_b + voff) = pk;                                                                      \
                else if (real_a) *reinterpret_cast<uint16_t*>(out_b + 2u * (RNA ? it.at0 - (uint32_t)g0 : it.at0 + (uint32_t)g0)) = (uint16_t)va; \
            } else *reinterpret_cast<uint32_t*>(dst_b + voff) = pk;                                                                               \
            if (SQG_PAIR_ABL != 3 && !okp && (real_a || real_b)) { /* ~2 % of the steps: which half, and is it mine */                            \
                const int ev_ = EV;                                                                                                               \
                const bool bad_a = real_a && !(fabsf(dd.x) < thr && c1a <= near_lim);                                                             \
                bool bad_b = real_b && !(fabsf(dd.y) < thr && c1b <= near_lim);                                                                   \
                if (bad_b && ev_ + 1 < ne && (int)(W.rec[ev_ + 1].w >> 18) <= g0 + 1) bad_b = false;   /* the next event's: the last phase makes it */ \
                PAIR_PARK(bad_a, g0, c1a, shfa, ev_)                                                                                              \
                PAIR_PARK(bad_b, g0 + 1, c1b, shfb, ev_)                                                                                          \
            }                                                                                                                                     \
            if ((DI) == 1) { idx4 += 1024u; voff = RNA ? voff - 512u : voff + 512u; tbp += 2; tb2p += 2; }                                        \
            { const uint32_t* mp_ = reinterpret_cast<const uint32_t*>(mult_b + ((DI) == 1 ? 0 : 512) + (idx4 - (RN.w >> 16)));                    \
              N0 = mp_[0]; N1 = mp_[1]; } }
        #define PAIR_B_TO_A { ra = rb; ma0 = mb0; ma1 = mb1; eva = evb; tqa = tqb; idx4 += 512u; voff = RNA ? voff - 256u : voff + 256u; tbp += 1; tb2p += 1; }
        LEAN_T(tr_e);
        uint4 ra, rb, tqa, tqb; uint32_t ma0, ma1, mb0, mb1; int eva, evb;
        const uint4* tbp = W.tb;                                        // table entry of the current group's first step
        const uint2* tb2p = W.tb2;
        {
            const uint4 tq0 = tbp[0];
            eva = PAIR_EVOF(tq0, tb2p, 0);
        }
        ra = W.rec[eva];
        tqa = tbp[1];
        { const uint32_t* mp_ = reinterpret_cast<const uint32_t*>(mult_b + (idx4 - (ra.w >> 16))); ma0 = mp_[0]; ma1 = mp_[1]; }
        int c = 0;                                                     // the current group's first step (scalar)
        const int nfull = (np_ - 1) >> 6, rem = np_ - 64 * nfull;      // the last step (1..64 pairs) always runs as EDGE: its last pair's second half may be the next item's
        #define PAIR_LOOP(SH)                                                                                       \
            if (sh && nfull > 0) { PAIR_STEP(SH, true, 0, ra, ma0, ma1, eva, rb, mb0, mb1, evb, tqa, tqb) PAIR_B_TO_A c = 1; }   \
            for (; c + 2 <= nfull; c += 2) {                                                                        \
                PAIR_STEP(SH, false, 0, ra, ma0, ma1, eva, rb, mb0, mb1, evb, tqa, tqb)                             \
                PAIR_STEP(SH, false, 1, rb, mb0, mb1, evb, ra, ma0, ma1, eva, tqb, tqa)                             \
            }                                                                                                       \
            if (c < nfull) { PAIR_STEP(SH, false, 0, ra, ma0, ma1, eva, rb, mb0, mb1, evb, tqa, tqb) PAIR_B_TO_A c++; }           \
            if (rem) PAIR_STEP(SH, true, 0, ra, ma0, ma1, eva, rb, mb0, mb1, evb, tqa, tqb)
        if (RNA && it.shift_hi > it.shift_lo) { PAIR_LOOP(true) } else { PAIR_LOOP(false) }
        #undef PAIR_LOOP
        #undef PAIR_STEP
        #undef PAIR_B_TO_A
        #undef PAIR_EVOF
        // the last phase: the first sample of every event that starts on the second half of a dword (for the item's first event: when sh) --
        // one sample as the one-sample loop makes it, a short store over the pair's second half once the dword stores have landed
#if SQG_PAIR_ABL != 1 && SQG_PAIR_ABL != 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll
        for (int q = 0; q < LEAN_EPL; q++) {
            const int e = lane + 64 * q;
            if (SQG_PAIR_ABL != 2 && e < ne) {
                const uint4 r_ = W.rec[e];
                const int so = (int)(r_.w >> 18);
                if ((so + sh) & 1) {
                    const uint32_t c1 = lcg_mul_dbl(r_.y, L.mult[0]);
                    const float x = box_muller_fast(c1);
                    const float vh = __builtin_fmaf(x, __uint_as_float(r_.z), __uint_as_float(r_.x));
                    const float t = vh + LEAN_MAGIC;
                    const float d = vh - (t - LEAN_MAGIC);
                    const bool shf = RNA && (uint32_t)(so - it.shift_lo) < (uint32_t)(it.shift_hi - it.shift_lo);
                    const uint32_t at_ = RNA ? it.at0 - (uint32_t)so : it.at0 + (uint32_t)so;
                    if (fabsf(d) < thr && c1 <= near_lim)
                        *reinterpret_cast<uint16_t*>(out_b + 2u * at_) = (uint16_t)((__float_as_uint(t) + r_.w - (shf ? (uint32_t)P.shift : 0u)) & 0xffffu);
                    else push_fix_one(P, it.sig_base + at_, c1, it.ev_first + e, it.read, shf ? 1 : 0);
                }
            }
        }
        #undef PAIR_PARK
#else
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");         // previous item's LDS reads are done
        W.tb[lane] = make_uint4(0u, 0u, 0u, 0u);
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
                W.rec[lane * LEAN_EPL + q] = make_uint4(er[q].x, ((uint32_t)so << 18) | ((uint32_t)(int)fl0 & 0xffffu),   // so < 4096
                                                        __float_as_uint(Fh), __float_as_uint(sdk));
                if ((e0 + q < ne) && (lane | q) != 0) {                // so >= 1: every earlier event has >= 1 sample
                    const uint32_t bit = (uint32_t)(so - 1);
                    atomicOr(reinterpret_cast<unsigned int*>(W.tb) + ((bit >> 6) << 2) + ((bit >> 5) & 1u), 1u << (bit & 31u));
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        {   // events begun before each step
            const uint4 mine = W.tb[lane];
            const int pc = __builtin_popcount(mine.x) + __builtin_popcount(mine.y);
            W.tb[lane].z = (uint32_t)(wave_incl_scan_dpp(pc) - pc);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        char* const out_b = reinterpret_cast<char*>(P.sig + it.sig_base);               // wave-uniform: global_store saddr + 32-bit lane offset
        // byte offset of my sample of step 0 within the read: generation index i is stored at at0 + i (RNA: at0 - i)
        uint32_t voff = RNA ? 2u * (it.at0 - (uint32_t)lane) : 2u * (it.at0 + (uint32_t)lane);
        uint32_t idx4 = (uint32_t)lane << 2;                           // 4 * (my sample index within the item), advanced every second step
#if defined(SQG_ABL_NOLOOP)
        const int nfull = 0, rem = wave_total & 1;
#else
        const int nfull = wave_total >> 6, rem = wave_total & 63;
#endif
        // event (within the item) of my sample in the step whose table entry is tq_: events begun in earlier steps + start bits
        // below my lane
        #define LEAN_EVOF(tq_) (int)__builtin_amdgcn_mbcnt_hi((tq_).y, __builtin_amdgcn_mbcnt_lo((tq_).x, (tq_).z))
        // one step: issue the loads of step c_+1 into (RN, MN, EN) and the table entry of step c_+2 into TQN, then the arithmetic
        // of step c_ from (RA, MU, EV).  TQ: the table entry of step c_+1, loaded one step ago.
        // DI: position of the step in its group of ST (4, or 2 for the remainder); idx4, voff and tbp advance once per group,
        // the other steps' +64*DI samples ride in the immediate offsets of their LDS reads and stores.
        #define LEAN_STEP(SH, TAIL, c_, DI, ST, RA, MU, EV, RN, MN, EN, TQ, TQN) {                                       \
            EN = LEAN_EVOF(TQ);                                                                                   \
            RN = W.rec[EN];                                                                                       \
            TQN = tbp[2 + (DI)];                                                                                  \
            LEAN_ARITH(RA, MU)                                                                                    \
            const float vh = __builtin_fmaf(x, __uint_as_float(RA.w), __uint_as_float(RA.z));                     \
            const float t = vh + LEAN_MAGIC;                                                                      \
            const float d = vh - (t - LEAN_MAGIC);                                                                \
            const int si_ = (int)(idx4 >> 2) + 64 * (DI);              /* my sample index within the item */         \
            const bool act = !(TAIL) || si_ < wave_total;                                                         \
            const bool ok = fabsf(d) < thr LEAN_NEARONE_TEST;                                                     \
            /* RNA adaptor window: the ADC value gets -(int16)(30*dig/range) with int16 wrap (src/genread.c:79-86) */ \
            const bool shf = (SH) && (uint32_t)(si_ - it.shift_lo) < (uint32_t)(it.shift_hi - it.shift_lo);       \
            char* const dst_b = out_b + (RNA ? -128 * (DI) : 128 * (DI));                                         \
            if (act && ok) { LEAN_STORE_STMT(DI, RA) }                                                              \
            else if (act) {                                        /* ~1 % of steps: park the undecided samples (no round trip) */ \
                const unsigned long long am = __builtin_amdgcn_ballot_w64(true);                                  \
                const int n0 = W.nfix;                                                                            \
                const int slot = n0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u)); \
                const uint32_t at_ = (RNA ? voff - 128u * (DI) : voff + 128u * (DI)) >> 1;                        \
                const int ev_ = EV;                                                                               \
                if (slot < FIX_SLOTS) W.park[slot] = make_uint4(at_, c1, (uint32_t)ev_, shf ? 1u : 0u);              \
                else push_fix_one(P, it.sig_base + at_, c1, it.ev_first + ev_, it.read, shf ? 1 : 0);   /* overflow (never in practice): global list */ \
                if (slot + 1 == n0 + __popcll(am)) W.nfix = slot + 1;          /* the last of them publishes the new count */ \
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                            \
            }                                                                                                     \
            if ((DI) == (ST) - 1) { idx4 += 256u * (ST); voff = RNA ? voff - 128u * (ST) : voff + 128u * (ST); tbp += (ST); } \
            MN = *reinterpret_cast<const uint32_t*>(mult_b + ((DI) == (ST) - 1 ? 0 : 256 * ((DI) + 1)) + (idx4 - (RN.y >> 16))); }

        /* ablation builds (tools/ab_variants.sh; results are wrong): -DSQG_ABL_NOARITH, -DSQG_ABL_NOSTORE, -DSQG_ABL_NOLOOP */
#if SQG_NEARONE
        #define LEAN_NEARONE_TEST                                      /* box_muller_fast returns NaN there: |d| < thr is false */
#else
        #define LEAN_NEARONE_TEST && c1 <= LCG_M - (1u << NEAR_ONE_BITS)
#endif
#if defined(SQG_ABL_NOSTORE)
        #define LEAN_STORE_COND && (__float_as_uint(t) == 0x12345u)
#else
        #define LEAN_STORE_COND
#endif
        #define LEAN_STORE_VAL(RA_) (uint16_t)((__float_as_uint(t) + RA_.y - (shf ? (uint32_t)P.shift : 0u)) & 0xffffu)
#if defined(SQG_ABL_STORE2)      /* timing-only ablation (results are wrong): the same bytes with half as many store instructions (a dword per lane every other step) */
        #define LEAN_STORE_STMT(DI_, RA_) if (((DI_) & 1) == 0) *reinterpret_cast<uint32_t*>(dst_b + voff + 2u * (uint32_t)lane) = (uint32_t)LEAN_STORE_VAL(RA_) * 0x10001u;
#elif defined(SQG_ABL_STORE4)    /* ... a quarter (8 bytes per lane every fourth step) */
        #define LEAN_STORE_STMT(DI_, RA_) if (((DI_) & 3) == 0) *reinterpret_cast<uint2*>(dst_b + voff + 6u * (uint32_t)lane) = make_uint2((uint32_t)LEAN_STORE_VAL(RA_) * 0x10001u, c1);
#else
#if SQG_LEAN_NT
        #define LEAN_STORE_STMT(DI_, RA_) if (true LEAN_STORE_COND) __builtin_nontemporal_store(LEAN_STORE_VAL(RA_), reinterpret_cast<uint16_t*>(dst_b + voff));
#else
        #define LEAN_STORE_STMT(DI_, RA_) if (true LEAN_STORE_COND) *reinterpret_cast<uint16_t*>(dst_b + voff) = LEAN_STORE_VAL(RA_);
#endif
#endif
#if defined(SQG_ABL_NOARITH)
        #define LEAN_ARITH(RA, MU) const uint32_t c1 = (RA.x ^ MU) & 0x3fffffffu; const float x = __uint_as_float((RA.x + MU) & 0x3fffffffu);
#else
        #define LEAN_ARITH(RA, MU) const uint32_t c1 = lcg_mul_dbl(RA.x, MU); const float x = box_muller_fast(c1);
#endif
        LEAN_T(tr_e);
        uint4 ra, rb, tqa, tqb; uint32_t ma, mb; int eva, evb;
        const uint4* tbp = W.tb;                                        // table entry of the current pair's first step
        {
            const uint4 tq0 = tbp[0];
            eva = LEAN_EVOF(tq0);
        }
        ra = W.rec[eva];
        tqa = tbp[1];
        ma = *reinterpret_cast<const uint32_t*>(mult_b + (idx4 - (ra.y >> 16)));
        int c = 0;
        // the (few) items that overlap the RNA level-shift window run the variant that tests every sample against it
        #define LEAN_LOOP(SH)                                                                                    \
            for (; c + 4 <= nfull; c += 4) {                                                                      \
                LEAN_STEP(SH, false, c, 0, 4, ra, ma, eva, rb, mb, evb, tqa, tqb)                                 \
                LEAN_STEP(SH, false, c + 1, 1, 4, rb, mb, evb, ra, ma, eva, tqb, tqa)                             \
                LEAN_STEP(SH, false, c + 2, 2, 4, ra, ma, eva, rb, mb, evb, tqa, tqb)                             \
                LEAN_STEP(SH, false, c + 3, 3, 4, rb, mb, evb, ra, ma, eva, tqb, tqa)                             \
            }                                                                                                     \
            if (c + 2 <= nfull) {                                                                                 \
                LEAN_STEP(SH, false, c, 0, 2, ra, ma, eva, rb, mb, evb, tqa, tqb)                                 \
                LEAN_STEP(SH, false, c + 1, 1, 2, rb, mb, evb, ra, ma, eva, tqb, tqa)                             \
                c += 2;                                                                                           \
            }                                                                                                     \
            if (c < nfull) {                                                                                      \
                LEAN_STEP(SH, false, c, 0, 2, ra, ma, eva, rb, mb, evb, tqa, tqb)                                 \
                ra = rb; ma = mb; eva = evb; tqa = tqb; c++;                                                      \
                idx4 += 256u; voff = RNA ? voff - 128u : voff + 128u; tbp += 1;                                   \
            }                                                                                                     \
            if (rem) LEAN_STEP(SH, true, c, 0, 2, ra, ma, eva, rb, mb, evb, tqa, tqb)
        if (RNA && it.shift_hi > it.shift_lo) { LEAN_LOOP(true) } else { LEAN_LOOP(false) }
        #undef LEAN_LOOP
        #undef LEAN_STEP
        #undef LEAN_EVOF
        #undef LEAN_ARITH
        #undef LEAN_STORE_COND
        #undef LEAN_STORE_STMT
        #undef LEAN_STORE_VAL
        #undef LEAN_NEARONE_TEST
#endif
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        LEAN_T(tr_f);
        // the item's parked samples (every other item has one: 4.5e-4 of the samples) join one of the batch's FIX_SHARDS lists for k_fixup: ONE returning
        // atomic per such item, at its end (per-item lists walked by a kernel of their own cost that kernel 0.28 ms and 0.4 GB per
        // batch of scattered look-ups next to the following batch's event pass; ONE list, 9e4 atomics on one address, 3.4 ms)
        const int nfix = min(W.nfix, FIX_SLOTS);
        if (nfix) {
            const unsigned int sh = blockIdx.x & (FIX_SHARDS - 1);
            unsigned int at0 = 0;
            if (lane == 0) at0 = atomicAdd(P.fix_sh_count + sh * FIX_SHARD_STRIDE, (unsigned int)nfix);
            at0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)at0);
            FixEntry* dst = P.fix_sh + (size_t)sh * P.fix_sh_cap + at0;
            bool room = at0 + (unsigned int)nfix <= P.fix_sh_cap;
            if (!room) {                                               // the list is full (forced fix-ups in tests): the global list
                if (lane == 0) at0 = atomicAdd(P.fix_count, (unsigned int)nfix);
                at0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)at0);
                dst = P.fix + at0;
                room = at0 + (unsigned int)nfix <= P.fix_cap;
                if (!room && lane == 0) atomicOr(P.err, 8u);
            }
            if (room && lane < nfix) {
                const uint4 pk = W.park[lane];
                FixEntry fe; fe.at = it.sig_base + pk.x; fe.c1 = pk.y; fe.ev = it.ev_first + pk.z; fe.read = it.read; fe.shifted = (int)pk.w; fe.pad = P.fix_tag;
                dst[lane] = fe;
            }
        }
#if defined(SQG_LEAN_TRACE)
        {
            LEAN_T(tr_g);
            if (lane == 0) {
                unsigned long long* const T = g_lean_trace + (size_t)((blockIdx.x * 4 + wid) & (LEAN_TRACE_SHARDS - 1)) * 16;
                atomicAdd(&T[0], 1ull); atomicAdd(&T[1], (unsigned long long)((wave_total + 63) >> 6));
                atomicAdd(&T[2], tr_b - tr_a); atomicAdd(&T[3], tr_c - tr_b); atomicAdd(&T[4], tr_d - tr_c);
                atomicAdd(&T[5], tr_e - tr_d); atomicAdd(&T[6], tr_f - tr_e); atomicAdd(&T[7], tr_g - tr_f);
                if (tr_prev) atomicAdd(&T[8], tr_a - tr_prev); else atomicAdd(&T[9], tr_a - tr_k0);
            }
            tr_prev = tr_g;
        }
#endif
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
            if (P.one) {                                               // one-partition hand-out: rank and dwell from part[]
                const uint32_t sl = (uint32_t)rd.slot0 + (uint32_t)e, w_ = P.part[sl];
                er = make_uint2(sl, w_ & 0xffffu);
                sps = (int)(w_ >> 16);
            } else if (P.evrec32) {                                    // bucketed hand-out, 4-B event records: {rank, slot within the (link, partition)}
                const uint32_t w_ = P.evrec32[rd.ev_off + e], rank = w_ >> EVR_REL_BITS;
                er = make_uint2(P.lbase[(size_t)(rd.slot0 >= 0 ? rd.slot0 : P.tile_link[g]) * PART_MAX + (rank >> PART_SUB_BITS)] + (w_ & ((1u << EVR_REL_BITS) - 1u)), rank);
                sps = P.dwell ? (int)P.dwell[rd.ev_off + e] : P.const_sps;
            } else {
                er = P.evrec[rd.ev_off + e];
                sps = P.dwell ? (int)P.dwell[rd.ev_off + e] : P.const_sps;
            }
            if (P.part_state) er.x = P.part_state[er.x];                // few workers, bucketed hand-out (k_part.h): slot -> state
        }
        const float2 md = valid ? P.model[er.y] : make_float2(0.f, 0.f);
        const int incl = wave_incl_scan(sps, lane);
        const int wave_total = __shfl(incl, 63);
        const int so = incl - sps;                                     // first sample of my event within the tile
        const long long sig_base = P.sig_off[r];
        if (P.sig_off[r + 1] - sig_base >= 4294967295LL) continue;     // a read of >= UINT32_MAX samples fails the batch (k_scan; src/sim.c:559-562): its
                                                                       // 32-bit positions would wrap, nobody writes it
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

// ---- k_fixup: FP64 path for the samples the certified kernels left undecided ------------------
// one list per batch (SigParams.fix): the lean kernel appends an item's parked samples at the item's end, the generic kernel per step
__device__ static inline void fixup_one(const SigParams& P, const FixEntry& fe) {
    const ReadDesc rd = P.reads[fe.read];
    const int e = (int)(fe.ev - rd.ev_off);
    const uint8_t* bp = P.bases + rd.base_off + (e < rd.ne0 ? (long long)e : (long long)rd.len0 + (e - rd.ne0));
    const uint32_t rank = kmer_rank_wide(bp, P.k, P.meth);
    const float2 md = P.model[rank];
#if defined(SQG_ABL_FIXMATH)                                       /* timing-only ablation: the fix-up kernel without its FP64 arithmetic */
    int16_t q = (int16_t)(fe.c1 + (uint32_t)md.x + (uint32_t)md.y);
#else
    int16_t q = sample_exact(fe.c1, md.x, md.y, P.dig, P.range, rd.offset);
#endif
    if (fe.shifted) q = (int16_t)(uint16_t)(((int)q - P.shift) & 0xffff);
    P.sig[fe.at] = q;
}
// grid: FIX_SHARDS workgroups.  Workgroup s: list s of the lean kernel (entries of this batch carry its tag), and its share of the
// global list (the generic kernel's samples, and what did not fit a list)
__global__ __launch_bounds__(256) void k_fixup(const SigParams P) {
    if (P.fix_sh) {
        const unsigned int ns = min(P.fix_sh_count[blockIdx.x * FIX_SHARD_STRIDE], P.fix_sh_cap);
        const FixEntry* lst = P.fix_sh + (size_t)blockIdx.x * P.fix_sh_cap;
        unsigned int mine = 0;
#if SQG_ABL_FIXK != 1
        for (unsigned int i = threadIdx.x; i < ns; i += 256) {
            const FixEntry fe = lst[i];
            if (fe.pad != P.fix_tag) continue;                         // (counted, but written to the global list: the list was full)
#if SQG_ABL_FIXK != 2
            fixup_one(P, fe);
#endif
            mine++;
        }
#endif
        // statistics (sqg_get_timing): samples that went through this list (a word per list, summed by the host: 4096 atomics on one
        // counter took this kernel from 15 to 124 us)
        __shared__ unsigned int wsum[4];
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
        __syncthreads();
        if (threadIdx.x == 0) { const unsigned int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3]; P.fix_sh_stat[blockIdx.x] = tot; if (P.host_res) P.host_res[4 + blockIdx.x] = tot; }
    }
    const unsigned int n_all = *P.fix_count, n = min(n_all, P.fix_cap);
    // (the batch's last kernel, behind everything that reports into the error word: the host finds both without a read-back)
    if (P.host_res && blockIdx.x == 0 && threadIdx.x == 0) { P.host_res[0] = *P.err; P.host_res[1] = n_all; }
    for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) fixup_one(P, P.fix[i]);
    // the batch's last kernel: the per-read sample totals have been read by everything that needs them, and the slot's next batch
    // wants them zero (the pieces of a read add theirs up: k_part_events.h) -- saves that batch a fill kernel in front of its first pass
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P.seglen_zero; i += gridDim.x * 256) P.seglen_out[i] = 0ull;
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

