// k_part_events.h -- few workers (the reference's `-t 1`, `-t 8`): the event passes of the hand-out over bucketed events, one wavefront per link
// Part of the device code of the per-read signal path; included through sqg_kernels.h (see there for the overview).
//
// With the hand-out left to k_part_hist / k_part_scan / k_part_hand (k_part.h) an event pass has nothing sequential left but
// the sample offsets inside a read: a link (a few whole reads of a worker chain, or pieces of long ones) is walked by ONE
// wavefront, the links of a workgroup share nothing but two read-only tables, and there is no workgroup barrier after the set-up.
// A read is walked in segments of 512 events; event 64 q + lane of the segment is the lane's q-th, so that an LDS atomic issued per q
// sees the segment's events in event order (instruction order, then lane order: what k_lds_order_check certifies on the device).
//
//   COUNT   (k > 6, first pass)  dwell draws (src/gensig.c:254-257) -> dwell[], per-read sample totals, first sample of every
//           64-event tile; k-mer ranks; events per (link, partition) -> pcnt
//   SCATTER (k > 6, second pass) ranks again, dwell from memory; slot = fetch-add on the (link, partition)'s next slot -- stable by
//           the order above; the records {dwell, low 12 bits of the rank} wait in per-partition rings in LDS until a whole 64-B line
//           of part[] can be written (a partition gets 8 of a segment's 512 events: written as they come, every line of part[]
//           would be written in pieces, which costs the memory system 1.5x the time: measured); evrec32 = rank | slot - first slot of the
//           (link, partition): 2 B per event for the sample kernels, which take the rank from the bases (lbase, tile_link)
//   ONE     (k <= 6: one partition, the only pass) COUNT's work; an event's slot is its position in the worker chain, which staging
//           knows for every link and read: part[slot] = {dwell, rank} is written straight away, and the sample kernels need no evrec
//
// The k-mer ranks come from the segment's bases packed two bits each, first base in the top bits of a 32-bit word (src/seq.h:31-42
// puts the first base in the top digits of the rank): an event's rank is a 2k-bit window of two consecutive words -- one
// two-word LDS read, one 64-bit shift.
// k_events<.., PART> (k_events.h) does COUNT and SCATTER with a workgroup per link and lane masks instead of ordered atomics; it
// stays for the 5-letter methylation alphabet and for devices that do not pass the order check.
#pragma once

#ifndef PEV_EPL
#define PEV_EPL 8                        // events per lane and segment (a multiple of 4)
#endif
#define PEV_SEG (64 * PEV_EPL)           // events per segment
#define PEV_WAVES 4                      // links (wavefronts) per workgroup, first pass
#ifndef PEV_COUNT_OCC
#define PEV_COUNT_OCC 1                  // first pass: wavefronts per SIMD the register allocation aims at
#endif
#define PEV_WAVES_SCATTER 2              // ... second pass (10 KiB of LDS per link)
#ifndef PEV_PACK_SUMS
#define PEV_PACK_SUMS 1                  // two tile sums per DPP scan (A/B)
#endif
#ifndef PEV_SCATTER_FASTRANK
#define PEV_SCATTER_FASTRANK 0          // A/B: the scatter pass with the per-segment test of the rank look-up as well
#endif
#define PEV_HALO 24                      // bases behind the segment's own: 2 (k - 1) <= 16 (the k-mers of the RNA stall start k - 1 bases further on)
#define PEV_WORDS ((PEV_SEG + PEV_HALO) / 16 + 2)

#define PEV_RING 32                      // slots a partition's ring holds: two 64-B lines of part[]
#define PEV_LINE 16                      // slots per 64-B line
#define PEV_TASKS (3 * PART_MAX)
#define PEV_FLUSH_IT 10                  // lines / 4 a segment's flush always writes (more: a loop)
template <bool SCATTER>
struct PevWave {
    uint32_t codes[PEV_WORDS];           // the segment's 2-bit base codes, 16 per word, first base in the top bits
    uint32_t wslot[PART_MAX];            // COUNT: the link's events per partition so far; SCATTER: the partition's next slot in part[]
    // SCATTER: a partition's records wait in its ring (slot s at ring[p][s % 32]) until a whole 64-B line of part[] can be written
    uint32_t wbase[SCATTER ? PART_MAX : 1];              // SCATTER: first slot of the (link, partition): an event's record carries its slot minus this
    uint32_t flu[SCATTER ? PART_MAX : 1];                // first slot of the partition not yet written to part[]
    uint32_t ring[SCATTER ? PART_MAX * PEV_RING : 1];    // (a segment that does not fit the rings borrows them as its sort buffer)
    uint2 tasks[SCATTER ? PEV_TASKS : 1];                // lines to write: {line, partition | first element << 8 | end element << 16}
};
struct PevTables {                        // read-only, shared by the wavefronts of a workgroup
    uint32_t jump[256];                  // 2 * a^(2j)
    uint8_t lut[256];                    // base -> 2-bit code (src/seq.h:14-27)
};
template <bool SCATTER>
struct PevLds {
    PevTables t;
    PevWave<SCATTER> w[SCATTER ? PEV_WAVES_SCATTER : PEV_WAVES];
};
__device__ static inline void pev_tables(PevTables& T, const uint32_t* __restrict__ pw, const int tid, const int nthreads) {
    for (int i = tid; i < 256; i += nthreads) { T.jump[i] = pw[2 * POW_N + i] << 1; T.lut[i] = (uint8_t)base_code((uint8_t)i); }   // (doubled: lcg_mul_dbl)
}

// sum over the wavefront, in every lane's SGPR-to-be
__device__ static inline int pev_wave_sum(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_dpp(v), 63); }
// four of them, step by step side by side: a DPP step reads the register the step before wrote, which costs the chain two idle
// cycles per step -- the other three chains fill them
__device__ static inline void pev_wave_sum4(int (&v)[4]) {
#define PEV_DPP_STEP(ctrl_, rows_) _Pragma("unroll") for (int i = 0; i < 4; i++) v[i] += __builtin_amdgcn_update_dpp(0, v[i], ctrl_, rows_, 0xf, false);
    PEV_DPP_STEP(0x111, 0xf) PEV_DPP_STEP(0x112, 0xf) PEV_DPP_STEP(0x114, 0xf) PEV_DPP_STEP(0x118, 0xf) PEV_DPP_STEP(0x142, 0xa) PEV_DPP_STEP(0x143, 0xc)
#undef PEV_DPP_STEP
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = __builtin_amdgcn_readlane(v[i], 63);
}

// DW as in k_events: 0 = dwell from memory (or constant), 1 = drawn here, certified fp32 path, 2 = drawn here in FP64.
// grid: ceil(links / waves per workgroup).  dump: first of 64 slots behind part[]'s last (a flush without a line writes there)
#define PEV_COUNT 0
#define PEV_SCATTER 1
#define PEV_ONE 2
// One link, one wavefront: link li of the launch order (P.chain_order), tables T (pev_tables), wave-private LDS W.  No workgroup
// barrier inside: the wavefronts of a workgroup -- of k_part_events, or of a kernel that runs links next to other work
// (k_part_hand_count, below) -- walk their links independently.
template <int DW, int MODE>
__device__ static __forceinline__ void pev_link(const SigParams& P, const PevTables& L, PevWave<MODE == PEV_SCATTER>& W, const int li, const uint32_t dump, const int lane) {
    constexpr bool SCATTER = MODE == PEV_SCATTER, ONE = MODE == PEV_ONE;
    static_assert(!SCATTER || DW == 0, "the second pass reads the dwells the first one drew");
    const int chain = P.chain_order[li];
    const int c_lo = P.chain_off[chain], c_hi = P.chain_off[chain + 1];
    const int k = P.k;
    const uint32_t kmask = (1u << (2 * k)) - 1u;                  // (k <= 9 here)
    W.wslot[lane] = (SCATTER && lane < P.n_part) ? P.poff[(size_t)lane * P.n_links + chain] + P.pstart[(size_t)P.link_q[chain] * P.n_part + lane] : 0u;
    uint32_t slot0 = ONE ? P.poff[chain] : 0u;                     // ONE: the link's first slot (absolute: staging knows every link's events); then the read's
    if (SCATTER) {
        W.flu[lane] = W.wslot[lane];
        W.wbase[lane] = W.wslot[lane];
        P.lbase[(size_t)chain * PART_MAX + lane] = W.wslot[lane];     // what the sample kernels add to an event's 16-bit record
    }
    // SCATTER: the slots [a, b) of lane p's partition go from the ring to part[], 64-B line by line (whole lines but for a link's
    // first and last): every lane lists its lines, then 16 lanes write one line each
    auto flush = [&](const uint32_t a, const uint32_t b) {
        const int nl = b > a ? (int)(((b - 1) >> 4) - (a >> 4)) + 1 : 0;      // <= 3: the ring holds 32 slots
        const int incl = wave_incl_scan_dpp(nl);
        const int n_task = __builtin_amdgcn_readlane(incl, 63);
        const int at = incl - nl;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (i < nl) {
                const uint32_t line = (a >> 4) + (uint32_t)i;
                const uint32_t lo_e = i == 0 ? (a & 15u) : 0u, hi_e = i == nl - 1 ? ((b - 1) & 15u) + 1u : 16u;
                W.tasks[at + i] = make_uint2(line, (uint32_t)lane | lo_e << 8 | hi_e << 16);
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        // PEV_FLUSH_IT store instructions whatever the number of lines (a segment completes 32 on average): the compiler can then
        // count the stores between a prefetch and its use, and waits for the prefetch only (a wavefront's loads and stores retire in
        // order).  A lane without an element of its own repeats a neighbour's store (same address, same value).
        auto put = [&](const int t) {
            const uint2 tk = W.tasks[max(min(t, n_task - 1), 0)];
            const uint32_t e = min(max((uint32_t)lane & 15u, (tk.y >> 8) & 0xffu), (tk.y >> 16) - 1u), sl = tk.x * PEV_LINE + e;
            const uint32_t v = W.ring[(tk.y & (PART_MAX - 1)) * PEV_RING + (sl & (PEV_RING - 1))];
            P.part[n_task > 0 ? sl : dump + (uint32_t)lane] = v;
        };
#pragma unroll
        for (int it = 0; it < PEV_FLUSH_IT; it++) put(4 * it + (lane >> 4));
        for (int t0 = 4 * PEV_FLUSH_IT; t0 < n_task; t0 += 4) put(t0 + (lane >> 4));
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    };
    const uint32_t a2seg = DW ? lcg_jump2(P.pw, (uint32_t)PEV_SEG) : 0u, a2half = DW ? lcg_jump2(P.pw, 256u) : 0u;
    const float dw_sf = (float)P.dstd, dw_mf = (float)P.dmean;
    const float dw_eps = P.delta_x * fabsf(dw_sf) + 4.0f * 5.9604645e-8f * (fabsf(dw_mf) + 7.0f * fabsf(dw_sf) + 1.0f) + 1e-6f;   // as in k_events

    for (int ci = c_lo; ci < c_hi; ci++) {
        // a piece: the events [e_lo, ne) of read r -- a whole read, or whole segments of a read too long for one link.  What a piece
        // needs from the pieces before it: the time stream's position (arithmetic), and the samples before its first event, which
        // only the tile offsets and the read's totals contain: those are written relative to the piece, the totals added up with
        // atomics, and k_part_tile_bases shifts the tile offsets of the later pieces
        const int4 pc = P.pieces[ci];
        const int r = pc.x, e_lo = pc.y;
        const ReadDesc rd = P.reads[r];
        const int ne = pc.z;                                       // (the piece's end: "the read's end" for everything below)
        const bool whole = e_lo == 0 && ne == rd.ne0 + rd.ne1;
        const uint8_t* rbases = P.bases + rd.base_off;
        const int shift1 = rd.len0 - rd.ne0;                      // base of event e >= ne0: e + shift1 (src/genread.c:87-88)
        unsigned long long done = 0;                              // samples before this segment
        long long n1 = -1;                                        // samples of the read's first part (read + prefix), once known
        uint32_t c_seg = DW ? (uint32_t)__builtin_amdgcn_readfirstlane((int)lcg_mul(lcg_mul(rd.time_c0, LCG_A), e_lo ? lcg_jump2(P.pw, (uint32_t)e_lo) : 1u)) : 0u;   // a * (time-stream state at the segment's first event)
        // a segment's inputs: 8 base bytes per lane (+ the halo in three lanes), SCATTER: 8 dwells per lane.  The next segment's are
        // requested BEFORE this segment's stores are issued: a wavefront's loads and stores retire in order (one vmcnt), so a load
        // issued behind the stores would wait for them as well
        // (unconditional: the batch's base buffer ends with 1 KiB of slack; what lies behind a read's last base only reaches the
        // ranks of events that do not exist)
        auto load8 = [&](const int bi) -> unsigned long long {
            unsigned long long v;
            __builtin_memcpy(&v, rbases + bi, 8);
            return v;
        };
        unsigned long long b_cur = 0, b_halo = 0;
        uint16_t d_cur[PEV_EPL];
        auto fetch = [&](const int s) {
            const int bs = s + (s >= rd.ne0 ? shift1 : 0);
            b_cur = load8(bs + 8 * lane);
            if (lane < PEV_HALO / 8) b_halo = load8(bs + PEV_SEG + 8 * lane);
            if (DW == 0 && P.dwell) {
#pragma unroll
                for (int q = 0; q < PEV_EPL; q++) d_cur[q] = P.dwell[rd.ev_off + s + 64 * q + lane];   // (unconditional: the array ends with slack)
            }
        };
#pragma unroll
        for (int q = 0; q < PEV_EPL; q++) d_cur[q] = 0;
        if (ne > e_lo) fetch(e_lo);
        // (these loads have landed before the loop: a use inside it then only waits for the prefetch of the iteration before, which
        // has this segment's stores behind it -- the compiler counts them -- instead of for everything in flight)
        __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0)
        // One segment.  FULL: every event of the segment exists (all but a read's last segment): no per-lane tests, no masked stores
        #define PEV_IN(j_) (FULL || (j_) < n_seg)
        auto segment = [&](auto full_tag, const int s0) {
            constexpr bool FULL = decltype(full_tag)::value;
            const int n_seg = FULL ? PEV_SEG : ne - s0;
            const bool second = s0 >= rd.ne0;                     // the whole segment lies in the read's second part
            // ---- the segment's bases, packed
            {
                auto pack8 = [&](const unsigned long long v) -> uint32_t {
                    uint32_t h = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) h = (h << 2) | L.lut[(uint32_t)(v >> (8 * i)) & 0xffu];
                    return h;
                };
                // lane l: bases 8l .. 8l+7 -> one 16-bit half; the first of two halves is the word's upper one
                reinterpret_cast<uint16_t*>(W.codes)[lane ^ 1] = (uint16_t)pack8(b_cur);
                if (lane < PEV_HALO / 8) reinterpret_cast<uint16_t*>(W.codes)[(64 + lane) ^ 1] = (uint16_t)pack8(b_halo);
            }
            int sps[PEV_EPL];
#pragma unroll
            for (int q = 0; q < PEV_EPL; q++) sps[q] = (DW == 0) ? (PEV_IN(64 * q + lane) ? (P.dwell ? (int)d_cur[q] : P.const_sps) : 0) : 0;
            if (s0 + PEV_SEG < ne) fetch(s0 + PEV_SEG);
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            // ---- ranks
            uint32_t rank[PEV_EPL];
            // (wave-uniform) does the read's second part -- the RNA stall behind the adaptor -- start inside this segment?  Else event j's
            // k-mer starts at base j of the segment: word and shift depend on (q, lane) alone and cost the segment nothing (with the
            // test per event: 9 of the pass's 60 VALU instructions per event)
            const bool straddle = (SCATTER && !PEV_SCATTER_FASTRANK) || (rd.ne1 > 0 && s0 < rd.ne0 && s0 + PEV_SEG > rd.ne0);   // (the scatter pass is not short of VALU cycles, and the second copy of the loop costs it 15 registers)
            if (!straddle) {
#pragma unroll
                for (int q = 0; q < PEV_EPL; q++) {
                    const int cb = 64 * q + lane;
                    const uint32_t hi = W.codes[cb >> 4], lo = W.codes[(cb >> 4) + 1];
                    rank[q] = (uint32_t)(((unsigned long long)hi << 32 | lo) >> (64 - 2 * (cb & 15) - 2 * k)) & kmask;
                }
            } else {
#pragma unroll
                for (int q = 0; q < PEV_EPL; q++) {
                    const int j = 64 * q + lane;
                    const int cb = j + ((!second && s0 + j >= rd.ne0) ? shift1 : 0);
                    const uint32_t hi = W.codes[cb >> 4], lo = W.codes[(cb >> 4) + 1];
                    rank[q] = (uint32_t)(((unsigned long long)hi << 32 | lo) >> (64 - 2 * (cb & 15) - 2 * k)) & kmask;
                }
            }
            // ---- dwells
            uint32_t c_blk[PEV_EPL / 4];                              // ... at the segment's 256th, 512th, ... event (scalar)
            c_blk[0] = c_seg;
#pragma unroll
            for (int h = 1; h < PEV_EPL / 4; h++) c_blk[h] = DW ? (uint32_t)__builtin_amdgcn_readfirstlane((int)lcg_mul(c_blk[h - 1], a2half)) : 0u;
#pragma unroll
            for (int q = 0; q < PEV_EPL; q++) {
                const int j = 64 * q + lane;
                const bool valid = PEV_IN(j);
                if (DW != 0 && valid) {
                    // event s0 + j uses draws 2j+1, 2j+2 of the time stream after the segment's first state
                    const uint32_t c1 = lcg_mul_dbl(c_blk[q >> 2], L.jump[64 * (q & 3) + lane]);
                    bool decided = false;
                    int v = 0;
                    if (DW == 1) {
                        const float x = box_muller_fast(c1);
                        const float g = __builtin_fmaf(x, dw_sf, dw_mf);
                        const float t = g + LEAN_MAGIC;              // |g| < 2^22: the host takes the FP64 variant (DW 2) when dwell_hi >= 1e6
                        const float fl = t - LEAN_MAGIC;
                        if (fabsf(g - fl) < 0.5f - dw_eps && c1 <= LCG_M - (1u << NEAR_ONE_BITS)) { v = (int)__float_as_uint(t) - 0x4b400000; decided = true; }
                    }
                    if (!decided) v = dwell_exact(c1, P.dstd, P.dmean);      // src/gensig.c:255
                    v = max(v, 1 - v);                                       // src/gensig.c:256
                    sps[q] = v;
                    P.dwell_out[rd.ev_off + s0 + j] = (uint16_t)v;
                }
            }
            if (DW != 0 && P.dwell_unbounded) {                    // (wave-uniform, rare profile: dwells beyond 16 bits are reported and clamped: a test per segment, not per draw)
#pragma unroll
                for (int q = 0; q < PEV_EPL; q++) {
                    if (PEV_IN(64 * q + lane) && sps[q] > 65535) { atomicOr(P.err, 1u); sps[q] = 65535; P.dwell_out[rd.ev_off + s0 + 64 * q + lane] = (uint16_t)65535; }
                }
            }
            if (DW) c_seg = (uint32_t)__builtin_amdgcn_readfirstlane((int)lcg_mul(c_seg, a2seg));
            if (!SCATTER) {
                // ---- first sample of every 64-event tile, the read's totals, events per partition
                // (the samples of each of the segment's eight tiles: a sum over the wavefront -- two tiles per DPP scan, 16 bits each,
                // while 64 dwells fit 16 bits: 7 of the pass's 69 instructions per event)
                uint32_t tsum[PEV_EPL];
                if (PEV_PACK_SUMS && P.dwell_pack) {
                    static_assert(PEV_EPL == 8, "four packed sums");
                    int two[4];
#pragma unroll
                    for (int h = 0; h < 4; h++) two[h] = (int)((uint32_t)sps[2 * h] | ((uint32_t)sps[2 * h + 1] << 16));
                    pev_wave_sum4(two);
#pragma unroll
                    for (int h = 0; h < 4; h++) { tsum[2 * h] = (uint32_t)two[h] & 0xffffu; tsum[2 * h + 1] = (uint32_t)two[h] >> 16; }
                } else {
#pragma unroll
                    for (int q = 0; q < PEV_EPL; q++) tsum[q] = (uint32_t)pev_wave_sum(sps[q]);
                }
                uint32_t run = 0;
#pragma unroll
                for (int q = 0; q < PEV_EPL; q++) {
                    const int e_q = s0 + 64 * q;                      // (wave-uniform)
                    if (e_q < ne) {
                        if (lane == 0) P.tile_so[rd.tile_off + (e_q >> 6)] = (uint32_t)done + run;
                        if (rd.ne1 > 0 && rd.ne0 >= e_q && rd.ne0 < e_q + 64)       // the second part starts in this tile
                            n1 = (long long)done + run + pev_wave_sum(lane < rd.ne0 - e_q ? sps[q] : 0);
                        run += tsum[q];
                        if (ONE) {
                            if (PEV_IN(64 * q + lane)) {
                                const uint32_t sl = slot0 + (uint32_t)(s0 - e_lo + 64 * q + lane);
                                P.part[sl] = rank[q] | ((uint32_t)sps[q] << 16);   // (all the sample kernels need besides state[sl]: no evrec)
                            }
                        } else if (PEV_IN(64 * q + lane)) atomicAdd(&W.wslot[rank[q] >> PART_SUB_BITS], 1u);
                    }
                }
                done += run;
            } else {
                // ---- slots, stable in event order
                const uint32_t start = W.wslot[lane], flu = W.flu[lane];
                uint32_t slot[PEV_EPL];
                asm volatile("" ::: "memory");
#pragma unroll
                for (int q = 0; q < PEV_EPL; q++) {
                    slot[q] = 0;
                    if (PEV_IN(64 * q + lane)) slot[q] = __hip_atomic_fetch_add(&W.wslot[rank[q] >> PART_SUB_BITS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    asm volatile("" ::: "memory");                    // (the compiler keeps the atomics in this order)
                }
                const uint32_t end = W.wslot[lane];
                // the event's record for the sample kernels: rank | its slot relative to the (link, partition)'s first (14 bits: staging
                // keeps a link below 16384 events).  Non-temporal: not read before the sample kernel
#pragma unroll
                for (int q = 0; q < PEV_EPL; q++) {
                    if (PEV_IN(64 * q + lane))
                        __builtin_nontemporal_store((rank[q] << EVR_REL_BITS) | (slot[q] - W.wbase[rank[q] >> PART_SUB_BITS]), P.evrec32 + rd.ev_off + s0 + 64 * q + lane);
                }
                if (lane < PEV_EPL && s0 + 64 * lane < ne) P.tile_link[rd.tile_off + (s0 >> 6) + lane] = chain;   // (one tile per 64 events of the segment)
                if (__builtin_amdgcn_ballot_w64(end - flu > (uint32_t)PEV_RING) == 0ull) {
                    // the records join their partitions' rings; what completes a line of part[] goes out
#pragma unroll
                    for (int q = 0; q < PEV_EPL; q++) {
                        if (PEV_IN(64 * q + lane))
                            W.ring[(rank[q] >> PART_SUB_BITS) * PEV_RING + (slot[q] & (PEV_RING - 1))] = (rank[q] & (PART_SUB - 1)) | ((uint32_t)sps[q] << 16);
                    }
                    const uint32_t upto = max(flu, end & ~(uint32_t)(PEV_LINE - 1));
                    W.flu[lane] = upto;
                    flush(flu, upto);
                } else {
                    // a partition with more events in this segment than its ring holds (homopolymers, adaptors): the rings are
                    // written out as they stand, and the segment's records go out sorted by partition, a run per partition
                    flush(flu, start);
                    W.flu[lane] = end;
                    uint2* const sorted = reinterpret_cast<uint2*>(W.ring);      // {slot, record}, PEV_SEG of them
                    uint2* const segbase = W.tasks;                               // per partition: {first slot in this segment, position of its run}
                    const uint32_t cnt = end - start;
                    segbase[lane] = make_uint2(start, (uint32_t)wave_incl_scan_dpp((int)cnt) - cnt);
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int q = 0; q < PEV_EPL; q++) {
                        if (PEV_IN(64 * q + lane)) {
                            const uint2 sb = segbase[rank[q] >> PART_SUB_BITS];
                            sorted[sb.y + (slot[q] - sb.x)] = make_uint2(slot[q], (rank[q] & (PART_SUB - 1)) | ((uint32_t)sps[q] << 16));
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int q = 0; q < PEV_EPL; q++) {
                        if (PEV_IN(64 * q + lane)) { const uint2 v = sorted[64 * q + lane]; P.part[v.x] = v.y; }
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
                }
            }
        };
        #undef PEV_IN
        for (int s0 = e_lo; s0 < ne; s0 += PEV_SEG) {
            if (s0 + PEV_SEG <= ne) segment(std::true_type{}, s0); else segment(std::false_type{}, s0);
        }
        slot0 += (uint32_t)(ne - e_lo);
        if (!SCATTER && lane == 0) {
            // samples of the piece's events before / from the read's second part (n1: known when the second part starts inside the piece)
            const long long tot = (long long)done, first = n1 >= 0 ? n1 : (rd.ne0 >= ne ? tot : 0);
            if (DW) {
                if (whole) { P.seglen_out[2 * r] = (unsigned long long)first; P.seglen_out[2 * r + 1] = (unsigned long long)(tot - first); }
                else { atomicAdd(&P.seglen_out[2 * r], (unsigned long long)first); atomicAdd(&P.seglen_out[2 * r + 1], (unsigned long long)(tot - first)); }
            }
            P.piece_total[ci] = (uint32_t)done;
        }
    }
    if (SCATTER) flush(W.flu[lane], W.wslot[lane]);                 // what is left in the rings: each partition's last, partial line
    if (MODE == PEV_COUNT && lane < P.n_part) P.pcnt[(size_t)lane * P.n_links + chain] = W.wslot[lane];
}

template <int DW, int MODE>
__global__ __launch_bounds__(64 * (MODE == PEV_SCATTER ? PEV_WAVES_SCATTER : PEV_WAVES), MODE == PEV_SCATTER ? 1 : PEV_COUNT_OCC) void k_part_events(const SigParams P, const int n_links, const uint32_t dump) {
    constexpr bool SCATTER = MODE == PEV_SCATTER;
    constexpr int NWV = SCATTER ? PEV_WAVES_SCATTER : PEV_WAVES;
    __shared__ PevLds<SCATTER> L;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    pev_tables(L.t, P.pw, tid, 64 * NWV);
    __syncthreads();
    const int li = blockIdx.x * NWV + wid;
    if (li >= n_links) return;                                    // (no barrier below)
    pev_link<DW, MODE>(P, L.t, L.w[wid], li, dump, lane);
}

// ---- the hand-out of batch i and the counting pass of batch i+1 in ONE launch ---------------------------------------------
// k_part_hand_ord is bound by the 1.4 GB it moves, at one wavefront per SIMD (a slice's two 16-KiB tables leave room for four per
// CU); the counting pass is bound by the VALU (the dwells' Box-Muller draws) and needs nothing but the staged reads of its batch.
// Queued one behind the other each leaves idle what the other wants; kernels of two queues share this GPU worse than they follow
// each other.  So, when the next batch is staged by the time a batch is run (sqg_batch_run, h_run.h): a persistent grid of
// four workgroups per CU, in each ONE wavefront that takes slices of batch i (hand_slice) and THREE that take links of batch i+1
// (pev_link<DW, COUNT>), the hand-out wavefront on a different SIMD in each of a CU's workgroups.  16 wavefronts per CU: what the counting pass' registers allow.
#ifndef PHC_COUNT_WAVES
#define PHC_COUNT_WAVES 3
#endif
#ifndef PHC_PRIO
#define PHC_PRIO 0                       // A/B: the hand-out wavefront at a raised issue priority (s_setprio)
#endif
template <int DW, int MODE>
__global__ __launch_bounds__(64 * (1 + PHC_COUNT_WAVES), 4) void k_part_hand_count(
        const uint32_t* __restrict__ part, uint32_t* __restrict__ state_out, const uint32_t* __restrict__ slice_lo, const uint32_t* __restrict__ slice_hi,
        const uint32_t* __restrict__ n_slices, const uint32_t* __restrict__ phist, const uint32_t* __restrict__ pw, unsigned int* __restrict__ err, const int fault,
        const SigParams Pn, const int n_links_n, const uint32_t dump_n, const int ncu) {
    __shared__ HandLds H;
    __shared__ PevTables T;
    __shared__ PevWave<false> W[PHC_COUNT_WAVES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ int simd_of[1 + PHC_COUNT_WAVES];
    hand_tables(H, pw, tid, 64 * (1 + PHC_COUNT_WAVES));
    pev_tables(T, Pn.pw, tid, 64 * (1 + PHC_COUNT_WAVES));
    // Which wavefront hands out: the one on SIMD j of the CU, j = this workgroup's place among the (four) workgroups its CU holds --
    // workgroups one CU count apart, when four per CU are dispatched round-robin over XCDs, then CUs (measured: tools/hwid_probe.hip;
    // a workgroup's wavefronts sit on four different SIMDs, in an order that differs from workgroup to workgroup: by wavefront INDEX two
    // SIMDs of a CU got two hand-out wavefronts each and two none, 470 instead of 290 us for the hand-out alone).  The SIMD is read
    // from the hardware (HW_REG_HW_ID bits 4-5); should no wavefront of the workgroup sit on SIMD j, the j-th does it.
    if (lane == 0) simd_of[wid] = (int)(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3u);
    __syncthreads();
    const int want = (int)((blockIdx.x / (unsigned)ncu) & 3u);
    int hw = want % (1 + PHC_COUNT_WAVES);
#pragma unroll
    for (int w = PHC_COUNT_WAVES; w >= 0; w--) if (simd_of[w] == want) hw = w;
    hw = __builtin_amdgcn_readfirstlane(hw);
    // Work is dealt out statically, a stride of the grid apart (slices are of equal length, links of equal size, the launch order
    // lists the longest first): a queue -- one counter for the slices, one for the links -- was measured first, and 30 000 returning
    // atomics on one cache line took longer than the work they handed out (the fused launch 705 us against 296 + 285 for the two
    // kernels one after the other)
    if (wid == hw) {
#if PHC_PRIO
        __builtin_amdgcn_s_setprio(PHC_PRIO);
#endif
        const uint32_t ns = *n_slices;
        for (uint32_t sl = blockIdx.x; sl < ns; sl += gridDim.x)
            hand_slice(H, part, state_out, slice_lo[sl], slice_hi[sl], phist + (size_t)sl * PART_SUB, pw, err, fault, lane);
    } else {
        const int m = wid - (wid > hw ? 1 : 0);
        PevWave<false>& Wm = W[m];
        for (int li = (int)blockIdx.x * PHC_COUNT_WAVES + m; li < n_links_n; li += (int)gridDim.x * PHC_COUNT_WAVES)
            pev_link<DW, MODE>(Pn, T, Wm, li, dump_n, lane);                 // MODE: COUNT (k > 6), or ONE (k <= 6: the only event pass)
    }
}

// Reads cut into several pieces: the first pass wrote every piece's tile offsets relative to the piece's own first sample; the pieces
// behind a read's first get the samples of the pieces before them added.  A read's pieces are consecutive in P.pieces.
// grid: pieces, 64 threads.
__device__ static inline void part_tile_bases_body(const SigParams& P, const int pi, const int lane) {
    const int4 pc = P.pieces[pi];
    if (pc.y == 0) return;
    uint32_t base = 0;
    for (int i = pi - 1; i >= 0; i--) {
        const int4 o = P.pieces[i];
        if (o.x != pc.x) break;
        base += P.piece_total[i];
        if (o.y == 0) break;
    }
    const ReadDesc rd = P.reads[pc.x];
    for (int t = (pc.y >> 6) + lane; t < (pc.z + 63) >> 6; t += 64) P.tile_so[rd.tile_off + t] += base;
}
__global__ __launch_bounds__(64) void k_part_tile_bases(const SigParams P) { part_tile_bases_body(P, blockIdx.x, threadIdx.x); }

// What lies between the two event passes in ONE launch (each of the four kernels it replaces is a few microseconds of work behind
// a launch: 25 us of a 1000-read batch's 360): workgroups [0, n_off) do k_part_offsets' (partition, worker chain) pairs, the
// ones behind them k_part_tile_bases' pieces (a wavefront each); the offsets workgroup that finishes LAST -- a counter, agent-scope
// release / acquire: the others' totals are then visible to it -- goes on to k_part_slices and the slices' bounds.
__global__ __launch_bounds__(1024) void k_part_mid(const SigParams P, const uint32_t* __restrict__ pcnt, uint32_t* __restrict__ poff, const int n_part, const int n_links,
                                                   const int* __restrict__ wlink_off, uint32_t* ptotal, uint32_t* pstart, const int n_pairs, const uint32_t slice_len,
                                                   uint32_t* pfirst, uint32_t* slice_lo, uint32_t* slice_hi, const int n_off, const int n_pieces, unsigned int* done,
                                                   const ScanArgs SA, const int n_scan) {
    __shared__ int last;
    const int g = blockIdx.x;
    const int n_pcb = (n_pieces + 15) / 16;
    if (g >= n_off + n_pcb) {
        // (round 5) the scan of the reads' sample totals (k_scan: 8 us behind a launch gap): its inputs are the first event pass', it needs
        // nothing of this kernel; its workgroups come last in the grid, in order (a scan workgroup waits for the ones before it only)
        if (g - n_off - n_pcb < n_scan) scan_body(SA, g - n_off - n_pcb, n_scan);
        return;
    }
    if (g >= n_off) {
        const int pi = (g - n_off) * 16 + (int)(threadIdx.x >> 6);
        if (pi < n_pieces) part_tile_bases_body(P, pi, threadIdx.x & 63);
        return;
    }
    part_offsets_body(g % n_part, g / n_part, pcnt, poff, n_part, n_links, wlink_off, ptotal);
    __syncthreads();                                               // (thread 0 has stored the pair's total)
    if (threadIdx.x == 0) {
        const unsigned int before = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = before + 1u == (unsigned int)n_off;
        if (last) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (for the next launch: the stream orders them)
    }
    __syncthreads();
    if (!last) return;
    part_slices_body(pstart, ptotal, n_pairs, slice_len, pfirst);
    part_slice_bounds_wg(pstart, ptotal, n_pairs, slice_len, pfirst, slice_lo, slice_hi);
}
