// h_run.h -- the launch sequence of a batch (sqg_batch_run, and its two halves for range sharding)
// Host side of include/sqg.h; included by sqg_hip.hip (one translation unit with the kernels), in the order listed there.
#pragma once

// (sqg_batch_run_end), `before` / `after` being what the other ranges of the batch draw from each stream
// what the first event pass of a batch (pev_link<DW, COUNT>) reads and writes: filled the same way for the batch being run and for the
// batch behind it, whose pass may ride along with this batch's hand-out (precount below)
static void count_params(sqg_ctx* c, const sqg_batch* b, sqg_ctx::CountSet& Q, const int n_part, uint32_t* d_pcnt, SigParams& P) {
    const sqg_profile_t& p = c->cfg.profile;
    P.reads = b->d_reads; P.chain_off = b->d_chain_off; P.chain_reads = b->d_chain_reads; P.chain_order = b->d_chain_order; P.bases = b->d_bases;
    P.dwell = c->use_dwell_stream ? Q.d_dwell : nullptr; P.dwell_out = Q.d_dwell; P.seglen_out = Q.d_seglen; P.seglen = Q.d_seglen; P.tile_so = Q.d_tile_so;
    P.dmean = p.dwell_mean; P.dstd = p.dwell_std; P.pw = c->d_pow; P.err = b->d_err; P.delta_x = c->delta_x;
    P.k = c->k; P.num_kmer = c->num_kmer; P.const_sps = (int)p.dwell_mean;
    P.dwell_unbounded = c->dwell_hi > 65535.0 ? 1 : 0;
    P.dwell_pack = c->dwell_hi < 1024.0 ? 1 : 0;
    P.pieces = b->d_pieces; P.piece_total = b->d_piece_total;
    P.pcnt = d_pcnt; P.n_part = n_part; P.n_links = b->n_chains;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// A batch's launch sequence is built as a LIST of named steps by one builder per regime and queued by one executor (round 5: it was
// one 480-line function).  The regimes, by what staging decided for the batch (h_stage.h):
//   plan_bucketed     few workers, chains cut into links, the stream hand-out over bucketed events (k_part.h): k > 6 -- count, mid,
//                     scatter, hist, scan, hand-out (+ the next batch's count) -- and its one-partition case, k <= 6
//   plan_link_rows    few workers, per-link rows (round 1's path: 5^8 / 5^9 tables, many workers with few reads, SQG_NO_PART)
//   plan_chains       one workgroup per worker chain (T = K)
// in front of them plan_first_pass (stand-alone dwell kernel / constant dwells), behind them plan_samples (scan, lean / generic sample
// kernels, fix-ups).  Range sharding runs a plan in two halves (phase 1: up to the per-stream counts; phase 2: the rest).
struct RunStep { const char* name; std::function<int()> go; };    // go(): queues the step; SQG_OK or an error code
typedef std::vector<RunStep> RunPlan;

struct Run {                                                      // one sqg_batch_run / _run_begin / _run_end call
    sqg_ctx* c; sqg_batch* b; int phase; const uint32_t* before; const uint32_t* after;
    sqg_ctx::Slot* S; sqg_ctx::Slot* other; sqg_ctx::CountSet* Q;
    int n = 0, n_part = 0, dw = 0;
    bool certified = false, inline_dwell = false, direct = false, fold = false, untimed = false, wave_links = false;
    size_t n_pairs = 0, n_rows = 0;
    unsigned scan_wgs = 0, pgrid = 0;
    uint32_t *d_pcnt = nullptr, *slice_lo = nullptr, *slice_hi = nullptr, *pfirst = nullptr, *pstart = nullptr, *ptotal = nullptr;
    sqg_batch* pre_nb = nullptr;                                  // the batch whose first event pass rides along with this one's hand-out
    SigParams P; ScanArgs SA;
    hipStream_t tail = nullptr;                                   // the stream the batch's last kernel runs on
    bool seglen_zeroed = false;
};

static int run_execute(sqg_ctx* c, const RunPlan& plan) {
    for (const RunStep& st : plan) {
        const int rc = st.go();
        if (rc != SQG_OK) return rc;
        HIPCHK(c, hipGetLastError());
        const int rs = dbg_sync(c, st.name);
        if (rs != SQG_OK) return rs;
    }
    return SQG_OK;
}

// ---- who may run, and in which of the context's buffer sets ---------------------------------------------------------------------
static int run_admit(Run& R) {
    sqg_ctx* c = R.c; sqg_batch* b = R.b; const int phase = R.phase;
    if (phase != 2) skip_abandoned(c);
    if (phase == 2 ? (!b->begun || b->ran) : (b->ran || b->begun || b->seq != c->next_run)) return SQG_ESEQUENCE;
    if ((R.before == nullptr) != (R.after == nullptr)) return SQG_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    R.n = b->n;
    R.certified = c->cfg.mode == SQG_MODE_CERTIFIED;
    if (phase != 2) {
        b->run_idx = c->runs;
        for (auto it = c->staged_q.begin(); it != c->staged_q.end(); ++it) if (*it == b) { c->staged_q.erase(it); break; }
        b->fixup_launched = false;                                // (a batch that is run again after a failed run: the first attempt's report is not this one's)
        // a first pass that ran ahead is only taken if it wrote what this run reads: the set of this run index, still in the generation the
        // pass took it in, and -- one partition: the pass writes part[] itself -- the slot this batch runs in.  (The run order guarantees all
        // three; a batch that fails them is simply counted again.)
        if (b->precounted && (b->cset != (int)(b->run_idx % 3) || c->cset[b->cset].gen != b->cset_gen || (b->one && b->pre_slot != (int)(b->run_idx & 1)))) b->precounted = false;
        if (!b->precounted) { b->cset = (int)(b->run_idx % 3); b->cset_gen = ++c->cset[b->cset].gen; }   // from here on the set's buffers belong to this batch
    }
    b->slot = (int)(b->run_idx & 1);
    R.S = &c->slot[b->slot]; R.other = &c->slot[b->slot ^ 1]; R.Q = &c->cset[b->cset];
    if (phase != 2) b->slot_gen = ++R.S->gen;                   // from here on the slot's buffers belong to this batch
    return SQG_OK;
}

// ---- every buffer the sequence writes, made large enough BEFORE its first launch (a reallocation synchronises the streams) ------------
static int run_buffers(Run& R) {
    sqg_ctx* c = R.c; sqg_batch* b = R.b; const int phase = R.phase, n = R.n;
    sqg_ctx::Slot& S = *R.S; sqg_ctx::Slot& other = *R.other; sqg_ctx::CountSet& Q = *R.Q;
    int rc;
    R.n_part = (c->num_kmer + PART_SUB - 1) >> PART_SUB_BITS;
    R.n_pairs = (size_t)b->n_wchains * (size_t)R.n_part;          // (worker chain, partition)
    R.n_rows = (size_t)c->nw * (size_t)c->num_kmer;
    R.scan_wgs = n > 0 ? (unsigned)((n + SCAN_WG - 1) / SCAN_WG) : 0u;
    if (phase != 2) {
        // this slot's buffers were last used by the sample kernels of batch seq-2 (stream2)
        HIPCHK(c, hipStreamWaitEvent(c->stream, S.done, 0));
        if (b->ev_staged && hipEventQuery(b->ev_staged) != hipSuccess) HIPCHK(c, hipStreamWaitEvent(c->stream, b->ev_staged, 0));     // the batch's uploads and staging kernels (a barrier packet: not queued if they are done)
        b->other_fresh = other.reads_cap == 0 && n > 0;
        if ((rc = grow_slot(c, S, b, /*with_output=*/false))) return rc;
        if (b->other_fresh && (rc = grow_slot(c, other, b, /*with_output=*/false))) return rc;
        if ((rc = grow_cset(c, Q, b))) return rc;
    }
    // precount (plan_bucketed): the batch staged behind this one, if its first event pass can ride along with this batch's hand-out.  What
    // the pass writes is made large enough HERE (a reallocation must not happen between the kernels that advance the rows and the
    // hand-out); it is an optimisation: should a buffer not be had, the plain hand-out runs and the batch behind counts for itself.
    R.pre_nb = nullptr;
    if (phase == 0 && n > 0 && b->n_chains > 0 && b->part && b->pieces && c->lds_ordered && c->use_dwell_stream && !SQG_DEV_ENV("SQG_SEPARATE_DWELL") &&
        !c->range_mode && !c->staged_q.empty() && !SQG_DEV_ENV("SQG_NO_PRECOUNT")) {
        sqg_batch* cand = c->staged_q.front();
        if (cand->seq > b->seq && cand->staged && !cand->ran && !cand->begun && !cand->precounted && cand->part && cand->pieces && cand->one == b->one &&
            cand->n > 0 && cand->n_chains > 0) {
            sqg_ctx::CountSet& NQ = c->cset[(b->run_idx + 1) % 3];
            bool ok = grow_cset(c, NQ, cand) == SQG_OK;
            const int nx = (int)((b->run_idx + 1) & 1);          // (the counts go to the buffer of the NEXT run index: this batch's own -- which a pass that ran ahead may have filled already -- stays)
            if (ok && !cand->one) ok = ensure(c, (void**)&c->d_pcnt[nx], &c->pcnt_cap[nx], (size_t)2 * cand->n_chains * (size_t)R.n_part, sizeof(uint32_t)) == SQG_OK;
            if (ok && cand->one) ok = ensure(c, (void**)&other.d_part, &other.part_cap, (size_t)cand->n_events + PART_SLACK, sizeof(uint32_t)) == SQG_OK;
            if (ok) R.pre_nb = cand;
            else c->err.clear();
        }
    }
    if (phase != 2 && b->split && !b->part && (rc = ensure(c, (void**)&c->d_link_rows, &c->link_rows_cap, (size_t)b->n_chains * (size_t)c->num_kmer, sizeof(uint32_t)))) return rc;
    if (phase != 2 && b->part) {
        if ((rc = ensure(c, (void**)&c->d_pcnt[b->run_idx & 1], &c->pcnt_cap[b->run_idx & 1], (size_t)2 * b->n_chains * (size_t)R.n_part, sizeof(uint32_t)))) return rc;   // counts, offsets (no-op for a batch whose first pass ran ahead: the pass' launch sized it)
        if ((rc = ensure(c, (void**)&c->d_slice, &c->slice_cap, (size_t)2 * (size_t)b->max_slices + (size_t)3 * R.n_pairs + 1, sizeof(uint32_t)))) return rc;
        if ((rc = ensure(c, (void**)&c->d_phist, &c->phist_cap, (size_t)b->max_slices * (size_t)PART_SUB, sizeof(uint32_t)))) return rc;
    }
    if (phase == 1 && (rc = ensure(c, (void**)&c->d_xcounts, &c->xcounts_cap, R.n_rows, sizeof(uint32_t)))) return rc;
    if (n > 0 && phase != 1) {
        const size_t cap0 = c->scan_part_cap;
        if ((rc = ensure(c, (void**)&c->d_scan_part, &c->scan_part_cap, (size_t)2 * R.scan_wgs, sizeof(unsigned long long)))) return rc;
        if (c->scan_part_cap != cap0)                         // tickets start at 1: a fresh array must not hold one by accident
            HIPCHK(c, hipMemsetAsync(c->d_scan_part, 0, c->scan_part_cap * sizeof(unsigned long long), c->stream));
    }
    return SQG_OK;
}

// ---- what the kernels are told (SigParams; the sample kernels' output fields follow in plan_samples) --------------------------------
static void run_params(Run& R) {
    sqg_ctx* c = R.c; sqg_batch* b = R.b; const int phase = R.phase, n = R.n;
    sqg_ctx::Slot& S = *R.S; sqg_ctx::CountSet& Q = *R.Q;
    const sqg_profile_t& p = c->cfg.profile;
    // Dwell draws are made inside k_events (SQG_SEPARATE_DWELL=1 keeps the stand-alone k_dwell for A/B runs).
    R.inline_dwell = c->use_dwell_stream && !SQG_DEV_ENV("SQG_SEPARATE_DWELL");
    R.direct = c->num_kmer <= 4096;                             // the worker's whole row of stream states fits in LDS
    R.dw = R.inline_dwell ? (R.certified && c->dwell_hi < 1.0e6 ? 1 : 2) : 0;
    R.wave_links = b->pieces;                                   // (decided at staging: the links are then runs of pieces of reads)
    const int n_part = R.n_part;
    R.d_pcnt = c->d_pcnt[b->run_idx & 1];
    R.pgrid = (unsigned)b->max_slices;
    R.slice_lo = c->d_slice;
    R.slice_hi = R.slice_lo ? R.slice_lo + (size_t)b->max_slices : nullptr;
    R.pfirst = R.slice_hi ? R.slice_hi + (size_t)b->max_slices : nullptr;      // pfirst[n_pairs]: the number of slices
    R.pstart = R.pfirst ? R.pfirst + R.n_pairs + 1 : nullptr;
    R.ptotal = R.pstart ? R.pstart + R.n_pairs : nullptr;
    SigParams& P = R.P;
    memset(&P, 0, sizeof P);
    P.link_rows = (b->split && !b->part) ? c->d_link_rows : nullptr;
    P.part = S.d_part; P.part_state = b->part ? S.d_part_state : nullptr;
    P.poff = b->one ? b->d_link_slot : R.d_pcnt ? R.d_pcnt + (size_t)b->n_chains * n_part : nullptr;
    P.link_q = b->d_link_q;
    P.one = b->one ? 1 : 0;
    count_params(c, b, Q, n_part, R.d_pcnt, P);
    P.pstart = R.pstart;
    P.sig_off = S.d_sigoff; P.model = c->d_model; P.rows = c->d_rows;
    P.seed_base = canon((long long)c->cfg.seed + (long long)c->wlo * ((long long)c->num_kmer + 10)); P.seed_step = canon((long long)c->num_kmer + 10);
    P.dig = p.digitisation; P.range = p.range; P.kd = p.digitisation / p.range;
    P.thr_all = c->thr_all;
    P.meth = (c->cfg.flags & SQG_METH) ? 1 : 0; P.num_kmer_pad = n_part * PART_SUB;
    P.meth_top = 1; for (int i = 1; i < c->k; i++) P.meth_top *= 5u;
    P.use_streams = c->use_kmer_streams ? 1 : 0;
    P.rna = (c->cfg.flags & SQG_RNA) ? 1 : 0;
    P.evrec = S.d_evrec;
    // bucketed hand-out with the wavefront-per-link passes: 4 B per event between the scatter pass and the sample kernels (k_part_events.h)
    if (b->part && b->pieces && !b->one) { P.evrec32 = reinterpret_cast<uint32_t*>(S.d_evrec); P.lbase = S.d_lbase; P.tile_link = S.d_tile_link; }
    P.tile_read = b->d_tile_read; P.stile_read = b->d_stile_read;
    // what k_items needs (it may run inside k_part_hist)
    const bool rna_prefix = (c->cfg.flags & SQG_RNA) && (c->cfg.flags & SQG_PREFIX);
    P.shift_len = rna_prefix ? (int)strlen(kAdaptorRna) * (int)p.dwell_mean : 0;
    {   // int16_t off = 30*dig/range (src/genread.c:82): double -> int16 as the CPU does it
        const double v = 30 * p.digitisation / p.range;
        int32_t t = (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : (int32_t)0x80000000u;
        P.shift = (int)(int16_t)(uint16_t)((uint32_t)t & 0xffffu);
    }
    P.slow_count = S.d_fix_count + 1; P.items = S.d_items; P.lean_epl = c->lean_epl;
    P.slow_tiles = (R.certified && c->use_kmer_streams) ? S.d_slow : nullptr;
    // (round 5) bucketed hand-out, one launch sequence (no range sharding): the scan of the reads' totals runs as extra workgroups of
    // k_part_mid and the lean kernel's work items are prepared by extra workgroups of k_part_hist -- two launches and their gaps less per
    // batch (36 us of a 3.6-ms step).  SQG_NO_FOLD=1 (development build) keeps the two kernels for A/B runs.
    R.fold = phase == 0 && n > 0 && b->n_chains > 0 && b->part && !b->one && b->pieces && R.certified && c->use_kmer_streams &&
             (c->lean_epl < 4 || SQG_LEAN_ITEMS4) && !SQG_DEV_ENV("SQG_MID_SPLIT") && !SQG_DEV_ENV("SQG_NO_FOLD");
    memset(&R.SA, 0, sizeof R.SA);
    if (n > 0 && phase != 1) {
        // the scan also writes the offsets through the batch's pinned host mapping (no copy between kernels)
        ScanArgs& SA = R.SA;
        SA.seglen = Q.d_seglen; SA.n_reads = n; SA.sig_off = S.d_sigoff; SA.host_off = b->h_sigoff_dev; SA.err = b->d_err; SA.counters = S.d_fix_count;
        SA.part = c->d_scan_part; SA.ticket = ++c->scan_tickets; SA.shard_counters = S.d_fix_sh_count;   // (a ticket per launch, also after a failed run)
    }
    // the phase boundaries are recorded events: barrier packets between the kernels, 3-8 us of idle GPU each (tools/ext_event_probe.hip;
    // events attached to a launch cost more, not less) -- 1.2 % of a 16384-read step, 5 % of a 1000-read one.  sqg_set_phase_timing
    // says how many batches carry them.
    if (phase != 2) b->untimed = c->phase_timing_every <= 0 || (b->run_idx % c->phase_timing_every) != 0;
    R.untimed = b->untimed;
}

// k_events, one workgroup per chain / link: dw as in the kernel (0: dwell from memory, 1: drawn, certified, 2: drawn in FP64); hist: the
// counting form (per-link rows: samples per (link, k-mer); bucketed: events per (link, partition))
static void launch_events(Run& R, const int dw, const bool hist) {
    sqg_ctx* c = R.c; sqg_batch* b = R.b; const SigParams& P = R.P;
    constexpr int NT = SQG_EVENT_THREADS, NT_WIDE = 1024;
    // few chains (the reference's default -K 1000 with one worker per read): a chain is a sequence of segments, each with
    // its barriers and LDS round trips, and there are not enough chains to hide them -- 1024 threads per chain walk it in a
    // quarter of the steps
    const int wide_max = dev_env_int(SQG_DEV_ENV("SQG_EVENTS_WIDE_MAX"), 1200);   // A/B knob
    const dim3 g((unsigned)b->n_chains);
    const bool wide = !hist && b->n_chains <= wide_max, direct = R.direct;
#define EVL(N, D, W, H, PT) hipLaunchKernelGGL((k_events<N, D, W, SQG_EVENT_EPT, H, PT>), g, dim3(N), 0, c->stream, P)
#define EVD(N, D, H, PT) do { if (dw == 0) EVL(N, D, 0, H, PT); else if (dw == 1) EVL(N, D, 1, H, PT); else EVL(N, D, 2, H, PT); } while (0)
    if (b->part) { if (hist) EVD(NT, false, true, true); else EVD(NT, false, false, true); }
    else if (wide) { if (direct) EVD(NT_WIDE, true, false, false); else EVD(NT_WIDE, false, false, false); }
    else if (direct) { if (hist) EVD(NT, true, true, false); else EVD(NT, true, false, false); }
    else { if (hist) EVD(NT, false, true, false); else EVD(NT, false, false, false); }
#undef EVD
#undef EVL
}
// the event passes of the bucketed hand-out: one wavefront per link, ordered LDS atomics (k_part_events.h); the workgroup-per-link passes
// of k_events stay for the 5-letter alphabet and for devices that do not pass the order check.  count: the first pass (with the dwell
// draws; one partition: the ONLY pass), else the scatter pass
static void launch_part_events(Run& R, const int dwm, const bool count) {
    sqg_ctx* c = R.c; sqg_batch* b = R.b; const SigParams& P = R.P;
    if (!R.wave_links) { launch_events(R, dwm, count); return; }
    const dim3 g((unsigned)((b->n_chains + PEV_WAVES - 1) / PEV_WAVES)), t(64 * PEV_WAVES);
    if (!count) hipLaunchKernelGGL((k_part_events<0, PEV_SCATTER>), dim3((unsigned)((b->n_chains + PEV_WAVES_SCATTER - 1) / PEV_WAVES_SCATTER)), dim3(64 * PEV_WAVES_SCATTER), 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
    else if (b->one) {                                            // one partition: the only event pass
        if (dwm == 0) hipLaunchKernelGGL((k_part_events<0, PEV_ONE>), g, t, 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
        else if (dwm == 1) hipLaunchKernelGGL((k_part_events<1, PEV_ONE>), g, t, 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
        else hipLaunchKernelGGL((k_part_events<2, PEV_ONE>), g, t, 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
    }
    else if (dwm == 0) hipLaunchKernelGGL((k_part_events<0, PEV_COUNT>), g, t, 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
    else if (dwm == 1) hipLaunchKernelGGL((k_part_events<1, PEV_COUNT>), g, t, 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
    else hipLaunchKernelGGL((k_part_events<2, PEV_COUNT>), g, t, 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
}
// Placement calibration.  The scatter pass writes two arrays at once -- part[] in 64-byte runs at ~4000 places, evrec32 as one
// stream per link -- and runs in one of several modes (825 / 890 / 1000-1130 us per 32768 10-kb reads) for the life of a slot's buffers: which one is
// decided by where hipMalloc put `evrec` relative to the slot's part[] (physical pages: nothing a virtual address shows, and no probe kernel
// reproduces the pass' own mode; profiles/r05_summary.md).  So the pass itself is the probe: over the first twelve large batches of a context it is
// timed between two events, and each slot tries PLACE_TRIES more allocations of `evrec`, one per run, keeping the one the pass ran fastest on
// (the previous one stays allocated until the candidate has been measured: never worse than where it started).  Costs those batches a host
// synchronisation each; changes no result (the scatter pass writes every entry the sample kernels read).  SQG_NO_PLACE=1 (development build) turns it off.
#define PLACE_TRIES 3
static int place_calibrate(Run& R) {
    sqg_ctx* c = R.c; sqg_batch* b = R.b; sqg_ctx::Slot& S = *R.S;
    static const bool off = SQG_DEV_ENV("SQG_NO_PLACE") != nullptr;
    if (off || c->cal_runs_left <= 0 || !R.wave_links || !R.P.evrec32 || b->n_events < (1ll << 24)) return SQG_OK;   // (small batches: microseconds, nothing to gain)
    if (!S.cal_a) { HIPCHK(c, hipEventCreate(&S.cal_a)); HIPCHK(c, hipEventCreate(&S.cal_b)); }
    if (S.cal_pending) {
        float ms = 0.f;
        HIPCHK(c, hipEventSynchronize(S.cal_b));
        HIPCHK(c, hipEventElapsedTime(&ms, S.cal_a, S.cal_b));
        S.cal_pending = false;
        float ps = ms * 1.0e9f / (float)S.cal_events;                // picoseconds per event: ~2.5-2.7 in the fast modes, 3.0-3.4 in the slow ones
        const char* what = "";
        if (S.cal_prev) {                                             // a candidate was measured: keep the better of the two
            if (ps > S.cal_prev_ps) {
                HIPCHK(c, hipEventSynchronize(S.done));               // (the slot's previous batch has read the candidate)
                HIPCHK(c, hipFree(S.d_evrec));
                S.d_evrec = S.cal_prev; ps = S.cal_prev_ps; what = ": slower than the allocation before it, which is taken back";
                R.P.evrec = S.d_evrec; R.P.evrec32 = reinterpret_cast<uint32_t*>(S.d_evrec);
            } else { HIPCHK(c, hipFree(S.cal_prev)); what = ": kept"; }
            S.cal_prev = nullptr;
        }
        S.cal_ps = ps;
        if (getenv("SQG_VERBOSE")) fprintf(stderr, "[sqg] placement: slot %d scatter pass %.0f us = %.3f ps per event%s\n", b->slot, ms * 1.0e3f, ms * 1.0e9f / (float)S.cal_events, what);
        if (S.cal_tries < PLACE_TRIES && c->cal_runs_left >= 3 && S.d_evrec && S.evrec_cap) {   // (>= 3: this slot runs once more inside the calibration, to measure the candidate)
            uint2* fresh = nullptr;
            if (hipMalloc(&fresh, S.evrec_cap * sizeof(uint2)) == hipSuccess) {         // (no memory for a second copy: stay where we are)
                S.cal_prev = S.d_evrec; S.cal_prev_ps = S.cal_ps;
                S.d_evrec = fresh; S.cal_tries++;
                R.P.evrec = S.d_evrec; R.P.evrec32 = reinterpret_cast<uint32_t*>(S.d_evrec);
            } else (void)hipGetLastError();
        }
    }
    c->cal_runs_left--;
    return SQG_OK;
}

// every worker's row moves past the whole batch, all ranges (range sharding)
static void launch_rows_advance(Run& R) {
    sqg_ctx* c = R.c;
    const dim3 ag((unsigned)((R.n_rows + 255) / 256));
    if (R.direct) hipLaunchKernelGGL(k_rows_advance<true>, ag, dim3(256), 0, c->stream, c->d_rows, c->d_pow, R.n_rows, R.before, c->d_xcounts, R.after);
    else hipLaunchKernelGGL(k_rows_advance<false>, ag, dim3(256), 0, c->stream, c->d_rows, c->d_pow, R.n_rows, R.before, c->d_xcounts, R.after);
}

// ---- in front of every regime: the rows' counts kept in range, the timing event, dwells that are not drawn inside the event kernels ----
static void plan_first_pass(Run& R, RunPlan& plan) {
    sqg_ctx* c = R.c; sqg_batch* b = R.b; const int n = R.n;
    if (R.phase == 2) return;
    if (!R.direct && c->use_kmer_streams && n > 0) {
        // the rows count samples in 32 bits; only the count mod (M-1)/2 matters (range mode: the other ranges' counts are not
        // known here, so the rows are reduced before every batch)
        const double bnd = (double)b->max_wchain_ev * c->dwell_hi;
        if (b->part && !c->range_mode) {
            // k_part_scan reduces the rows itself and reports a stream that is asked for >= 2^32 samples by one batch
            c->row_bound = std::min((double)LCG_ORD2 + bnd, 4294967295.0) - bnd;
        } else if (c->range_mode || c->row_bound + bnd >= 4294967295.0) {
            plan.push_back({"k_rows_normalize", [&R]() -> int {
                const size_t nrow = (size_t)R.c->nw * (size_t)R.c->num_kmer;
                hipLaunchKernelGGL(k_rows_normalize, dim3((unsigned)((nrow + 255) / 256)), dim3(256), 0, R.c->stream, R.c->d_rows, nrow);
                return SQG_OK; }});
            c->row_bound = (double)LCG_ORD2;
        }
        c->row_bound += bnd;
    }
    if (!R.untimed) plan.push_back({"event: start", [&R]() -> int { HIPCHK(R.c, hipEventRecord(R.b->ev[0], R.c->stream)); return SQG_OK; }});
    if (n > 0 && c->use_dwell_stream && !R.inline_dwell)
        plan.push_back({"k_dwell", [&R]() -> int {
            sqg_ctx* c = R.c; sqg_batch* b = R.b; const sqg_profile_t& p = c->cfg.profile;
            HIPCHK(c, hipMemsetAsync(R.Q->d_seglen, 0, (size_t)2 * R.n * sizeof(unsigned long long), c->stream));
            const long long nblk = (b->n_events + DW_EPB - 1) / DW_EPB;
            if (nblk > 0) {
                if (R.certified)
                    hipLaunchKernelGGL(k_dwell<1>, dim3((unsigned)nblk), dim3(256), 0, c->stream, b->d_reads, R.n, b->d_blk_read,
                                       b->n_events, c->d_pow, p.dwell_mean, p.dwell_std, c->delta_x, R.Q->d_dwell, R.Q->d_seglen, b->d_err);
                else
                    hipLaunchKernelGGL(k_dwell<0>, dim3((unsigned)nblk), dim3(256), 0, c->stream, b->d_reads, R.n, b->d_blk_read,
                                       b->n_events, c->d_pow, p.dwell_mean, p.dwell_std, 0.f, R.Q->d_dwell, R.Q->d_seglen, b->d_err);
            }
            return SQG_OK; }});
    else if (n > 0 && !c->use_dwell_stream)
        plan.push_back({"upload: constant dwells' totals", [&R]() -> int {
            HIPCHK(R.c, hipMemcpyAsync(R.Q->d_seglen, R.b->seglen_host.data(), (size_t)2 * R.n * sizeof(unsigned long long), hipMemcpyHostToDevice, R.c->stream));
            return SQG_OK; }});
    b->dwell_timed = c->use_dwell_stream && !R.inline_dwell && !R.untimed;        // stand-alone k_dwell (A/B runs): two more timing events
    if (b->dwell_timed)
        plan.push_back({"events: dwell kernel done", [&R]() -> int { HIPCHK(R.c, hipEventRecord(R.b->ev[1], R.c->stream)); HIPCHK(R.c, hipEventRecord(R.b->ev[2], R.c->stream)); return SQG_OK; }});
}

// ---- few workers, the hand-out over events bucketed by the top bits of the rank (k_part.h); one partition for k <= 6 ---------------------
static void plan_bucketed(Run& R, RunPlan& plan) {
    sqg_ctx* c = R.c; sqg_batch* b = R.b; const int phase = R.phase;
    if (phase != 2) {
        if (!b->precounted)                                       // (else: the pass ran next to the hand-out of the batch before this one)
            plan.push_back({b->one ? "k_part_events<one>: the event pass" : "k_part_events<count>: dwell draws, events per (link, partition)", [&R]() -> int {
                sqg_ctx* c = R.c; sqg_batch* b = R.b; sqg_ctx::CountSet& Q = *R.Q;
                if (b->split_reads && R.dw && Q.seglen_dirty > 0) {   // pieces add up; (usually the set's previous batch has left the words zero: k_fixup)
                    HIPCHK(c, hipMemsetAsync(Q.d_seglen, 0, (size_t)2 * std::max<size_t>((size_t)R.n, Q.seglen_dirty) * sizeof(unsigned long long), c->stream));
                    Q.seglen_dirty = 0;
                }
                if (b->split_reads && R.dw) Q.seglen_dirty = (size_t)R.n;   // (until this batch's k_fixup is queued: a run that fails half-way leaves them dirty)
                launch_part_events(R, R.dw, true);
                return SQG_OK; }});
        const bool mid_split = SQG_DEV_ENV("SQG_MID_SPLIT") != nullptr;   // A/B: the four small kernels between the passes, one by one
        if (b->split_reads && (b->one || mid_split))
            plan.push_back({"k_part_tile_bases", [&R]() -> int { hipLaunchKernelGGL(k_part_tile_bases, dim3((unsigned)R.b->n_pieces), dim3(64), 0, R.c->stream, R.P); return SQG_OK; }});
        if (b->one) {
            // one partition: the pass above has written part[] in chain order; the slices of every worker chain's events
            plan.push_back({"k_part_slices", [&R]() -> int {
                hipLaunchKernelGGL(k_part_slices, dim3(1), dim3(1024), 0, R.c->stream, R.pstart, R.b->d_wchain_total, (int)R.n_pairs, R.b->slice_len, R.pfirst, R.slice_lo, R.slice_hi);
                return SQG_OK; }});
        } else {
            if (mid_split)
                plan.push_back({"k_part_offsets + k_part_slices + k_part_slice_bounds", [&R]() -> int {
                    sqg_ctx* c = R.c; sqg_batch* b = R.b;
                    hipLaunchKernelGGL(k_part_offsets, dim3((unsigned)R.n_part, (unsigned)b->n_wchains), dim3(1024), 0, c->stream, R.d_pcnt,
                                       R.d_pcnt + (size_t)b->n_chains * R.n_part, R.n_part, b->n_chains, b->d_wlink_off, R.ptotal);
                    hipLaunchKernelGGL(k_part_slices, dim3(1), dim3(1024), 0, c->stream, R.pstart, R.ptotal, (int)R.n_pairs, b->slice_len, R.pfirst, nullptr, nullptr);
                    hipLaunchKernelGGL(k_part_slice_bounds, dim3((R.pgrid + 255) / 256), dim3(256), 0, c->stream, R.pstart, R.ptotal, (int)R.n_pairs, b->slice_len, R.pfirst, R.slice_lo, R.slice_hi);
                    return SQG_OK; }});
            else
                // offsets per (partition, worker chain), the tile offsets of split reads, the slices and their bounds -- and (fold) the scan of
                // the reads' totals: one launch
                plan.push_back({R.fold ? "k_part_mid (+ the scan)" : "k_part_mid", [&R]() -> int {
                    sqg_ctx* c = R.c; sqg_batch* b = R.b;
                    const int n_off = (int)R.n_pairs, n_pc = b->split_reads ? b->n_pieces : 0, n_sc = R.fold ? (int)R.scan_wgs : 0;
                    hipLaunchKernelGGL(k_part_mid, dim3((unsigned)(n_off + (n_pc + 15) / 16 + n_sc)), dim3(1024), 0, c->stream, R.P, R.d_pcnt,
                                       R.d_pcnt + (size_t)b->n_chains * R.n_part, R.n_part, b->n_chains, b->d_wlink_off, R.ptotal, R.pstart, (int)R.n_pairs, b->slice_len,
                                       R.pfirst, R.slice_lo, R.slice_hi, n_off, n_pc, c->d_mid_done, R.SA, n_sc);
                    return SQG_OK; }});
            plan.push_back({"k_part_events<scatter>: every event to its slot", [&R]() -> int {             // (the dwell is in memory now)
                const int left = R.c->cal_runs_left;
                int rc = place_calibrate(R); if (rc) return rc;
                const bool timed = R.c->cal_runs_left < left;         // a calibration run: the pass between two events
                if (timed) HIPCHK(R.c, hipEventRecord(R.S->cal_a, R.c->stream));
                launch_part_events(R, 0, false);
                if (timed) { HIPCHK(R.c, hipEventRecord(R.S->cal_b, R.c->stream)); R.S->cal_pending = true; R.S->cal_events = R.b->n_events; }
                return SQG_OK; }});
        }
        plan.push_back({R.fold ? "k_part_hist (+ the lean kernel's work items)" : "k_part_hist", [&R]() -> int {
            const int n_st = R.fold ? (int)R.b->n_stiles : 0;
            hipLaunchKernelGGL(k_part_hist, dim3(R.pgrid + (unsigned)((n_st + 255) / 256)), dim3(256), 0, R.c->stream, R.S->d_part, R.slice_lo, R.slice_hi, R.pfirst + R.n_pairs,
                               R.c->d_phist, R.P, n_st, R.pgrid);
            return SQG_OK; }});
        if (phase == 1)                                           // range sharding: what this range draws per stream, for the exchange
            plan.push_back({"k_part_totals", [&R]() -> int {
                sqg_ctx* c = R.c;
                const dim3 sg((unsigned)((c->num_kmer + 255) / 256), (unsigned)R.b->n_wchains);
                HIPCHK(c, hipMemsetAsync(c->d_xcounts, 0, R.n_rows * sizeof(uint32_t), c->stream));
                hipLaunchKernelGGL(k_part_totals, sg, dim3(256), 0, c->stream, c->d_phist, c->num_kmer, R.n_part, R.pfirst, R.b->d_wlink_worker, c->d_xcounts);
                return SQG_OK; }});
    }
    if (phase == 1) return;
    plan.push_back({"k_part_scan", [&R]() -> int {
        sqg_ctx* c = R.c; sqg_batch* b = R.b;
        const dim3 pg((unsigned)((c->num_kmer + 63) / 64), (unsigned)b->n_wchains);
#define SCANL(R_, G_) hipLaunchKernelGGL((k_part_scan<R_, G_>), dim3((unsigned)((c->num_kmer + R_ - 1) / R_), (unsigned)b->n_wchains), dim3(R_ * G_), 0, c->stream, \
                                        c->d_phist, c->d_rows, c->num_kmer, R.n_part, R.pfirst, b->d_wlink_worker, R.before, c->d_pow, R.P.seed_base, R.P.seed_step, R.direct ? 1 : 0, b->d_err)
        const int scan_g4 = dev_env_int(SQG_DEV_ENV("SQG_SCAN_G4"), 32);   // A/B knob
        if ((long long)b->max_slices <= 8 * (long long)R.n_pairs) SCANL(256, 1);       // a slice or two per pair: one thread per rank walks them
        else if ((long long)b->max_slices <= scan_g4 * (long long)R.n_pairs && (size_t)pg.x * pg.y >= 512) SCANL(64, 4);   // a dozen (small batches): 16 runs would be 16 x the wavefronts, most of them idle
        else if ((size_t)pg.x * pg.y >= 512) SCANL(64, 16);
        else SCANL(16, 64);
#undef SCANL
        return SQG_OK; }});
    if (R.before) plan.push_back({"k_rows_advance", [&R]() -> int { launch_rows_advance(R); return SQG_OK; }});
    // the hand-out; precount: with the first event pass of the batch staged behind this one (k_part_hand_count, k_part_events.h), chosen --
    // and its buffers made -- in run_buffers
    sqg_batch* nb = (R.pre_nb && R.wave_links && R.dw != 0) ? R.pre_nb : nullptr;
    if (nb)
        plan.push_back({"k_part_hand_count: the hand-out + the next batch's first event pass", [&R, nb]() -> int {
            sqg_ctx* c = R.c; sqg_batch* b = R.b; sqg_ctx::Slot& S = *R.S; sqg_ctx::Slot& other = *R.other;
            const int order_fault = SQG_DEV_ENV("SQG_TEST_ORDER_FAULT") ? 1 : 0;         // (tests: the per-batch order check has to fire)
            const int ncs = (int)((b->run_idx + 1) % 3);
            sqg_ctx::CountSet& NQ = c->cset[ncs];
            // (one partition, k <= 6: the pass writes part[] of the next batch's slot -- the other one, which the fix-ups of the batch
            // before this one (fix_stream) may still be reading)
            if (nb->one) HIPCHK(c, hipStreamWaitEvent(c->stream, other.done, 0));
            if (nb->ev_staged && hipEventQuery(nb->ev_staged) != hipSuccess) HIPCHK(c, hipStreamWaitEvent(c->stream, nb->ev_staged, 0));
            if (nb->split_reads && NQ.seglen_dirty > 0) {
                HIPCHK(c, hipMemsetAsync(NQ.d_seglen, 0, (size_t)2 * std::max<size_t>((size_t)nb->n, NQ.seglen_dirty) * sizeof(unsigned long long), c->stream));
                NQ.seglen_dirty = 0;
            }
            if (nb->split_reads) NQ.seglen_dirty = (size_t)nb->n;
            SigParams Pn;
            memset(&Pn, 0, sizeof Pn);
            count_params(c, nb, NQ, R.n_part, c->d_pcnt[(b->run_idx + 1) & 1], Pn);
            if (nb->one) { Pn.one = 1; Pn.part = other.d_part; Pn.poff = nb->d_link_slot; }
            const dim3 fg((unsigned)(dev_env_int(SQG_DEV_ENV("SQG_PHC_GRID"), 4) * c->num_cu)), ft(64 * (1 + PHC_COUNT_WAVES));   // (A/B: workgroups per CU)
            // (development build, timing experiments: 1 -- the fused launch hands out only, the next batch's pass follows as a launch
            // of its own; 2 -- the plain hand-out first, the fused launch counts only)
            const int phc_abl = dev_env_int(SQG_DEV_ENV("SQG_PHC_ABL"), 0);
            const int dw = R.dw;
            if (phc_abl == 2) hipLaunchKernelGGL(k_part_hand_ord, dim3(R.pgrid), dim3(64), 0, c->stream, S.d_part, S.d_part_state, R.slice_lo, R.slice_hi, R.pfirst + R.n_pairs, c->d_phist, c->d_pow, b->d_err, order_fault);
            const uint32_t* const ns_ptr = phc_abl == 2 ? c->d_zero : R.pfirst + R.n_pairs;
            const int nl_fused = phc_abl == 1 ? 0 : nb->n_chains;
#define PHCL(D_, M_) hipLaunchKernelGGL((k_part_hand_count<D_, M_>), fg, ft, 0, c->stream, S.d_part, S.d_part_state, R.slice_lo, R.slice_hi, ns_ptr, c->d_phist, c->d_pow, \
                                    b->d_err, order_fault, Pn, nl_fused, (uint32_t)nb->n_events, c->num_cu)
            if (nb->one) { if (dw == 1) PHCL(1, PEV_ONE); else PHCL(2, PEV_ONE); }
            else { if (dw == 1) PHCL(1, PEV_COUNT); else PHCL(2, PEV_COUNT); }
#undef PHCL
            if (phc_abl == 1) {
                const dim3 g1((unsigned)((nb->n_chains + PEV_WAVES - 1) / PEV_WAVES)), t1(64 * PEV_WAVES);
                if (nb->one) { if (dw == 1) hipLaunchKernelGGL((k_part_events<1, PEV_ONE>), g1, t1, 0, c->stream, Pn, nb->n_chains, (uint32_t)nb->n_events);
                               else hipLaunchKernelGGL((k_part_events<2, PEV_ONE>), g1, t1, 0, c->stream, Pn, nb->n_chains, (uint32_t)nb->n_events); }
                else if (dw == 1) hipLaunchKernelGGL((k_part_events<1, PEV_COUNT>), g1, t1, 0, c->stream, Pn, nb->n_chains, (uint32_t)nb->n_events);
                else hipLaunchKernelGGL((k_part_events<2, PEV_COUNT>), g1, t1, 0, c->stream, Pn, nb->n_chains, (uint32_t)nb->n_events);
            }
            nb->precounted = true; nb->cset = ncs; nb->cset_gen = ++NQ.gen; nb->pre_slot = b->slot ^ 1;
            b->carried_precount = true;                           // (sqg_get_timing: this batch's event side holds the successor's first pass)
            return SQG_OK; }});
    else
        plan.push_back({c->lds_ordered ? "k_part_hand_ord" : "k_part_hand (claims)", [&R]() -> int {
            sqg_ctx* c = R.c; sqg_batch* b = R.b; sqg_ctx::Slot& S = *R.S;
            const int order_fault = SQG_DEV_ENV("SQG_TEST_ORDER_FAULT") ? 1 : 0;
            if (c->lds_ordered) hipLaunchKernelGGL(k_part_hand_ord, dim3(R.pgrid), dim3(64), 0, c->stream, S.d_part, S.d_part_state, R.slice_lo, R.slice_hi, R.pfirst + R.n_pairs, c->d_phist, c->d_pow, b->d_err, order_fault);
            else if (c->dwell_hi >= (double)PART_JT) hipLaunchKernelGGL(k_part_hand<true>, dim3(R.pgrid), dim3(64), 0, c->stream, S.d_part, S.d_part_state, R.slice_lo, R.slice_hi, R.pfirst + R.n_pairs, c->d_phist, c->d_pow, (uint32_t)b->n_events);
            else hipLaunchKernelGGL(k_part_hand<false>, dim3(R.pgrid), dim3(64), 0, c->stream, S.d_part, S.d_part_state, R.slice_lo, R.slice_hi, R.pfirst + R.n_pairs, c->d_phist, c->d_pow, (uint32_t)b->n_events);
            return SQG_OK; }});
}

// ---- few workers, per-link rows: samples per (link, k-mer), a prefix over a worker's links, then the ordinary k_events per link ------------
static void plan_link_rows(Run& R, RunPlan& plan) {
    sqg_ctx* c = R.c; const int phase = R.phase;
    if (phase != 2) {
        plan.push_back({"k_events<hist>: samples per (link, k-mer)", [&R]() -> int {
            sqg_ctx* c = R.c; sqg_batch* b = R.b;
            if (!R.direct) HIPCHK(c, hipMemsetAsync(c->d_link_rows, 0, (size_t)b->n_chains * (size_t)c->num_kmer * sizeof(uint32_t), c->stream));
            launch_events(R, R.dw, true);
            return SQG_OK; }});
        if (phase == 1)                                           // ... summed per worker for the exchange
            plan.push_back({"k_link_totals", [&R]() -> int {
                sqg_ctx* c = R.c; sqg_batch* b = R.b;
                const dim3 pg((unsigned)((c->num_kmer + 63) / 64), (unsigned)b->n_wchains);
                HIPCHK(c, hipMemsetAsync(c->d_xcounts, 0, R.n_rows * sizeof(uint32_t), c->stream));
                hipLaunchKernelGGL(k_link_totals, pg, dim3(1024), 0, c->stream, R.P, b->d_wlink_off, b->d_wlink_worker, c->d_xcounts);
                return SQG_OK; }});
    }
    if (phase == 1) return;
    // ... then each link's view of its worker's streams (the worker's own row is read in front of the barrier its first group rewrites it
    // behind: k_events.h)
    plan.push_back({"k_link_prefix", [&R]() -> int {
        sqg_ctx* c = R.c; sqg_batch* b = R.b;
        const dim3 pg((unsigned)((c->num_kmer + 63) / 64), (unsigned)b->n_wchains);
        if (R.direct) hipLaunchKernelGGL(k_link_prefix<true>, pg, dim3(1024), 0, c->stream, R.P, b->d_wlink_off, b->d_wlink_worker, R.before);
        else hipLaunchKernelGGL(k_link_prefix<false>, pg, dim3(1024), 0, c->stream, R.P, b->d_wlink_off, b->d_wlink_worker, R.before);
        return SQG_OK; }});
    if (R.before) plan.push_back({"k_rows_advance", [&R]() -> int { launch_rows_advance(R); return SQG_OK; }});
    plan.push_back({"k_events: the links", [&R]() -> int { launch_events(R, 0, false); return SQG_OK; }});   // (the dwell is in memory now)
    (void)c;
}

// ---- one workgroup per worker chain (T = K; few chains that are not cut) ----------------------------------------------------------------
static void plan_chains(Run& R, RunPlan& plan) {
    if (R.phase == 1) return;
    plan.push_back({"k_events", [&R]() -> int { launch_events(R, R.dw, false); return SQG_OK; }});
}

// ---- the sample side: scan (unless it ran inside k_part_mid), lean / generic sample kernels, FP64 fix-ups ---------------------------------
static int plan_samples(Run& R, RunPlan& plan) {
    sqg_ctx* c = R.c; sqg_batch* b = R.b; const int n = R.n;
    sqg_ctx::Slot& S = *R.S; sqg_ctx::CountSet& Q = *R.Q; SigParams& P = R.P;
    R.tail = c->stream2;
    R.seglen_zeroed = false;
    auto split_streams = [&R]() -> int {                          // event side done: the sample kernels may start on their own stream (SQG_OVERLAP), next to the next batch's event side
        if (R.c->stream2 != R.c->stream) { HIPCHK(R.c, hipEventRecord(R.b->ev[7], R.c->stream)); HIPCHK(R.c, hipStreamWaitEvent(R.c->stream2, R.b->ev[7], 0)); }
        return SQG_OK; };
    if (n > 0 && b->n_chains > 0) {
        P.sig = S.d_sig; P.fix = S.d_fix; P.fix_count = S.d_fix_count;
        P.fix_sh = S.d_fix_sh; P.fix_sh_cap = S.fix_sh_per; P.fix_sh_count = S.d_fix_sh_count; P.fix_sh_stat = S.d_fix_count + 4; P.host_res = reinterpret_cast<unsigned int*>(b->h_sigoff_dev + (b->h_n - SQG_HRES_LL)); P.fix_tag = (int)(++c->fix_tickets & 0x3fffffffull) + 1;   // (a tag per launch, also when a batch is run again after a failed run: stale entries of the first attempt must not match)
        P.fix_cap = (unsigned int)std::min<size_t>(S.fix_cap, 0xffffffffu);
        if (b->run_idx < 8 && getenv("SQG_VERBOSE"))
            fprintf(stderr, "[sqg] batch %lld slot %d: sig %p part %p evrec %p part_state %p dwell %p bases %p seglen %p\n", (long long)b->run_idx, b->slot,
                    (void*)S.d_sig, (void*)S.d_part, (void*)S.d_evrec, (void*)S.d_part_state, (void*)Q.d_dwell, (void*)b->d_bases, (void*)Q.d_seglen);
        if (b->run_idx < 8 && getenv("SQG_VERBOSE"))
            fprintf(stderr, "[sqg] batch %lld: %d reads, %lld events, %d links in %d worker chains, %d pieces, %lld slices of %u events at most%s\n", (long long)b->run_idx, n, (long long)b->n_events,
                    b->n_chains, b->n_wchains, b->n_pieces, (long long)b->max_slices, b->slice_len, b->precounted ? "; first event pass: ran ahead, with the previous batch's hand-out" : "");
        if (R.certified && c->use_kmer_streams) {
            // work items of 256 events (4 per lane) look their descriptor up themselves, on the scalar unit, unless k_items prepared them;
            // with shorter items (profiles with long dwells) the look-up chain per item is always worth the kernel
            if (R.fold) {}                                        // (the work items were prepared inside k_part_hist)
            else if (c->lean_epl < 4 || SQG_LEAN_ITEMS4)
                plan.push_back({"k_items", [&R]() -> int { const int ns = (int)R.b->n_stiles; hipLaunchKernelGGL(k_items, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, R.c->stream, R.P, ns); return SQG_OK; }});
            else P.items = nullptr;
            plan.push_back({"k_samples_lean", [&R, split_streams]() -> int {
                sqg_ctx* c = R.c; sqg_batch* b = R.b; const SigParams& P = R.P;
                int rs = split_streams(); if (rs) return rs;
                const int n_stiles = (int)b->n_stiles;
                unsigned lgrid = (unsigned)((n_stiles + 3) / 4);
                const int lean_grid_cap = dev_env_int(SQG_DEV_ENV("SQG_LEAN_GRID"), 0);   // A/B knob
                if (lean_grid_cap > 0) lgrid = std::min(lgrid, (unsigned)lean_grid_cap);
                if (!R.untimed) HIPCHK(c, hipEventRecord(b->ev[5], c->stream2));
                const unsigned lean_dynlds = (unsigned)dev_env_int(SQG_DEV_ENV("SQG_LEAN_DYNLDS"), 0);   // A/B: bytes of LDS a workgroup reserves on top (fewer workgroups per CU)
#define LEANL(R_, E_) hipLaunchKernelGGL((k_samples_lean<R_, E_>), dim3(lgrid), dim3(256), lean_dynlds, c->stream2, P, n_stiles)
                if (P.rna) { if (c->lean_epl == 4) LEANL(true, 4); else if (c->lean_epl == 2) LEANL(true, 2); else LEANL(true, 1); }
                else { if (c->lean_epl == 4) LEANL(false, 4); else if (c->lean_epl == 2) LEANL(false, 2); else LEANL(false, 1); }
#undef LEANL
                if (!R.untimed) HIPCHK(c, hipEventRecord(b->ev[6], c->stream2));
                b->lean_timed = !R.untimed;
                return SQG_OK; }});
            // what is left -- the items the lean kernel did not take (usually none) and the FP64 fix-ups, small latency-bound
            // kernels -- goes to a stream of its own: the next batch's event side does not wait for it
            plan.push_back({"k_samples<generic> + k_fixup (their own stream)", [&R]() -> int {
                sqg_ctx* c = R.c; sqg_batch* b = R.b; SigParams& P = R.P;
                const bool fix_inline = SQG_DEV_ENV("SQG_FIX_INLINE") != nullptr;   // A/B: the left-over kernels on the batch's own stream
                HIPCHK(c, hipEventRecord(R.S->sampled, c->stream2));
                if (!fix_inline) { HIPCHK(c, hipStreamWaitEvent(c->fix_stream, R.S->sampled, 0)); R.tail = c->fix_stream; }
                if (SQG_DEV_ENV("SQG_ABL_NOFIX")) return SQG_OK;  // timing-only ablation (results are wrong): what the left-over kernels cost the step
                const int n_tiles = (int)b->n_tiles;
                const unsigned sgrid = (unsigned)((n_tiles + 3) / 4);
                hipLaunchKernelGGL((k_samples<1, true>), dim3(std::min(sgrid, 4096u)), dim3(256), 0, R.tail, P, n_tiles);
                P.seglen_zero = 2 * R.n;
                hipLaunchKernelGGL(k_fixup, dim3(FIX_SHARDS), dim3(256), 0, R.tail, P);
                R.seglen_zeroed = true;
                b->fixup_launched = true;                         // (its per-list statistics words are this batch's)
                return SQG_OK; }});
        } else {
            plan.push_back({"k_samples<generic>", [&R, split_streams]() -> int {
                sqg_ctx* c = R.c; const int n_tiles = (int)R.b->n_tiles;
                int rs = split_streams(); if (rs) return rs;
                const unsigned sgrid = (unsigned)((n_tiles + 3) / 4);
                if (R.certified) hipLaunchKernelGGL((k_samples<1, true>), dim3(sgrid), dim3(256), 0, c->stream2, R.P, n_tiles);
                else hipLaunchKernelGGL((k_samples<0, true>), dim3(sgrid), dim3(256), 0, c->stream2, R.P, n_tiles);
                return SQG_OK; }});
        }
    } else plan.push_back({"streams: event side done", split_streams});
    return SQG_OK;
}

// phase 0: the whole run; 1: up to the per-stream sample counts of a split batch (sqg_batch_run_begin); 2: the rest (sqg_batch_run_end),
// `before` / `after` being what the other ranges of the batch draw from each stream
static int run_impl(sqg_ctx* c, sqg_batch* b, const int phase, const uint32_t* before, const uint32_t* after) {
    if (!c || !b) return SQG_EINVAL;
    Run R;
    R.c = c; R.b = b; R.phase = phase; R.before = before; R.after = after;
    int rc;
    if ((rc = run_admit(R))) return rc;
    if ((rc = run_buffers(R))) return rc;
    run_params(R);
    const int n = R.n;
    sqg_ctx::Slot& S = *R.S; sqg_ctx::Slot& other = *R.other; sqg_ctx::CountSet& Q = *R.Q;

    // ---- the event side
    RunPlan ev;
    plan_first_pass(R, ev);
    if (n > 0 && b->n_chains > 0) {
        if (b->part) plan_bucketed(R, ev);
        else if (b->split) plan_link_rows(R, ev);
        else plan_chains(R, ev);
    } else if (phase == 1 && c->d_xcounts)
        ev.push_back({"memset: no local reads, no counts", [&R]() -> int { HIPCHK(R.c, hipMemsetAsync(R.c->d_xcounts, 0, R.n_rows * sizeof(uint32_t), R.c->stream)); return SQG_OK; }});
    if (phase != 1 && before && !(n > 0 && b->n_chains > 0 && b->split) && c->use_kmer_streams)
        // no local reads in this batch: the rows still move past what the other ranges draw
        ev.push_back({"k_rows_advance (no local reads)", [&R]() -> int { launch_rows_advance(R); return SQG_OK; }});
    if (phase != 1) {
        if (!R.untimed) ev.push_back({"event: event side done", [&R]() -> int { HIPCHK(R.c, hipEventRecord(R.b->ev[3], R.c->stream)); return SQG_OK; }});
        if (n > 0 && !R.fold)                                     // (fold: the scan ran inside k_part_mid)
            ev.push_back({"k_scan", [&R]() -> int { hipLaunchKernelGGL(k_scan, dim3(R.scan_wgs), dim3(SCAN_WG), 0, R.c->stream, R.SA); return SQG_OK; }});
        else if (n == 0)
            ev.push_back({"memset: the slot's counters", [&R]() -> int { HIPCHK(R.c, hipMemsetAsync(R.S->d_fix_count, 0, 4 * sizeof(unsigned int), R.c->stream)); return SQG_OK; }});
    }
    if ((rc = run_execute(c, ev))) return rc;
    if (phase == 1) { b->begun = true; return SQG_OK; }

    // ---- the output slab.  Its size is data-dependent.  A hard bound exists (|z| <= sqrt(2 ln(2^31-1)) = 6.5556 for any
    // draw), so the slab is sized by it and the launches continue without a host round trip; only
    // if that bound is unreasonable (huge dwell spread) is the scan read back first.
    if (n == 0) b->h_sigoff[0] = 0;
    size_t need_samples;
    {
        const double bound = c->dwell_hi * (double)b->n_events;
        if (bound <= 4.0e10) need_samples = (size_t)bound;
        else {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            need_samples = (size_t)b->h_sigoff[n];
        }
    }
    const sqg_profile_t& p = c->cfg.profile;
    for (int z = 0; z < (b->other_fresh ? 2 : 1); z++) {
        sqg_ctx::Slot& Z = z ? other : S;
        if ((rc = ensure(c, (void**)&Z.d_sig, &Z.sig_cap, need_samples + 64, sizeof(int16_t)))) return rc;
        if (R.certified && c->use_kmer_streams) {
            if ((rc = ensure(c, (void**)&Z.d_fix, &Z.fix_cap, (c->force_fix ? need_samples : need_samples / 256) + 65536, sizeof(FixEntry)))) return rc;
            // the lean kernel's lists: four times the undecided samples a batch of this size expects (5e-4 of its samples on the profiles
            // measured), at least FIX_SHARD_CAP_MIN per list -- a list that is full falls back to the ONE global list, whose counter
            // serialises the sample kernel (65536-read batches ran at 15.7 instead of 9.6 ms with lists of a fixed 2048 entries)
            const double expect = (double)b->n_events * (std::fabs(p.dwell_mean) + 1.0) * 2.0e-3;
            const size_t per = std::max<size_t>(FIX_SHARD_CAP_MIN, (size_t)(expect / FIX_SHARDS) + 1);
            if (per > Z.fix_sh_per) {
                const size_t cap0 = Z.fix_sh_cap;
                if ((rc = ensure(c, (void**)&Z.d_fix_sh, &Z.fix_sh_cap, (size_t)FIX_SHARDS * (per + per / 4), sizeof(FixEntry)))) return rc;
                if (Z.fix_sh_cap != cap0) HIPCHK(c, hipMemsetAsync(Z.d_fix_sh, 0, Z.fix_sh_cap * sizeof(FixEntry), c->stream));   // (tags of no batch)
                Z.fix_sh_per = (unsigned int)std::min<size_t>(Z.fix_sh_cap / FIX_SHARDS, 0x7fffffffu);
            }
        }
    }

    // ---- the sample side
    RunPlan sm;
    if ((rc = plan_samples(R, sm))) return rc;
    if ((rc = run_execute(c, sm))) return rc;
    if (n > 0) Q.seglen_dirty = R.seglen_zeroed ? (Q.seglen_dirty > (size_t)n ? Q.seglen_dirty : 0) : std::max(Q.seglen_dirty, (size_t)n);
    HIPCHK(c, hipEventRecord(b->ev[4], R.tail));
    HIPCHK(c, hipEventRecord(S.done, R.tail));
    b->ran = true;
    c->next_run++;
    c->runs++;
    return SQG_OK;
}

extern "C" int sqg_batch_run(sqg_ctx_t* c, sqg_batch_t* b) { return run_impl(c, b, 0, nullptr, nullptr); }

extern "C" int sqg_batch_run_begin(sqg_ctx_t* c, sqg_batch_t* b, const uint32_t** d_counts) {
    if (!c || !b || !d_counts) return SQG_EINVAL;
    if (!c->range_mode) { c->err = "sqg_batch_run_begin needs sqg_set_range_mode(ctx, 1) before the batch is staged"; return SQG_EINVAL; }
    if (!c->use_kmer_streams) { c->err = "no k-mer streams in --ideal / --ideal-amp: nothing to exchange, use sqg_batch_run"; return SQG_EINVAL; }
    const int rc = run_impl(c, b, 1, nullptr, nullptr);
    if (rc != SQG_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));                  // the counts are complete on return: the exchange runs on the caller's stream
    *d_counts = c->d_xcounts;
    return SQG_OK;
}

extern "C" int sqg_batch_run_end(sqg_ctx_t* c, sqg_batch_t* b, const uint32_t* d_before, const uint32_t* d_after) {
    return run_impl(c, b, 2, d_before, d_after);
}

extern "C" int sqg_set_range_mode(sqg_ctx_t* c, int on) {
    if (!c) return SQG_EINVAL;
    skip_abandoned(c);
    if (c->next_stage != c->next_run) return SQG_ESEQUENCE;       // staged batches pending
    c->range_mode = on != 0;
    return SQG_OK;
}
