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

static int run_impl(sqg_ctx* c, sqg_batch* b, const int phase, const uint32_t* before, const uint32_t* after) {
    if (!c || !b) return SQG_EINVAL;
    if (phase != 2) skip_abandoned(c);
    if (phase == 2 ? (!b->begun || b->ran) : (b->ran || b->begun || b->seq != c->next_run)) return SQG_ESEQUENCE;
    if ((before == nullptr) != (after == nullptr)) return SQG_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const sqg_profile_t& p = c->cfg.profile;
    const int n = b->n;
    const bool certified = c->cfg.mode == SQG_MODE_CERTIFIED;
    int rc;
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
    sqg_ctx::Slot& S = c->slot[b->slot];
    sqg_ctx::CountSet& Q = c->cset[b->cset];
    if (phase != 2) b->slot_gen = ++S.gen;                      // from here on the slot's buffers belong to this batch
    sqg_ctx::Slot& other = c->slot[b->slot ^ 1];
    if (phase != 2) {
        // this slot's buffers were last used by the sample kernels of batch seq-2 (stream2)
        HIPCHK(c, hipStreamWaitEvent(c->stream, S.done, 0));
        if (b->ev_staged && hipEventQuery(b->ev_staged) != hipSuccess) HIPCHK(c, hipStreamWaitEvent(c->stream, b->ev_staged, 0));     // the batch's uploads and staging kernels (a barrier packet: not queued if they are done)
        auto grow = [&](sqg_ctx::Slot& Z) -> int { return grow_slot(c, Z, b, /*with_output=*/false); };
        b->other_fresh = other.reads_cap == 0 && n > 0;
        if ((rc = grow(S))) return rc;
        if (b->other_fresh && (rc = grow(other))) return rc;
        if ((rc = grow_cset(c, Q, b))) return rc;
    }
    const bool other_fresh = b->other_fresh;
    // precount (below): the batch staged behind this one, if its first event pass can ride along with this batch's hand-out.  What the
    // pass writes is made large enough HERE, before this batch's first launch (a reallocation synchronises the streams and must not
    // happen between the kernels that advance the rows and the hand-out); it is an optimisation: should a buffer not be had, the plain
    // hand-out runs and the batch behind counts for itself.
    sqg_batch* pre_nb = nullptr;
    if (phase == 0 && n > 0 && b->n_chains > 0 && b->part && b->pieces && c->lds_ordered && c->use_dwell_stream && !SQG_DEV_ENV("SQG_SEPARATE_DWELL") &&
        !c->range_mode && !c->staged_q.empty() && !SQG_DEV_ENV("SQG_NO_PRECOUNT")) {
        sqg_batch* cand = c->staged_q.front();
        if (cand->seq > b->seq && cand->staged && !cand->ran && !cand->begun && !cand->precounted && cand->part && cand->pieces && cand->one == b->one &&
            cand->n > 0 && cand->n_chains > 0) {
            const int n_part0 = (c->num_kmer + PART_SUB - 1) >> PART_SUB_BITS;
            sqg_ctx::CountSet& NQ = c->cset[(b->run_idx + 1) % 3];
            bool ok = grow_cset(c, NQ, cand) == SQG_OK;
            const int nx = (int)((b->run_idx + 1) & 1);          // (the counts go to the buffer of the NEXT run index: this batch's own -- which a pass that ran ahead may have filled already -- stays)
            if (ok && !cand->one) ok = ensure(c, (void**)&c->d_pcnt[nx], &c->pcnt_cap[nx], (size_t)2 * cand->n_chains * (size_t)n_part0, sizeof(uint32_t)) == SQG_OK;
            if (ok && cand->one) ok = ensure(c, (void**)&other.d_part, &other.part_cap, (size_t)cand->n_events + PART_SLACK, sizeof(uint32_t)) == SQG_OK;
            if (ok) pre_nb = cand;
            else c->err.clear();
        }
    }

    // Dwell draws are made inside k_events (SQG_SEPARATE_DWELL=1 keeps the stand-alone k_dwell for A/B runs).
    const bool separate_dwell = SQG_DEV_ENV("SQG_SEPARATE_DWELL") != nullptr;
    const bool inline_dwell = c->use_dwell_stream && !separate_dwell;
    const bool direct = c->num_kmer <= 4096;                    // the worker's whole row of stream states fits in LDS
    if (phase != 2 && !direct && c->use_kmer_streams && n > 0) {
        // the rows count samples in 32 bits; only the count mod (M-1)/2 matters (range mode: the other ranges' counts are not
        // known here, so the rows are reduced before every batch)
        const double bnd = (double)b->max_wchain_ev * c->dwell_hi;
        if (b->part && !c->range_mode) {
            // k_part_scan reduces the rows itself and reports a stream that is asked for >= 2^32 samples by one batch
            c->row_bound = std::min((double)LCG_ORD2 + bnd, 4294967295.0) - bnd;
        } else if (c->range_mode || c->row_bound + bnd >= 4294967295.0) {
            const size_t nrow = (size_t)c->nw * (size_t)c->num_kmer;
            hipLaunchKernelGGL(k_rows_normalize, dim3((unsigned)((nrow + 255) / 256)), dim3(256), 0, c->stream, c->d_rows, nrow);
            HIPCHK(c, hipGetLastError());
            c->row_bound = (double)LCG_ORD2;
        }
        c->row_bound += bnd;
    }
    if (phase != 2 && b->split && !b->part && (rc = ensure(c, (void**)&c->d_link_rows, &c->link_rows_cap, (size_t)b->n_chains * (size_t)c->num_kmer, sizeof(uint32_t)))) return rc;
    const int n_part = (c->num_kmer + PART_SUB - 1) >> PART_SUB_BITS;
    const int kmer_pad = n_part * PART_SUB;
    const size_t n_pairs = (size_t)b->n_wchains * (size_t)n_part;   // (worker chain, partition)
    if (phase != 2 && b->part) {
        if ((rc = ensure(c, (void**)&c->d_pcnt[b->run_idx & 1], &c->pcnt_cap[b->run_idx & 1], (size_t)2 * b->n_chains * (size_t)n_part, sizeof(uint32_t)))) return rc;   // counts, offsets (no-op for a batch whose first pass ran ahead: the pass' launch sized it)
        if ((rc = ensure(c, (void**)&c->d_slice, &c->slice_cap, (size_t)2 * (size_t)b->max_slices + (size_t)3 * n_pairs + 1, sizeof(uint32_t)))) return rc;
        if ((rc = ensure(c, (void**)&c->d_phist, &c->phist_cap, (size_t)b->max_slices * (size_t)PART_SUB, sizeof(uint32_t)))) return rc;
    }
    const size_t n_rows = (size_t)c->nw * (size_t)c->num_kmer;
    if (phase == 1 && (rc = ensure(c, (void**)&c->d_xcounts, &c->xcounts_cap, n_rows, sizeof(uint32_t)))) return rc;
    uint32_t* const d_pcnt = c->d_pcnt[b->run_idx & 1];
    SigParams P;
    memset(&P, 0, sizeof P);
    P.link_rows = (b->split && !b->part) ? c->d_link_rows : nullptr;
    P.part = S.d_part; P.part_state = b->part ? S.d_part_state : nullptr; P.pcnt = d_pcnt; P.poff = b->one ? b->d_link_slot : d_pcnt ? d_pcnt + (size_t)b->n_chains * n_part : nullptr; P.n_part = n_part; P.n_links = b->n_chains;
    P.link_q = b->d_link_q; P.pieces = b->d_pieces; P.piece_total = b->d_piece_total;
    P.one = b->one ? 1 : 0;
    count_params(c, b, Q, n_part, d_pcnt, P);
    P.sig_off = S.d_sigoff; P.model = c->d_model; P.pw = c->d_pow; P.rows = c->d_rows;
    P.seed_base = canon((long long)c->cfg.seed + (long long)c->wlo * ((long long)c->num_kmer + 10)); P.seed_step = canon((long long)c->num_kmer + 10);
    P.dig = p.digitisation; P.range = p.range; P.kd = p.digitisation / p.range;
    P.thr_all = c->thr_all;
    P.meth = (c->cfg.flags & SQG_METH) ? 1 : 0; P.num_kmer_pad = kmer_pad;
    P.meth_top = 1; for (int i = 1; i < c->k; i++) P.meth_top *= 5u;
    P.use_streams = c->use_kmer_streams ? 1 : 0;
    P.rna = (c->cfg.flags & SQG_RNA) ? 1 : 0;
    P.evrec = S.d_evrec;
    // bucketed hand-out with the wavefront-per-link passes: 4 B per event between the scatter pass and the sample kernels (k_part_events.h)
    if (b->part && b->pieces && !b->one) { P.evrec32 = reinterpret_cast<uint32_t*>(S.d_evrec); P.lbase = S.d_lbase; P.tile_link = S.d_tile_link; } P.tile_read = b->d_tile_read; P.stile_read = b->d_stile_read;
    // what k_items needs of P (the rest of the sample kernels' parameters follows further down): k_items may run inside k_part_hist
    {
        const bool rna_prefix0 = (c->cfg.flags & SQG_RNA) && (c->cfg.flags & SQG_PREFIX);
        P.shift_len = rna_prefix0 ? (int)strlen(kAdaptorRna) * (int)p.dwell_mean : 0;
        P.slow_count = S.d_fix_count + 1; P.items = S.d_items; P.lean_epl = c->lean_epl;
        P.slow_tiles = (certified && c->use_kmer_streams) ? S.d_slow : nullptr;
    }
    // (round 5) bucketed hand-out, one launch sequence (no range sharding): the scan of the reads' totals runs as extra workgroups of
    // k_part_mid and the lean kernel's work items are prepared by extra workgroups of k_part_hist -- two launches and their gaps less per
    // batch (35 us of a 3.7-ms step).  SQG_NO_FOLD=1 (development build) keeps the two kernels for A/B runs.
    const bool fold = phase == 0 && n > 0 && b->n_chains > 0 && b->part && !b->one && b->pieces && certified && c->use_kmer_streams &&
                      (c->lean_epl < 4 || SQG_LEAN_ITEMS4) && !SQG_DEV_ENV("SQG_MID_SPLIT") && !SQG_DEV_ENV("SQG_NO_FOLD");
    const unsigned scan_wgs = n > 0 ? (unsigned)((n + SCAN_WG - 1) / SCAN_WG) : 0u;
    ScanArgs SA;
    memset(&SA, 0, sizeof SA);
    if (n > 0 && phase != 1) {
        // the scan also writes the offsets through the batch's pinned host mapping (no copy between kernels)
        const size_t cap0 = c->scan_part_cap;
        if ((rc = ensure(c, (void**)&c->d_scan_part, &c->scan_part_cap, (size_t)2 * scan_wgs, sizeof(unsigned long long)))) return rc;
        if (c->scan_part_cap != cap0)                         // tickets start at 1: a fresh array must not hold one by accident
            HIPCHK(c, hipMemsetAsync(c->d_scan_part, 0, c->scan_part_cap * sizeof(unsigned long long), c->stream));
        SA.seglen = Q.d_seglen; SA.n_reads = n; SA.sig_off = S.d_sigoff; SA.host_off = b->h_sigoff_dev; SA.err = b->d_err; SA.counters = S.d_fix_count;
        SA.part = c->d_scan_part; SA.ticket = ++c->scan_tickets; SA.shard_counters = S.d_fix_sh_count;   // (a ticket per launch, also after a failed run)
    }
    constexpr int NT = SQG_EVENT_THREADS, NT_WIDE = 1024;
    // few chains (the reference's default -K 1000 with one worker per read): a chain is a sequence of segments, each with
    // its barriers and LDS round trips, and there are not enough chains to hide them -- 1024 threads per chain walk it in a
    // quarter of the steps
    const int wide_max = dev_env_int(SQG_DEV_ENV("SQG_EVENTS_WIDE_MAX"), 1200);   // A/B knob
    auto launch_events = [&](int dw, bool hist) {
        const dim3 g((unsigned)b->n_chains);
        const bool wide = !hist && b->n_chains <= wide_max;
#define EVL(N, D, W, H, PT) hipLaunchKernelGGL((k_events<N, D, W, SQG_EVENT_EPT, H, PT>), g, dim3(N), 0, c->stream, P)
#define EVD(N, D, H, PT) do { if (dw == 0) EVL(N, D, 0, H, PT); else if (dw == 1) EVL(N, D, 1, H, PT); else EVL(N, D, 2, H, PT); } while (0)
        if (b->part) { if (hist) EVD(NT, false, true, true); else EVD(NT, false, false, true); }
        else if (wide) { if (direct) EVD(NT_WIDE, true, false, false); else EVD(NT_WIDE, false, false, false); }
        else if (direct) { if (hist) EVD(NT, true, true, false); else EVD(NT, true, false, false); }
        else { if (hist) EVD(NT, false, true, false); else EVD(NT, false, false, false); }
#undef EVD
#undef EVL
    };

    // the phase boundaries are recorded events: barrier packets between the kernels, 3-8 us of idle GPU each (tools/ext_event_probe.hip;
    // events attached to a launch cost more, not less) -- 1.2 % of a 16384-read step, 5 % of a 1000-read one.  sqg_set_phase_timing
    // says how many batches carry them.
    if (phase != 2) b->untimed = c->phase_timing_every <= 0 || (b->run_idx % c->phase_timing_every) != 0;
    const bool untimed = b->untimed;
    if (phase != 2) {
        if (!untimed) HIPCHK(c, hipEventRecord(b->ev[0], c->stream));
        if (n > 0) {
            if (c->use_dwell_stream && !inline_dwell) {
                HIPCHK(c, hipMemsetAsync(Q.d_seglen, 0, (size_t)2 * n * sizeof(unsigned long long), c->stream));
                const long long nblk = (b->n_events + DW_EPB - 1) / DW_EPB;
                if (nblk > 0) {
                    if (certified)
                        hipLaunchKernelGGL(k_dwell<1>, dim3((unsigned)nblk), dim3(256), 0, c->stream, b->d_reads, n, b->d_blk_read,
                                           b->n_events, c->d_pow, p.dwell_mean, p.dwell_std, c->delta_x, Q.d_dwell, Q.d_seglen, b->d_err);
                    else
                        hipLaunchKernelGGL(k_dwell<0>, dim3((unsigned)nblk), dim3(256), 0, c->stream, b->d_reads, n, b->d_blk_read,
                                           b->n_events, c->d_pow, p.dwell_mean, p.dwell_std, 0.f, Q.d_dwell, Q.d_seglen, b->d_err);
                }
                if ((rc = dbg_sync(c, "k_dwell"))) return rc;
            } else if (!c->use_dwell_stream) {
                HIPCHK(c, hipMemcpyAsync(Q.d_seglen, b->seglen_host.data(), (size_t)2 * n * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
            }
        }
        b->dwell_timed = c->use_dwell_stream && !inline_dwell && !untimed;        // stand-alone k_dwell (A/B runs): two more timing events
        if (b->dwell_timed) { HIPCHK(c, hipEventRecord(b->ev[1], c->stream)); HIPCHK(c, hipEventRecord(b->ev[2], c->stream)); }
    }
    if (n > 0 && b->n_chains > 0) {
        const int dw = inline_dwell ? (certified && c->dwell_hi < 1.0e6 ? 1 : 2) : 0;
        const dim3 pg((unsigned)((c->num_kmer + 63) / 64), (unsigned)b->n_wchains);
        const dim3 sg((unsigned)((c->num_kmer + 255) / 256), (unsigned)b->n_wchains);
        const unsigned pgrid = (unsigned)b->max_slices;
        uint32_t* const slice_lo = c->d_slice;
        uint32_t* const slice_hi = slice_lo + (size_t)b->max_slices;
        uint32_t* const pfirst = slice_hi + (size_t)b->max_slices;   // pfirst[n_pairs]: the number of slices
        uint32_t* const pstart = pfirst + n_pairs + 1;
        uint32_t* const ptotal = pstart + n_pairs;
        P.pstart = pstart;
        // one wavefront per link, ordered LDS atomics (k_part_events.h); the workgroup-per-link passes of k_events stay for the
        // 5-letter alphabet and for devices that do not pass the order check
        const bool wave_links = b->pieces;                       // (decided at staging: the links are then runs of pieces of reads)
        auto launch_part_events = [&](int dwm, bool count) {
            if (!wave_links) { launch_events(dwm, count); return; }
            const dim3 g((unsigned)((b->n_chains + PEV_WAVES - 1) / PEV_WAVES)), t(64 * PEV_WAVES);
            if (!count) hipLaunchKernelGGL((k_part_events<0, PEV_SCATTER>), dim3((unsigned)((b->n_chains + PEV_WAVES_SCATTER - 1) / PEV_WAVES_SCATTER)), dim3(64 * PEV_WAVES_SCATTER), 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
            else if (b->one) {                                    // one partition: the only event pass
                if (dwm == 0) hipLaunchKernelGGL((k_part_events<0, PEV_ONE>), g, t, 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
                else if (dwm == 1) hipLaunchKernelGGL((k_part_events<1, PEV_ONE>), g, t, 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
                else hipLaunchKernelGGL((k_part_events<2, PEV_ONE>), g, t, 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
            }
            else if (dwm == 0) hipLaunchKernelGGL((k_part_events<0, PEV_COUNT>), g, t, 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
            else if (dwm == 1) hipLaunchKernelGGL((k_part_events<1, PEV_COUNT>), g, t, 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
            else hipLaunchKernelGGL((k_part_events<2, PEV_COUNT>), g, t, 0, c->stream, P, b->n_chains, (uint32_t)b->n_events);
        };
        if (b->part) {
            // k > 6, split chains: the hand-out over events bucketed by the top bits of the rank (k_part.h)
            if (phase != 2) {
                if (!b->precounted) {                             // (else: the pass ran next to the hand-out of the batch before this one)
                    if (b->split_reads && dw && Q.seglen_dirty > 0) {   // pieces add up; (usually the set's previous batch has left the words zero: k_fixup)
                        HIPCHK(c, hipMemsetAsync(Q.d_seglen, 0, (size_t)2 * std::max<size_t>((size_t)n, Q.seglen_dirty) * sizeof(unsigned long long), c->stream));
                        Q.seglen_dirty = 0;
                    }
                    if (b->split_reads && dw) Q.seglen_dirty = (size_t)n;   // (until this batch's k_fixup is queued: a run that fails half-way leaves them dirty)
                    launch_part_events(dw, true);                 // dwell draws; events per (link, partition)
                }
                const bool mid_split = SQG_DEV_ENV("SQG_MID_SPLIT") != nullptr;   // A/B: the four small kernels between the passes, one by one
                if (b->split_reads && (b->one || mid_split)) hipLaunchKernelGGL(k_part_tile_bases, dim3((unsigned)b->n_pieces), dim3(64), 0, c->stream, P);
                if (b->one) {
                    // one partition: the pass above has written part[] in chain order; the slices of every worker chain's events
                    hipLaunchKernelGGL(k_part_slices, dim3(1), dim3(1024), 0, c->stream, pstart, b->d_wchain_total, (int)n_pairs, b->slice_len, pfirst, slice_lo, slice_hi);
                    HIPCHK(c, hipGetLastError());
                    if ((rc = dbg_sync(c, "k_part_events<one>/k_part_slices"))) return rc;
                } else {
                    if (mid_split) {
                        hipLaunchKernelGGL(k_part_offsets, dim3((unsigned)n_part, (unsigned)b->n_wchains), dim3(1024), 0, c->stream, d_pcnt,
                                           d_pcnt + (size_t)b->n_chains * n_part, n_part, b->n_chains, b->d_wlink_off, ptotal);
                        hipLaunchKernelGGL(k_part_slices, dim3(1), dim3(1024), 0, c->stream, pstart, ptotal, (int)n_pairs, b->slice_len, pfirst, nullptr, nullptr);
                        hipLaunchKernelGGL(k_part_slice_bounds, dim3((pgrid + 255) / 256), dim3(256), 0, c->stream, pstart, ptotal, (int)n_pairs, b->slice_len, pfirst, slice_lo, slice_hi);
                    } else {
                        // offsets per (partition, worker chain), the tile offsets of split reads, the slices and their bounds: one launch
                        const int n_off = (int)n_pairs, n_pc = b->split_reads ? b->n_pieces : 0;
                        const int n_sc = fold ? (int)scan_wgs : 0;
                        hipLaunchKernelGGL(k_part_mid, dim3((unsigned)(n_off + (n_pc + 15) / 16 + n_sc)), dim3(1024), 0, c->stream, P, d_pcnt,
                                           d_pcnt + (size_t)b->n_chains * n_part, n_part, b->n_chains, b->d_wlink_off, ptotal, pstart, (int)n_pairs, b->slice_len,
                                           pfirst, slice_lo, slice_hi, n_off, n_pc, c->d_mid_done, SA, n_sc);
                    }
                    HIPCHK(c, hipGetLastError());
                    if ((rc = dbg_sync(c, "k_events<count>/k_part_offsets"))) return rc;
                    launch_part_events(0, false);                 // every event to its slot (the dwell is in memory now)
                }
                {
                    const int n_st = fold ? (int)b->n_stiles : 0;
                    hipLaunchKernelGGL(k_part_hist, dim3(pgrid + (unsigned)((n_st + 255) / 256)), dim3(256), 0, c->stream, S.d_part, slice_lo, slice_hi, pfirst + n_pairs,
                                       c->d_phist, P, n_st, pgrid);
                }
                HIPCHK(c, hipGetLastError());
                if ((rc = dbg_sync(c, "k_events<scatter>/k_part_hist"))) return rc;
                if (phase == 1) {                                 // range sharding: what this range draws per stream, for the exchange
                    HIPCHK(c, hipMemsetAsync(c->d_xcounts, 0, n_rows * sizeof(uint32_t), c->stream));
                    hipLaunchKernelGGL(k_part_totals, sg, dim3(256), 0, c->stream, c->d_phist, c->num_kmer, n_part, pfirst, b->d_wlink_worker, c->d_xcounts);
                    HIPCHK(c, hipGetLastError());
                }
            }
            if (phase != 1) {
#define SCANL(R_, G_) hipLaunchKernelGGL((k_part_scan<R_, G_>), dim3((unsigned)((c->num_kmer + R_ - 1) / R_), (unsigned)b->n_wchains), dim3(R_ * G_), 0, c->stream, \
                                        c->d_phist, c->d_rows, c->num_kmer, n_part, pfirst, b->d_wlink_worker, before, c->d_pow, P.seed_base, P.seed_step, direct ? 1 : 0, b->d_err)
                const int scan_g4 = dev_env_int(SQG_DEV_ENV("SQG_SCAN_G4"), 32);   // A/B knob
                if ((long long)b->max_slices <= 8 * (long long)n_pairs) SCANL(256, 1);       // a slice or two per pair: one thread per rank walks them
                else if ((long long)b->max_slices <= scan_g4 * (long long)n_pairs && (size_t)pg.x * pg.y >= 512) SCANL(64, 4);   // a dozen (small batches): 16 runs would be 16 x the wavefronts, most of them idle
                else if ((size_t)pg.x * pg.y >= 512) SCANL(64, 16);
                else SCANL(16, 64);
#undef SCANL
                if (before) {                                     // every worker's row moves past the whole batch, all ranges
                    const dim3 ag((unsigned)((n_rows + 255) / 256));
                    if (direct) hipLaunchKernelGGL(k_rows_advance<true>, ag, dim3(256), 0, c->stream, c->d_rows, c->d_pow, n_rows, before, c->d_xcounts, after);
                    else hipLaunchKernelGGL(k_rows_advance<false>, ag, dim3(256), 0, c->stream, c->d_rows, c->d_pow, n_rows, before, c->d_xcounts, after);
                }
                const int order_fault = SQG_DEV_ENV("SQG_TEST_ORDER_FAULT") ? 1 : 0;         // (tests: the per-batch order check has to fire)
                // precount: the batch staged behind this one, if its first event pass can ride along with this batch's hand-out
                // (k_part_hand_count, k_part_events.h): same kind of batch, nothing in between, the plain launch sequence
                sqg_batch* nb = (pre_nb && wave_links && dw != 0) ? pre_nb : nullptr;       // (chosen, and its buffers made, before this batch's first launch)
                if (nb) {
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
                    count_params(c, nb, NQ, n_part, c->d_pcnt[(b->run_idx + 1) & 1], Pn);
                    if (nb->one) { Pn.one = 1; Pn.part = other.d_part; Pn.poff = nb->d_link_slot; }
                    const dim3 fg((unsigned)(dev_env_int(SQG_DEV_ENV("SQG_PHC_GRID"), 4) * c->num_cu)), ft(64 * (1 + PHC_COUNT_WAVES));   // (A/B: workgroups per CU)
                    // (development build, timing experiments: 1 -- the fused launch hands out only, the next batch's pass follows as a launch
                    // of its own; 2 -- the plain hand-out first, the fused launch counts only)
                    const int phc_abl = dev_env_int(SQG_DEV_ENV("SQG_PHC_ABL"), 0);
                    if (phc_abl == 2) hipLaunchKernelGGL(k_part_hand_ord, dim3(pgrid), dim3(64), 0, c->stream, S.d_part, S.d_part_state, slice_lo, slice_hi, pfirst + n_pairs, c->d_phist, c->d_pow, b->d_err, order_fault);
                    const uint32_t* const ns_ptr = phc_abl == 2 ? c->d_zero : pfirst + n_pairs;
                    const int nl_fused = phc_abl == 1 ? 0 : nb->n_chains;
#define PHCL(D_, M_) hipLaunchKernelGGL((k_part_hand_count<D_, M_>), fg, ft, 0, c->stream, S.d_part, S.d_part_state, slice_lo, slice_hi, ns_ptr, c->d_phist, c->d_pow, \
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
                    b->carried_precount = true;                       // (sqg_get_timing: this batch's event side holds the successor's first pass)
                }
                else if (c->lds_ordered) hipLaunchKernelGGL(k_part_hand_ord, dim3(pgrid), dim3(64), 0, c->stream, S.d_part, S.d_part_state, slice_lo, slice_hi, pfirst + n_pairs, c->d_phist, c->d_pow, b->d_err, order_fault);
                else if (c->dwell_hi >= (double)PART_JT) hipLaunchKernelGGL(k_part_hand<true>, dim3(pgrid), dim3(64), 0, c->stream, S.d_part, S.d_part_state, slice_lo, slice_hi, pfirst + n_pairs, c->d_phist, c->d_pow, (uint32_t)b->n_events);
                else hipLaunchKernelGGL(k_part_hand<false>, dim3(pgrid), dim3(64), 0, c->stream, S.d_part, S.d_part_state, slice_lo, slice_hi, pfirst + n_pairs, c->d_phist, c->d_pow, (uint32_t)b->n_events);
                HIPCHK(c, hipGetLastError());
                if ((rc = dbg_sync(c, "k_part_scan/k_part_hand"))) return rc;
            }
        } else if (b->split && phase != 2) {
            // links: samples per (link, k-mer) with the dwell draws ...
            if (!direct) HIPCHK(c, hipMemsetAsync(c->d_link_rows, 0, (size_t)b->n_chains * (size_t)c->num_kmer * sizeof(uint32_t), c->stream));
            launch_events(dw, true);
            HIPCHK(c, hipGetLastError());
            if (phase == 1) {                                     // ... summed per worker for the exchange
                HIPCHK(c, hipMemsetAsync(c->d_xcounts, 0, n_rows * sizeof(uint32_t), c->stream));
                hipLaunchKernelGGL(k_link_totals, pg, dim3(1024), 0, c->stream, P, b->d_wlink_off, b->d_wlink_worker, c->d_xcounts);
                HIPCHK(c, hipGetLastError());
            }
        }
        if (b->part) {
        } else if (b->split && phase != 1) {
            // ... then each link's view of its worker's streams
            if (direct) hipLaunchKernelGGL(k_link_prefix<true>, pg, dim3(1024), 0, c->stream, P, b->d_wlink_off, b->d_wlink_worker, before);
            else hipLaunchKernelGGL(k_link_prefix<false>, pg, dim3(1024), 0, c->stream, P, b->d_wlink_off, b->d_wlink_worker, before);
            if (before) {                                         // every worker's row moves past the whole batch, all ranges
                const dim3 ag((unsigned)((n_rows + 255) / 256));
                if (direct) hipLaunchKernelGGL(k_rows_advance<true>, ag, dim3(256), 0, c->stream, c->d_rows, c->d_pow, n_rows, before, c->d_xcounts, after);
                else hipLaunchKernelGGL(k_rows_advance<false>, ag, dim3(256), 0, c->stream, c->d_rows, c->d_pow, n_rows, before, c->d_xcounts, after);
            }
            HIPCHK(c, hipGetLastError());
            if ((rc = dbg_sync(c, "k_events<hist>/k_link_prefix"))) return rc;
            launch_events(0, false);                              // the dwell is in memory now
        } else if (!b->split && phase != 1) launch_events(dw, false);
        HIPCHK(c, hipGetLastError());
        if ((rc = dbg_sync(c, "k_events"))) return rc;
    } else if (phase == 1 && c->d_xcounts) {
        HIPCHK(c, hipMemsetAsync(c->d_xcounts, 0, n_rows * sizeof(uint32_t), c->stream));
    }
    if (phase == 1) { b->begun = true; return SQG_OK; }
    if (before && !(n > 0 && b->n_chains > 0 && b->split) && c->use_kmer_streams) {
        // no local reads in this batch: the rows still move past what the other ranges draw
        const dim3 ag((unsigned)((n_rows + 255) / 256));
        if (direct) hipLaunchKernelGGL(k_rows_advance<true>, ag, dim3(256), 0, c->stream, c->d_rows, c->d_pow, n_rows, before, c->d_xcounts, after);
        else hipLaunchKernelGGL(k_rows_advance<false>, ag, dim3(256), 0, c->stream, c->d_rows, c->d_pow, n_rows, before, c->d_xcounts, after);
        HIPCHK(c, hipGetLastError());
    }
    if (!untimed) HIPCHK(c, hipEventRecord(b->ev[3], c->stream));
    if (n > 0) {
        if (!fold) {                                              // (else: the scan ran inside k_part_mid)
            hipLaunchKernelGGL(k_scan, dim3(scan_wgs), dim3(SCAN_WG), 0, c->stream, SA);
            HIPCHK(c, hipGetLastError());
            if ((rc = dbg_sync(c, "k_scan"))) return rc;
        }
    } else HIPCHK(c, hipMemsetAsync(S.d_fix_count, 0, 4 * sizeof(unsigned int), c->stream));
    // Output size is data-dependent.  A hard bound exists (|z| <= sqrt(2 ln(2^31-1)) = 6.5556 for any
    // draw), so the slab is sized by it and the launches continue without a host round trip; only
    // if that bound is unreasonable (huge dwell spread) is the scan read back first.
    if (n == 0) b->h_sigoff[0] = 0;
    size_t need_samples;
    {
        const double hi = c->dwell_hi;
        const double bound = hi * (double)b->n_events;
        if (bound <= 4.0e10) need_samples = (size_t)bound;
        else {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            need_samples = (size_t)b->h_sigoff[n];
        }
    }
    for (int z = 0; z < (other_fresh ? 2 : 1); z++) {
        sqg_ctx::Slot& Z = z ? other : S;
        if ((rc = ensure(c, (void**)&Z.d_sig, &Z.sig_cap, need_samples + 64, sizeof(int16_t)))) return rc;
        if (certified && c->use_kmer_streams) {
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

    hipStream_t tail = c->stream2;                               // the stream the batch's last kernel runs on
    bool seglen_zeroed = false;
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
        const bool rna_prefix = (c->cfg.flags & SQG_RNA) && (c->cfg.flags & SQG_PREFIX);
        P.shift_len = rna_prefix ? (int)strlen(kAdaptorRna) * (int)p.dwell_mean : 0;
        {   // int16_t off = 30*dig/range (src/genread.c:82): double -> int16 as the CPU does it
            const double v = 30 * p.digitisation / p.range;
            int32_t t = (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : (int32_t)0x80000000u;
            P.shift = (int)(int16_t)(uint16_t)((uint32_t)t & 0xffffu);
        }
        P.slow_tiles = nullptr; P.slow_count = S.d_fix_count + 1; P.items = S.d_items; P.lean_epl = c->lean_epl;
        const int n_tiles = (int)b->n_tiles;
        const unsigned sgrid = (unsigned)((n_tiles + 3) / 4);
        if (certified && c->use_kmer_streams) {
            P.slow_tiles = S.d_slow;
            const int n_stiles = (int)b->n_stiles;
            unsigned lgrid = (unsigned)((n_stiles + 3) / 4);
            const int lean_grid_cap = dev_env_int(SQG_DEV_ENV("SQG_LEAN_GRID"), 0);   // A/B knob
            if (lean_grid_cap > 0) lgrid = std::min(lgrid, (unsigned)lean_grid_cap);
            // work items of 256 events (4 per lane) look their descriptor up themselves, on the scalar unit; with shorter
            // items (profiles with long dwells) the look-up chain per item is worth a kernel of its own
            if (fold) {}                                          // (the work items were prepared inside k_part_hist)
            else if (c->lean_epl < 4 || SQG_LEAN_ITEMS4) hipLaunchKernelGGL(k_items, dim3((unsigned)((n_stiles + 255) / 256)), dim3(256), 0, c->stream, P, n_stiles);
            else P.items = nullptr;
            if (c->stream2 != c->stream) {
                HIPCHK(c, hipEventRecord(b->ev[7], c->stream));              // event side done: the sample kernels may start ...
                HIPCHK(c, hipStreamWaitEvent(c->stream2, b->ev[7], 0));      // ... on their own stream, next to the next batch's k_events
            }
            if (!untimed) HIPCHK(c, hipEventRecord(b->ev[5], c->stream2));
            const unsigned lean_dynlds = (unsigned)dev_env_int(SQG_DEV_ENV("SQG_LEAN_DYNLDS"), 0);   // A/B: bytes of LDS a workgroup reserves on top (fewer workgroups per CU)
#define LEANL(R, E) hipLaunchKernelGGL((k_samples_lean<R, E>), dim3(lgrid), dim3(256), lean_dynlds, c->stream2, P, n_stiles)
            if (P.rna) { if (c->lean_epl == 4) LEANL(true, 4); else if (c->lean_epl == 2) LEANL(true, 2); else LEANL(true, 1); }
            else { if (c->lean_epl == 4) LEANL(false, 4); else if (c->lean_epl == 2) LEANL(false, 2); else LEANL(false, 1); }
#undef LEANL
            if (!untimed) HIPCHK(c, hipEventRecord(b->ev[6], c->stream2));
            b->lean_timed = !untimed;
            if ((rc = dbg_sync(c, "k_samples_lean"))) return rc;
            // what is left -- the items the lean kernel did not take (usually none) and the FP64 fix-ups, small latency-bound
            // kernels -- goes to a stream of its own: the next batch's k_events does not wait for it
            const bool fix_inline = SQG_DEV_ENV("SQG_FIX_INLINE") != nullptr;   // A/B: the left-over kernels on the batch's own stream
            HIPCHK(c, hipEventRecord(S.sampled, c->stream2));
            if (!fix_inline) {
                HIPCHK(c, hipStreamWaitEvent(c->fix_stream, S.sampled, 0));
                tail = c->fix_stream;
            }
            const bool abl_nofix = SQG_DEV_ENV("SQG_ABL_NOFIX") != nullptr;   // timing-only ablation (results are wrong): what the left-over kernels cost the step
            if (!abl_nofix) {
            hipLaunchKernelGGL((k_samples<1, true>), dim3(std::min(sgrid, 4096u)), dim3(256), 0, tail, P, n_tiles);
            if ((rc = dbg_sync(c, "k_samples<generic>"))) return rc;
            P.seglen_zero = 2 * n;
            hipLaunchKernelGGL(k_fixup, dim3(FIX_SHARDS), dim3(256), 0, tail, P);
            seglen_zeroed = true;
            b->fixup_launched = true;                                // (its per-list statistics words are this batch's)
            }
            if ((rc = dbg_sync(c, "k_fixup"))) return rc;
        } else {
            if (c->stream2 != c->stream) {
                HIPCHK(c, hipEventRecord(b->ev[7], c->stream));
                HIPCHK(c, hipStreamWaitEvent(c->stream2, b->ev[7], 0));
            }
            if (certified) hipLaunchKernelGGL((k_samples<1, true>), dim3(sgrid), dim3(256), 0, c->stream2, P, n_tiles);
            else hipLaunchKernelGGL((k_samples<0, true>), dim3(sgrid), dim3(256), 0, c->stream2, P, n_tiles);
        }
        HIPCHK(c, hipGetLastError());
    } else if (c->stream2 != c->stream) {
        HIPCHK(c, hipEventRecord(b->ev[7], c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, b->ev[7], 0));
    }
    if (n > 0) Q.seglen_dirty = seglen_zeroed ? (Q.seglen_dirty > (size_t)n ? Q.seglen_dirty : 0) : std::max(Q.seglen_dirty, (size_t)n);
    HIPCHK(c, hipEventRecord(b->ev[4], tail));
    HIPCHK(c, hipEventRecord(S.done, tail));
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
