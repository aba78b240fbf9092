// h_stage.h -- staging a batch: descriptors, worker chains and links, per-read host draws, uploads; per-batch and per-slot device memory
// Host side of include/sqg.h; included by sqg_hip.hip (one translation unit with the kernels), in the order listed there.
#pragma once

extern "C" void sqg_batch_free(sqg_ctx_t* ctx, sqg_batch_t* b) {
    if (!b) return;
    if (ctx) {
        (void)hipSetDevice(ctx->cfg.device);
        if (b->ran && b->ev[4]) (void)hipEventSynchronize(b->ev[4]);      // this batch's kernels only, not the ones queued after it
    }
    // never run (or only begun): the batches staged after it must not wait for it.  Its share of the workers' scalar streams
    // is spent all the same: what follows is no longer the reference's sequence.
    if (ctx && b->staged && !b->ran) {
        if ((b->begun || b->precounted) && ctx->stream) (void)hipStreamSynchronize(ctx->stream);   // (its first pass may be running: it reads the batch's block)
        ctx->abandoned.insert(b->seq);
    }
    if (ctx) for (auto it = ctx->staged_q.begin(); it != ctx->staged_q.end(); ++it) if (*it == b) { ctx->staged_q.erase(it); break; }
    if (ctx && !b->ran && ctx->stage_stream) (void)hipStreamSynchronize(ctx->stage_stream);   // its uploads may still be in flight
    if (b->h_svboff) (void)hipHostFree(b->h_svboff);
    if (ctx && b->d_block && b->h_sigoff && b->ev[0] && ctx->pool.size() < 4) {
        sqg_ctx::Recycled r;
        r.d_block = b->d_block; r.block_bytes = b->block_bytes; r.h_sigoff = b->h_sigoff; r.h_sigoff_dev = b->h_sigoff_dev; r.h_n = b->h_n;
        for (int i = 0; i < 8; i++) r.ev[i] = b->ev[i];
        r.h_meta = b->h_meta; r.h_meta_bytes = b->h_meta_bytes; r.ev_staged = b->ev_staged;
        ctx->pool.push_back(r);
    } else {
        (void)hipFree(b->d_block);
        if (b->h_sigoff) (void)hipHostFree(b->h_sigoff);
        if (b->h_meta) (void)hipHostFree(b->h_meta);
        for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
        if (b->ev_staged) (void)hipEventDestroy(b->ev_staged);
    }
    delete b;
}

// Per-slot device buffers for a batch of this geometry.  with_output: also the signal slab and the fix-up list, sized
// by the hard bound on the dwell (skipped when that bound is unreasonable; sqg_batch_run then reads the scan back).
static int grow_cset(sqg_ctx* c, sqg_ctx::CountSet& Q, const sqg_batch* b) {
    int rc2;
    const size_t n = (size_t)b->n;
    if (2 * (n + 1) > Q.seglen_cap) {
        if ((rc2 = ensure(c, (void**)&Q.d_seglen, &Q.seglen_cap, 2 * (n + 1 + n / 2), sizeof(unsigned long long)))) return rc2;
        Q.seglen_dirty = Q.seglen_cap / 2;                         // (fresh memory)
    }
    if ((rc2 = ensure(c, (void**)&Q.d_dwell, &Q.dwell_cap, (size_t)b->n_events + 1024, sizeof(uint16_t)))) return rc2;
    if ((rc2 = ensure(c, (void**)&Q.d_tile_so, &Q.tile_cap, (size_t)b->n_tiles + 64, sizeof(uint32_t)))) return rc2;
    return SQG_OK;
}
static int grow_slot(sqg_ctx* c, sqg_ctx::Slot& Z, const sqg_batch* b, bool with_output) {
    int rc2;
    const int n = b->n;
    const bool certified = c->cfg.mode == SQG_MODE_CERTIFIED;
    if ((size_t)n + 1 > Z.reads_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream2));
        (void)hipFree(Z.d_sigoff); Z.d_sigoff = nullptr;
        const size_t cap = (size_t)n + 1 + (size_t)n / 2;
        HIPCHK(c, hipMalloc(&Z.d_sigoff, cap * sizeof(long long)));
        Z.reads_cap = cap;
    }
    if (Z.cal_prev && (size_t)b->n_events + 64 > Z.evrec_cap) {    // (the event records grow: a placement candidate's predecessor is not coming back)
        HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream2)); HIPCHK(c, hipStreamSynchronize(c->fix_stream));
        (void)hipFree(Z.cal_prev); Z.cal_prev = nullptr; Z.cal_pending = false;
    }
    if ((rc2 = ensure(c, (void**)&Z.d_evrec, &Z.evrec_cap, (size_t)b->n_events + 64, sizeof(uint2)))) return rc2;
    if (b->part && (rc2 = ensure(c, (void**)&Z.d_part, &Z.part_cap, (size_t)b->n_events + PART_SLACK, sizeof(uint32_t)))) return rc2;
    if (b->part && b->pieces && !b->one) {
        if ((rc2 = ensure(c, (void**)&Z.d_lbase, &Z.lbase_cap, (size_t)b->n_chains * PART_MAX, sizeof(uint32_t)))) return rc2;
        if ((rc2 = ensure(c, (void**)&Z.d_tile_link, &Z.tile_link_cap, (size_t)b->n_tiles + 64, sizeof(int)))) return rc2;
    }
#if defined(SQG_ABL_HANDOVER)
    if (b->part && (rc2 = ensure(c, (void**)&Z.d_part_state, &Z.part_state_cap, (size_t)4 * ((size_t)b->n_events + PART_SLACK), sizeof(uint32_t)))) return rc2;
#else
    if (b->part && (rc2 = ensure(c, (void**)&Z.d_part_state, &Z.part_state_cap, (size_t)b->n_events + PART_SLACK, sizeof(uint32_t)))) return rc2;
#endif
    if ((rc2 = ensure(c, (void**)&Z.d_slow, &Z.slow_cap, (size_t)b->n_tiles + 64, sizeof(int)))) return rc2;
    if (certified && c->use_kmer_streams) {
        if ((rc2 = ensure(c, (void**)&Z.d_items, &Z.items_cap, (size_t)b->n_stiles + 64, sizeof(ItemDesc)))) return rc2;
    }
    if (with_output) {
        const double bound = c->dwell_hi * (double)b->n_events;
        if (bound <= 4.0e10) {
            const size_t need = (size_t)bound;
            if ((rc2 = ensure(c, (void**)&Z.d_sig, &Z.sig_cap, need + 64, sizeof(int16_t)))) return rc2;
            if (certified && c->use_kmer_streams)
                if ((rc2 = ensure(c, (void**)&Z.d_fix, &Z.fix_cap, (c->force_fix ? need : need / 256) + 65536, sizeof(FixEntry)))) return rc2;
        }
    }
    return SQG_OK;
}

// Staging shared by sqg_batch_stage (reads come from the host: seqs != null) and sqg_batch_sample (reads were
// sampled on the device: seqs == null, d_rec holds one SampleRec per read and k_copy_reads fills the base buffer).
// One object per call, one method per pass (round 5: it was one 480-line function); stage_common below calls them in order.
#include <chrono>
struct Staging {
    // the call
    sqg_ctx* c; int n; const char* seqs; const int64_t* seq_off; const int32_t* worker; const SampleRec* d_rec; const uint32_t* d_mstate;
    const sqg_profile_t& p;
    const bool rna, prefix;
    const int k;
    sqg_batch* b = nullptr;
    // geometry(): descriptors, tiles
    std::vector<ReadDesc> rd; std::vector<int> wk;
    long long nb = 0, nev = 0, ntile = 0, nst = 0;
    int lean_ev = 0;
    std::vector<uint8_t> hb;                                     // base_buffer(): the reads with prefix / stall attached (host-provided reads)
    // chains(): the worker chains; links(): their cut into links (of whole reads, or of pieces of reads)
    std::vector<int> chain_of, chain_off, chain_reads, wchain_off, wlink_off, wlink_worker;
    int n_wchains = 0, forced = -1;
    std::vector<long long> wchain_ev, chain_ev;
    struct Piece { int read, e_lo, e_hi, pad; };                 // events [e_lo, e_hi) of a read (k_part_events)
    std::vector<Piece> pieces;
    bool part_one = false, part_ok = false;
    // slices(): the bucketed hand-out's geometry; launch_order()
    std::vector<int> link_q;                                     // the worker chain of every link
    std::vector<uint32_t> link_slot, wchain_total;               // one partition: every link's first slot, every worker chain's events
    std::vector<int> chain_order;
    // draws(): the workers' scalar streams as they stood (a staging that fails afterwards puts them back: bail())
    std::vector<uint32_t> snap_time; std::vector<long long> snap_off, snap_med;
    std::vector<int> blk_read;                                   // dwell_blocks()

    Staging(sqg_ctx* c_, int n_, const char* seqs_, const int64_t* seq_off_, const int32_t* worker_, const SampleRec* d_rec_, const uint32_t* d_mstate_)
        : c(c_), n(n_), seqs(seqs_), seq_off(seq_off_), worker(worker_), d_rec(d_rec_), d_mstate(d_mstate_), p(c_->cfg.profile),
          rna(c_->cfg.flags & SQG_RNA), prefix(c_->cfg.flags & SQG_PREFIX), k(c_->k) {}
    int bail(int code) { c->time_c = snap_time; c->off_x = snap_off; c->med_x = snap_med; sqg_batch_free(c, b); b = nullptr; return code; }

    // ---- pass 1: worker ids, segment geometry, tiles
    int geometry() {
        b = new (std::nothrow) sqg_batch();
        if (!b) return SQG_ENOMEM;
        b->n = n; b->seq = c->next_stage;
        b->ev_off.assign((size_t)n + 1, 0); b->sig_off.assign((size_t)n + 1, 0);
        b->offset.resize((size_t)n); b->median.resize((size_t)n);
        rd.resize((size_t)n);
        wk.resize((size_t)n);

        // pass 1: worker ids, segment geometry
        nb = 0; nev = 0;
        for (int i = 0; i < n; i++) {
            const int w = worker ? worker[i] : sqg_worker_of(i, n, c->T);
            if (w < c->wlo || w >= c->whi) { delete b; b = nullptr; c->err = "read assigned to a worker this context does not own"; return SQG_EINVAL; }
            wk[(size_t)i] = w - c->wlo;
            const long long len = seq_off[i + 1] - seq_off[i];
            if (len < 0 || len > 2000000000LL) { delete b; b = nullptr; return SQG_EINVAL; }
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
        // 64-event tiles (the work unit of k_samples); a tile never spans two reads
        ntile = 0;
        for (int i = 0; i < n; i++) { rd[(size_t)i].tile_off = (int)ntile; rd[(size_t)i].fast = 0; rd[(size_t)i].stile_off = 0; rd[(size_t)i].slot0 = 0; ntile += (rd[(size_t)i].ne0 + rd[(size_t)i].ne1 + 63) / 64; }
        if (ntile > 2000000000LL) { delete b; b = nullptr; c->err = "batch too large"; return SQG_EINVAL; }
        b->n_tiles = ntile;
        lean_ev = 64 * c->lean_epl;
        nst = 0;                                        // super tiles of 64*lean_epl events (work items of k_samples_lean)
        for (int i = 0; i < n; i++) { rd[(size_t)i].stile_off = (int)nst; rd[(size_t)i].slot0 = 0; nst += (rd[(size_t)i].ne0 + rd[(size_t)i].ne1 + lean_ev - 1) / lean_ev; }
        b->n_stiles = nst;
        // the tile -> read maps are filled on the device (k_fill_tiles) once the descriptors are there
            return SQG_OK;
    }

    // ---- pass 2: base buffer (prefix/stall attached as src/genread.c:95-123 does)
    void base_buffer() {
        hb.assign(seqs ? (size_t)nb + 16 : 0, (uint8_t)'A');
        for (int i = 0; seqs && i < n; i++) {
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
    }

    // ---- pass 3: per-worker chains in batch order; the host-side scalar streams advance in that order
    void chains() {
        std::vector<int> count((size_t)c->nw, 0);
        for (int i = 0; i < n; i++) count[(size_t)wk[(size_t)i]]++;
        chain_of.assign((size_t)c->nw, -1); chain_off.clear();
        chain_off.push_back(0);
        for (int w = 0; w < c->nw; w++) if (count[(size_t)w]) { chain_of[(size_t)w] = (int)chain_off.size() - 1; chain_off.push_back(chain_off.back() + count[(size_t)w]); }
        b->n_chains = (int)chain_off.size() - 1;
        std::vector<int> fill(chain_off.begin(), chain_off.end() - 1);
        chain_reads.resize((size_t)n);
        for (int i = 0; i < n; i++) chain_reads[(size_t)fill[(size_t)chain_of[(size_t)wk[(size_t)i]]]++] = i;

        // Few workers, many reads (`-t 1`, `-t 8 -K 1000`): a worker chain would be one workgroup walking its reads one after
        // the other.  It is cut into links of whole reads, which k_events walks concurrently after k_link_hist/k_link_prefix
        // have prepared each link's view of the worker's k-mer streams.  SQG_SPLIT_CHAINS=0 disables this, =N forces it
        // with N links as the target (tests).
        wchain_off = chain_off;                                                  // the worker chains: the host's scalar streams follow these
        n_wchains = b->n_chains;
        b->n_wchains = n_wchains;
        wchain_ev.assign((size_t)n_wchains, 0);
        for (int i = 0; i < n; i++) wchain_ev[(size_t)chain_of[(size_t)wk[(size_t)i]]] += rd[(size_t)i].ne0 + rd[(size_t)i].ne1;
        for (long long v : wchain_ev) b->max_wchain_ev = std::max(b->max_wchain_ev, v);
    }

    // ---- few workers: the chains cut into links
    void links() {
        wlink_off.assign(1, 0); wlink_worker.clear(); pieces.clear();
        // the hand-out over bucketed events (k_part.h) instead of per-link rows: k > 6 (up to PART_MAX partitions of 4096 streams), and
        // k <= 6 as its one-partition case -- nothing to bucket, the events stay in chain order -- on devices with ordered LDS atomics
        part_one = c->num_kmer <= PART_SUB && c->lds_ordered && !(c->cfg.flags & SQG_METH) && !SQG_DEV_ENV("SQG_PART_WG_EVENTS");
        const char* split_env = SQG_DEV_ENV("SQG_SPLIT_CHAINS");
        forced = split_env ? atoi(split_env) : -1;
        part_ok = ((c->num_kmer > PART_SUB && c->num_kmer <= PART_MAX * PART_SUB) || part_one) && nev < 4294967000LL && !SQG_DEV_ENV("SQG_NO_PART");
        // ... when the events outweigh the tables: every (worker chain, partition) costs at least one 16-KiB table in three passes
        // (hundreds of workers with a read or two each: the per-link rows, or no cut at all, are the better choice)
        if (forced < 0 && !c->range_mode && nev < (long long)n_wchains * (part_one ? 1 : (c->num_kmer + PART_SUB - 1) >> PART_SUB_BITS) * 1024) part_ok = false;
        {
            // (bucketed hand-out: a link is one wavefront of k_part_events -- 8 per SIMD -- and may hold pieces of reads, so that a few
            // long reads are worth cutting as well; one workgroup of k_events and whole reads otherwise)
            const bool wave_links = part_ok && c->lds_ordered && !(c->cfg.flags & SQG_METH) && !SQG_DEV_ENV("SQG_PART_WG_EVENTS");
            const bool multi = n > n_wchains;
            const bool want = c->range_mode ? n > 0 : forced >= 0 ? (forced > 0 && (multi || wave_links))
                                                                  : ((multi || (wave_links && n_wchains <= 16)) && n_wchains < (wave_links && part_one ? 2048 : 1024) && nev >= 65536);   // (measured: from 2048 / 1024 chains on, one workgroup of k_events per chain is as fast or faster)
            if (c->use_kmer_streams && want) {
                const size_t row_bytes = (size_t)c->num_kmer * sizeof(uint32_t);
                // (one wavefront per link: 8192 links fill the machine twice over; a small batch -- the reference's default -K 1000 is 1e7
                // events -- is better off with half as many links of twice the length: fewer ragged last segments, fewer cut reads.
                // Measured at 1000 reads per batch, 9-mers: event side 0.224 -> 0.214 ms; with the longer slices below 0.196)
                long long target = forced > 0 ? forced : wave_links ? (nev < 25000000LL ? 4096 : 8192) : part_ok ? 4096 : 2048;
                if (!part_ok) target = std::min<long long>(target, std::max<long long>(8, (long long)(((size_t)1 << 30) / row_bytes)));
                std::vector<int> link_off(1, 0);
                b->pieces = wave_links;
                const bool whole_links = !SQG_DEV_ENV("SQG_NO_WHOLE_LINKS");           // A/B (development library)
                for (int q = 0; q < n_wchains && wave_links; q++) {
                    // a link of k_part_events is a run of PIECES: whole reads, and the pieces (whole 512-event segments) of reads longer
                    // than a link should be -- one wavefront walks a link, and a read of 10^5 events would keep it busy ten times as
                    // long as the others (k_part_events.h: what a piece needs from the pieces before it)
                    const int lo = wchain_off[(size_t)q], hi = wchain_off[(size_t)q + 1];
                    const long long lq = std::max<long long>(1, nev > 0 ? (target * wchain_ev[(size_t)q] + nev - 1) / nev : 1);
                    // (a link stays below 2^14 events -- an event's record for the sample kernels carries its slot within the (link,
                    // partition) in EVR_REL_BITS bits, k_common.h: a link closes at less than per + (per + per / 4) events)
                    const long long per_cap = ((1 << EVR_REL_BITS) - 1) * 4 / 9;
                    const long long per = std::min<long long>(std::max<long long>(1, (wchain_ev[(size_t)q] + lq - 1) / lq), per_cap);
                    const long long chunk = std::max<long long>(PEV_SEG, per / PEV_SEG * PEV_SEG);
                    long long acc = 0;
                    auto close = [&]() { link_off.push_back((int)pieces.size()); acc = 0; };
                    for (int ci = lo; ci < hi; ci++) {
                        const int r = chain_reads[(size_t)ci];
                        const long long ne = rd[(size_t)r].ne0 + rd[(size_t)r].ne1;
                        if (ne <= 0) continue;
                        if (ne <= per + per / 4 || ne <= PEV_SEG) {
                            pieces.push_back(Piece{r, 0, (int)ne, 0});
                            if ((acc += ne) >= per) close();
                        } else if (whole_links && per == per_cap && ne < (1 << EVR_REL_BITS) - PEV_SEG) {
                            // a large batch -- the cap above, not the link target, sets `per` -- and a read that is longer than that but fits a link:
                            // a link of its own, whole (round 5: at the headline size every 10-kb read, 9992 events, was cut into 7168 + 2824: twice the
                            // links, half of them short, and the cut reads' tile offsets patched by k_part_tile_bases).  A small batch keeps cutting:
                            // there the pieces are what fills the machine
                            if (acc > 0) close();
                            pieces.push_back(Piece{r, 0, (int)ne, 0});
                            close();
                        } else {
                            for (long long e = 0; e < ne; e += chunk) {
                                const long long e_hi = std::min(e + chunk, ne);
                                if (acc > 0 && acc + (e_hi - e) > per + per / 4) close();
                                if (e > 0) b->split_reads = true;
                                pieces.push_back(Piece{r, (int)e, (int)e_hi, 0});
                                if ((acc += e_hi - e) >= per) close();
                            }
                        }
                    }
                    if (acc > 0) close();
                    wlink_off.push_back((int)link_off.size() - 1);
                    wlink_worker.push_back(rd[(size_t)chain_reads[(size_t)lo]].worker);
                }
                for (int q = 0; q < n_wchains && !wave_links; q++) {
                    const int lo = wchain_off[(size_t)q], hi = wchain_off[(size_t)q + 1];
                    long long lq = nev > 0 ? (target * wchain_ev[(size_t)q] + nev - 1) / nev : 1;
                    lq = std::max<long long>(1, std::min<long long>(lq, hi - lo));
                    const long long per = (wchain_ev[(size_t)q] + lq - 1) / lq;
                    long long acc = 0;
                    for (int ci = lo; ci < hi; ci++) {
                        const ReadDesc& d = rd[(size_t)chain_reads[(size_t)ci]];
                        acc += d.ne0 + d.ne1;
                        // a link's per-k-mer sample counts are 32-bit
                        const long long nxt = ci + 1 < hi ? rd[(size_t)chain_reads[(size_t)ci + 1]].ne0 + rd[(size_t)chain_reads[(size_t)ci + 1]].ne1 : 0;
                        if (ci + 1 == hi || acc >= per || (double)(acc + nxt) * c->dwell_hi >= 2147483648.0) { link_off.push_back(ci + 1); acc = 0; }
                    }
                    wlink_off.push_back((int)link_off.size() - 1);
                    wlink_worker.push_back(rd[(size_t)chain_reads[(size_t)lo]].worker);
                }
                chain_off.swap(link_off);
                b->n_chains = (int)chain_off.size() - 1;
                b->split = true;
            }
        }
    }

    // ---- the events per chain / link; the bucketed hand-out's slices and slots
    int slices() {
        // launch order: longest chain first, so the tail of the grid is made of short chains
        chain_ev.assign((size_t)b->n_chains, 0);
        for (int q = 0; q < b->n_chains; q++)
            for (int ci = chain_off[(size_t)q]; ci < chain_off[(size_t)q + 1]; ci++)
                chain_ev[(size_t)q] += b->pieces ? pieces[(size_t)ci].e_hi - pieces[(size_t)ci].e_lo : rd[(size_t)chain_reads[(size_t)ci]].ne0 + rd[(size_t)chain_reads[(size_t)ci]].ne1;
        // k > 6, split chains: the hand-out runs over events bucketed by the top bits of the rank (k_part.h).  The events of a
        // (worker chain, partition) are cut into slices of equal length, one workgroup of k_part_hist / k_part_hand each; the
        // slices' 4096-entry tables are what k_part_scan sweeps (SQG_PART_SLICE: events per slice, tests; SQG_NO_PART=1: the
        // per-link rows of round 1, for A/B runs).
        if (b->split && part_ok) {
            const int n_part = (c->num_kmer + PART_SUB - 1) >> PART_SUB_BITS;
            link_q.assign((size_t)b->n_chains, 0);
            for (int q = 0; q < n_wchains; q++) for (int l = wlink_off[(size_t)q]; l < wlink_off[(size_t)q + 1]; l++) link_q[(size_t)l] = q;
            const char* senv = SQG_DEV_ENV("SQG_PART_SLICE");
            // at most 4096 slices (a whole number of rounds of 4 wavefronts per CU for k_part_hand_ord), whole steps of the hand-out
            const long long n_pairs = (long long)n_wchains * n_part, want = std::max<long long>(1024, 4096 - n_pairs);
            long long len = senv ? atoll(senv) : ((nev + want - 1) / want + PART_STEP - 1) / PART_STEP * PART_STEP;
            // a slice's per-stream sample counts are 32-bit (a bucketed event carries its dwell in 16 bits)
            len = std::min<long long>(len, (long long)(4.0e9 / std::min(std::max(c->dwell_hi, 1.0), 65535.0)));
            len = std::max<long long>(PART_STEP, len / PART_STEP * PART_STEP);
            if (!senv) len = std::max<long long>(len, std::min<long long>(16 * PART_STEP, (long long)(4.0e9 / std::min(std::max(c->dwell_hi, 1.0), 65535.0)) / PART_STEP * PART_STEP));   // small batches: fewer, not shorter slices (a slice costs a 16-KiB table in three passes; 16384 instead of 8192 events: event side -4 % at 4096 reads per batch, -9 % at 1000)
            b->slice_len = (uint32_t)len;
            b->max_slices = (long long)(nev / len) + (long long)n_wchains * n_part;
            b->part = true;
            b->one = part_one;
            if (b->pieces && !part_one) {
                // the sample kernels need an event's link (SigParams.evrec32): a read that is one piece says it itself (ReadDesc.slot0, not
                // used otherwise with several partitions); the tiles of a read cut into pieces are looked up in tile_link (-1 here)
                for (int l = 0; l < b->n_chains; l++)
                    for (int ci = chain_off[(size_t)l]; ci < chain_off[(size_t)l + 1]; ci++) {
                        const Piece& pc = pieces[(size_t)ci];
                        ReadDesc& d = rd[(size_t)pc.read];
                        d.slot0 = (pc.e_lo == 0 && pc.e_hi == d.ne0 + d.ne1) ? l : -1;
                    }
            }
            if (b->one) {                                             // part[] is the worker chains one after the other, each in chain order
                link_slot.assign((size_t)b->n_chains, 0u);
                long long at = 0;
                for (int l = 0; l < b->n_chains; l++) { link_slot[(size_t)l] = (uint32_t)at; at += chain_ev[(size_t)l]; }
                for (int q = 0; q < n_wchains; q++) wchain_total.push_back((uint32_t)wchain_ev[(size_t)q]);
                at = 0;                                               // ... and every read's first slot: a read's events are consecutive slots
                for (int q = 0; q < n_wchains; q++)
                    for (int ci = wchain_off[(size_t)q]; ci < wchain_off[(size_t)q + 1]; ci++) {
                        ReadDesc& d = rd[(size_t)chain_reads[(size_t)ci]];
                        d.slot0 = (int)(uint32_t)at;
                        at += d.ne0 + d.ne1;
                    }
            }
        }
        if (c->use_kmer_streams && c->num_kmer > 4096 && !b->part && (double)b->max_wchain_ev * c->dwell_hi >= 4294967295.0 - (double)LCG_ORD2) {
            delete b; b = nullptr; c->err = "one worker's reads of a batch may draw more than 3.2e9 samples (k > 6): use smaller batches"; return SQG_EINVAL;
        }
            return SQG_OK;
    }

    // ---- launch order: longest chain first
    void launch_order() {
        // (a counting sort over 4096 length classes: exact order within a class does not matter for the tail)
        chain_order.resize((size_t)b->n_chains);
        {
            long long mx = 1;
            for (long long v : chain_ev) mx = std::max(mx, v);
            constexpr int NB = 4096;
            std::vector<int> cnt(NB + 1, 0);
            auto cls = [&](long long v) { return (int)((NB - 1) - (v * (NB - 1)) / mx); };    // longest -> class 0
            for (long long v : chain_ev) cnt[(size_t)cls(v) + 1]++;
            for (int q = 0; q < NB; q++) cnt[(size_t)q + 1] += cnt[(size_t)q];
            for (int q = 0; q < b->n_chains; q++) chain_order[(size_t)cnt[(size_t)cls(chain_ev[(size_t)q])]++] = q;
        }
    }

    // ---- the per-read scalar draws (host libm), the time stream, `fast`
    void draws() {
        const bool no_lean = SQG_DEV_ENV("SQG_TEST_NO_LEAN") != nullptr;
        // the workers' scalar streams advance below; a staging that fails afterwards (allocation) puts them back
        snap_time = c->time_c; snap_off = c->off_x; snap_med = c->med_x;
        // the per-read scalar draws (host libm, so that `offset` / `median_before` are the doubles the CPU reference prints): 2 x 16384
        // log / sqrt / cos per batch, 0.8 ms on one thread.  chain_reads[] lists the reads worker chain by worker chain, in batch order
        // within a chain; the helper threads of the context take ranges [lo, hi) of it and leave the draws in two compact arrays (they
        // touch nothing else: the descriptors sit in this thread's cache).  A range that starts in the middle of a chain starts from the
        // chain's streams moved past the reads before it: two steps of each per read (src/rand.h:87-94: nrng draws two uniforms, neither
        // can be 0).  The streams themselves, the time stream and the descriptors are left to this thread.
        const bool ideal = (c->cfg.flags & SQG_IDEAL) != 0;
        std::vector<double> off_d, med_d;                              // by position in chain_reads
        std::vector<int> rd_worker;                                    // ... and the worker of the read at that position
        std::vector<int> pre((size_t)n_wchains, 0);                   // draws of chain q that came from the context's draw-ahead thread (below)
        auto draw_range = [&](const int lo, const int hi) {
            int ci = lo;
            while (ci < hi) {
                const size_t w = (size_t)rd_worker[(size_t)ci];
                const int q = chain_of[w], c_lo = wchain_off[(size_t)q], c_hi = wchain_off[(size_t)q + 1];
                long long off = snap_off[w], med = snap_med[w];
                if (ci > c_lo) {
                    const uint32_t j = c->jump2((unsigned long long)(ci - c_lo));
                    off = (long long)lcg_mul(canon(off), j); med = (long long)lcg_mul(canon(med), j);
                }
                const int stop = std::min(hi, c_hi);
                if (ci < c_lo + pre[(size_t)q]) {                       // (the chain's first draws came from the draw-ahead thread)
                    ci = std::min(stop, c_lo + pre[(size_t)q]);
                    if (ci >= stop) continue;
                    const uint32_t j = c->jump2((unsigned long long)(ci - c_lo));
                    off = (long long)lcg_mul(canon(snap_off[w]), j); med = (long long)lcg_mul(canon(snap_med[w]), j);
                }
                for (; ci < stop; ci++) {
                    off_d[(size_t)ci] = host_nrng(p.offset_mean, p.offset_std, &off);                     // src/gensig.c:315
                    med_d[(size_t)ci] = host_nrng(p.median_before_mean, p.median_before_std, &med);       // src/gensig.c:316
                }
            }
        };
        // few workers: what the context's draw-ahead thread has ready (h_common.h, DrawAhead) comes first -- a prefix of every worker chain
        std::vector<long long> pre_off((size_t)n_wchains, 0), pre_med((size_t)n_wchains, 0);   // ... and the streams' states behind them
        int n_pre = 0;
        const bool ahead = !ideal && c->nw <= 16 && n > 0 && usable_cpus() >= 2 && !SQG_DEV_ENV("SQG_NO_DRAW_AHEAD");
        if (ahead) {
            if (!c->draw_ahead) c->draw_ahead = new DrawAhead(c->nw, p);
            off_d.resize((size_t)n); med_d.resize((size_t)n);
            for (int q = 0; q < n_wchains; q++) {
                const int c_lo = wchain_off[(size_t)q], c_hi = wchain_off[(size_t)q + 1];
                const size_t w = (size_t)rd[(size_t)chain_reads[(size_t)c_lo]].worker;
                pre_off[(size_t)q] = c->off_x[w]; pre_med[(size_t)q] = c->med_x[w];
                pre[(size_t)q] = (int)c->draw_ahead->take((int)w, (size_t)(c_hi - c_lo), &pre_off[(size_t)q], &pre_med[(size_t)q], off_d.data() + c_lo, med_d.data() + c_lo);
                n_pre += pre[(size_t)q];
            }
        }
        // (measured, 16384 reads per batch: 1.30 ms on one thread, 1.10 with two, 0.93 with four, 0.88 with six -- the helpers sleep for
        // milliseconds between two batches and wake slowly).  sqg_set_stage_threads fixes the number (a host that runs one context per GPU
        // on a CPU quota shared by eight of them); automatic: four from 8192 reads per batch on, never more than the CPUs this process may use.
        const int dev_th = dev_env_int(SQG_DEV_ENV("SQG_STAGE_THREADS"), 0);       // (development build: A/B runs)
        const int forced_th = dev_th > 0 ? dev_th : c->stage_threads;
        const int n_left = n - n_pre;                                  // draws still to be made here
        const int want_th = ideal ? 1 : forced_th > 0 ? forced_th : n_left >= 8192 ? std::min(4, usable_cpus()) : 1;
        const int nth = n_left > 0 ? std::max(1, std::min(want_th, std::max(n_left, 1))) : 1;
        c->stage_threads_last = nth;
        if (nth > 1) {
            off_d.resize((size_t)n); med_d.resize((size_t)n); rd_worker.resize((size_t)n);
            for (int ci = 0; ci < n; ci++) rd_worker[(size_t)ci] = rd[(size_t)chain_reads[(size_t)ci]].worker;
            std::vector<std::function<void()>> jobs;
            const int per = (n + nth - 1) / nth;
            for (int t = 1; t < nth; t++) jobs.push_back([&, t] { draw_range(std::min(t * per, n), std::min((t + 1) * per, n)); });
            c->pool_threads.post(std::move(jobs));
            draw_range(0, std::min(per, n));
            c->pool_threads.wait();
        }
        for (int q = 0; q < n_wchains; q++) {
            const size_t w = (size_t)rd[(size_t)chain_reads[(size_t)wchain_off[(size_t)q]]].worker;
            uint32_t tc = c->time_c[w];
            const int c_lo = wchain_off[(size_t)q], c_hi = wchain_off[(size_t)q + 1], c_pre = c_lo + pre[(size_t)q];
            if (nth == 1 && pre[(size_t)q] > 0) { c->off_x[w] = pre_off[(size_t)q]; c->med_x[w] = pre_med[(size_t)q]; }   // (the streams behind the prefix: where the draws made here go on)
            for (int ci = c_lo; ci < c_hi; ci++) {   // batch order within the worker
                const int i = chain_reads[(size_t)ci];
                ReadDesc& d = rd[(size_t)i];
                if (ideal) {                                          // src/gensig.c:311-313
                    d.offset = p.offset_mean; b->median[(size_t)i] = p.median_before_mean;
                } else if (nth > 1 || ci < c_pre) {
                    d.offset = off_d[(size_t)ci]; b->median[(size_t)i] = med_d[(size_t)ci];
                } else {                                              // src/gensig.c:315-316
                    d.offset = host_nrng(p.offset_mean, p.offset_std, &c->off_x[w]);
                    b->median[(size_t)i] = host_nrng(p.median_before_mean, p.median_before_std, &c->med_x[w]);
                }
                b->offset[(size_t)i] = d.offset;
                d.fast = (c->cfg.mode == SQG_MODE_CERTIFIED && c->use_kmer_streams && c->dwell_hi <= (double)MULT_N && !no_lean &&
                          c->amp_floor - d.offset > 4.0 && c->amp_ceil - d.offset < 65000.0) ? 1 : 0;
                d.time_c0 = tc;
                if (c->use_dwell_stream)                              // two draws per event (src/gensig.c:255)
                    tc = lcg_mul(tc, c->jump2((unsigned long long)(d.ne0 + d.ne1)));
            }
            c->time_c[w] = tc;
            if (nth > 1) {                                            // the streams behind the chain's reads
                const uint32_t j = c->jump2((unsigned long long)(c_hi - c_lo));
                c->off_x[w] = (long long)lcg_mul(canon(c->off_x[w]), j); c->med_x[w] = (long long)lcg_mul(canon(c->med_x[w]), j);
            }
            if (ahead) c->draw_ahead->rebase((int)w, c->off_x[w], c->med_x[w]);
        }
        if (!c->use_dwell_stream) {                           // constant dwell: lengths are known now
            const unsigned long long sps = (unsigned long long)(int)p.dwell_mean;
            b->seglen_host.resize((size_t)2 * n);
            for (int i = 0; i < n; i++) { b->seglen_host[(size_t)2 * i] = sps * rd[(size_t)i].ne0; b->seglen_host[(size_t)2 * i + 1] = sps * rd[(size_t)i].ne1; }
        }
    }

    // ---- dwell kernel launch geometry: first read of every DW_EPB-event block
    void dwell_blocks() {
        const long long nblk = (nev + DW_EPB - 1) / DW_EPB;
        blk_read.assign((size_t)std::max<long long>(nblk, 1), 0);
        {
            int r = 0;
            for (long long bi = 0; bi < nblk; bi++) {
                const long long g = bi * DW_EPB;
                while (r + 1 < n && g >= rd[(size_t)r + 1].ev_off) r++;
                blk_read[(size_t)bi] = r;
            }
        }
    }

    // ---- the batch's device block, the uploads, the staging kernels
    int upload() {
#define CHKB(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { c->err = std::string(#call) + ": " + hipGetErrorString(e_); return bail(e_ == hipErrorOutOfMemory ? SQG_ENOMEM : SQG_EDEVICE); } } while (0)
        size_t meta_bytes = 0, mo_err = 0, mo_reads = 0, mo_blk = 0, mo_coff = 0, mo_crd = 0, mo_ord = 0, mo_wlo = 0, mo_wlw = 0, mo_cb = 0, mo_ls = 0, mo_wt = 0, mo_pc = 0;
        {   // one device allocation per batch, carved into the batch's arrays (256-byte aligned)
            size_t off = 0;
            auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
            // the host-built arrays first and back to back: they go up in one copy from the batch's pinned mirror
            const size_t o_err = carve(256),
                         o_reads = carve(std::max<size_t>(1, rd.size()) * sizeof(ReadDesc)),
                         o_blk = carve(blk_read.size() * sizeof(int)), o_coff = carve(chain_off.size() * sizeof(int)),
                         o_crd = carve(std::max<size_t>(1, chain_reads.size()) * sizeof(int)),
                         o_ord = carve(std::max<size_t>(1, chain_order.size()) * sizeof(int)),
                         o_wlo = carve(wlink_off.size() * sizeof(int)), o_wlw = carve(std::max<size_t>(1, wlink_worker.size()) * sizeof(int)),
                         o_cb = carve(std::max<size_t>(1, link_q.size()) * sizeof(int)),
                         o_ls = carve(std::max<size_t>(1, link_slot.size()) * sizeof(uint32_t)), o_wt = carve(std::max<size_t>(1, wchain_total.size()) * sizeof(uint32_t)),
                         o_pc = carve(std::max<size_t>(1, pieces.size()) * sizeof(Piece));
            meta_bytes = off;
            const size_t o_ptot = carve(std::max<size_t>(1, pieces.size()) * sizeof(uint32_t));
            const size_t o_bases = carve((size_t)nb + 1024),        // (k_part_events reads a whole segment + halo from a read's last segment on)
                         o_st = carve((size_t)std::max<long long>(nst, 1) * sizeof(int)), o_t = carve((size_t)std::max<long long>(ntile, 1) * sizeof(int));
            mo_err = o_err; mo_reads = o_reads; mo_blk = o_blk; mo_coff = o_coff; mo_crd = o_crd; mo_ord = o_ord; mo_wlo = o_wlo; mo_wlw = o_wlw; mo_cb = o_cb; mo_ls = o_ls; mo_wt = o_wt; mo_pc = o_pc;
            // a freed batch's block, pinned offsets and events are reused when they are large enough
            for (size_t pi = 0; pi < c->pool.size(); pi++) {
                sqg_ctx::Recycled& r = c->pool[pi];
                if (r.block_bytes >= off && r.h_n >= (size_t)n + 1 + SQG_HRES_LL && r.h_meta_bytes >= meta_bytes) {
                    b->d_block = r.d_block; b->block_bytes = r.block_bytes; b->h_sigoff = r.h_sigoff; b->h_sigoff_dev = r.h_sigoff_dev; b->h_n = r.h_n;
                    for (int i = 0; i < 8; i++) b->ev[i] = r.ev[i];
                    b->h_meta = r.h_meta; b->h_meta_bytes = r.h_meta_bytes; b->ev_staged = r.ev_staged;
                    c->pool.erase(c->pool.begin() + (long)pi);
                    break;
                }
            }
            if (!b->d_block) {
                if (c->pool.size() >= 4) {                      // nothing fits: make room
                    sqg_ctx::Recycled& r = c->pool.front();
                    (void)hipFree(r.d_block); (void)hipHostFree(r.h_sigoff); (void)hipHostFree(r.h_meta); for (auto& e : r.ev) if (e) (void)hipEventDestroy(e);
                    if (r.ev_staged) (void)hipEventDestroy(r.ev_staged);
                    c->pool.erase(c->pool.begin());
                }
                b->block_bytes = off + off / 8;                 // slack: the next batches are about this size
                CHKB(hipMalloc(&b->d_block, b->block_bytes));
                b->h_meta_bytes = meta_bytes + meta_bytes / 8;
                CHKB(hipHostMalloc(&b->h_meta, b->h_meta_bytes, hipHostMallocDefault));
            }
            uint8_t* base = b->d_block;
            b->d_err = (unsigned int*)(base + o_err);
            b->d_bases = base + o_bases; b->d_reads = (ReadDesc*)(base + o_reads); b->d_blk_read = (int*)(base + o_blk);
            b->d_chain_off = (int*)(base + o_coff); b->d_chain_reads = (int*)(base + o_crd); b->d_stile_read = (int*)(base + o_st);
            b->d_tile_read = (int*)(base + o_t); b->d_chain_order = (int*)(base + o_ord);
            b->d_wlink_off = (int*)(base + o_wlo); b->d_wlink_worker = (int*)(base + o_wlw);
            b->d_link_q = (int*)(base + o_cb); b->d_link_slot = (uint32_t*)(base + o_ls); b->d_wchain_total = (uint32_t*)(base + o_wt);
            b->d_pieces = (int4*)(base + o_pc); b->d_piece_total = (uint32_t*)(base + o_ptot); b->n_pieces = (int)pieces.size();
        }
        {   // the host-built arrays -> pinned mirror -> one asynchronous copy
            uint8_t* m = b->h_meta;
            memset(m + mo_err, 0, 256);                         // the batch's error word starts clear
            if (n) memcpy(m + mo_reads, rd.data(), rd.size() * sizeof(ReadDesc));
            memcpy(m + mo_blk, blk_read.data(), blk_read.size() * sizeof(int));
            memcpy(m + mo_coff, chain_off.data(), chain_off.size() * sizeof(int));
            if (n) memcpy(m + mo_crd, chain_reads.data(), chain_reads.size() * sizeof(int));
            if (b->n_chains) memcpy(m + mo_ord, chain_order.data(), chain_order.size() * sizeof(int));
            if (b->split) {
                memcpy(m + mo_wlo, wlink_off.data(), wlink_off.size() * sizeof(int));
                memcpy(m + mo_wlw, wlink_worker.data(), wlink_worker.size() * sizeof(int));
            }
            if (b->pieces && !pieces.empty()) memcpy(m + mo_pc, pieces.data(), pieces.size() * sizeof(Piece));
            if (b->part) {
                memcpy(m + mo_cb, link_q.data(), link_q.size() * sizeof(int));
                if (b->one) { memcpy(m + mo_ls, link_slot.data(), link_slot.size() * sizeof(uint32_t)); memcpy(m + mo_wt, wchain_total.data(), wchain_total.size() * sizeof(uint32_t)); }
            }
            CHKB(hipMemcpyAsync(b->d_block, m, meta_bytes, hipMemcpyHostToDevice, c->stage_stream));
        }
        if (seqs) CHKB(hipMemcpyAsync(b->d_bases, hb.data(), hb.size(), hipMemcpyHostToDevice, c->stage_stream));
        else CHKB(hipMemsetAsync(b->d_bases + nb, 'A', 16, c->stage_stream));
        if (!seqs && n) {                                      // the reads come from the resident genome
            hipLaunchKernelGGL(k_copy_reads, dim3((unsigned)n), dim3(256), 0, c->stage_stream, c->genome, d_rec, b->d_reads, b->d_bases, n,
                               rna ? 1 : 0, prefix ? 1 : 0, d_mstate);
            CHKB(hipGetLastError());
            // reads shorter than k (only --full-contigs can produce them here): the reference generates 5 events of a fixed
            // sequence instead (src/gensig.c:242-245)
            for (int i = 0; i < n; i++) {
                const long long len = seq_off[i + 1] - seq_off[i];
                const long long len0 = len + (prefix ? (rna ? (kPolyA + (long long)strlen(kAdaptorRna)) : ((long long)strlen(kStallDna) + (long long)strlen(kAdaptorDna))) : 0);
                if (len0 < k) CHKB(hipMemcpyAsync(b->d_bases + rd[(size_t)i].base_off, kShortHack, (size_t)rd[(size_t)i].len0, hipMemcpyHostToDevice, c->stage_stream));
            }
        }
        if (n) {
            hipLaunchKernelGGL(k_fill_tiles, dim3((unsigned)n), dim3(64), 0, c->stage_stream, b->d_reads, n, lean_ev, b->d_tile_read, b->d_stile_read);
            CHKB(hipGetLastError());
        }
        b->n_bases_total = nb;
        b->h_base_off.resize((size_t)n);
        for (int i = 0; i < n; i++) b->h_base_off[(size_t)i] = rd[(size_t)i].base_off;
        if (!b->h_sigoff) {
            b->h_n = (size_t)n + 1 + (size_t)n / 8 + SQG_HRES_LL;
            CHKB(hipHostMalloc(&b->h_sigoff, b->h_n * sizeof(long long), hipHostMallocMapped));
            CHKB(hipHostGetDevicePointer((void**)&b->h_sigoff_dev, b->h_sigoff, 0));
            for (auto& e : b->ev) CHKB(hipEventCreate(&e));
            CHKB(hipEventCreateWithFlags(&b->ev_staged, hipEventDisableTiming));
        }
        {   // k_fixup's report (error word, fix-up counts: the last SQG_HRES_LL words of the mapped block) starts out as "nothing reported":
            // a recycled block holds an earlier batch's words, and sqg_batch_wait reads them whenever this batch's k_fixup was launched
            unsigned int* const hres = reinterpret_cast<unsigned int*>(b->h_sigoff + (b->h_n - SQG_HRES_LL));
            hres[0] = 0x80000000u;                              // (bit 31: "k_fixup has not written its report": an error, should a wait ever see it)
            for (int i = 1; i < 4 + FIX_SHARDS; i++) hres[i] = 0u;
        }
        CHKB(hipEventRecord(b->ev_staged, c->stage_stream));   // sqg_batch_run waits for it on its own stream
            return SQG_OK;
#undef CHKB
    }
};

static int stage_common(sqg_ctx_t* c, int32_t n, const char* seqs, const int64_t* seq_off,
                        const int32_t* worker, const SampleRec* d_rec, sqg_batch_t** out, const uint32_t* d_mstate = nullptr) {
    *out = nullptr;
    static const bool st_on = getenv("SQG_STAGE_TIMING") != nullptr;
    auto st_t0 = std::chrono::steady_clock::now();
    auto st_mark = [&](const char* what) { if (st_on) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[stage] %-22s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - st_t0).count()); st_t0 = t; } };
    HIPCHK(c, hipSetDevice(c->cfg.device));
    Staging s(c, n, seqs, seq_off, worker, d_rec, d_mstate);
    int rc;
    if ((rc = s.geometry())) return rc;
    st_mark("descriptors+tiles");
    s.base_buffer();
    st_mark("base buffer");
    s.chains();
    s.links();
    if ((rc = s.slices())) return rc;
    s.launch_order();
    s.draws();
    st_mark("per-read draws");
    s.dwell_blocks();
    st_mark("chains+streams+blocks");
    if ((rc = s.upload())) return rc;
    st_mark("mallocs+enqueue");
    sqg_batch* b = s.b;
    // the read bytes of sqg_batch_stage come from the caller's (pageable) buffer through a stack-owned copy: wait for that
    // upload.  Everything else sits in memory the batch owns, and the staging stream is left running.
    if (seqs && hipStreamSynchronize(c->stage_stream) != hipSuccess) { c->err = "hipStreamSynchronize(stage_stream) failed"; return s.bail(SQG_EDEVICE); }
    st_mark("sync");
    // slots that have never held a batch are sized now, so that not even the first run allocates
    for (auto& Z : c->slot)
        if (Z.reads_cap == 0 && n > 0) { const int rg = grow_slot(c, Z, b, /*with_output=*/true); if (rg) return s.bail(rg); }
    for (auto& Q : c->cset)
        if (Q.dwell_cap == 0 && n > 0) { const int rg = grow_cset(c, Q, b); if (rg) return s.bail(rg); }
    c->next_stage++;
    b->staged = true;
    c->staged_q.push_back(b);
    *out = b;
    return SQG_OK;
}


extern "C" int sqg_batch_stage(sqg_ctx_t* c, int32_t n, const char* seqs, const int64_t* seq_off,
                               const int32_t* worker, sqg_batch_t** out) {
    if (!c || !out || n < 0 || (n > 0 && (!seqs || !seq_off))) return SQG_EINVAL;
    static const char none[1] = {0};
    static const int64_t zero_off[1] = {0};
    return stage_common(c, n, n > 0 ? seqs : none, n > 0 ? seq_off : zero_off, worker, nullptr, out);
}
