// h_sampler.h -- resident genome and the device-side read sampler (sqg_genome_load, sqg_batch_sample*, sqg_skip_reads, sqg_fetch_reads)
// Host side of include/sqg.h; included by sqg_hip.hip (one translation unit with the kernels), in the order listed there.
#pragma once

// ---- resident genome + device-side read sampler ("next" row of SURVEY.md section 8f) ----
static int genome_load_impl(sqg_ctx_t* c, const sqg_genome_t* g, const bool on_device) {
    if (!c || !g || g->n_contigs <= 0 || !g->seqs || !g->contig_off || g->rlen <= 0) return SQG_EINVAL;
    if (g->n_trans < 0 || (g->n_trans > 0 && (!g->trans_csum || !g->trans_idx))) return SQG_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const int nc = g->n_contigs;
    const long long total = g->contig_off[nc] - g->contig_off[0];
    std::vector<long long> off((size_t)nc + 1), cum((size_t)nc);
    for (int i = 0; i <= nc; i++) off[(size_t)i] = g->contig_off[i] - g->contig_off[0];
    long long run = 0;
    for (int i = 0; i < nc; i++) {
        const long long len = off[(size_t)i + 1] - off[(size_t)i];
        if (len < 0 || len > 2000000000LL) return SQG_EINVAL;
        run += len; cum[(size_t)i] = run;
    }
    (void)hipFree(c->d_genome); (void)hipFree(c->d_contig_off); (void)hipFree(c->d_cum); (void)hipFree(c->d_nprefix); c->d_nprefix = nullptr;
    (void)hipFree(c->d_trans_csum); (void)hipFree(c->d_trans_idx); (void)hipFree(c->d_samp);
    c->d_genome = nullptr; c->d_contig_off = nullptr; c->d_cum = nullptr; c->d_trans_csum = nullptr; c->d_trans_idx = nullptr; c->d_samp = nullptr;
    HIPCHK(c, hipMalloc(&c->d_genome, (size_t)total + 16));
    HIPCHK(c, hipMemcpy(c->d_genome, g->seqs + g->contig_off[0], (size_t)total, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(c->d_genome + total, 0, 16));
    HIPCHK(c, hipMalloc(&c->d_contig_off, off.size() * sizeof(long long)));
    HIPCHK(c, hipMemcpy(c->d_contig_off, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->d_cum, cum.size() * sizeof(long long)));
    HIPCHK(c, hipMemcpy(c->d_cum, cum.data(), cum.size() * sizeof(long long), hipMemcpyHostToDevice));
    if (g->n_trans > 0) {
        HIPCHK(c, hipMalloc(&c->d_trans_csum, (size_t)g->n_trans * sizeof(float)));
        HIPCHK(c, hipMemcpy(c->d_trans_csum, g->trans_csum, (size_t)g->n_trans * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc(&c->d_trans_idx, (size_t)g->n_trans * sizeof(int)));
        HIPCHK(c, hipMemcpy(c->d_trans_idx, g->trans_idx, (size_t)g->n_trans * sizeof(int), hipMemcpyHostToDevice));
    }
    // the workers' sampler streams: ref_pos = s, rand_strand = s+1, rand_rlen = s+3 (src/sim.c:238-247)
    HIPCHK(c, hipMalloc(&c->d_samp, (size_t)c->nw * 3 * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_init_sampler, dim3((unsigned)((c->nw + 255) / 256)), dim3(256), 0, c->stage_stream, c->d_samp,
                       (long long)c->cfg.seed, c->wlo, c->nw, c->num_kmer);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stage_stream));
    GenomeParams& G = c->genome;
    {   // 'N's per 64-base block, summed: the sampler's 10 % test reads two entries instead of the candidate read (genomes below 2^32 'N's: all)
        const long long n_blocks = (total + 63) / 64;
        HIPCHK(c, hipMalloc(&c->d_nprefix, ((size_t)n_blocks + 2) * sizeof(uint32_t)));
        if (n_blocks > 0) hipLaunchKernelGGL(k_nprefix_count, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, c->stream, c->d_genome, (long long)total, n_blocks, c->d_nprefix);
        hipLaunchKernelGGL(k_nprefix_scan, dim3(1), dim3(1024), 0, c->stream, c->d_nprefix, n_blocks);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    G.seq = c->d_genome; G.contig_off = c->d_contig_off; G.cum = c->d_cum; G.nprefix = c->d_nprefix;
    G.trans_csum = c->d_trans_csum; G.trans_idx = c->d_trans_idx;
    G.sum = total; G.grng_b = (double)(g->rlen / 2); G.n_contigs = nc; G.n_trans = g->n_trans; G.rlen = g->rlen;
    G.flags = (int)g->mode;
    (void)hipFree(c->d_meth); (void)hipFree(c->d_meth_has); (void)hipFree(c->d_meth_st);
    c->d_meth = nullptr; c->d_meth_has = nullptr; c->d_meth_st = nullptr;
    G.meth = nullptr; G.meth_has = nullptr;
    c->h_contig_off = off;
    c->full_next = 0;
    c->genome_loaded = true;
    return SQG_OK;
}

// --meth-freq: load_meth_freq (src/ref.c:291-361) has turned the file into one byte per base; the workers' rand_meth streams
// start at seed + 6 (src/sim.c:252-254)
extern "C" int sqg_genome_set_meth(sqg_ctx_t* c, const uint8_t* freq, const uint8_t* contig_has) {
    if (!c || !freq || !contig_has) return SQG_EINVAL;
    if (!c->genome_loaded) { c->err = "sqg_genome_load has not been called"; return SQG_EINVAL; }
    if (!(c->cfg.flags & SQG_METH)) { c->err = "sqg_genome_set_meth needs a context created with SQG_METH (5-letter pore table)"; return SQG_EINVAL; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    (void)hipFree(c->d_meth); (void)hipFree(c->d_meth_has); (void)hipFree(c->d_meth_st);
    c->d_meth = nullptr; c->d_meth_has = nullptr; c->d_meth_st = nullptr;
    const size_t total = (size_t)c->genome.sum;
    HIPCHK(c, hipMalloc(&c->d_meth, total + 16));
    HIPCHK(c, hipMemcpy(c->d_meth, freq, total, hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->d_meth_has, (size_t)c->genome.n_contigs));
    HIPCHK(c, hipMemcpy(c->d_meth_has, contig_has, (size_t)c->genome.n_contigs, hipMemcpyHostToDevice));
    std::vector<uint32_t> st((size_t)c->nw);
    for (int w = 0; w < c->nw; w++) st[(size_t)w] = canon((long long)c->cfg.seed + (long long)(w + c->wlo) * ((long long)c->num_kmer + 10) + 6);
    HIPCHK(c, hipMalloc(&c->d_meth_st, st.size() * sizeof(uint32_t)));
    HIPCHK(c, hipMemcpy(c->d_meth_st, st.data(), st.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->genome.meth = c->d_meth; c->genome.meth_has = c->d_meth_has;
    return SQG_OK;
}

extern "C" int sqg_genome_load(sqg_ctx_t* c, const sqg_genome_t* g) { return genome_load_impl(c, g, false); }
extern "C" int sqg_genome_load_device(sqg_ctx_t* c, const sqg_genome_t* g) { return genome_load_impl(c, g, true); }

// events of a read of `len` bases once the prefix is attached (src/gensig.c:242-245, src/genread.c:87-123)
static long long read_events(const sqg_ctx* c, long long len) {
    const bool rna = c->cfg.flags & SQG_RNA, prefix = c->cfg.flags & SQG_PREFIX;
    long long len0 = len;
    if (prefix) len0 += rna ? (kPolyA + (long long)strlen(kAdaptorRna)) : ((long long)strlen(kStallDna) + (long long)strlen(kAdaptorDna));
    const long long ne0 = len0 < c->k ? 5 : len0 - c->k + 1;
    const long long ne1 = (prefix && rna) ? (long long)strlen(kStallRna) - c->k + 1 : 0;
    return ne0 + ne1;
}

// a read generated elsewhere (range sharding): local worker w's scalar streams move past it -- one offset and one
// median_before draw (src/gensig.c:315-316), two time-stream draws per event (src/gensig.c:255)
static void skip_read(sqg_ctx* c, int w, long long n_events) {
    const sqg_profile_t& p = c->cfg.profile;
    if (!(c->cfg.flags & SQG_IDEAL)) {
        (void)host_nrng(p.offset_mean, p.offset_std, &c->off_x[(size_t)w]);
        (void)host_nrng(p.median_before_mean, p.median_before_std, &c->med_x[(size_t)w]);
    }
    if (c->use_dwell_stream)
        c->time_c[(size_t)w] = lcg_mul(c->time_c[(size_t)w], c->jump2((unsigned long long)n_events));
}

extern "C" int sqg_skip_reads(sqg_ctx_t* c, int32_t n, const int64_t* seq_len, const int32_t* worker) {
    if (!c || n < 0 || (n > 0 && (!seq_len || !worker))) return SQG_EINVAL;
    for (int i = 0; i < n; i++)
        if (worker[i] < c->wlo || worker[i] >= c->whi || seq_len[i] < 0) { c->err = "sqg_skip_reads: worker not owned by this context, or negative length"; return SQG_EINVAL; }
    for (int i = 0; i < n; i++) skip_read(c, worker[i] - c->wlo, read_events(c, seq_len[i]));
    return SQG_OK;
}

static int sample_impl(sqg_ctx_t* c, int32_t n, const int32_t* worker, int32_t lo, int32_t hi, sqg_batch_t** out, sqg_sample_t* info);

extern "C" int sqg_batch_sample(sqg_ctx_t* c, int32_t n, const int32_t* worker, sqg_batch_t** out, sqg_sample_t* info) {
    return sample_impl(c, n, worker, 0, n, out, info);
}

extern "C" int sqg_batch_sample_range(sqg_ctx_t* c, int32_t n, const int32_t* worker, int32_t lo, int32_t hi, sqg_batch_t** out, sqg_sample_t* info) {
    if (lo < 0 || hi < lo || hi > n) return SQG_EINVAL;
    return sample_impl(c, n, worker, lo, hi, out, info);
}

// gen_read for all n reads of the batch (the sampler streams are consumed read by read); reads [lo, hi) are staged, the
// workers' scalar streams skip over the others
static int sample_impl(sqg_ctx_t* c, int32_t n, const int32_t* worker, int32_t lo, int32_t hi, sqg_batch_t** out, sqg_sample_t* info) {
    if (!c || !out || n < 0) return SQG_EINVAL;
    if (!c->genome_loaded) { c->err = "sqg_genome_load has not been called"; return SQG_EINVAL; }
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    // worker chains in batch order (a worker's reads are sampled in index order, like its signal streams)
    std::vector<int> wk((size_t)n), count((size_t)c->nw, 0);
    for (int i = 0; i < n; i++) {
        const int w = worker ? worker[i] : sqg_worker_of(i, n, c->T);
        if (w < c->wlo || w >= c->whi) { c->err = "read assigned to a worker this context does not own"; return SQG_EINVAL; }
        wk[(size_t)i] = w - c->wlo; count[(size_t)wk[(size_t)i]]++;
    }
    std::vector<int> chain_of((size_t)c->nw, -1), chain_off(1, 0), chain_worker;
    for (int w = 0; w < c->nw; w++) if (count[(size_t)w]) { chain_of[(size_t)w] = (int)chain_off.size() - 1; chain_off.push_back(chain_off.back() + count[(size_t)w]); chain_worker.push_back(w); }
    const int n_chains = (int)chain_off.size() - 1;
    std::vector<int> fill(chain_off.begin(), chain_off.end() - 1), chain_reads((size_t)n);
    for (int i = 0; i < n; i++) chain_reads[(size_t)fill[(size_t)chain_of[(size_t)wk[(size_t)i]]]++] = i;

    // device scratch of the sampler: one grow-only allocation of the context, carved per call (every call ends with the
    // staging stream drained, so nothing of the previous call is still in use)
    int max_m = 0;
    for (int q = 0; q < n_chains; q++) max_m = std::max(max_m, chain_off[(size_t)q + 1] - chain_off[(size_t)q]);
    const bool concurrent = n > 0 && !(c->genome.flags & SQG_SAMPLE_FULL) && max_m >= 16 && !SQG_DEV_ENV("SQG_SAMPLER_SERIAL");
    std::vector<long long> att_off((size_t)n_chains + 1, 0);
    long long max_a = 0;
    if (concurrent)    // long chains: the attempts are evaluated concurrently, 25 % more than the acceptance rate seen so far asks for
        for (int q = 0; q < n_chains; q++) {
            const long long m = chain_off[(size_t)q + 1] - chain_off[(size_t)q];
            const long long a = (long long)std::ceil((double)m * c->samp_ratio * 1.25) + 64;
            att_off[(size_t)q + 1] = att_off[(size_t)q] + a; max_a = std::max(max_a, a);
        }
    const size_t na = (size_t)att_off.back();
    size_t top = 0;
    auto room = [&](size_t bytes) { const size_t o = top; top += (bytes + 255) & ~(size_t)255; return o; };
    // (the host-built inputs -- chain lists, attempt offsets -- lie back to back: one upload from the call's pinned buffer)
    const size_t o_rec = room(std::max<size_t>(1, (size_t)n) * sizeof(SampleRec)), o_co = room(chain_off.size() * sizeof(int)),
                 o_cr = room(std::max<size_t>(1, chain_reads.size()) * sizeof(int)), o_cw = room(std::max<size_t>(1, chain_worker.size()) * sizeof(int)),
                 o_ao = room((att_off.size() + (size_t)n_chains) * sizeof(long long)), o_in_end = top,
                 o_try = room(na * sizeof(SampleRec)), o_ok = room(na),
                 o_mc = room(std::max<size_t>(1, (size_t)n) * sizeof(int)), o_ms = room(std::max<size_t>(1, (size_t)n) * sizeof(uint32_t));
    if (top > c->samp_scratch_cap) {
        (void)hipStreamSynchronize(c->stage_stream);
        (void)hipFree(c->d_samp_scratch); c->d_samp_scratch = nullptr; c->samp_scratch_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_samp_scratch, top + top / 4));
        c->samp_scratch_cap = top + top / 4;
    }
    uint8_t* const sb = c->d_samp_scratch;
    SampleRec* d_rec = (SampleRec*)(sb + o_rec);
    int *d_co = (int*)(sb + o_co), *d_cr = (int*)(sb + o_cr), *d_cw = (int*)(sb + o_cw);
    SampleRec* d_try = (SampleRec*)(sb + o_try); unsigned char* d_ok = sb + o_ok; long long* d_ao = (long long*)(sb + o_ao);
    int* d_mcnt = (int*)(sb + o_mc); uint32_t* d_mstate = (uint32_t*)(sb + o_ms);
    // CpG methylation (methylate_dna, src/genread.c:207-241) is part of gen_read_dna only
    const bool do_meth = c->d_meth && n > 0 && !(c->genome.flags & (SQG_SAMPLE_RNA | SQG_SAMPLE_CDNA | SQG_SAMPLE_FULL));
    std::vector<SampleRec> rec((size_t)n);
    int rc = SQG_OK;
    // a failed call must leave the context where it was: the workers' streams -- sampler streams on the device, scalar streams
    // on the host -- are put back (the reference's sequence would otherwise silently stop matching)
    const std::vector<uint32_t> snap_time = c->time_c;
    const std::vector<long long> snap_off = c->off_x, snap_med = c->med_x;
    const long long snap_full = c->full_next;
    // (the sampler streams' snapshot and, below, the error word travel through pinned memory on the staging stream, in front of and behind
    // the kernels: the call waits for the device ONCE -- three blocking copies were 60 us of a 1000-read batch's 330)
    const size_t n_snap = (size_t)c->nw * 3, n_snap_m = c->d_meth_st ? (size_t)c->nw : 0;
    const size_t in_bytes = o_in_end - o_co;                      // (256-byte granules, as on the device)
    const size_t hw_head = ((n_snap + n_snap_m + 4) * sizeof(uint32_t) + 255) & ~(size_t)255;
    const size_t hw_rec = hw_head + in_bytes, hw_used = hw_rec + ((std::max<size_t>(1, (size_t)n) * sizeof(SampleRec) + 255) & ~(size_t)255);
    const size_t hw_all = (hw_used + (size_t)n_chains * sizeof(long long) + 255) / sizeof(uint32_t);
    if (c->h_samp_cap < hw_all) {
        (void)hipStreamSynchronize(c->stage_stream);
        if (c->h_samp) (void)hipHostFree(c->h_samp);
        c->h_samp = nullptr; c->h_samp_cap = 0;
        HIPCHK(c, hipHostMalloc(&c->h_samp, (hw_all + hw_all / 4) * sizeof(uint32_t), hipHostMallocDefault));
        c->h_samp_cap = hw_all + hw_all / 4;
    }
    uint8_t* const hb = reinterpret_cast<uint8_t*>(c->h_samp);
    uint32_t* const snap_samp = c->h_samp; uint32_t* const snap_meth = c->h_samp + n_snap; uint32_t* const h_err = c->h_samp + n_snap + n_snap_m;
    uint8_t* const h_in = hb + hw_head;
    SampleRec* const h_rec = reinterpret_cast<SampleRec*>(hb + hw_rec);
    long long* const h_used = reinterpret_cast<long long*>(hb + hw_used);
    if (hipMemcpyAsync(snap_samp, c->d_samp, n_snap * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stage_stream) != hipSuccess) { c->err = "sampler streams unreadable"; return SQG_EDEVICE; }
    if (n_snap_m && hipMemcpyAsync(snap_meth, c->d_meth_st, n_snap_m * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stage_stream) != hipSuccess) { c->err = "sampler streams unreadable"; return SQG_EDEVICE; }
    bool failed = true;                                           // cleared on the way out of a successful call
    auto cleanup = [&]() {
        if (!failed) return;
        c->time_c = snap_time; c->off_x = snap_off; c->med_x = snap_med; c->full_next = snap_full;
        (void)hipStreamSynchronize(c->stage_stream);              // (the snapshot has landed by now)
        (void)hipMemcpy(c->d_samp, snap_samp, n_snap * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (n_snap_m) (void)hipMemcpy(c->d_meth_st, snap_meth, n_snap_m * sizeof(uint32_t), hipMemcpyHostToDevice);
    };
#define CHKS(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { c->err = std::string(#call) + ": " + hipGetErrorString(e_); cleanup(); return e_ == hipErrorOutOfMemory ? SQG_ENOMEM : SQG_EDEVICE; } } while (0)
    if (n > 0) {
        memcpy(h_in + (o_co - o_co), chain_off.data(), chain_off.size() * sizeof(int));
        memcpy(h_in + (o_cr - o_co), chain_reads.data(), chain_reads.size() * sizeof(int));
        memcpy(h_in + (o_cw - o_co), chain_worker.data(), chain_worker.size() * sizeof(int));
        memcpy(h_in + (o_ao - o_co), att_off.data(), att_off.size() * sizeof(long long));
        CHKS(hipMemcpyAsync(sb + o_co, h_in, in_bytes, hipMemcpyHostToDevice, c->stage_stream));
        bool have_used = false;
        if (c->genome.flags & SQG_SAMPLE_FULL) {
            // --full-contigs (src/sim.c:543-549): the reads are the contigs themselves, in order, as loaded ('N' stays 'N')
            if (c->full_next + n > c->genome.n_contigs) { c->err = "--full-contigs: more reads asked for than there are contigs left"; cleanup(); return SQG_EINVAL; }
            for (int i = 0; i < n; i++) {
                const long long q = c->full_next + i;
                SampleRec& r = rec[(size_t)i];
                r.src = c->h_contig_off[(size_t)q]; r.ref_idx = (int)q; r.ref_pos = 0;
                r.rlen = (int)(c->h_contig_off[(size_t)q + 1] - c->h_contig_off[(size_t)q]);
                r.strand = '+'; r.n_N = 0; r.ref_len = r.rlen;
            }
            c->full_next += n;
            CHKS(hipMemcpyAsync(d_rec, rec.data(), rec.size() * sizeof(SampleRec), hipMemcpyHostToDevice, c->stage_stream));
        } else if (concurrent) {
            long long* d_used = d_ao + att_off.size();
            hipLaunchKernelGGL(k_sample_try, dim3((unsigned)((max_a + 3) / 4), (unsigned)n_chains), dim3(256), 0, c->stage_stream, c->genome, c->d_samp,
                               d_cw, d_ao, d_try, d_ok);
            hipLaunchKernelGGL(k_sample_pick, dim3((unsigned)n_chains), dim3(256), 0, c->stage_stream, c->genome, c->d_samp, d_co, d_cr, d_cw,
                               d_ao, d_try, d_ok, d_rec, d_used, c->d_err);
            CHKS(hipGetLastError());
            have_used = true;
            CHKS(hipMemcpyAsync(h_used, d_used, (size_t)n_chains * sizeof(long long), hipMemcpyDeviceToHost, c->stage_stream));
        } else {
            hipLaunchKernelGGL(k_sample, dim3((unsigned)n_chains), dim3(64), 0, c->stage_stream, c->genome, c->d_samp, d_co, d_cr, d_cw,
                               n_chains, d_rec, c->d_err);
            CHKS(hipGetLastError());
        }
        if (do_meth) {                                         // every read of the batch, staged here or not: the stream is the worker's
            hipLaunchKernelGGL(k_meth_count, dim3((unsigned)n), dim3(256), 0, c->stage_stream, c->genome, d_rec, n, d_mcnt);
            hipLaunchKernelGGL(k_meth_scan, dim3((unsigned)n_chains), dim3(256), 0, c->stage_stream, d_co, d_cr, d_cw, d_mcnt, c->d_meth_st, d_mstate);
            CHKS(hipGetLastError());
        }
        CHKS(hipMemcpyAsync(h_rec, d_rec, rec.size() * sizeof(SampleRec), hipMemcpyDeviceToHost, c->stage_stream));
        CHKS(hipMemcpyAsync(h_err, c->d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stage_stream));
        CHKS(hipStreamSynchronize(c->stage_stream));
        memcpy(rec.data(), h_rec, rec.size() * sizeof(SampleRec));
        if (have_used) {
            double r = 1.0;
            for (int q = 0; q < n_chains; q++) {
                const int m = chain_off[(size_t)q + 1] - chain_off[(size_t)q];
                if (m >= 16) r = std::max(r, (double)h_used[q] / (double)m);
            }
            c->samp_ratio = r;
        }
        const unsigned int e = *h_err;
        if (e & 16u) { CHKS(hipMemset(c->d_err, 0, sizeof e)); c->err = "read sampler: no acceptable read after 100000 attempts"; cleanup(); return SQG_EINVAL; }
    }
#undef CHKS
    // lengths are known now: stage as sqg_batch_stage would, the base buffer being filled on the device
    const int m = hi - lo;                                       // reads staged here
    std::vector<int64_t> seq_off((size_t)m + 1, 0);
    for (int i = 0; i < m; i++) seq_off[(size_t)i + 1] = seq_off[(size_t)i] + rec[(size_t)(lo + i)].rlen;
    std::vector<int32_t> wk_glob;                                // global worker ids of the whole batch (the partition depends on n)
    if (m != n) {
        wk_glob.resize((size_t)n);
        for (int i = 0; i < n; i++) wk_glob[(size_t)i] = wk[(size_t)i] + c->wlo;
        for (int i = 0; i < lo; i++) skip_read(c, wk[(size_t)i], read_events(c, rec[(size_t)i].rlen));
    }
    rc = stage_common(c, m, nullptr, seq_off.data(), m != n ? wk_glob.data() + lo : worker, d_rec + lo, out, do_meth ? d_mstate + lo : nullptr);
    if (rc == SQG_OK && m != n)
        for (int i = hi; i < n; i++) skip_read(c, wk[(size_t)i], read_events(c, rec[(size_t)i].rlen));
    failed = rc != SQG_OK;
    cleanup();
    if (rc) return rc;
    sqg_batch* b = *out;
    const bool rna = c->cfg.flags & SQG_RNA, prefix = c->cfg.flags & SQG_PREFIX;
    const long long read_at = (prefix && !rna) ? (long long)(strlen(kStallDna) + strlen(kAdaptorDna)) : 0;
    rec.erase(rec.begin(), rec.begin() + lo); rec.resize((size_t)m);
    n = m;
    b->s_ref_idx.resize((size_t)n); b->s_ref_len.resize((size_t)n); b->s_ref_pos.resize((size_t)n); b->s_rlen.resize((size_t)n);
    b->s_strand.resize((size_t)n + 1); b->s_seq_off.assign(seq_off.begin(), seq_off.end()); b->s_read_at.resize((size_t)n);
    b->s_src.resize((size_t)n);
    for (int i = 0; i < n; i++) b->s_src[(size_t)i] = rec[(size_t)i].src;
    for (int i = 0; i < n; i++) {
        const SampleRec& q = rec[(size_t)i];
        b->s_ref_idx[(size_t)i] = q.ref_idx; b->s_ref_len[(size_t)i] = q.ref_len; b->s_ref_pos[(size_t)i] = q.ref_pos;
        b->s_rlen[(size_t)i] = q.rlen; b->s_strand[(size_t)i] = (char)q.strand; b->s_read_at[(size_t)i] = read_at;
    }
    if (info) {
        info->ref_idx = b->s_ref_idx.data(); info->ref_len = b->s_ref_len.data(); info->ref_pos = b->s_ref_pos.data();
        info->rlen = b->s_rlen.data(); info->strand = b->s_strand.data(); info->seq_off = (const int64_t*)b->s_seq_off.data();
    }
    return SQG_OK;
}

extern "C" int sqg_fetch_reads(sqg_ctx_t* c, sqg_batch_t* b, char* dst) {
    if (!c || !b || !dst || b->s_seq_off.empty()) return SQG_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (b->ev_staged) HIPCHK(c, hipEventSynchronize(b->ev_staged));     // the base buffer is filled on the staging stream
    std::vector<uint8_t> all((size_t)b->n_bases_total + 1);
    if (b->n_bases_total) HIPCHK(c, hipMemcpy(all.data(), b->d_bases, (size_t)b->n_bases_total, hipMemcpyDeviceToHost));
    const bool rna = c->cfg.flags & SQG_RNA, prefix = c->cfg.flags & SQG_PREFIX;
    const long long extra = prefix ? (rna ? (kPolyA + (long long)strlen(kAdaptorRna)) : ((long long)strlen(kStallDna) + (long long)strlen(kAdaptorDna))) : 0;
    for (int i = 0; i < b->n; i++) {
        const long long len = b->s_rlen[(size_t)i];
        if (len + extra >= c->k)
            memcpy(dst + b->s_seq_off[(size_t)i], all.data() + b->h_base_off[(size_t)i] + b->s_read_at[(size_t)i], (size_t)len);
        else if (len > 0)     // shorter than a k-mer (--full-contigs only): the base buffer holds the stand-in sequence of src/gensig.c:242-245
            HIPCHK(c, hipMemcpy(dst + b->s_seq_off[(size_t)i], c->d_genome + b->s_src[(size_t)i], (size_t)len, hipMemcpyDeviceToHost));
    }
    return SQG_OK;
}
