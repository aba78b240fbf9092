// h_results.h -- waiting for a batch, fetching results, svb-zd compression, timing, pinned host memory, the store probe
// Host side of include/sqg.h; included by sqg_hip.hip (one translation unit with the kernels), in the order listed there.
#pragma once

// the slot's buffers still hold this batch's results (two batches later they do not: range mode bumps the generation as soon
// as sqg_batch_run_begin of a later batch starts writing them)
static bool slot_is_mine(const sqg_ctx* c, const sqg_batch* b) { return c->slot[b->slot].gen == b->slot_gen; }
// ... and the set of first-pass outputs (dwells) this batch's (three batches later it does not: the first pass of batch i+3 may run
// inside sqg_batch_run of batch i+2)
static bool cset_is_mine(const sqg_ctx* c, const sqg_batch* b) { return c->cset[b->cset].gen == b->cset_gen; }

extern "C" int sqg_batch_wait(sqg_ctx_t* c, sqg_batch_t* b, sqg_result_t* res) {
    if (!c || !b || !b->ran) return SQG_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipEventSynchronize(b->ev[4]));              // this batch only: later batches keep running
    sqg_ctx::Slot& S = c->slot[b->slot];
    if (!b->waited) {
        b->n_samples = b->h_sigoff[b->n];
        for (int i = 0; i <= b->n; i++) b->sig_off[(size_t)i] = b->h_sigoff[i];
        unsigned int e = 0;
        const unsigned int* const hres = reinterpret_cast<const unsigned int*>(b->h_sigoff + (b->h_n - SQG_HRES_LL));   // k_fixup's report (mapped host memory)
        if (b->fixup_launched) e = hres[0];
        else HIPCHK(c, hipMemcpy(&e, b->d_err, sizeof e, hipMemcpyDeviceToHost));   // the batch's own word: never cleared, never shared
#if defined(SQG_ABL_EV_NOSTORE) || defined(SQG_ABL_NOSTORE)       /* timing-only ablation builds: results are garbage by design */
        e = 0;
#endif
        b->waited = true;
        if (e) {
            c->err = "device reported: " + std::string((e & 1) ? "dwell>65535 " : "") + ((e & 2) ? "read>=UINT32_MAX samples " : "") + ((e & 4) ? "internal length mismatch " : "") + ((e & 8) ? "FP64 fix-up list overflow " : "") + ((e & 32) ? "one k-mer stream asked for >= 2^32 samples by one batch " : "") +
                     ((e & 64) ? "LDS atomics were not served in lane order (the batch's sample of the stream hand-out; create the context with SQG_ORDER_FREE) " : "") +
                     ((e & 0x80000000u) ? "the batch's last kernel left no report" : "");
            b->wait_rc = (e & (12 | 64 | 0x80000000u)) ? SQG_EDEVICE : SQG_EOVERFLOW;
            return b->wait_rc;
        }
        float d = 0, s = 0, t = 0, ee = 0;
        if (!b->untimed) {                                  // (sqg_set_phase_timing: a batch without the phase events reports 0 ms)
            if (b->dwell_timed) HIPCHK(c, hipEventElapsedTime(&d, b->ev[0], b->ev[1]));
            HIPCHK(c, hipEventElapsedTime(&ee, b->ev[b->dwell_timed ? 2 : 0], b->ev[3]));
            HIPCHK(c, hipEventElapsedTime(&s, b->ev[3], b->ev[4]));
            HIPCHK(c, hipEventElapsedTime(&t, b->ev[0], b->ev[4]));
        }
        c->timing.events_ms = ee;
        c->timing.lean_ms = 0.f;
        if (b->lean_timed) HIPCHK(c, hipEventElapsedTime(&c->timing.lean_ms, b->ev[5], b->ev[6]));
        long long nfix = 0;
        if (c->cfg.mode == SQG_MODE_CERTIFIED && b->fixup_launched) {
            nfix = hres[1];                                 // the global list + the lean kernel's lists, as k_fixup reported them
            for (int i = 0; i < FIX_SHARDS; i++) nfix += hres[4 + i];
        }
        else if (c->cfg.mode == SQG_MODE_CERTIFIED && !slot_is_mine(c, b)) nfix = -1;   // the slot's counters belong to a later batch by now: not known
        else if (c->cfg.mode == SQG_MODE_CERTIFIED) {
            unsigned int cnt[4 + FIX_SHARDS];                // the counters and the lists' statistics in one read-back
            const bool lists = S.d_fix_sh_count != nullptr && b->fixup_launched;   // (an empty batch launches no k_fixup: the words are an earlier batch's)
            HIPCHK(c, hipMemcpy(cnt, S.d_fix_count, (lists ? 4 + FIX_SHARDS : 4) * sizeof(unsigned int), hipMemcpyDeviceToHost));
            nfix = cnt[0];                                  // the global list ...
            if (lists) for (int i = 0; i < FIX_SHARDS; i++) nfix += cnt[4 + i];   // ... + the lean kernel's lists (a word per list, written by k_fixup)
        }
        c->timing.dwell_ms = d; c->timing.samples_ms = s; c->timing.total_ms = t; c->timing.fallback_samples = nfix;
        c->timing.carried_first_pass = b->carried_precount ? 1 : 0; c->timing.first_pass_ran_ahead = b->precounted ? 1 : 0;
    } else if (b->wait_rc) {
        c->err = "this batch failed on the device (see the first sqg_batch_wait)";
        return b->wait_rc;
    }
    if (res) {
        const bool mine = slot_is_mine(c, b);               // else: the slabs have been handed to a later batch
        res->n_reads = b->n; res->n_events = b->n_events; res->n_samples = b->n_samples; res->n_bases = b->n_bases;
        res->sig_off = (const int64_t*)b->sig_off.data(); res->ev_off = (const int64_t*)b->ev_off.data();
        res->offset = b->offset.data(); res->median_before = b->median.data();
        res->d_signal = mine ? S.d_sig : nullptr; res->d_dwell = (mine && c->use_dwell_stream && cset_is_mine(c, b)) ? c->cset[b->cset].d_dwell : nullptr;
    }
    return SQG_OK;
}

extern "C" int sqg_fetch_signal(sqg_ctx_t* c, sqg_batch_t* b, int16_t* dst) {
    if (!c || !b || !b->ran || !dst) return SQG_EINVAL;
    if (b->run_idx + 2 < c->runs || !slot_is_mine(c, b)) return SQG_ESEQUENCE;       // slab already reused (two batches later)
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipEventSynchronize(b->ev[4]));
    if (!b->waited) b->n_samples = b->h_sigoff[b->n];
    if (b->n_samples) HIPCHK(c, hipMemcpy(dst, c->slot[b->slot].d_sig, (size_t)b->n_samples * sizeof(int16_t), hipMemcpyDeviceToHost));
    return SQG_OK;
}

extern "C" int sqg_fetch_dwell(sqg_ctx_t* c, sqg_batch_t* b, int32_t* dst) {
    if (!c || !b || !b->ran || !dst) return SQG_EINVAL;
    if (b->run_idx + 2 < c->runs || !slot_is_mine(c, b) || !cset_is_mine(c, b)) return SQG_ESEQUENCE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipEventSynchronize(b->ev[4]));
    if (!c->use_dwell_stream) {
        for (long long i = 0; i < b->n_events; i++) dst[i] = (int)c->cfg.profile.dwell_mean;
        return SQG_OK;
    }
    std::vector<uint16_t> tmp((size_t)b->n_events);
    if (b->n_events) HIPCHK(c, hipMemcpy(tmp.data(), c->cset[b->cset].d_dwell, tmp.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); i++) dst[i] = tmp[i];
    return SQG_OK;
}

extern "C" int sqg_get_timing(sqg_ctx_t* c, sqg_timing_t* t) {
    if (!c || !t) return SQG_EINVAL;
    *t = c->timing;
    return SQG_OK;
}

extern "C" int sqg_set_phase_timing(sqg_ctx_t* c, int every) {
    if (!c || every < 0) return SQG_EINVAL;
    c->phase_timing_every = every;
    return SQG_OK;
}

extern "C" int sqg_set_stage_threads(sqg_ctx_t* c, int n) {
    if (!c || n < 0 || n > 64) return SQG_EINVAL;
    c->stage_threads = n;
    return c->stage_threads_last;
}

extern "C" int sqg_submit(sqg_ctx_t* c, int32_t n, const char* seqs, const int64_t* seq_off,
                          const int32_t* worker, sqg_batch_t** out, sqg_result_t* res) {
    if (!out) return SQG_EINVAL;
    int rc = sqg_batch_stage(c, n, seqs, seq_off, worker, out);
    if (rc) return rc;
    if ((rc = sqg_batch_run(c, *out)) || (rc = sqg_batch_wait(c, *out, res))) { sqg_batch_free(c, *out); *out = nullptr; }
    return rc;
}

extern "C" int sqg_batch_compress(sqg_ctx_t* c, sqg_batch_t* b, sqg_svb_t* out) {
    if (!c || !b || !b->ran || !out) return SQG_EINVAL;
    if (b->run_idx + 2 < c->runs || !slot_is_mine(c, b)) return SQG_ESEQUENCE;       // the signals of an older batch are gone
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipEventSynchronize(b->ev[4]));
    sqg_ctx::Slot& S = c->slot[b->slot];
    int rc;
    const int n = b->n;
    if (!b->h_svboff) {
        HIPCHK(c, hipHostMalloc(&b->h_svboff, ((size_t)n + 1) * sizeof(long long), hipHostMallocMapped));
    }
    long long* h_dev = nullptr;
    HIPCHK(c, hipHostGetDevicePointer((void**)&h_dev, b->h_svboff, 0));
    b->h_svboff[0] = 0;
    if (n > 0) {
        if ((rc = ensure(c, (void**)&c->d_svb_size, &c->svb_size_cap, (size_t)n + 64, sizeof(long long)))) return rc;
        if ((rc = ensure(c, (void**)&c->d_svb_off, &c->svb_off_cap, (size_t)n + 64, sizeof(long long)))) return rc;
        hipLaunchKernelGGL(k_svb_size, dim3((unsigned)n), dim3(256), 0, c->stream2, S.d_sig, S.d_sigoff, n, c->d_svb_size);
        hipLaunchKernelGGL(k_svb_scan, dim3(1), dim3(1024), 0, c->stream2, c->d_svb_size, n, c->d_svb_off, h_dev);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream2));           // the total sizes the output buffer
        const long long total = b->h_svboff[n];
        if ((rc = ensure(c, (void**)&c->d_svb, &c->svb_cap, (size_t)total + 64, 1))) return rc;
        hipLaunchKernelGGL(k_svb_encode, dim3((unsigned)n), dim3(256), 0, c->stream2, S.d_sig, S.d_sigoff, n, c->d_svb_off, c->d_svb);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream2));
    }
    b->n_svb = b->h_svboff[n];
    b->compress_seq = ++c->compress_seq;
    out->n_bytes = b->n_svb;
    out->svb_off = (const int64_t*)b->h_svboff;
    out->d_svb = c->d_svb;
    return SQG_OK;
}

extern "C" int sqg_fetch_svb(sqg_ctx_t* c, sqg_batch_t* b, uint8_t* dst) {
    if (!c || !b || !dst || b->n_svb < 0) return SQG_EINVAL;
    if (b->compress_seq != c->compress_seq) return SQG_ESEQUENCE;      // a later sqg_batch_compress reused the buffer
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (b->n_svb > 0) HIPCHK(c, hipMemcpy(dst, c->d_svb, (size_t)b->n_svb, hipMemcpyDeviceToHost));
    return SQG_OK;
}

// The batch's BLOW5 records -- slow5_rec_to_mem's layout (slow5lib/src/slow5.c:3928-4072) around the device's svb-zd bytes, each in a zlib
// stream of stored blocks (k_blow5.h) -- framed on the device and copied to pinned host memory of the context: what a writer appends to
// the file as it is.  Compresses the batch first unless its encodings are still the context's.
extern "C" int sqg_batch_blow5_records(sqg_ctx_t* c, sqg_batch_t* b, const sqg_profile_t* profile, uint32_t flags, const char* read_ids,
                                       const int64_t* id_off, int64_t read_number0, uint64_t start_time0,
                                       const uint8_t** records, int64_t* n_bytes, const int64_t** rec_off) {
    if (!c || !b || !b->ran || !profile || !records || !n_bytes || (b->n > 0 && (!read_ids || !id_off))) return SQG_EINVAL;
    if (b->run_idx + 2 < c->runs || !slot_is_mine(c, b)) return SQG_ESEQUENCE;
    int rc;
    if (b->n_svb < 0 || b->compress_seq != c->compress_seq) { sqg_svb_t sv; if ((rc = sqg_batch_compress(c, b, &sv))) return rc; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const int n = b->n;
    const bool ont = (flags & SQG_ONT) != 0;
    std::vector<int64_t>& ro = c->b5_rec_off;
    ro.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) {
        const int64_t idl = id_off[i + 1] - id_off[i];
        if (idl < 0 || idl > B5_ID_MAX) { c->err = "sqg_batch_blow5_records: read id longer than 4096 bytes (use the host-zlib writer)"; return SQG_EINVAL; }
        const unsigned long long R = (unsigned long long)(2 + idl + 4 + 32 + 8) + (unsigned long long)(b->h_svboff[i + 1] - b->h_svboff[i]) + (unsigned long long)(30 + (ont ? 1 : 0));
        ro[(size_t)i + 1] = ro[(size_t)i] + (int64_t)b5_stored_size(R);
    }
    const size_t total = (size_t)ro[(size_t)n];
    *records = nullptr; *n_bytes = (int64_t)total;
    if (rec_off) *rec_off = ro.data();
    if (n == 0) return SQG_OK;
    // one upload: id offsets | record offsets | offsets | medians | id bytes
    const size_t id_bytes = (size_t)(id_off[n] - id_off[0]);
    const size_t o_io = 0, o_ro = o_io + ((size_t)n + 1) * 8, o_of = o_ro + ((size_t)n + 1) * 8, o_md = o_of + (size_t)n * 8, o_id = o_md + (size_t)n * 8, meta = o_id + id_bytes + 16;
    if ((rc = ensure(c, (void**)&c->d_b5meta, &c->b5meta_cap, meta, 1))) return rc;
    if ((rc = ensure(c, (void**)&c->d_b5out, &c->b5out_cap, total + 64, 1))) return rc;
    if (c->h_b5meta_cap < meta) {
        if (c->h_b5meta) (void)hipHostFree(c->h_b5meta);
        c->h_b5meta = nullptr; c->h_b5meta_cap = 0;
        HIPCHK(c, hipHostMalloc(&c->h_b5meta, meta + meta / 2, hipHostMallocDefault));
        c->h_b5meta_cap = meta + meta / 2;
    }
    const int fl = c->b5_flip ^= 1;
    if (c->b5_reader_drain && c->b5_reader_buf == fl) c->b5_reader_drain(c->b5_reader, false);   // (a writer's background write still reads this buffer: only when somebody else called in between)
    if (!c->b5_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->b5_stream, hipStreamNonBlocking));
    // the encodings are complete (sqg_batch_compress synchronises) and so is the slot's sig_off (the batch was waited for): the framing needs nothing of
    // the main stream, where the NEXT batch's kernels are queued by now -- a stream of its own, and only that one is waited for
    const hipStream_t bs = c->b5_stream;
    if (c->h_b5out_cap[fl] < total) {
        if (c->h_b5out[fl]) (void)hipHostFree(c->h_b5out[fl]);
        c->h_b5out[fl] = nullptr; c->h_b5out_cap[fl] = 0;
        HIPCHK(c, hipHostMalloc(&c->h_b5out[fl], total + total / 4, hipHostMallocDefault));
        c->h_b5out_cap[fl] = total + total / 4;
    }
    {
        long long* io = reinterpret_cast<long long*>(c->h_b5meta + o_io);
        for (int i = 0; i <= n; i++) io[i] = (long long)(id_off[i] - id_off[0]);
        memcpy(c->h_b5meta + o_ro, ro.data(), ((size_t)n + 1) * 8);
        memcpy(c->h_b5meta + o_of, b->offset.data(), (size_t)n * 8);
        memcpy(c->h_b5meta + o_md, b->median.data(), (size_t)n * 8);
        memcpy(c->h_b5meta + o_id, read_ids + id_off[0], id_bytes);
    }
    HIPCHK(c, hipMemcpyAsync(c->d_b5meta, c->h_b5meta, meta, hipMemcpyHostToDevice, bs));
    Blow5Params P;
    P.svb = c->d_svb; P.svb_off = c->d_svb_off; P.sig_off = c->slot[b->slot].d_sigoff;
    P.id_off = reinterpret_cast<const long long*>(c->d_b5meta + o_io); P.rec_off = reinterpret_cast<const long long*>(c->d_b5meta + o_ro);
    P.offset = reinterpret_cast<const double*>(c->d_b5meta + o_of); P.median = reinterpret_cast<const double*>(c->d_b5meta + o_md);
    P.ids = c->d_b5meta + o_id; P.out = c->d_b5out;
    P.digitisation = profile->digitisation; P.range = profile->range; P.sample_rate = profile->sample_rate;
    P.read_number0 = read_number0; P.start_time0 = start_time0; P.ont = ont ? 1 : 0; P.n = n;
    hipLaunchKernelGGL(k_blow5_frame, dim3((unsigned)n), dim3(256), 0, bs, P);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_b5out[fl], c->d_b5out, total, hipMemcpyDeviceToHost, bs));
    HIPCHK(c, hipStreamSynchronize(bs));
    *records = c->h_b5out[fl];
    return SQG_OK;
}

extern "C" void* sqg_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

extern "C" void sqg_host_free(void* p) { if (p) (void)hipHostFree(p); }

extern "C" int sqg_probe_store_bandwidth(sqg_ctx_t* c, size_t bytes, int iters, float* ms_per_pass) {
    if (!c || !ms_per_pass || iters < 1 || bytes < 4096) return SQG_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    void* buf = nullptr;
    HIPCHK(c, hipMalloc(&buf, bytes));
    const size_t n16 = bytes / 16;
    hipEvent_t a, z;
    HIPCHK(c, hipEventCreate(&a)); HIPCHK(c, hipEventCreate(&z));
    hipLaunchKernelGGL(k_store_probe, dim3(256 * 8), dim3(256), 0, c->stream, (uint4*)buf, n16, 1u);   // warm-up
    HIPCHK(c, hipEventRecord(a, c->stream));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_store_probe, dim3(256 * 8), dim3(256), 0, c->stream, (uint4*)buf, n16, (uint32_t)i);
    HIPCHK(c, hipEventRecord(z, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, a, z));
    *ms_per_pass = ms / iters;
    (void)hipEventDestroy(a); (void)hipEventDestroy(z); (void)hipFree(buf);
    return SQG_OK;
}


extern "C" int sqg_probe_lds_order(sqg_ctx_t* c, int workgroups, int rounds, unsigned int* mismatches, int* in_use) {
    if (!c || !mismatches || workgroups < 1 || rounds < 1) return SQG_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    unsigned int* d_bad = nullptr;
    HIPCHK(c, hipMalloc(&d_bad, sizeof(unsigned int)));
    HIPCHK(c, hipMemsetAsync(d_bad, 0, sizeof(unsigned int), c->stream));
    hipLaunchKernelGGL(k_lds_order_check, dim3((unsigned)workgroups), dim3(64), 0, c->stream, rounds, d_bad);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(mismatches, d_bad, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d_bad);
    if (in_use) *in_use = c->lds_ordered ? 1 : 0;
    return SQG_OK;
}
