// h_context.h -- sqg_create / sqg_destroy, error strings, the reference's static read -> worker partition
// Host side of include/sqg.h; included by sqg_hip.hip (one translation unit with the kernels), in the order listed there.
#pragma once

extern "C" const char* sqg_strerror(int code) {
    switch (code) {
    case SQG_OK: return "ok";
    case SQG_EINVAL: return "invalid argument or unsupported configuration";
    case SQG_ENOMEM: return "out of memory";
    case SQG_EDEVICE: return "HIP runtime error";
    case SQG_ESEQUENCE: return "batches must be run in staging order";
    case SQG_ENODEVICE: return "no usable HIP device";
    case SQG_EOVERFLOW: return "read too long (>= UINT32_MAX samples) or dwell > 65535";
    case SQG_EIO: return "file I/O error";
    default: return "unknown error";
    }
}

extern "C" const char* sqg_last_error(const sqg_ctx_t* ctx) { return ctx ? ctx->err.c_str() : ""; }

extern "C" int sqg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return SQG_ENODEVICE;
    return n;
}

extern "C" int32_t sqg_worker_of(int32_t i, int32_t n_rec, int32_t T) {
    if (T <= 1 || n_rec <= 0) return 0;                    // src/thread.c:122-125
    const int32_t step = (n_rec + T - 1) / T;              // src/thread.c:80
    return i / step;
}

extern "C" void sqg_destroy(sqg_ctx_t* ctx) {
    if (!ctx) return;
    if (ctx->b5_reader_drain) ctx->b5_reader_drain(ctx->b5_reader, true);       // a writer's background write still reads this context's pinned records: let it finish, unbind
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->b5_stream) { (void)hipStreamSynchronize(ctx->b5_stream); (void)hipStreamDestroy(ctx->b5_stream); ctx->b5_stream = nullptr; }
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->fix_stream) (void)hipStreamSynchronize(ctx->fix_stream);
#if defined(SQG_LEAN_TRACE)
    {   // (timing-only build) where an item of k_samples_lean spends its time: k_samples.h
        std::vector<unsigned long long> all((size_t)LEAN_TRACE_SHARDS * 16, 0ull);
        unsigned long long t[16] = {0};
        if (hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(g_lean_trace), all.size() * sizeof(unsigned long long)) == hipSuccess)
            for (size_t i = 0; i < all.size(); i++) t[i & 15] += all[i];
        if (t[0]) {
            static const char* nm[] = {"drain of the previous item's stores", "first-level loads", "second-level gathers", "tables", "sample loop", "item end", "between items", "workgroup prologue"};
            fprintf(stderr, "[sqg] lean trace (build %d): %llu items, %.1f steps per item; shader-clock ticks per item:", (int)SQG_LEAN_TRACE, t[0], (double)t[1] / (double)t[0]);
            double tot = 0; for (int i = 2; i <= 9; i++) tot += (double)t[i] / (double)t[0];
            for (int i = 2; i <= 9; i++) fprintf(stderr, " %s %.0f (%.1f %%);", nm[i - 2], (double)t[i] / (double)t[0], 100.0 * (double)t[i] / (double)t[0] / tot);
            fprintf(stderr, " total %.0f; loop ticks per step %.1f\n", tot, (double)t[6] / (double)t[1]);
        }
    }
#endif
    delete ctx->draw_ahead; ctx->draw_ahead = nullptr;
    (void)hipFree(ctx->d_pcnt[0]); (void)hipFree(ctx->d_pcnt[1]); (void)hipFree(ctx->d_slice); (void)hipFree(ctx->d_phist);
    (void)hipFree(ctx->d_rows); (void)hipFree(ctx->d_link_rows); (void)hipFree(ctx->d_scan_part); (void)hipFree(ctx->d_xcounts); (void)hipFree(ctx->d_model); (void)hipFree(ctx->d_samp_scratch); (void)hipFree(ctx->d_pow); (void)hipFree(ctx->d_err); (void)hipFree(ctx->d_mid_done); (void)hipFree(ctx->d_zero); if (ctx->h_samp) (void)hipHostFree(ctx->h_samp);
    for (auto& Q : ctx->cset) { (void)hipFree(Q.d_dwell); (void)hipFree(Q.d_tile_so); (void)hipFree(Q.d_seglen); }
    for (auto& S : ctx->slot) {
        (void)hipFree(S.d_sig); (void)hipFree(S.d_sigoff);
        (void)hipFree(S.d_fix); (void)hipFree(S.d_fix_count); (void)hipFree(S.d_fix_sh); (void)hipFree(S.d_fix_sh_count);
        (void)hipFree(S.d_evrec); (void)hipFree(S.d_slow);
        (void)hipFree(S.d_items); (void)hipFree(S.d_part_state); (void)hipFree(S.d_part); (void)hipFree(S.d_lbase); (void)hipFree(S.d_tile_link);
        (void)hipFree(S.cal_prev);
        if (S.cal_a) (void)hipEventDestroy(S.cal_a);
        if (S.cal_b) (void)hipEventDestroy(S.cal_b);
        if (S.done) (void)hipEventDestroy(S.done);
        if (S.sampled) (void)hipEventDestroy(S.sampled);
    }
    (void)hipFree(ctx->d_svb); (void)hipFree(ctx->d_svb_size); (void)hipFree(ctx->d_svb_off);
    (void)hipFree(ctx->d_b5meta); (void)hipFree(ctx->d_b5out); for (auto* q : ctx->h_b5out) if (q) (void)hipHostFree(q); if (ctx->h_b5meta) (void)hipHostFree(ctx->h_b5meta);
    (void)hipFree(ctx->d_genome); (void)hipFree(ctx->d_contig_off); (void)hipFree(ctx->d_cum); (void)hipFree(ctx->d_nprefix);
    (void)hipFree(ctx->d_trans_csum); (void)hipFree(ctx->d_trans_idx); (void)hipFree(ctx->d_samp);
    (void)hipFree(ctx->d_meth); (void)hipFree(ctx->d_meth_has); (void)hipFree(ctx->d_meth_st);
    if (ctx->stage_stream) { (void)hipStreamSynchronize(ctx->stage_stream); (void)hipStreamDestroy(ctx->stage_stream); }
    for (auto& r : ctx->pool) { (void)hipFree(r.d_block); (void)hipHostFree(r.h_sigoff); (void)hipHostFree(r.h_meta); for (auto& e : r.ev) if (e) (void)hipEventDestroy(e); if (r.ev_staged) (void)hipEventDestroy(r.ev_staged); }
    ctx->pool.clear();
    if (ctx->stream2 && ctx->stream2 != ctx->stream) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->fix_stream) (void)hipStreamDestroy(ctx->fix_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ---- (development build) CU-masked streams -------------------------------------------------------------------------------
// What a bit of a stream's CU mask stands for (measured: tools/cumask_probe.hip, profiles/r05_cumask.md): bit i is CU slot i / 8 of XCC
// i % 8 (MI355X, 8 XCCs of 32 CUs); an XCC whose bits are ALL clear runs on all of its CUs, so every XCC gets the same slots.
static int cu_split_streams(sqg_ctx* c, int n_ev, const bool same) {
    const int ncu = c->num_cu, words = (ncu + 31) / 32, nx = 8, per = ncu / nx;
    if (ncu % nx || n_ev < (same ? 0 : 1) || n_ev >= per) return SQG_EINVAL;
    std::vector<uint32_t> m_ev((size_t)words, 0u), m_smp((size_t)words, 0u);
    for (int i = 0; i < ncu; i++) ((i / nx) < n_ev ? m_ev : m_smp)[(size_t)i >> 5] |= 1u << (i & 31);
    const int n_e = n_ev * nx, n_s = ncu - n_e;
    fprintf(stderr, "[sqg] CU split: %d CUs for the event side, %d for the sample kernels%s\n", n_e, n_s, same ? " (one stream on the latter)" : "");
    hipStream_t old = c->stream;
    hipStream_t s_ev = nullptr, s_smp = nullptr;
    if (hipExtStreamCreateWithCUMask(&s_smp, (uint32_t)words, m_smp.data()) != hipSuccess) return SQG_EDEVICE;
    if (same) { c->stream = s_smp; c->stream2 = s_smp; c->num_cu = n_s; }
    else {
        if (hipExtStreamCreateWithCUMask(&s_ev, (uint32_t)words, m_ev.data()) != hipSuccess) { (void)hipStreamDestroy(s_smp); return SQG_EDEVICE; }
        c->stream = s_ev; c->stream2 = s_smp; c->num_cu = n_e;       // (num_cu: what the persistent event-side grids are sized by)
    }
    if (old) { (void)hipStreamSynchronize(old); (void)hipStreamDestroy(old); }
    return SQG_OK;
}

extern "C" int sqg_create(const sqg_cfg_t* cfg, sqg_ctx_t** out) {
    if (!cfg || !out) return SQG_EINVAL;
    *out = nullptr;
    if (cfg->abi_version != SQG_ABI_VERSION) return SQG_EINVAL;
    if (cfg->kmer_size < 1 || cfg->kmer_size > 9 || !cfg->model) return SQG_EINVAL;
    if (cfg->num_workers < 1 || cfg->worker_lo < 0 || cfg->worker_hi > cfg->num_workers || cfg->worker_lo >= cfg->worker_hi) return SQG_EINVAL;
    if (!(cfg->profile.range != 0.0) || !(cfg->profile.dwell_mean >= 1.0)) return SQG_EINVAL;
    if (cfg->profile.dwell_mean + 8.0 * std::fabs(cfg->profile.dwell_std) > 60000.0) return SQG_EINVAL;
    if (cfg->mode != SQG_MODE_EXACT && cfg->mode != SQG_MODE_CERTIFIED) return SQG_EINVAL;
    long long nk = 1LL << (2 * cfg->kmer_size);
    if (cfg->flags & SQG_METH) { nk = 1; for (uint32_t i = 0; i < cfg->kmer_size; i++) nk *= 5; }     // (uint32_t)pow(5,k), src/sim.c:325
    // canonical-form validity: |seed| + T*(nk+10) must stay where Schrage's uncorrected state is
    // within (-M, M) after one step (see DESIGN.md "LCG")
    const double span = std::fabs((double)cfg->seed) + (double)cfg->num_workers * (double)(nk + 10);
    if (span > 9.0e10) return SQG_EINVAL;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SQG_ENODEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return SQG_EINVAL;

    sqg_ctx* c = new (std::nothrow) sqg_ctx();
    if (!c) return SQG_ENOMEM;
    c->cfg = *cfg;
    c->cfg.model = nullptr;
    c->k = (int)cfg->kmer_size; c->num_kmer = (int)nk; c->T = cfg->num_workers;
    c->wlo = cfg->worker_lo; c->whi = cfg->worker_hi; c->nw = c->whi - c->wlo;
    c->use_dwell_stream = !(cfg->flags & (SQG_IDEAL | SQG_IDEAL_TIME));
    c->use_kmer_streams = !(cfg->flags & (SQG_IDEAL | SQG_IDEAL_AMP));
    int rc = SQG_OK;
    auto fail = [&](int code) { sqg_destroy(c); return code; };
#define CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = (e_ == hipErrorOutOfMemory) ? SQG_ENOMEM : SQG_EDEVICE; fprintf(stderr, "[sqg] %s: %s\n", #call, hipGetErrorString(e_)); return fail(rc); } } while (0)
    CHK(hipSetDevice(cfg->device));
    CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));


    // pore model: {level_mean, (float)(level_stdv*amp_noise)}  (src/sim.c:249)
    std::vector<float2> hm((size_t)nk);
    for (long long j = 0; j < nk; j++) {
        const float sd = cfg->model[j].level_stdv * cfg->amp_noise;
        hm[(size_t)j] = make_float2(cfg->model[j].level_mean, sd);
    }
    CHK(hipMalloc(&c->d_model, (size_t)nk * sizeof(float2)));
    CHK(hipMemcpy(c->d_model, hm.data(), (size_t)nk * sizeof(float2), hipMemcpyHostToDevice));

    // jump tables
    std::vector<uint32_t> pw((size_t)POW_WORDS);
    {
        const uint32_t a2 = lcg_mul(LCG_A, LCG_A);
        uint32_t p = 1;                                     // a^(2j)
        for (int j = 0; j < POW_N; j++) {
            pw[2 * POW_N + j] = p;
            pw[0 * POW_N + j] = lcg_mul(p, LCG_A);
            pw[1 * POW_N + j] = lcg_mul(p, a2);
            p = lcg_mul(p, a2);
        }
        const uint32_t step1 = p;                           // a^(2*1024)
        p = 1;
        for (int j = 0; j < POW_N; j++) { pw[3 * POW_N + j] = p; p = lcg_mul(p, step1); }
        const uint32_t step2 = p;                           // a^(2*1024*1024)
        p = 1;
        for (int j = 0; j < POW_TOP; j++) { pw[4 * POW_N + j] = p; p = lcg_mul(p, step2); }
    }
    CHK(hipMalloc(&c->d_pow, pw.size() * sizeof(uint32_t)));
    CHK(hipMemcpy(c->d_pow, pw.data(), pw.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->h_pow = pw;
    CHK(hipMalloc(&c->d_err, sizeof(unsigned int)));
    CHK(hipMalloc(&c->d_mid_done, sizeof(unsigned int)));
    CHK(hipMemset(c->d_mid_done, 0, sizeof(unsigned int)));
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && ncu > 0) c->num_cu = ncu; }
    CHK(hipMalloc(&c->d_zero, sizeof(unsigned int)));
    CHK(hipMemset(c->d_zero, 0, sizeof(unsigned int)));
    CHK(hipMemset(c->d_err, 0, sizeof(unsigned int)));
    CHK(hipStreamCreateWithFlags(&c->stage_stream, hipStreamNonBlocking));
    // SQG_OVERLAP=1: the sample kernels get their own stream, so that the event kernels of the next batch run next to
    // them.  Both are VALU-bound: measured +2 % with earlier kernels, -2..4 % with the current ones, and it stretches every
    // kernel's duration -- the default keeps one stream and clean per-kernel timings.  Batches are double-buffered either way.
    if (const char* split = SQG_DEV_ENV("SQG_CU_SPLIT")) {
        // (development build, experiment) CU partitioning: the event side on `n` CUs of every XCD, the sample kernels on the others, each
        // through a stream with a CU mask of its own -- two queues that cannot take each other's CUs.  SQG_CU_SPLIT="n" or "n,same"
        // (same: ONE stream on the 32 - n CUs per XCD: how a kernel scales with the CUs it gets).
        int rc2 = cu_split_streams(c, atoi(split), strstr(split, "same") != nullptr);
        if (rc2 != SQG_OK) return fail(rc2);
    }
    else if (SQG_DEV_ENV("SQG_OVERLAP")) CHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    else c->stream2 = c->stream;
    {   // lowest priority: the fix-ups fill the gaps of the next batch's k_events, they must not take its slots
        int prio_lo = 0, prio_hi = 0;
        CHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        CHK(hipStreamCreateWithPriority(&c->fix_stream, hipStreamNonBlocking, prio_lo));
    }
    for (auto& S : c->slot) {
        // (the four counters and, behind them, one word of statistics per list of the lean kernel: one read-back at sqg_batch_wait)
        CHK(hipMalloc(&S.d_fix_count, (4 + FIX_SHARDS) * sizeof(unsigned int)));
        CHK(hipMemset(S.d_fix_count, 0, (4 + FIX_SHARDS) * sizeof(unsigned int)));
        if (cfg->mode == SQG_MODE_CERTIFIED && c->use_kmer_streams) {
            CHK(hipMalloc(&S.d_fix_sh_count, (size_t)FIX_SHARDS * FIX_SHARD_STRIDE * sizeof(unsigned int)));          // (counters, FIX_SHARD_STRIDE words apart)
            CHK(hipMemset(S.d_fix_sh_count, 0, (size_t)FIX_SHARDS * FIX_SHARD_STRIDE * sizeof(unsigned int)));
        }
        CHK(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
        CHK(hipEventCreateWithFlags(&S.sampled, hipEventDisableTiming));
        CHK(hipEventRecord(S.done, c->stream2));
    }
    if (c->use_kmer_streams && !SQG_DEV_ENV("SQG_PART_CLAIMS") && !(cfg->flags & SQG_ORDER_FREE)) {
        // the hand-out over bucketed events by ordered LDS atomics (few workers).  Three lines of defence (k_part.h): an allow-list of
        // architectures on which the property was verified offline (2e8 fetch-adds, tests/test_split_chains.py, tools/stress_few.py:
        // ordered against order-free kernels over 4e11 samples); this check on THIS device, in the production shape (4096-entry
        // table, 16-bit addends, four wavefronts per CU, ~1.5 ms; sixteen times as long on an architecture that is not on the list);
        // and a sample of every slice of every batch (k_part_hand_ord).  SQG_ORDER_FREE in cfg.flags skips all of it.
        hipDeviceProp_t prop;
        CHK(hipGetDeviceProperties(&prop, cfg->device));
        const bool listed = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        if (!listed) fprintf(stderr, "[sqg] %s is not on the list of architectures verified for lane-ordered LDS atomics: extended probe\n", prop.gcnArchName);
        unsigned int* d_bad = nullptr;
        CHK(hipMalloc(&d_bad, sizeof(unsigned int)));
        CHK(hipMemset(d_bad, 0, sizeof(unsigned int)));
        hipLaunchKernelGGL(k_lds_order_check, dim3(1024), dim3(64), 0, c->stream, listed ? 4 : 64, d_bad);   // (four wavefronts per CU: the LDS pipeline is shared, as in the kernels)
        CHK(hipGetLastError());
        unsigned int bad = 1;
        CHK(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, c->stream));
        CHK(hipStreamSynchronize(c->stream));
        (void)hipFree(d_bad);
        c->lds_ordered = bad == 0;
        if (!c->lds_ordered) fprintf(stderr, "[sqg] LDS atomics are not served in lane order on this device (%u mismatches): the claim protocol hands the k-mer streams out\n", bad);
    }
    if (cfg->mode == SQG_MODE_CERTIFIED) {
        // exhaustive sweep of the fp32 deviate against the FP64 one on THIS device (~25 ms):
        // the bound the acceptance test uses is measured, not assumed
        unsigned int* d_max = nullptr;
        CHK(hipMalloc(&d_max, sizeof(unsigned int)));
        CHK(hipMemset(d_max, 0, sizeof(unsigned int)));
        hipLaunchKernelGGL(k_certify, dim3(256 * 16), dim3(256), 0, c->stream, d_max);
        CHK(hipGetLastError());
        unsigned int bits = 0;
        CHK(hipMemcpyAsync(&bits, d_max, sizeof bits, hipMemcpyDeviceToHost, c->stream));
        CHK(hipStreamSynchronize(c->stream));
        (void)hipFree(d_max);
        float m; memcpy(&m, &bits, sizeof m);
        c->delta_x_measured = m;
        if (!(m < 1.0e-4f)) { fprintf(stderr, "[sqg] certification sweep failed: max error %g\n", (double)m); return fail(SQG_EDEVICE); }
        c->delta_x = m * 1.25f + 1.0e-7f;
        if (getenv("SQG_VERBOSE")) fprintf(stderr, "[sqg] certified fp32 deviate: max |x_fp32 - x_fp64| over all states = %.3e, bound used %.3e\n", (double)m, (double)c->delta_x);
        // testing knob: inflate the bound so that (almost) every sample takes the FP64 fix-up path
        if (const char* ov = SQG_DEV_ENV("SQG_TEST_DELTA_X")) { c->delta_x = (float)atof(ov); c->force_fix = true; }
        // table-wide quantities of the lean kernel (same eps formula as k_samples<1, GENERIC>, per k-mer)
        const double kd = cfg->profile.digitisation / cfg->profile.range;
        double lo = 1e300, hi = -1e300, eps_max = 0;
        for (long long j = 0; j < nk; j++) {
            const double mkd = (double)hm[(size_t)j].x * kd;
            const float sdk = (float)((double)hm[(size_t)j].y * kd);
            const float asdk = std::fabs(sdk);
            lo = std::min(lo, mkd - 7.0 * asdk); hi = std::max(hi, mkd + 7.0 * asdk);
            const float eps = c->delta_x * asdk + 5.9604645e-8f * ((float)std::fabs(mkd) + 21.0f * asdk + 3.0f) + 2.0e-7f;
            eps_max = std::max(eps_max, (double)eps);
        }
        c->amp_floor = lo; c->amp_ceil = hi;
        c->thr_all = std::nextafterf((float)(0.5 - eps_max * 1.000001), 0.0f);
    }
    {
        const sqg_profile_t& q = cfg->profile;
        const double a = std::floor(q.dwell_mean + 6.5556 * std::fabs(q.dwell_std) + 0.5);
        const double z = std::floor(std::fabs(q.dwell_mean - 6.5556 * std::fabs(q.dwell_std)) + 0.5) + 1.0;
        c->dwell_hi = c->use_dwell_stream ? std::max(std::max(a, z), 1.0) + 1.0 : (double)(int)q.dwell_mean;
        // lean-kernel work item = 64*epl events: the largest epl whose items stay below LEAN_MAX_SAMPLES samples
        // (mean + 6 sigma of the item total; the rare longer item is left to the generic kernel)
        const double mu = std::fabs(q.dwell_mean) + 0.5, sg = c->use_dwell_stream ? std::fabs(q.dwell_std) : 0.0;
        c->lean_epl = 1;
        for (int epl = LEAN_EPL_MAX; epl >= 1; epl >>= 1) {
            const double nev = 64.0 * epl;
            if (nev * mu + 6.0 * std::sqrt(nev) * sg <= 0.97 * LEAN_MAX_SAMPLES) { c->lean_epl = epl; break; }
        }
        if (const char* ov = SQG_DEV_ENV("SQG_LEAN_EPL")) { const int v = atoi(ov); if (v == 1 || v == 2 || v == 4) c->lean_epl = v; }   // A/B knob
    }

    // per-(worker,k-mer) stream states
    if (c->use_kmer_streams) {
        const long long total = (long long)c->nw * nk;
        CHK(hipMalloc(&c->d_rows, (size_t)total * sizeof(uint32_t)));
        if (c->num_kmer <= 4096) {
            // rows hold the stream STATES; a chain moves its whole row through LDS (src/sim.c:248-256)
            const int blocks = (int)((total + 255) / 256);
            hipLaunchKernelGGL(k_init_rows, dim3(blocks), dim3(256), 0, c->stream, c->d_rows, (int)nk, (long long)cfg->seed, c->wlo, total);
            CHK(hipGetLastError());
        } else {
            // 1 MiB per worker, ~4 % of it used by a read: rows hold the number of SAMPLES each stream has produced, so
            // that one returning atomic add per k-mer bin replaces a load and a store; the state is seed * a^(2*count)
            // (test hook SQG_TEST_ROW_TURNS=t: start every count at t*(M-1)/2, which is the same stream position; t = 3
            // makes the first batch normalise the counts, t = 2 exercises the top of the jump tables)
            const char* turns_env = SQG_DEV_ENV("SQG_TEST_ROW_TURNS");
            const int turns = turns_env ? std::min(3, std::max(0, atoi(turns_env))) : 0;
            CHK(hipMemsetD32Async((hipDeviceptr_t)c->d_rows, (int)((unsigned)turns * LCG_ORD2), (size_t)total, c->stream));
            c->row_bound = (double)turns * (double)LCG_ORD2 + (turns == 3 ? (double)LCG_ORD2 : 0.0);
        }
    }
    // scalar streams (src/sim.c:241-247): time = s+2, offset = s+4, median = s+5
    c->time_c.resize((size_t)c->nw); c->off_x.resize((size_t)c->nw); c->med_x.resize((size_t)c->nw);
    for (int w = 0; w < c->nw; w++) {
        const long long s = (long long)cfg->seed + (long long)(w + c->wlo) * (nk + 10);
        c->time_c[(size_t)w] = canon(s + 2);
        c->off_x[(size_t)w] = s + 4;
        c->med_x[(size_t)w] = s + 5;
    }
    CHK(hipStreamSynchronize(c->stream));
#undef CHK
    *out = c;
    return SQG_OK;
}
