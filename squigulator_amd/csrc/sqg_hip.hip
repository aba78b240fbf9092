// sqg_hip.hip -- MI355X (gfx950) implementation of include/sqg.h.
//
// Host side of the C ABI: context/batch management, staging, launches, timing.  The gfx950
// kernels are in sqg_kernels.h (k_common.h, k_events.h, k_samples.h, k_sampler.h, k_svb.h).
// No MFMA anywhere: this is an integer-LCG / transcendental / streaming-store path.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see squigulator_amd/build.py).
//
// The product path never touches oracle/: this file is self-contained.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sqg.h"

#include "sqg_kernels.h"

static uint32_t lcg_pow(uint32_t base, unsigned long long e) {
    uint32_t r = 1, b = base;
    while (e) { if (e & 1) r = lcg_mul(r, b); b = lcg_mul(b, b); e >>= 1; }
    return r;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct sqg_ctx {
    sqg_cfg_t cfg;
    int k = 0, num_kmer = 0, T = 0, wlo = 0, whi = 0, nw = 0;
    hipStream_t stream = nullptr;
    uint32_t* d_rows = nullptr;
    float2* d_model = nullptr;
    uint32_t* d_pow = nullptr;
    unsigned int* d_err = nullptr;
    // Everything a batch's kernels write lives in one of two SLOTS (batch seq & 1): a batch's results stay valid while
    // the next one runs (sqg_batch_wait / sqg_fetch_* of batch i do not wait for batch i+1), and with SQG_OVERLAP=1 the
    // event kernels of batch i+1 (stream) run while the sample kernels of batch i (stream2) are still busy.
    struct Slot {
        int16_t* d_sig = nullptr; size_t sig_cap = 0;
        uint16_t* d_dwell = nullptr; size_t dwell_cap = 0;
        unsigned long long* d_seglen = nullptr; long long* d_sigoff = nullptr; size_t reads_cap = 0;
        FixEntry* d_fix = nullptr; size_t fix_cap = 0;
        unsigned int* d_fix_count = nullptr;       // [0] fix-up entries, [1] slow tiles
        uint2* d_evrec = nullptr; size_t evrec_cap = 0;
        uint32_t* d_tile_so = nullptr; size_t tile_cap = 0;
        int* d_slow = nullptr; size_t slow_cap = 0;
        uint4* d_tfix = nullptr; size_t tfix_cap = 0;
        unsigned char* d_tfix_n = nullptr; size_t tfixn_cap = 0;
        ItemDesc* d_items = nullptr; size_t items_cap = 0;          // [n_stiles] work items of the lean kernel (k_items)
        hipEvent_t done = nullptr;                 // recorded on stream2 after the slot's last sample kernel
    } slot[2];
    hipStream_t stream2 = nullptr;                 // the sample kernels (k_samples_lean, generic, fix-ups)
    uint32_t* d_link_rows = nullptr; size_t link_rows_cap = 0;   // split chains: one row per link of the running batch
    double row_bound = 0;                          // k > 6: upper bound of any sample count held in d_rows
    bool range_mode = false;                       // range sharding (sqg_set_range_mode): every batch is cut into links and run in two phases
    uint32_t* d_xcounts = nullptr; size_t xcounts_cap = 0;       // [nw][num_kmer] samples the running batch draws per stream (sqg_batch_run_begin)
    // device block, pinned offsets and events of freed batches, kept for the next sqg_batch_stage / sqg_batch_sample
    struct Recycled { uint8_t* d_block; size_t block_bytes; long long* h_sigoff; long long* h_sigoff_dev; size_t h_n; hipEvent_t ev[8]; };
    std::vector<Recycled> pool;
    hipStream_t stage_stream = nullptr;            // uploads and the staging kernels (k_sample, k_copy_reads, k_fill_tiles): a host
                                                   // can stage batch i+1 while batch i runs
    std::vector<uint32_t> time_c;          // canonical time-stream state per local worker
    std::vector<long long> off_x, med_x;   // raw Schrage states (as the reference keeps them)
    unsigned long long next_stage = 0, next_run = 0, compress_seq = 0;
    sqg_timing_t timing = {0, 0, 0, 0, 0, 0};
    bool use_dwell_stream = true, use_kmer_streams = true;
    float delta_x = 0.f;                   // certified mode: swept |x_fast - x_exact| bound incl. margin
    float delta_x_measured = 0.f;
    double amp_floor = 0, amp_ceil = 0;    // min/max over k-mers of m*kd -/+ 7|sd*kd| (ADC value range before the offset)
    float thr_all = -1.f;                  // lean-kernel acceptance threshold (0.5 - largest eps over the table)
    int lean_epl = 4;                      // events per lane of the lean kernel (work item = 64*lean_epl events)
    double dwell_hi = 1;                   // hard upper bound of a dwell draw
    bool force_fix = false;
    uint8_t* d_genome = nullptr;                                // resident reference (sqg_genome_load)
    long long* d_contig_off = nullptr; long long* d_cum = nullptr;
    float* d_trans_csum = nullptr; int* d_trans_idx = nullptr;
    uint32_t* d_samp = nullptr;                                 // [nw][3] sampler stream states: ref_pos, rand_strand, rand_rlen
    GenomeParams genome{};
    bool genome_loaded = false;
    double samp_ratio = 1.1;                                    // attempts per accepted read seen so far (long chains)
    uint8_t* d_svb = nullptr; size_t svb_cap = 0;               // svb-zd encodings of the last compressed batch
    long long* d_svb_size = nullptr; size_t svb_size_cap = 0;   // per read
    long long* d_svb_off = nullptr; size_t svb_off_cap = 0;
    long long tile_fix = 0;                // undecided samples parked per tile in the last batch (timing info)
    std::string err;
};

struct sqg_batch {
    unsigned long long seq = 0;
    int n = 0;
    long long n_events = 0, n_bases = 0, n_samples = 0;
    int n_chains = 0;
    std::vector<long long> ev_off, sig_off;
    std::vector<double> offset, median;
    std::vector<unsigned long long> seglen_host;   // only when dwell is constant
    uint8_t* d_block = nullptr;          // the batch's one device allocation; the pointers below point into it
    size_t block_bytes = 0, h_n = 0;     // its size; entries of h_sigoff
    uint8_t* d_bases = nullptr;
    ReadDesc* d_reads = nullptr;
    int* d_blk_read = nullptr;
    int* d_chain_off = nullptr;
    int* d_chain_reads = nullptr;
    int* d_chain_order = nullptr;
    bool split = false;                  // the worker chains are cut into links (d_chain_off describes the links)
    int n_wchains = 0;                   // workers with reads in this batch
    int* d_wlink_off = nullptr;          // [n_wchains+1] links of each worker chain
    int* d_wlink_worker = nullptr;       // [n_wchains]
    long long max_wchain_ev = 0;         // events of the longest worker chain
    int* d_tile_read = nullptr;
    int* d_stile_read = nullptr;
    long long n_tiles = 0, n_stiles = 0;
    long long* h_sigoff = nullptr;   // pinned, device-mapped: k_scan writes it directly
    long long* h_sigoff_dev = nullptr;   // its device-side address
    long long n_bases_total = 0;         // bytes in d_bases
    std::vector<long long> h_base_off;   // per read: its segment 0 in d_bases
    std::vector<int32_t> s_ref_idx, s_ref_len, s_ref_pos, s_rlen;   // sqg_batch_sample: what gen_read returned
    std::vector<char> s_strand;
    std::vector<long long> s_seq_off, s_read_at;                    // offsets of the reads in sqg_fetch_reads / in d_bases
    long long* h_svboff = nullptr;       // pinned, device-mapped: offsets of the svb-zd encodings (sqg_batch_compress)
    long long n_svb = -1;
    unsigned long long compress_seq = 0;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // kernel-phase boundaries; [7]: the event side is done
    int slot = 0;                        // which of the context's two buffer sets this batch runs in
    bool ran = false, waited = false, lean_timed = false, dwell_timed = false;
    bool begun = false, other_fresh = false;   // sqg_batch_run_begin has run; the other slot had never held a batch then
};

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return e_ == hipErrorOutOfMemory ? SQG_ENOMEM : SQG_EDEVICE;                       \
        }                                                                                      \
    } while (0)

// the reference's rng()/nrng() on the host, for the two per-read scalar draws that are
// RETURNED as doubles (offset, median_before; src/gensig.c:311-317): made with the host's libm
// so they are the very doubles the CPU reference produces on this machine.
static double host_rng(long long* xp) {            // src/rand.h:79-85
    const long long x = *xp;
    const long long nx = 16807LL * (x % 127773LL) - 2836LL * (x / 127773LL);
    *xp = nx;
    return (double)(nx > 0 ? nx : nx + 2147483647LL) / 2147483647;
}
static double host_nrng(double m, double s, long long* xp) {   // src/rand.h:87-94
    double u = 0.0, t = 0.0;
    while (u == 0.0) u = host_rng(xp);
    while (t == 0.0) t = 2.0 * 3.14159265 * host_rng(xp);
    const double z = std::sqrt(-2.0 * std::log(u)) * std::cos(t);
    return (z * s) + m;
}

static uint32_t canon(long long s) {
    s %= (long long)LCG_M;
    if (s < 0) s += LCG_M;
    return (uint32_t)s;
}

extern "C" const char* sqg_strerror(int code) {
    switch (code) {
    case SQG_OK: return "ok";
    case SQG_EINVAL: return "invalid argument or unsupported configuration";
    case SQG_ENOMEM: return "out of memory";
    case SQG_EDEVICE: return "HIP runtime error";
    case SQG_ESEQUENCE: return "batches must be run in staging order";
    case SQG_ENODEVICE: return "no usable HIP device";
    case SQG_EOVERFLOW: return "read too long (>= UINT32_MAX samples) or dwell > 65535";
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
    if (T <= 1) return 0;                                  // src/thread.c:122-125
    const int32_t step = (n_rec + T - 1) / T;              // src/thread.c:80
    return i / step;
}

extern "C" void sqg_destroy(sqg_ctx_t* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    (void)hipFree(ctx->d_rows); (void)hipFree(ctx->d_link_rows); (void)hipFree(ctx->d_xcounts); (void)hipFree(ctx->d_model); (void)hipFree(ctx->d_pow); (void)hipFree(ctx->d_err);
    for (auto& S : ctx->slot) {
        (void)hipFree(S.d_sig); (void)hipFree(S.d_dwell); (void)hipFree(S.d_seglen); (void)hipFree(S.d_sigoff);
        (void)hipFree(S.d_fix); (void)hipFree(S.d_fix_count);
        (void)hipFree(S.d_evrec); (void)hipFree(S.d_tile_so); (void)hipFree(S.d_slow);
        (void)hipFree(S.d_tfix); (void)hipFree(S.d_tfix_n); (void)hipFree(S.d_items);
        if (S.done) (void)hipEventDestroy(S.done);
    }
    (void)hipFree(ctx->d_svb); (void)hipFree(ctx->d_svb_size); (void)hipFree(ctx->d_svb_off);
    (void)hipFree(ctx->d_genome); (void)hipFree(ctx->d_contig_off); (void)hipFree(ctx->d_cum);
    (void)hipFree(ctx->d_trans_csum); (void)hipFree(ctx->d_trans_idx); (void)hipFree(ctx->d_samp);
    if (ctx->stage_stream) { (void)hipStreamSynchronize(ctx->stage_stream); (void)hipStreamDestroy(ctx->stage_stream); }
    for (auto& r : ctx->pool) { (void)hipFree(r.d_block); (void)hipHostFree(r.h_sigoff); for (auto& e : r.ev) if (e) (void)hipEventDestroy(e); }
    ctx->pool.clear();
    if (ctx->stream2 && ctx->stream2 != ctx->stream) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
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
    const long long nk = 1LL << (2 * cfg->kmer_size);
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
    CHK(hipMalloc(&c->d_err, sizeof(unsigned int)));
    CHK(hipMemset(c->d_err, 0, sizeof(unsigned int)));
    CHK(hipStreamCreateWithFlags(&c->stage_stream, hipStreamNonBlocking));
    // SQG_OVERLAP=1: the sample kernels get their own stream, so that the event kernels of the next batch run next to
    // them (measured +2 % throughput on the bench workload; it stretches every kernel's duration, which is why the
    // default keeps one stream and clean per-kernel timings).  Batches are double-buffered either way.
    if (getenv("SQG_OVERLAP")) CHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    else c->stream2 = c->stream;
    for (auto& S : c->slot) {
        CHK(hipMalloc(&S.d_fix_count, 4 * sizeof(unsigned int)));
        CHK(hipMemset(S.d_fix_count, 0, 4 * sizeof(unsigned int)));
        CHK(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
        CHK(hipEventRecord(S.done, c->stream2));
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
        // testing knob: inflate the bound so that (almost) every sample takes the FP64 fix-up path
        if (const char* ov = getenv("SQG_TEST_DELTA_X")) { c->delta_x = (float)atof(ov); c->force_fix = true; }
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
        const double a = std::floor(q.dwell_mean + 6.5546 * std::fabs(q.dwell_std) + 0.5);
        const double z = std::floor(std::fabs(q.dwell_mean - 6.5546 * std::fabs(q.dwell_std)) + 0.5) + 1.0;
        c->dwell_hi = c->use_dwell_stream ? std::max(std::max(a, z), 1.0) + 1.0 : (double)(int)q.dwell_mean;
        // lean-kernel work item = 64*epl events: the largest epl whose items stay below LEAN_MAX_SAMPLES samples
        // (mean + 6 sigma of the item total; the rare longer item is left to the generic kernel)
        const double mu = std::fabs(q.dwell_mean) + 0.5, sg = c->use_dwell_stream ? std::fabs(q.dwell_std) : 0.0;
        c->lean_epl = 1;
        for (int epl = LEAN_EPL_MAX; epl >= 1; epl >>= 1) {
            const double nev = 64.0 * epl;
            if (nev * mu + 6.0 * std::sqrt(nev) * sg <= 0.97 * LEAN_MAX_SAMPLES) { c->lean_epl = epl; break; }
        }
        if (const char* ov = getenv("SQG_LEAN_EPL")) { const int v = atoi(ov); if (v == 1 || v == 2 || v == 4) c->lean_epl = v; }   // A/B knob
    }

    // per-(worker,k-mer) stream states
    if (c->use_kmer_streams) {
        const long long total = (long long)c->nw * nk;
        CHK(hipMalloc(&c->d_rows, (size_t)total * sizeof(uint32_t)));
        if (c->k <= 6) {
            // rows hold the stream STATES; a chain moves its whole row through LDS (src/sim.c:248-256)
            const int blocks = (int)((total + 255) / 256);
            hipLaunchKernelGGL(k_init_rows, dim3(blocks), dim3(256), 0, c->stream, c->d_rows, (int)nk, (long long)cfg->seed, c->wlo, total);
            CHK(hipGetLastError());
        } else {
            // 1 MiB per worker, ~4 % of it used by a read: rows hold the number of SAMPLES each stream has produced, so
            // that one returning atomic add per k-mer bin replaces a load and a store; the state is seed * a^(2*count)
            // (test hook SQG_TEST_ROW_TURNS=t: start every count at t*(M-1)/2, which is the same stream position; t = 3
            // makes the first batch normalise the counts, t = 2 exercises the top of the jump tables)
            const char* turns_env = getenv("SQG_TEST_ROW_TURNS");
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

static int ensure(sqg_ctx* c, void** p, size_t* cap, size_t need, size_t elem) {
    if (need <= *cap) return SQG_OK;
    size_t ncap = std::max(need + need / 4, *cap + *cap / 2);     // slack: batches of similar size never re-allocate
    if (*p) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream2)); HIPCHK(c, hipFree(*p)); *p = nullptr; *cap = 0; }
    HIPCHK(c, hipMalloc(p, ncap * elem));
    *cap = ncap;
    return SQG_OK;
}

static const char kStallRna[] = "AAAAAGAAAAAACCCCCCCCCCCCCCCCCC";                  // src/genread.c:87
static const char kStallDna[] = "TTTTTTTTTTTTTTTTTTAATCAA";                       // src/genread.c:110
static const char kAdaptorDna[] = "GGCGTCTGCTTGGGTGTTTAACCTTTTTTTTTTAATGTACTTCGTTCAGTTACGTATTGCT";  // src/genread.c:38
static const char kAdaptorRna[] = "TGATGATGAGGGATAGACGATGGTTGTTTCTGTTGGTGCTGATATTGCTTTTTTTTTTTTTATGATGCAAGATACGCAC";  // src/genread.c:39
static const int kPolyA = 158;                                                   // src/genread.c:37
static const char kShortHack[] = "ACGTACGTACGTA";   // src/gensig.c:242-245: "ACGTACGTACGT" + its NUL (rank 0)

extern "C" void sqg_batch_free(sqg_ctx_t* ctx, sqg_batch_t* b) {
    if (!b) return;
    if (ctx) {
        (void)hipSetDevice(ctx->cfg.device);
        if (b->ran && b->ev[4]) (void)hipEventSynchronize(b->ev[4]);      // this batch's kernels only, not the ones queued after it
    }
    if (b->h_svboff) (void)hipHostFree(b->h_svboff);
    if (ctx && b->d_block && b->h_sigoff && b->ev[0] && ctx->pool.size() < 4) {
        sqg_ctx::Recycled r;
        r.d_block = b->d_block; r.block_bytes = b->block_bytes; r.h_sigoff = b->h_sigoff; r.h_sigoff_dev = b->h_sigoff_dev; r.h_n = b->h_n;
        for (int i = 0; i < 8; i++) r.ev[i] = b->ev[i];
        ctx->pool.push_back(r);
    } else {
        (void)hipFree(b->d_block);
        if (b->h_sigoff) (void)hipHostFree(b->h_sigoff);
        for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
    }
    delete b;
}

// Per-slot device buffers for a batch of this geometry.  with_output: also the signal slab and the fix-up list, sized
// by the hard bound on the dwell (skipped when that bound is unreasonable; sqg_batch_run then reads the scan back).
static int grow_slot(sqg_ctx* c, sqg_ctx::Slot& Z, const sqg_batch* b, bool with_output) {
    int rc2;
    const int n = b->n;
    const bool certified = c->cfg.mode == SQG_MODE_CERTIFIED;
    if ((size_t)n + 1 > Z.reads_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream2));
        (void)hipFree(Z.d_seglen); (void)hipFree(Z.d_sigoff); Z.d_seglen = nullptr; Z.d_sigoff = nullptr;
        const size_t cap = (size_t)n + 1 + (size_t)n / 2;
        HIPCHK(c, hipMalloc(&Z.d_seglen, 2 * cap * sizeof(unsigned long long)));
        HIPCHK(c, hipMalloc(&Z.d_sigoff, cap * sizeof(long long)));
        Z.reads_cap = cap;
    }
    if ((rc2 = ensure(c, (void**)&Z.d_dwell, &Z.dwell_cap, (size_t)b->n_events + 64, sizeof(uint16_t)))) return rc2;
    if ((rc2 = ensure(c, (void**)&Z.d_evrec, &Z.evrec_cap, (size_t)b->n_events + 64, sizeof(uint2)))) return rc2;
    if ((rc2 = ensure(c, (void**)&Z.d_tile_so, &Z.tile_cap, (size_t)b->n_tiles + 64, sizeof(uint32_t)))) return rc2;
    if ((rc2 = ensure(c, (void**)&Z.d_slow, &Z.slow_cap, (size_t)b->n_tiles + 64, sizeof(int)))) return rc2;
    if (certified && c->use_kmer_streams) {
        if ((rc2 = ensure(c, (void**)&Z.d_tfix, &Z.tfix_cap, (size_t)b->n_stiles * FIX_SLOTS + 64, sizeof(uint4)))) return rc2;
        if ((rc2 = ensure(c, (void**)&Z.d_tfix_n, &Z.tfixn_cap, (size_t)b->n_stiles + 64, 1))) return rc2;
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
#include <chrono>
static int stage_common(sqg_ctx_t* c, int32_t n, const char* seqs, const int64_t* seq_off,
                        const int32_t* worker, const SampleRec* d_rec, sqg_batch_t** out) {
    *out = nullptr;
    static const bool st_on = getenv("SQG_STAGE_TIMING") != nullptr;
    auto st_t0 = std::chrono::steady_clock::now();
    auto st_mark = [&](const char* what) { if (st_on) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[stage] %-22s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - st_t0).count()); st_t0 = t; } };
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const sqg_profile_t& p = c->cfg.profile;
    const bool rna = c->cfg.flags & SQG_RNA, prefix = c->cfg.flags & SQG_PREFIX;
    const int k = c->k;

    sqg_batch* b = new (std::nothrow) sqg_batch();
    if (!b) return SQG_ENOMEM;
    b->n = n; b->seq = c->next_stage;
    b->ev_off.assign((size_t)n + 1, 0); b->sig_off.assign((size_t)n + 1, 0);
    b->offset.resize((size_t)n); b->median.resize((size_t)n);
    std::vector<ReadDesc> rd((size_t)n);
    std::vector<int> wk((size_t)n);

    // pass 1: worker ids, segment geometry
    long long nb = 0, nev = 0;
    for (int i = 0; i < n; i++) {
        const int w = worker ? worker[i] : sqg_worker_of(i, n, c->T);
        if (w < c->wlo || w >= c->whi) { delete b; c->err = "read assigned to a worker this context does not own"; return SQG_EINVAL; }
        wk[(size_t)i] = w - c->wlo;
        const long long len = seq_off[i + 1] - seq_off[i];
        if (len < 0 || len > 2000000000LL) { delete b; return SQG_EINVAL; }
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
    long long ntile = 0;
    for (int i = 0; i < n; i++) { rd[(size_t)i].tile_off = (int)ntile; rd[(size_t)i].fast = 0; rd[(size_t)i].stile_off = 0; rd[(size_t)i].pad = 0; ntile += (rd[(size_t)i].ne0 + rd[(size_t)i].ne1 + 63) / 64; }
    if (ntile > 2000000000LL) { delete b; c->err = "batch too large"; return SQG_EINVAL; }
    b->n_tiles = ntile;
    const int lean_ev = 64 * c->lean_epl;
    long long nst = 0;                                        // super tiles of 64*lean_epl events (work items of k_samples_lean)
    for (int i = 0; i < n; i++) { rd[(size_t)i].stile_off = (int)nst; rd[(size_t)i].pad = 0; nst += (rd[(size_t)i].ne0 + rd[(size_t)i].ne1 + lean_ev - 1) / lean_ev; }
    b->n_stiles = nst;
    // the tile -> read maps are filled on the device (k_fill_tiles) once the descriptors are there

    st_mark("descriptors+tiles");
    // pass 2: base buffer (prefix/stall attached as src/genread.c:95-123 does)
    std::vector<uint8_t> hb(seqs ? (size_t)nb + 16 : 0, (uint8_t)'A');
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

    st_mark("base buffer");
    // pass 3: per-worker chains in batch order; host-side scalar streams advance in that order
    std::vector<int> count((size_t)c->nw, 0);
    for (int i = 0; i < n; i++) count[(size_t)wk[(size_t)i]]++;
    std::vector<int> chain_of((size_t)c->nw, -1), chain_off;
    chain_off.push_back(0);
    for (int w = 0; w < c->nw; w++) if (count[(size_t)w]) { chain_of[(size_t)w] = (int)chain_off.size() - 1; chain_off.push_back(chain_off.back() + count[(size_t)w]); }
    b->n_chains = (int)chain_off.size() - 1;
    std::vector<int> fill(chain_off.begin(), chain_off.end() - 1), chain_reads((size_t)n);
    for (int i = 0; i < n; i++) chain_reads[(size_t)fill[(size_t)chain_of[(size_t)wk[(size_t)i]]]++] = i;

    // Few workers, many reads (`-t 1`, `-t 8 -K 1000`): a worker chain would be one workgroup walking its reads one after
    // the other.  It is cut into links of whole reads, which k_events walks concurrently after k_link_hist/k_link_prefix
    // have prepared each link's view of the worker's k-mer streams.  SQG_SPLIT_CHAINS=0 disables this, =N forces it
    // with N links as the target (tests).
    const std::vector<int> wchain_off = chain_off;              // the worker chains: the host's scalar streams follow these
    const int n_wchains = b->n_chains;
    b->n_wchains = n_wchains;
    std::vector<long long> wchain_ev((size_t)n_wchains, 0);
    for (int i = 0; i < n; i++) wchain_ev[(size_t)chain_of[(size_t)wk[(size_t)i]]] += rd[(size_t)i].ne0 + rd[(size_t)i].ne1;
    for (long long v : wchain_ev) b->max_wchain_ev = std::max(b->max_wchain_ev, v);
    if (c->use_kmer_streams && k > 6 && (double)b->max_wchain_ev * c->dwell_hi >= 4294967295.0 - (double)LCG_ORD2) {
        delete b; c->err = "one worker's reads of a batch may draw more than 3.2e9 samples (k > 6): use smaller batches"; return SQG_EINVAL;
    }
    std::vector<int> wlink_off(1, 0), wlink_worker;
    {
        const char* env = getenv("SQG_SPLIT_CHAINS");
        const int forced = env ? atoi(env) : -1;
        const bool multi = n > n_wchains;
        const bool want = c->range_mode ? n > 0 : forced >= 0 ? (forced > 0 && multi) : (multi && n_wchains < 1024 && nev >= 65536);
        if (c->use_kmer_streams && want) {
            const size_t row_bytes = (size_t)c->num_kmer * sizeof(uint32_t);
            long long target = forced > 0 ? forced : 2048;
            target = std::min<long long>(target, std::max<long long>(8, (long long)(((size_t)1 << 30) / row_bytes)));
            std::vector<int> link_off(1, 0);
            for (int q = 0; q < n_wchains; q++) {
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
    // launch order: longest chain first, so the tail of the grid is made of short chains
    std::vector<long long> chain_ev((size_t)b->n_chains, 0);
    for (int q = 0; q < b->n_chains; q++)
        for (int ci = chain_off[(size_t)q]; ci < chain_off[(size_t)q + 1]; ci++) chain_ev[(size_t)q] += rd[(size_t)chain_reads[(size_t)ci]].ne0 + rd[(size_t)chain_reads[(size_t)ci]].ne1;
    // (a counting sort over 4096 length classes: exact order within a class does not matter for the tail)
    std::vector<int> chain_order((size_t)b->n_chains);
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

    const uint32_t a2 = lcg_mul(LCG_A, LCG_A);
    const bool no_lean = getenv("SQG_TEST_NO_LEAN") != nullptr;
    // the per-read scalar draws (host libm, so that `offset` / `median_before` are the doubles the CPU reference prints):
    // chains are independent, a few host threads share them
    auto chain_range = [&](int q_lo, int q_hi) {
        for (int q = q_lo; q < q_hi; q++)
            for (int ci = wchain_off[(size_t)q]; ci < wchain_off[(size_t)q + 1]; ci++) {   // batch order within the worker
                const int i = chain_reads[(size_t)ci];
                ReadDesc& d = rd[(size_t)i];
                const size_t w = (size_t)d.worker;
                if (c->cfg.flags & SQG_IDEAL) {                   // src/gensig.c:311-313
                    d.offset = p.offset_mean; b->median[(size_t)i] = p.median_before_mean;
                } else {                                          // src/gensig.c:315-316
                    d.offset = host_nrng(p.offset_mean, p.offset_std, &c->off_x[w]);
                    b->median[(size_t)i] = host_nrng(p.median_before_mean, p.median_before_std, &c->med_x[w]);
                }
                b->offset[(size_t)i] = d.offset;
                d.fast = (c->cfg.mode == SQG_MODE_CERTIFIED && c->use_kmer_streams && c->dwell_hi <= (double)MULT_N && !no_lean &&
                          c->amp_floor - d.offset > 4.0 && c->amp_ceil - d.offset < 65000.0) ? 1 : 0;
                d.time_c0 = c->time_c[w];
                if (c->use_dwell_stream)                          // two draws per event (src/gensig.c:255)
                    c->time_c[w] = lcg_mul(c->time_c[w], lcg_pow(a2, (unsigned long long)(d.ne0 + d.ne1)));
            }
    };
    {
        const int nth = (n_wchains >= 1024) ? (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())) : 1;
        if (nth <= 1) chain_range(0, n_wchains);
        else {
            std::vector<std::thread> th;
            const int per = (n_wchains + nth - 1) / nth;
            for (int t = 0; t < nth; t++) th.emplace_back(chain_range, std::min(t * per, n_wchains), std::min((t + 1) * per, n_wchains));
            for (auto& t : th) t.join();
        }
    }
    if (!c->use_dwell_stream) {                           // constant dwell: lengths are known now
        const unsigned long long sps = (unsigned long long)(int)p.dwell_mean;
        b->seglen_host.resize((size_t)2 * n);
        for (int i = 0; i < n; i++) { b->seglen_host[(size_t)2 * i] = sps * rd[(size_t)i].ne0; b->seglen_host[(size_t)2 * i + 1] = sps * rd[(size_t)i].ne1; }
    }

    // dwell kernel launch geometry: first read of every DW_EPB-event block
    const long long nblk = (nev + DW_EPB - 1) / DW_EPB;
    std::vector<int> blk_read((size_t)std::max<long long>(nblk, 1), 0);
    {
        int r = 0;
        for (long long bi = 0; bi < nblk; bi++) {
            const long long g = bi * DW_EPB;
            while (r + 1 < n && g >= rd[(size_t)r + 1].ev_off) r++;
            blk_read[(size_t)bi] = r;
        }
    }

    st_mark("chains+streams+blocks");
    auto bail = [&](int code) { sqg_batch_free(c, b); return code; };
#define CHKB(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { c->err = std::string(#call) + ": " + hipGetErrorString(e_); return bail(e_ == hipErrorOutOfMemory ? SQG_ENOMEM : SQG_EDEVICE); } } while (0)
    {   // one device allocation per batch, carved into the batch's arrays (256-byte aligned)
        size_t off = 0;
        auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        const size_t o_bases = carve((size_t)nb + 16), o_reads = carve(std::max<size_t>(1, rd.size()) * sizeof(ReadDesc)),
                     o_blk = carve(blk_read.size() * sizeof(int)), o_coff = carve(chain_off.size() * sizeof(int)),
                     o_crd = carve(std::max<size_t>(1, chain_reads.size()) * sizeof(int)),
                     o_st = carve((size_t)std::max<long long>(nst, 1) * sizeof(int)), o_t = carve((size_t)std::max<long long>(ntile, 1) * sizeof(int)),
                     o_ord = carve(std::max<size_t>(1, chain_order.size()) * sizeof(int)),
                     o_wlo = carve(wlink_off.size() * sizeof(int)), o_wlw = carve(std::max<size_t>(1, wlink_worker.size()) * sizeof(int));
        // a freed batch's block, pinned offsets and events are reused when they are large enough
        for (size_t pi = 0; pi < c->pool.size(); pi++) {
            sqg_ctx::Recycled& r = c->pool[pi];
            if (r.block_bytes >= off && r.h_n >= (size_t)n + 1) {
                b->d_block = r.d_block; b->block_bytes = r.block_bytes; b->h_sigoff = r.h_sigoff; b->h_sigoff_dev = r.h_sigoff_dev; b->h_n = r.h_n;
                for (int i = 0; i < 8; i++) b->ev[i] = r.ev[i];
                c->pool.erase(c->pool.begin() + (long)pi);
                break;
            }
        }
        if (!b->d_block) {
            if (c->pool.size() >= 4) {                      // nothing fits: make room
                sqg_ctx::Recycled& r = c->pool.front();
                (void)hipFree(r.d_block); (void)hipHostFree(r.h_sigoff); for (auto& e : r.ev) if (e) (void)hipEventDestroy(e);
                c->pool.erase(c->pool.begin());
            }
            b->block_bytes = off + off / 8;                 // slack: the next batches are about this size
            CHKB(hipMalloc(&b->d_block, b->block_bytes));
        }
        uint8_t* base = b->d_block;
        b->d_bases = base + o_bases; b->d_reads = (ReadDesc*)(base + o_reads); b->d_blk_read = (int*)(base + o_blk);
        b->d_chain_off = (int*)(base + o_coff); b->d_chain_reads = (int*)(base + o_crd); b->d_stile_read = (int*)(base + o_st);
        b->d_tile_read = (int*)(base + o_t); b->d_chain_order = (int*)(base + o_ord);
        b->d_wlink_off = (int*)(base + o_wlo); b->d_wlink_worker = (int*)(base + o_wlw);
    }
    if (seqs) CHKB(hipMemcpyAsync(b->d_bases, hb.data(), hb.size(), hipMemcpyHostToDevice, c->stage_stream));
    else CHKB(hipMemsetAsync(b->d_bases + nb, 'A', 16, c->stage_stream));
    if (n) CHKB(hipMemcpyAsync(b->d_reads, rd.data(), rd.size() * sizeof(ReadDesc), hipMemcpyHostToDevice, c->stage_stream));
    if (!seqs && n) {                                      // the reads come from the resident genome
        hipLaunchKernelGGL(k_copy_reads, dim3((unsigned)n), dim3(256), 0, c->stage_stream, c->genome, d_rec, b->d_reads, b->d_bases, n,
                           rna ? 1 : 0, prefix ? 1 : 0);
        CHKB(hipGetLastError());
    }
    if (n) {
        hipLaunchKernelGGL(k_fill_tiles, dim3((unsigned)n), dim3(64), 0, c->stage_stream, b->d_reads, n, lean_ev, b->d_tile_read, b->d_stile_read);
        CHKB(hipGetLastError());
    }
    b->n_bases_total = nb;
    b->h_base_off.resize((size_t)n);
    for (int i = 0; i < n; i++) b->h_base_off[(size_t)i] = rd[(size_t)i].base_off;
    CHKB(hipMemcpyAsync(b->d_blk_read, blk_read.data(), blk_read.size() * sizeof(int), hipMemcpyHostToDevice, c->stage_stream));
    CHKB(hipMemcpyAsync(b->d_chain_off, chain_off.data(), chain_off.size() * sizeof(int), hipMemcpyHostToDevice, c->stage_stream));
    if (n) CHKB(hipMemcpyAsync(b->d_chain_reads, chain_reads.data(), chain_reads.size() * sizeof(int), hipMemcpyHostToDevice, c->stage_stream));
    if (b->n_chains) CHKB(hipMemcpyAsync(b->d_chain_order, chain_order.data(), chain_order.size() * sizeof(int), hipMemcpyHostToDevice, c->stage_stream));
    if (b->split) {
        CHKB(hipMemcpyAsync(b->d_wlink_off, wlink_off.data(), wlink_off.size() * sizeof(int), hipMemcpyHostToDevice, c->stage_stream));
        CHKB(hipMemcpyAsync(b->d_wlink_worker, wlink_worker.data(), wlink_worker.size() * sizeof(int), hipMemcpyHostToDevice, c->stage_stream));
    }
    if (!b->h_sigoff) {
        b->h_n = (size_t)n + 1 + (size_t)n / 8;
        CHKB(hipHostMalloc(&b->h_sigoff, b->h_n * sizeof(long long), hipHostMallocMapped));
        CHKB(hipHostGetDevicePointer((void**)&b->h_sigoff_dev, b->h_sigoff, 0));
        for (auto& e : b->ev) CHKB(hipEventCreate(&e));
    }
    st_mark("mallocs+enqueue");
    CHKB(hipStreamSynchronize(c->stage_stream));     // staging buffers above are stack-owned (only the staging stream: a running batch is not waited for)
    st_mark("sync");
#undef CHKB
    // slots that have never held a batch are sized now, so that not even the first run allocates
    for (auto& Z : c->slot)
        if (Z.reads_cap == 0 && n > 0) { const int rg = grow_slot(c, Z, b, /*with_output=*/true); if (rg) return bail(rg); }
    c->next_stage++;
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

// ---- resident genome + device-side read sampler ("next" row of SURVEY.md section 8f) ----
extern "C" int sqg_genome_load(sqg_ctx_t* c, const sqg_genome_t* g) {
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
    (void)hipFree(c->d_genome); (void)hipFree(c->d_contig_off); (void)hipFree(c->d_cum);
    (void)hipFree(c->d_trans_csum); (void)hipFree(c->d_trans_idx); (void)hipFree(c->d_samp);
    c->d_genome = nullptr; c->d_contig_off = nullptr; c->d_cum = nullptr; c->d_trans_csum = nullptr; c->d_trans_idx = nullptr; c->d_samp = nullptr;
    HIPCHK(c, hipMalloc(&c->d_genome, (size_t)total + 16));
    HIPCHK(c, hipMemcpy(c->d_genome, g->seqs + g->contig_off[0], (size_t)total, hipMemcpyHostToDevice));
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
                       (long long)c->cfg.seed, c->wlo, c->nw, (int)(1u << (2 * c->k)));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stage_stream));
    GenomeParams& G = c->genome;
    G.seq = c->d_genome; G.contig_off = c->d_contig_off; G.cum = c->d_cum;
    G.trans_csum = c->d_trans_csum; G.trans_idx = c->d_trans_idx;
    G.sum = total; G.grng_b = (double)(g->rlen / 2); G.n_contigs = nc; G.n_trans = g->n_trans; G.rlen = g->rlen;
    G.flags = (int)g->mode;
    c->genome_loaded = true;
    return SQG_OK;
}

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
        c->time_c[(size_t)w] = lcg_mul(c->time_c[(size_t)w], lcg_pow(lcg_mul(LCG_A, LCG_A), (unsigned long long)n_events));
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

    SampleRec* d_rec = nullptr;
    int *d_co = nullptr, *d_cr = nullptr, *d_cw = nullptr;
    SampleRec* d_try = nullptr; unsigned char* d_ok = nullptr; long long* d_ao = nullptr;
    std::vector<SampleRec> rec((size_t)n);
    int rc = SQG_OK;
    auto cleanup = [&]() { (void)hipFree(d_rec); (void)hipFree(d_co); (void)hipFree(d_cr); (void)hipFree(d_cw); (void)hipFree(d_try); (void)hipFree(d_ok); (void)hipFree(d_ao); };
#define CHKS(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { c->err = std::string(#call) + ": " + hipGetErrorString(e_); cleanup(); return e_ == hipErrorOutOfMemory ? SQG_ENOMEM : SQG_EDEVICE; } } while (0)
    CHKS(hipMalloc(&d_rec, std::max<size_t>(1, (size_t)n) * sizeof(SampleRec)));
    if (n > 0) {
        CHKS(hipMalloc(&d_co, chain_off.size() * sizeof(int)));
        CHKS(hipMalloc(&d_cr, chain_reads.size() * sizeof(int)));
        CHKS(hipMalloc(&d_cw, chain_worker.size() * sizeof(int)));
        CHKS(hipMemcpyAsync(d_co, chain_off.data(), chain_off.size() * sizeof(int), hipMemcpyHostToDevice, c->stage_stream));
        CHKS(hipMemcpyAsync(d_cr, chain_reads.data(), chain_reads.size() * sizeof(int), hipMemcpyHostToDevice, c->stage_stream));
        CHKS(hipMemcpyAsync(d_cw, chain_worker.data(), chain_worker.size() * sizeof(int), hipMemcpyHostToDevice, c->stage_stream));
        int max_m = 0;
        for (int q = 0; q < n_chains; q++) max_m = std::max(max_m, chain_off[(size_t)q + 1] - chain_off[(size_t)q]);
        std::vector<long long> att_used;
        if (max_m >= 16 && !getenv("SQG_SAMPLER_SERIAL")) {
            // long chains: the attempts are evaluated concurrently, 25 % more than the acceptance rate seen so far asks for
            std::vector<long long> att_off((size_t)n_chains + 1, 0);
            long long max_a = 0;
            for (int q = 0; q < n_chains; q++) {
                const long long m = chain_off[(size_t)q + 1] - chain_off[(size_t)q];
                const long long a = (long long)std::ceil((double)m * c->samp_ratio * 1.25) + 64;
                att_off[(size_t)q + 1] = att_off[(size_t)q] + a; max_a = std::max(max_a, a);
            }
            const size_t na = (size_t)att_off.back();
            CHKS(hipMalloc(&d_try, na * sizeof(SampleRec)));
            CHKS(hipMalloc(&d_ok, na));
            CHKS(hipMalloc(&d_ao, (att_off.size() + (size_t)n_chains) * sizeof(long long)));
            long long* d_used = d_ao + att_off.size();
            CHKS(hipMemcpyAsync(d_ao, att_off.data(), att_off.size() * sizeof(long long), hipMemcpyHostToDevice, c->stage_stream));
            hipLaunchKernelGGL(k_sample_try, dim3((unsigned)((max_a + 3) / 4), (unsigned)n_chains), dim3(256), 0, c->stage_stream, c->genome, c->d_samp,
                               d_cw, d_ao, d_try, d_ok);
            hipLaunchKernelGGL(k_sample_pick, dim3((unsigned)n_chains), dim3(256), 0, c->stage_stream, c->genome, c->d_samp, d_co, d_cr, d_cw,
                               d_ao, d_try, d_ok, d_rec, d_used, c->d_err);
            CHKS(hipGetLastError());
            att_used.resize((size_t)n_chains);
            CHKS(hipMemcpyAsync(att_used.data(), d_used, att_used.size() * sizeof(long long), hipMemcpyDeviceToHost, c->stage_stream));
        } else {
            hipLaunchKernelGGL(k_sample, dim3((unsigned)n_chains), dim3(64), 0, c->stage_stream, c->genome, c->d_samp, d_co, d_cr, d_cw,
                               n_chains, d_rec, c->d_err);
            CHKS(hipGetLastError());
        }
        CHKS(hipMemcpyAsync(rec.data(), d_rec, rec.size() * sizeof(SampleRec), hipMemcpyDeviceToHost, c->stage_stream));
        CHKS(hipStreamSynchronize(c->stage_stream));
        if (!att_used.empty()) {
            double r = 1.0;
            for (int q = 0; q < n_chains; q++) {
                const int m = chain_off[(size_t)q + 1] - chain_off[(size_t)q];
                if (m >= 16) r = std::max(r, (double)att_used[(size_t)q] / (double)m);
            }
            c->samp_ratio = r;
        }
        unsigned int e = 0;
        CHKS(hipMemcpy(&e, c->d_err, sizeof e, hipMemcpyDeviceToHost));
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
    rc = stage_common(c, m, nullptr, seq_off.data(), m != n ? wk_glob.data() + lo : worker, d_rec + lo, out);
    if (rc == SQG_OK && m != n)
        for (int i = hi; i < n; i++) skip_read(c, wk[(size_t)i], read_events(c, rec[(size_t)i].rlen));
    cleanup();
    if (rc) return rc;
    sqg_batch* b = *out;
    const bool rna = c->cfg.flags & SQG_RNA, prefix = c->cfg.flags & SQG_PREFIX;
    const long long read_at = (prefix && !rna) ? (long long)(strlen(kStallDna) + strlen(kAdaptorDna)) : 0;
    rec.erase(rec.begin(), rec.begin() + lo); rec.resize((size_t)m);
    n = m;
    b->s_ref_idx.resize((size_t)n); b->s_ref_len.resize((size_t)n); b->s_ref_pos.resize((size_t)n); b->s_rlen.resize((size_t)n);
    b->s_strand.resize((size_t)n + 1); b->s_seq_off.assign(seq_off.begin(), seq_off.end()); b->s_read_at.resize((size_t)n);
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
    std::vector<uint8_t> all((size_t)b->n_bases_total + 1);
    if (b->n_bases_total) HIPCHK(c, hipMemcpy(all.data(), b->d_bases, (size_t)b->n_bases_total, hipMemcpyDeviceToHost));
    for (int i = 0; i < b->n; i++)
        memcpy(dst + b->s_seq_off[(size_t)i], all.data() + b->h_base_off[(size_t)i] + b->s_read_at[(size_t)i], (size_t)b->s_rlen[(size_t)i]);
    return SQG_OK;
}

// debugging aid: SQG_DEBUG_SYNC=1 synchronises after every launch and names the kernel that faulted
static int dbg_sync(sqg_ctx* c, const char* what) {
    static const bool on = getenv("SQG_DEBUG_SYNC") != nullptr;
    if (!on) return SQG_OK;
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream2);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { c->err = std::string(what) + ": " + hipGetErrorString(e); fprintf(stderr, "[sqg] %s\n", c->err.c_str()); return SQG_EDEVICE; }
    fprintf(stderr, "[sqg] %s ok\n", what);
    return SQG_OK;
}

// phase 0: the whole run; 1: up to the per-stream sample counts of a split batch (sqg_batch_run_begin); 2: the rest
// (sqg_batch_run_end), `before` / `after` being what the other ranges of the batch draw from each stream
static int run_impl(sqg_ctx* c, sqg_batch* b, const int phase, const uint32_t* before, const uint32_t* after) {
    if (!c || !b) return SQG_EINVAL;
    if (phase == 2 ? (!b->begun || b->ran) : (b->ran || b->begun || b->seq != c->next_run)) return SQG_ESEQUENCE;
    if ((before == nullptr) != (after == nullptr)) return SQG_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const sqg_profile_t& p = c->cfg.profile;
    const int n = b->n;
    const bool certified = c->cfg.mode == SQG_MODE_CERTIFIED;
    int rc;
    b->slot = (int)(b->seq & 1);
    sqg_ctx::Slot& S = c->slot[b->slot];
    sqg_ctx::Slot& other = c->slot[b->slot ^ 1];
    if (phase != 2) {
        // this slot's buffers were last used by the sample kernels of batch seq-2 (stream2)
        HIPCHK(c, hipStreamWaitEvent(c->stream, S.done, 0));
        auto grow = [&](sqg_ctx::Slot& Z) -> int { return grow_slot(c, Z, b, /*with_output=*/false); };
        b->other_fresh = other.reads_cap == 0 && n > 0;
        if ((rc = grow(S))) return rc;
        if (b->other_fresh && (rc = grow(other))) return rc;
    }
    const bool other_fresh = b->other_fresh;

    // Dwell draws are made inside k_events (SQG_SEPARATE_DWELL=1 keeps the stand-alone k_dwell for A/B runs).
    static const bool separate_dwell = getenv("SQG_SEPARATE_DWELL") != nullptr;
    const bool inline_dwell = c->use_dwell_stream && !separate_dwell;
    const bool direct = c->k <= 6;
    if (phase != 2 && !direct && c->use_kmer_streams && n > 0) {
        // the rows count samples in 32 bits; only the count mod (M-1)/2 matters (range mode: the other ranges' counts are not
        // known here, so the rows are reduced before every batch)
        const double bnd = (double)b->max_wchain_ev * c->dwell_hi;
        if (c->range_mode || c->row_bound + bnd >= 4294967295.0) {
            const size_t nrow = (size_t)c->nw * (size_t)c->num_kmer;
            hipLaunchKernelGGL(k_rows_normalize, dim3((unsigned)((nrow + 255) / 256)), dim3(256), 0, c->stream, c->d_rows, nrow);
            HIPCHK(c, hipGetLastError());
            c->row_bound = (double)LCG_ORD2;
        }
        c->row_bound += bnd;
    }
    if (phase != 2 && b->split && (rc = ensure(c, (void**)&c->d_link_rows, &c->link_rows_cap, (size_t)b->n_chains * (size_t)c->num_kmer, sizeof(uint32_t)))) return rc;
    const size_t n_rows = (size_t)c->nw * (size_t)c->num_kmer;
    if (phase == 1 && (rc = ensure(c, (void**)&c->d_xcounts, &c->xcounts_cap, n_rows, sizeof(uint32_t)))) return rc;
    SigParams P;
    memset(&P, 0, sizeof P);
    P.link_rows = b->split ? c->d_link_rows : nullptr;
    P.reads = b->d_reads; P.chain_off = b->d_chain_off; P.chain_reads = b->d_chain_reads; P.bases = b->d_bases;
    P.dwell = c->use_dwell_stream ? S.d_dwell : nullptr; P.dwell_out = S.d_dwell; P.seglen_out = S.d_seglen;
    P.dmean = p.dwell_mean; P.dstd = p.dwell_std;
    P.seglen = S.d_seglen; P.sig_off = S.d_sigoff; P.model = c->d_model; P.pw = c->d_pow; P.rows = c->d_rows;
    P.seed_base = canon((long long)c->cfg.seed + (long long)c->wlo * ((long long)(1u << (2 * c->k)) + 10)); P.seed_step = canon((long long)(1u << (2 * c->k)) + 10);
    P.err = c->d_err; P.dig = p.digitisation; P.range = p.range; P.kd = p.digitisation / p.range;
    P.chain_order = b->d_chain_order; P.delta_x = c->delta_x; P.thr_all = c->thr_all;
    P.k = c->k; P.num_kmer = c->num_kmer; P.const_sps = (int)p.dwell_mean;
    P.use_streams = c->use_kmer_streams ? 1 : 0;
    P.rna = (c->cfg.flags & SQG_RNA) ? 1 : 0;
    P.evrec = S.d_evrec; P.tile_so = S.d_tile_so; P.tile_read = b->d_tile_read; P.stile_read = b->d_stile_read;
    constexpr int NT = SQG_EVENT_THREADS;
    auto launch_events = [&](int dw, bool hist) {
        const dim3 g((unsigned)b->n_chains), t(NT);
#define EVL(D, W, H) hipLaunchKernelGGL((k_events<NT, D, W, SQG_EVENT_EPT, H>), g, t, 0, c->stream, P)
#define EVD(D, H) do { if (dw == 0) EVL(D, 0, H); else if (dw == 1) EVL(D, 1, H); else EVL(D, 2, H); } while (0)
        if (direct) { if (hist) EVD(true, true); else EVD(true, false); }
        else { if (hist) EVD(false, true); else EVD(false, false); }
#undef EVD
#undef EVL
    };

    if (phase != 2) {
        HIPCHK(c, hipEventRecord(b->ev[0], c->stream));
        if (n > 0) {
            if (c->use_dwell_stream && !inline_dwell) {
                HIPCHK(c, hipMemsetAsync(S.d_seglen, 0, (size_t)2 * n * sizeof(unsigned long long), c->stream));
                const long long nblk = (b->n_events + DW_EPB - 1) / DW_EPB;
                if (nblk > 0) {
                    if (certified)
                        hipLaunchKernelGGL(k_dwell<1>, dim3((unsigned)nblk), dim3(256), 0, c->stream, b->d_reads, n, b->d_blk_read,
                                           b->n_events, c->d_pow, p.dwell_mean, p.dwell_std, c->delta_x, S.d_dwell, S.d_seglen, c->d_err);
                    else
                        hipLaunchKernelGGL(k_dwell<0>, dim3((unsigned)nblk), dim3(256), 0, c->stream, b->d_reads, n, b->d_blk_read,
                                           b->n_events, c->d_pow, p.dwell_mean, p.dwell_std, 0.f, S.d_dwell, S.d_seglen, c->d_err);
                }
                if ((rc = dbg_sync(c, "k_dwell"))) return rc;
            } else if (!c->use_dwell_stream) {
                HIPCHK(c, hipMemcpyAsync(S.d_seglen, b->seglen_host.data(), (size_t)2 * n * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
            }
        }
        b->dwell_timed = c->use_dwell_stream && !inline_dwell;        // stand-alone k_dwell (A/B runs): two more timing events
        if (b->dwell_timed) { HIPCHK(c, hipEventRecord(b->ev[1], c->stream)); HIPCHK(c, hipEventRecord(b->ev[2], c->stream)); }
    }
    if (n > 0 && b->n_chains > 0) {
        const int dw = inline_dwell ? (certified && c->dwell_hi < 1.0e6 ? 1 : 2) : 0;
        const dim3 pg((unsigned)((c->num_kmer + 63) / 64), (unsigned)b->n_wchains);
        if (b->split && phase != 2) {
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
        if (b->split && phase != 1) {
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
    HIPCHK(c, hipEventRecord(b->ev[3], c->stream));
    if (n > 0) {
        // the scan also writes the offsets through the batch's pinned host mapping (no copy between kernels)
        // (k_items, when it runs, does that part with more parallelism)
        const bool items_run = certified && c->use_kmer_streams && b->n_chains > 0 && c->dwell_hi * (double)b->n_events <= 4.0e10;
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, c->stream, S.d_seglen, n, S.d_sigoff, items_run ? nullptr : b->h_sigoff_dev, c->d_err, S.d_fix_count);
        HIPCHK(c, hipGetLastError());
        if ((rc = dbg_sync(c, "k_scan"))) return rc;
    } else HIPCHK(c, hipMemsetAsync(S.d_fix_count, 0, 4 * sizeof(unsigned int), c->stream));
    // Output size is data-dependent.  A hard bound exists (|z| <= sqrt(2 ln(2^31-1)) = 6.5546 for any
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
        }
    }

    if (n > 0 && b->n_chains > 0) {
        P.sig = S.d_sig; P.fix = S.d_fix; P.fix_count = S.d_fix_count;
        P.fix_cap = (unsigned int)std::min<size_t>(S.fix_cap, 0xffffffffu);
        const bool rna_prefix = (c->cfg.flags & SQG_RNA) && (c->cfg.flags & SQG_PREFIX);
        P.shift_len = rna_prefix ? (int)strlen(kAdaptorRna) * (int)p.dwell_mean : 0;
        {   // int16_t off = 30*dig/range (src/genread.c:82): double -> int16 as the CPU does it
            const double v = 30 * p.digitisation / p.range;
            int32_t t = (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : (int32_t)0x80000000u;
            P.shift = (int)(int16_t)(uint16_t)((uint32_t)t & 0xffffu);
        }
        P.slow_tiles = nullptr; P.slow_count = S.d_fix_count + 1; P.tfix = S.d_tfix; P.tfix_n = S.d_tfix_n; P.items = S.d_items; P.lean_epl = c->lean_epl;
        const int n_tiles = (int)b->n_tiles;
        const unsigned sgrid = (unsigned)((n_tiles + 3) / 4);
        if (certified && c->use_kmer_streams) {
            P.slow_tiles = S.d_slow;
            const int n_stiles = (int)b->n_stiles;
            unsigned lgrid = (unsigned)((n_stiles + 3) / 4);
            static const int lean_grid_cap = getenv("SQG_LEAN_GRID") ? atoi(getenv("SQG_LEAN_GRID")) : 0;   // A/B knob
            if (lean_grid_cap > 0) lgrid = std::min(lgrid, (unsigned)lean_grid_cap);
            hipLaunchKernelGGL(k_items, dim3((unsigned)((std::max(n_stiles, n + 1) + 255) / 256)), dim3(256), 0, c->stream, P, n_stiles, n, b->h_sigoff_dev);
            HIPCHK(c, hipEventRecord(b->ev[7], c->stream));              // event side done: the sample kernels may start ...
            HIPCHK(c, hipStreamWaitEvent(c->stream2, b->ev[7], 0));      // ... on their own stream, next to the next batch's k_events
            HIPCHK(c, hipEventRecord(b->ev[5], c->stream2));
#define LEANL(R, E) hipLaunchKernelGGL((k_samples_lean<R, E>), dim3(lgrid), dim3(256), 0, c->stream2, P, n_stiles)
            if (P.rna) { if (c->lean_epl == 4) LEANL(true, 4); else if (c->lean_epl == 2) LEANL(true, 2); else LEANL(true, 1); }
            else { if (c->lean_epl == 4) LEANL(false, 4); else if (c->lean_epl == 2) LEANL(false, 2); else LEANL(false, 1); }
#undef LEANL
            HIPCHK(c, hipEventRecord(b->ev[6], c->stream2));
            b->lean_timed = true;
            if ((rc = dbg_sync(c, "k_samples_lean"))) return rc;
            hipLaunchKernelGGL((k_samples<1, true>), dim3(std::min(sgrid, 4096u)), dim3(256), 0, c->stream2, P, n_tiles);
            if ((rc = dbg_sync(c, "k_samples<generic>"))) return rc;
            hipLaunchKernelGGL(k_fixup, dim3(512), dim3(256), 0, c->stream2, P);
            hipLaunchKernelGGL(k_fixup_tiles, dim3((unsigned)((n_stiles + 255) / 256)), dim3(256), 0, c->stream2, P, n_stiles);
            if ((rc = dbg_sync(c, "k_fixup"))) return rc;
        } else {
            HIPCHK(c, hipEventRecord(b->ev[7], c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream2, b->ev[7], 0));
            if (certified) hipLaunchKernelGGL((k_samples<1, true>), dim3(sgrid), dim3(256), 0, c->stream2, P, n_tiles);
            else hipLaunchKernelGGL((k_samples<0, true>), dim3(sgrid), dim3(256), 0, c->stream2, P, n_tiles);
        }
        HIPCHK(c, hipGetLastError());
    } else {
        HIPCHK(c, hipEventRecord(b->ev[7], c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, b->ev[7], 0));
    }
    HIPCHK(c, hipEventRecord(b->ev[4], c->stream2));
    HIPCHK(c, hipEventRecord(S.done, c->stream2));
    b->ran = true;
    c->next_run++;
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
    if (c->next_stage != c->next_run) return SQG_ESEQUENCE;       // staged batches pending
    c->range_mode = on != 0;
    return SQG_OK;
}

extern "C" int sqg_batch_wait(sqg_ctx_t* c, sqg_batch_t* b, sqg_result_t* res) {
    if (!c || !b || !b->ran) return SQG_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipEventSynchronize(b->ev[4]));              // this batch only: later batches keep running
    sqg_ctx::Slot& S = c->slot[b->slot];
    if (!b->waited) {
        b->n_samples = b->h_sigoff[b->n];
        for (int i = 0; i <= b->n; i++) b->sig_off[(size_t)i] = b->h_sigoff[i];
        unsigned int e = 0;
        HIPCHK(c, hipMemcpy(&e, c->d_err, sizeof e, hipMemcpyDeviceToHost));
#if defined(SQG_ABL_EV_NOSTORE) || defined(SQG_ABL_NOSTORE)       /* timing-only ablation builds: results are garbage by design */
        if (e) { HIPCHK(c, hipMemset(c->d_err, 0, sizeof e)); e = 0; }
#endif
        if (e) {
            HIPCHK(c, hipMemset(c->d_err, 0, sizeof e));
            c->err = "device reported: " + std::string((e & 1) ? "dwell>65535 " : "") + ((e & 2) ? "read>=UINT32_MAX samples " : "") + ((e & 4) ? "internal length mismatch " : "") + ((e & 8) ? "FP64 fix-up list overflow" : "");
            return (e & 12) ? SQG_EDEVICE : SQG_EOVERFLOW;
        }
        float d = 0, s = 0, t = 0, ee = 0;
        if (b->dwell_timed) HIPCHK(c, hipEventElapsedTime(&d, b->ev[0], b->ev[1]));
        HIPCHK(c, hipEventElapsedTime(&ee, b->ev[b->dwell_timed ? 2 : 0], b->ev[3]));
        HIPCHK(c, hipEventElapsedTime(&s, b->ev[3], b->ev[4]));
        HIPCHK(c, hipEventElapsedTime(&t, b->ev[0], b->ev[4]));
        c->timing.events_ms = ee;
        c->timing.lean_ms = 0.f;
        if (b->lean_timed) HIPCHK(c, hipEventElapsedTime(&c->timing.lean_ms, b->ev[5], b->ev[6]));
        unsigned int nfix = 0;
        if (c->cfg.mode == SQG_MODE_CERTIFIED) {
            unsigned int cnt[4] = {0, 0, 0, 0};
            HIPCHK(c, hipMemcpy(cnt, S.d_fix_count, sizeof cnt, hipMemcpyDeviceToHost));
            nfix = cnt[0];                                  // global list ...
            if (c->use_kmer_streams && b->n_stiles > 0) {   // ... plus the per-tile slots of the lean kernel
                std::vector<unsigned char> tn((size_t)b->n_stiles);
                HIPCHK(c, hipMemcpy(tn.data(), S.d_tfix_n, tn.size(), hipMemcpyDeviceToHost));
                for (unsigned char v : tn) nfix += v;
            }
        }
        c->timing.dwell_ms = d; c->timing.samples_ms = s; c->timing.total_ms = t; c->timing.fallback_samples = nfix;
        b->waited = true;
    }
    if (res) {
        res->n_reads = b->n; res->n_events = b->n_events; res->n_samples = b->n_samples; res->n_bases = b->n_bases;
        res->sig_off = (const int64_t*)b->sig_off.data(); res->ev_off = (const int64_t*)b->ev_off.data();
        res->offset = b->offset.data(); res->median_before = b->median.data();
        res->d_signal = S.d_sig; res->d_dwell = c->use_dwell_stream ? S.d_dwell : nullptr;
    }
    return SQG_OK;
}

extern "C" int sqg_fetch_signal(sqg_ctx_t* c, sqg_batch_t* b, int16_t* dst) {
    if (!c || !b || !b->ran || !dst) return SQG_EINVAL;
    if (b->seq + 2 < c->next_run) return SQG_ESEQUENCE;       // slab already reused (two batches later)
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipEventSynchronize(b->ev[4]));
    if (!b->waited) b->n_samples = b->h_sigoff[b->n];
    if (b->n_samples) HIPCHK(c, hipMemcpy(dst, c->slot[b->slot].d_sig, (size_t)b->n_samples * sizeof(int16_t), hipMemcpyDeviceToHost));
    return SQG_OK;
}

extern "C" int sqg_fetch_dwell(sqg_ctx_t* c, sqg_batch_t* b, int32_t* dst) {
    if (!c || !b || !b->ran || !dst) return SQG_EINVAL;
    if (b->seq + 2 < c->next_run) return SQG_ESEQUENCE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipEventSynchronize(b->ev[4]));
    if (!c->use_dwell_stream) {
        for (long long i = 0; i < b->n_events; i++) dst[i] = (int)c->cfg.profile.dwell_mean;
        return SQG_OK;
    }
    std::vector<uint16_t> tmp((size_t)b->n_events);
    if (b->n_events) HIPCHK(c, hipMemcpy(tmp.data(), c->slot[b->slot].d_dwell, tmp.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); i++) dst[i] = tmp[i];
    return SQG_OK;
}

extern "C" int sqg_get_timing(sqg_ctx_t* c, sqg_timing_t* t) {
    if (!c || !t) return SQG_EINVAL;
    *t = c->timing;
    return SQG_OK;
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
    if (b->seq + 2 < c->next_run) return SQG_ESEQUENCE;       // the signals of an older batch are gone
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
