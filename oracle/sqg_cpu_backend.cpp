/*
 * sqg_cpu_backend.cpp -- the C ABI of include/sqg.h implemented on the CPU oracle ("CPU backend
 * with identical symbols").
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/: libsqg_cpu.so exports the very symbols libsqg_hip.so exports,
 * so that ONE host program -- the ctypes binding in squigulator_amd/api.py with lib_path=..., or examples/process_db_gpu.c
 * linked against it -- can be driven through both and the outputs compared (tests/test_cpu_backend.py), and so that a
 * host can be developed without a GPU.  The product never loads it: libsqg_hip.so has no CPU fallback.
 *
 * Semantics: those of include/sqg.h.  "Device" pointers in the result structs are host pointers here.  Not provided (they
 * return SQG_EINVAL): range sharding (sqg_set_range_mode and friends) -- one CPU process owns all workers anyway.
 * The BLOW5 writer is the product's own host code (squigulator_amd/csrc/h_blow5.h), compiled in unchanged.
 */
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../include/sqg.h"
#include "sqg_oracle.h"

struct sqg_ctx {
    sqg_cfg_t cfg;
    orc_core_t* core = nullptr;
    orc_ref_t* ref = nullptr;           // resident genome (sqg_genome_load)
    unsigned long long next_stage = 0, next_run = 0;
    sqg_timing_t timing = {0, 0, 0, 0, 0, 0, 0, 0};
    std::string err;
    // (h_blow5.h's stored-mode writer registers itself with the context whose records its background write reads: same fields as the HIP library's context)
    void* b5_reader = nullptr; int b5_reader_buf = -1; void (*b5_reader_drain)(void* writer, bool unbind) = nullptr; int b5_flip = 0;
};

struct sqg_batch {
    unsigned long long seq = 0;
    int n = 0;
    std::vector<std::string> reads;
    std::vector<int32_t> worker;        // context-global worker ids
    bool explicit_workers = false;
    orc_batch_t* out = nullptr;
    std::vector<int16_t> sig;
    std::vector<uint16_t> dwell;
    std::vector<int64_t> sig_off, ev_off, svb_off, seq_off;
    std::vector<double> offset, median;
    std::vector<uint8_t> svb;
    std::vector<int32_t> s_ref_idx, s_ref_len, s_ref_pos, s_rlen;
    std::vector<char> s_strand;
    long long n_bases = 0;
    bool ran = false;
};

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
extern "C" const char* sqg_last_error(const sqg_ctx_t* c) { return c ? c->err.c_str() : ""; }
extern "C" int sqg_device_count(void) { return 1; }        /* "the CPU" */
extern "C" const char* sqg_build_info(void) { return "source_hash=cpu-backend;dev=0"; }
extern "C" int32_t sqg_worker_of(int32_t i, int32_t n_rec, int32_t T) { return (T <= 1 || n_rec <= 0) ? 0 : orc_worker_of(i, n_rec, T); }

extern "C" void sqg_destroy(sqg_ctx_t* c) {
    if (!c) return;
    if (c->b5_reader_drain) c->b5_reader_drain(c->b5_reader, true);
    if (c->core) orc_core_free(c->core);
    if (c->ref) orc_ref_free(c->ref);
    delete c;
}

extern "C" int sqg_create(const sqg_cfg_t* cfg, sqg_ctx_t** out) {
    if (!cfg || !out) return SQG_EINVAL;
    *out = nullptr;
    if (cfg->abi_version != SQG_ABI_VERSION) return SQG_EINVAL;
    if (cfg->kmer_size < 1 || cfg->kmer_size > 9 || !cfg->model) return SQG_EINVAL;
    if (cfg->num_workers < 1 || cfg->worker_lo < 0 || cfg->worker_hi > cfg->num_workers || cfg->worker_lo >= cfg->worker_hi) return SQG_EINVAL;
    if (!(cfg->profile.range != 0.0) || !(cfg->profile.dwell_mean >= 1.0)) return SQG_EINVAL;
    sqg_ctx* c = new (std::nothrow) sqg_ctx();
    if (!c) return SQG_ENOMEM;
    c->cfg = *cfg; c->cfg.model = nullptr;
    orc_profile_t p;
    memcpy(&p, &cfg->profile, sizeof p);                   /* same ten doubles, same order */
    c->core = orc_core_new(&p, cfg->flags, cfg->amp_noise, cfg->kmer_size, reinterpret_cast<const orc_kmer_t*>(cfg->model), cfg->seed,
                           cfg->num_workers, 10000);
    if (!c->core) { delete c; return SQG_ENOMEM; }
    *out = c;
    return SQG_OK;
}

static int stage(sqg_ctx* c, int32_t n, std::vector<std::string>&& reads, const int32_t* worker, sqg_batch_t** out) {
    sqg_batch* b = new (std::nothrow) sqg_batch();
    if (!b) return SQG_ENOMEM;
    b->n = n; b->seq = c->next_stage++;
    b->reads = std::move(reads);
    b->worker.resize((size_t)n);
    b->explicit_workers = worker != nullptr;
    for (int i = 0; i < n; i++) {
        const int w = worker ? worker[i] : sqg_worker_of(i, n, c->cfg.num_workers);
        if (w < c->cfg.worker_lo || w >= c->cfg.worker_hi) { delete b; c->err = "read assigned to a worker this context does not own"; return SQG_EINVAL; }
        b->worker[(size_t)i] = w;
        b->n_bases += (long long)b->reads[(size_t)i].size();
    }
    *out = b;
    return SQG_OK;
}

extern "C" int sqg_batch_stage(sqg_ctx_t* c, int32_t n, const char* seqs, const int64_t* seq_off, const int32_t* worker, sqg_batch_t** out) {
    if (!c || !out || n < 0 || (n > 0 && (!seqs || !seq_off))) return SQG_EINVAL;
    std::vector<std::string> reads((size_t)n);
    for (int i = 0; i < n; i++) reads[(size_t)i].assign(seqs + seq_off[i], (size_t)(seq_off[i + 1] - seq_off[i]));
    return stage(c, n, std::move(reads), worker, out);
}

extern "C" int sqg_batch_run(sqg_ctx_t* c, sqg_batch_t* b) {
    if (!c || !b) return SQG_EINVAL;
    if (b->ran || b->seq != c->next_run) return SQG_ESEQUENCE;
    const int n = b->n;
    std::vector<const char*> ptr((size_t)std::max(n, 1));
    std::vector<int32_t> len((size_t)std::max(n, 1));
    for (int i = 0; i < n; i++) { ptr[(size_t)i] = b->reads[(size_t)i].c_str(); len[(size_t)i] = (int32_t)b->reads[(size_t)i].size(); }
    b->out = orc_batch_run_assigned(c->core, n, ptr.data(), len.data(), b->worker.data(), 1);
    b->sig_off.assign((size_t)n + 1, 0); b->ev_off.assign((size_t)n + 1, 0);
    b->offset.resize((size_t)n); b->median.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        const orc_read_t& r = b->out->reads[i];
        b->sig_off[(size_t)i + 1] = b->sig_off[(size_t)i] + r.len_raw_signal;
        b->ev_off[(size_t)i + 1] = b->ev_off[(size_t)i] + r.ss_n;
        b->offset[(size_t)i] = r.offset; b->median[(size_t)i] = r.median_before;
    }
    b->sig.resize((size_t)b->sig_off[(size_t)n]); b->dwell.resize((size_t)b->ev_off[(size_t)n]);
    for (int i = 0; i < n; i++) {
        const orc_read_t& r = b->out->reads[i];
        if (r.len_raw_signal) memcpy(b->sig.data() + b->sig_off[(size_t)i], r.raw_signal, (size_t)r.len_raw_signal * 2);
        for (int64_t e = 0; e < r.ss_n; e++) b->dwell[(size_t)(b->ev_off[(size_t)i] + e)] = (uint16_t)r.ss[e];
    }
    b->ran = true;
    c->next_run++;
    return SQG_OK;
}

extern "C" int sqg_batch_wait(sqg_ctx_t* c, sqg_batch_t* b, sqg_result_t* res) {
    if (!c || !b || !b->ran) return SQG_EINVAL;
    if (res) {
        res->n_reads = b->n; res->n_events = b->ev_off[(size_t)b->n]; res->n_samples = b->sig_off[(size_t)b->n]; res->n_bases = b->n_bases;
        res->sig_off = b->sig_off.data(); res->ev_off = b->ev_off.data();
        res->offset = b->offset.data(); res->median_before = b->median.data();
        res->d_signal = b->sig.data(); res->d_dwell = b->dwell.data();
    }
    return SQG_OK;
}

extern "C" int sqg_fetch_signal(sqg_ctx_t* c, sqg_batch_t* b, int16_t* dst) {
    if (!c || !b || !b->ran || !dst) return SQG_EINVAL;
    if (!b->sig.empty()) memcpy(dst, b->sig.data(), b->sig.size() * 2);
    return SQG_OK;
}
extern "C" int sqg_fetch_dwell(sqg_ctx_t* c, sqg_batch_t* b, int32_t* dst) {
    if (!c || !b || !b->ran || !dst) return SQG_EINVAL;
    for (size_t i = 0; i < b->dwell.size(); i++) dst[i] = b->dwell[i];
    return SQG_OK;
}
extern "C" void sqg_batch_free(sqg_ctx_t* c, sqg_batch_t* b) {
    if (!b) return;
    if (c && !b->ran && b->seq == c->next_run) c->next_run++;      /* freed without a run: the next batch may go */
    if (b->out) orc_batch_free(b->out);
    delete b;
}
extern "C" int sqg_get_timing(sqg_ctx_t* c, sqg_timing_t* t) { if (!c || !t) return SQG_EINVAL; *t = c->timing; return SQG_OK; }
extern "C" int sqg_set_stage_threads(sqg_ctx_t* c, int n) { return (!c || n < 0 || n > 64) ? SQG_EINVAL : 1; }   // (one thread stages here)
extern "C" int sqg_set_phase_timing(sqg_ctx_t* c, int every) { return (!c || every < 0) ? SQG_EINVAL : SQG_OK; }   // (no phases to time here)

extern "C" int sqg_submit(sqg_ctx_t* c, int32_t n, const char* seqs, const int64_t* seq_off, const int32_t* worker, sqg_batch_t** out, sqg_result_t* res) {
    if (!out) return SQG_EINVAL;
    int rc = sqg_batch_stage(c, n, seqs, seq_off, worker, out);
    if (rc) return rc;
    if ((rc = sqg_batch_run(c, *out)) || (rc = sqg_batch_wait(c, *out, res))) { sqg_batch_free(c, *out); *out = nullptr; }
    return rc;
}

extern "C" int sqg_batch_compress(sqg_ctx_t* c, sqg_batch_t* b, sqg_svb_t* out) {
    if (!c || !b || !b->ran || !out) return SQG_EINVAL;
    const int n = b->n;
    b->svb_off.assign((size_t)n + 1, 0);
    b->svb.clear();
    for (int i = 0; i < n; i++) {
        const int64_t len = b->sig_off[(size_t)i + 1] - b->sig_off[(size_t)i];
        const size_t at = b->svb.size();
        b->svb.resize(at + orc_svb_zd_bound(len));
        const size_t used = orc_svb_zd(b->sig.data() + b->sig_off[(size_t)i], len, b->svb.data() + at);
        b->svb.resize(at + used);
        b->svb_off[(size_t)i + 1] = (int64_t)b->svb.size();
    }
    out->n_bytes = (int64_t)b->svb.size(); out->svb_off = b->svb_off.data(); out->d_svb = b->svb.data();
    return SQG_OK;
}
extern "C" int sqg_fetch_svb(sqg_ctx_t* c, sqg_batch_t* b, uint8_t* dst) {
    if (!c || !b || !dst || b->svb_off.empty()) return SQG_EINVAL;
    if (!b->svb.empty()) memcpy(dst, b->svb.data(), b->svb.size());
    return SQG_OK;
}

/* ---- resident genome + sampler ---- */
static int genome_load(sqg_ctx_t* c, const sqg_genome_t* g) {
    if (!c || !g || g->n_contigs <= 0 || !g->seqs || !g->contig_off || g->rlen <= 0) return SQG_EINVAL;
    if (c->ref) { orc_ref_free(c->ref); c->ref = nullptr; }
    orc_ref_t* r = (orc_ref_t*)calloc(1, sizeof *r);
    r->num_ref = g->n_contigs;
    r->names = (char**)calloc((size_t)g->n_contigs, sizeof(char*));
    r->seqs = (char**)calloc((size_t)g->n_contigs, sizeof(char*));
    r->lengths = (int32_t*)calloc((size_t)g->n_contigs, sizeof(int32_t));
    for (int i = 0; i < g->n_contigs; i++) {
        const int64_t len = g->contig_off[i + 1] - g->contig_off[i];
        r->lengths[i] = (int32_t)len; r->sum += len;
        r->seqs[i] = (char*)malloc((size_t)len + 1);
        memcpy(r->seqs[i], (const char*)g->seqs + g->contig_off[i], (size_t)len); r->seqs[i][len] = '\0';
        char nm[32]; snprintf(nm, sizeof nm, "contig%d", i);
        r->names[i] = strdup(nm);
    }
    if (g->n_trans > 0) {
        r->trans_n = g->n_trans;
        r->trans_csum = (float*)malloc(sizeof(float) * (size_t)g->n_trans);
        r->trans_idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)g->n_trans);
        memcpy(r->trans_csum, g->trans_csum, sizeof(float) * (size_t)g->n_trans);
        memcpy(r->trans_idx, g->trans_idx, sizeof(int32_t) * (size_t)g->n_trans);
    }
    c->ref = r;
    /* the sampler variant and -r live in the core (src/sq.h:71-87) */
    c->core->rlen = g->rlen;
    for (int t = 0; t < c->core->num_workers; t++) c->core->workers[t].rlen.b = (double)(g->rlen / 2);
    uint32_t f = c->core->flags & ~(uint32_t)(ORC_CDNA | ORC_TRANS_TRUNC | ORC_FULL_CONTIG);
    if (g->mode & SQG_SAMPLE_CDNA) f |= ORC_CDNA;
    if (g->mode & SQG_SAMPLE_TRUNC) f |= ORC_TRANS_TRUNC;
    if (g->mode & SQG_SAMPLE_FULL) f |= ORC_FULL_CONTIG;
    c->core->flags = f;
    return SQG_OK;
}
extern "C" int sqg_genome_set_meth(sqg_ctx_t* c, const uint8_t* freq, const uint8_t* contig_has) {
    if (!c || !c->ref || !freq || !contig_has) return SQG_EINVAL;
    orc_ref_t* r = c->ref;
    if (r->meth) { for (int i = 0; i < r->num_ref; i++) free(r->meth[i]); free(r->meth); }
    r->meth = (uint8_t**)calloc((size_t)r->num_ref, sizeof(uint8_t*));
    size_t at = 0;
    for (int i = 0; i < r->num_ref; i++) {
        if (contig_has[i]) { r->meth[i] = (uint8_t*)malloc((size_t)r->lengths[i] + 1); memcpy(r->meth[i], freq + at, (size_t)r->lengths[i]); }
        at += (size_t)r->lengths[i];
    }
    return SQG_OK;
}
extern "C" int sqg_genome_load(sqg_ctx_t* c, const sqg_genome_t* g) { return genome_load(c, g); }
extern "C" int sqg_genome_load_device(sqg_ctx_t* c, const sqg_genome_t* g) { return genome_load(c, g); }   /* (host memory here) */

extern "C" int sqg_batch_sample(sqg_ctx_t* c, int32_t n, const int32_t* worker, sqg_batch_t** out, sqg_sample_t* info) {
    if (!c || !out || n < 0) return SQG_EINVAL;
    if (!c->ref) { c->err = "sqg_genome_load has not been called"; return SQG_EINVAL; }
    if (c->core->flags & ORC_FULL_CONTIG) { c->err = "--full-contigs is not provided by the CPU backend"; return SQG_EINVAL; }
    std::vector<std::string> reads((size_t)n);
    std::vector<int32_t> ri((size_t)n), rl((size_t)n), rp((size_t)n), ln((size_t)n);
    std::vector<char> st((size_t)n + 1);
    for (int i = 0; i < n; i++) {                              /* index order == every worker's own order */
        const int w = worker ? worker[i] : sqg_worker_of(i, n, c->cfg.num_workers);
        if (w < c->cfg.worker_lo || w >= c->cfg.worker_hi) { c->err = "read assigned to a worker this context does not own"; return SQG_EINVAL; }
        char strand = '+';
        char* s = orc_gen_read(c->core, c->ref, w, &ri[(size_t)i], &rl[(size_t)i], &rp[(size_t)i], &ln[(size_t)i], &strand);
        reads[(size_t)i].assign(s, (size_t)ln[(size_t)i]);
        st[(size_t)i] = strand;
        free(s);
    }
    const int rc = stage(c, n, std::move(reads), worker, out);
    if (rc) return rc;
    sqg_batch* b = *out;
    b->s_ref_idx = ri; b->s_ref_len = rl; b->s_ref_pos = rp; b->s_rlen = ln; b->s_strand = st;
    b->seq_off.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) b->seq_off[(size_t)i + 1] = b->seq_off[(size_t)i] + ln[(size_t)i];
    if (info) {
        info->ref_idx = b->s_ref_idx.data(); info->ref_len = b->s_ref_len.data(); info->ref_pos = b->s_ref_pos.data();
        info->rlen = b->s_rlen.data(); info->strand = b->s_strand.data(); info->seq_off = b->seq_off.data();
    }
    return SQG_OK;
}
extern "C" int sqg_fetch_reads(sqg_ctx_t* c, sqg_batch_t* b, char* dst) {
    if (!c || !b || !dst || b->seq_off.empty()) return SQG_EINVAL;
    for (int i = 0; i < b->n; i++) memcpy(dst + b->seq_off[(size_t)i], b->reads[(size_t)i].data(), b->reads[(size_t)i].size());
    return SQG_OK;
}

/* ---- not in the CPU backend ---- */
static int no(sqg_ctx_t* c, const char* what) { if (c) c->err = std::string(what) + ": not provided by the CPU backend"; return SQG_EINVAL; }
extern "C" int sqg_set_range_mode(sqg_ctx_t* c, int) { return no(c, "sqg_set_range_mode"); }
extern "C" int sqg_skip_reads(sqg_ctx_t* c, int32_t, const int64_t*, const int32_t*) { return no(c, "sqg_skip_reads"); }
extern "C" int sqg_batch_sample_range(sqg_ctx_t* c, int32_t, const int32_t*, int32_t, int32_t, sqg_batch_t**, sqg_sample_t*) { return no(c, "sqg_batch_sample_range"); }
extern "C" int sqg_batch_run_begin(sqg_ctx_t* c, sqg_batch_t*, const uint32_t**) { return no(c, "sqg_batch_run_begin"); }
extern "C" int sqg_batch_run_end(sqg_ctx_t* c, sqg_batch_t*, const uint32_t*, const uint32_t*) { return no(c, "sqg_batch_run_end"); }
extern "C" int sqg_probe_store_bandwidth(sqg_ctx_t* c, size_t, int, float*) { return no(c, "sqg_probe_store_bandwidth"); }
extern "C" int sqg_probe_lds_order(sqg_ctx_t* c, int, int, unsigned int*, int*) { return no(c, "sqg_probe_lds_order"); }

extern "C" void* sqg_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
extern "C" void sqg_host_free(void* p) { free(p); }

/* the BLOW5 writer: the product's own host code */
#include "../squigulator_amd/csrc/h_blow5.h"

/* the records of the writer's stored-block mode: the same host framing (blow5_record_stored), batch-wide */
extern "C" int sqg_batch_blow5_records(sqg_ctx_t* c, sqg_batch_t* b, const sqg_profile_t* profile, uint32_t flags, const char* read_ids,
                                       const int64_t* id_off, int64_t read_number0, uint64_t start_time0,
                                       const uint8_t** records, int64_t* n_bytes, const int64_t** rec_off) {
    if (!c || !b || !b->ran || !profile || !records || !n_bytes || (b->n > 0 && (!read_ids || !id_off))) return SQG_EINVAL;
    if (b->svb_off.empty()) { sqg_svb_t sv; const int rc = sqg_batch_compress(c, b, &sv); if (rc) return rc; }
    if (c->b5_reader_drain) c->b5_reader_drain(c->b5_reader, false);     // (ONE buffer here: a writer's background write of the previous call's records finishes first)
    sqg_blow5 w; w.profile = *profile; w.flags = flags;
    static thread_local std::vector<uint8_t> out, raw;
    static thread_local std::vector<int64_t> ro;
    out.clear(); ro.assign((size_t)b->n + 1, 0);
    for (int i = 0; i < b->n; i++) {
        if (id_off[i + 1] - id_off[i] > 4096) { c->err = "sqg_batch_blow5_records: read id longer than 4096 bytes"; return SQG_EINVAL; }
        blow5_record_stored(raw, out, &w, read_ids + id_off[i], (size_t)(id_off[i + 1] - id_off[i]), b->offset[(size_t)i], b->median[(size_t)i],
                            b->svb.data() + b->svb_off[(size_t)i], (uint64_t)(b->svb_off[(size_t)i + 1] - b->svb_off[(size_t)i]),
                            (int32_t)(read_number0 + i), start_time0 + (uint64_t)b->sig_off[(size_t)i]);
        ro[(size_t)i + 1] = (int64_t)out.size();
    }
    *records = out.data(); *n_bytes = (int64_t)out.size();
    if (rec_off) *rec_off = ro.data();
    return SQG_OK;
}
