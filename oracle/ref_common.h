/*
 * ref_common.h -- what the two programs that drive the REFERENCE's own translation units share (ref_harness.c: the reference's
 * per-read path as it is; ref_host_gpu.c: the same program with process_db() replaced by the C ABI of include/sqg.h).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it contains no reference code.  It is included after the reference's headers
 * (sq.h, format.h, error.h) and holds: the key=value configuration both programs take, the f5c-format pore-table reader, and the
 * per-worker stream seeding of src/sim.c:238-257 (src/sim.c itself cannot be linked: src/model.c needs the absent src/model.h).
 */
#ifndef SQG_REF_COMMON_H
#define SQG_REF_COMMON_H

typedef struct {
    char fasta[4096], model[4096], out[4096], slow5[4096], fasta_out[4096], paf[4096], sam[4096], trans_count[4096];
    char svb_out[4096];      /* per read: int64 nbytes + slow5lib's own svb-zd encoding of the raw signal */
    char meth_freq[4096];    /* --meth-freq: CpG methylation; `model` is then the 5-letter table (5^k rows, src/sim.c:297-326) */
    char svb_in[4096];       /* stand-alone mode: int32 n, then n x (int64 len, int16[len]) -> svb_out in the same framing */
    profile_t p;
    uint32_t flags;
    float amp_noise;
    long seed, threads, batch, nreads, rlen;
    long device, exact;      /* ref_host_gpu only: the GPU, and mode=exact (every draw in FP64) instead of the certified fp32 path */
    double time_s;           /* > 0: timed mode -- reads are generated until this many seconds have gone into the read loop
                                (model/FASTA load and stream seeding excluded); one "SQGTIME ..." line on stdout */
} cfg_t;

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static void parse_cfg(const char *path, cfg_t *c) {
    memset(c, 0, sizeof *c);
    c->amp_noise = 1; c->threads = 1; c->batch = 1000; c->nreads = 1; c->rlen = 10000;
    FILE *fp = fopen(path, "r");
    if (!fp) { perror(path); exit(2); }
    char line[8192];
    while (fgets(line, sizeof line, fp)) {
        char *eq = strchr(line, '=');
        if (!eq || line[0] == '#') continue;
        *eq = '\0';
        char *v = eq + 1;
        v[strcspn(v, "\r\n")] = '\0';
        const char *k = line;
#define STR(name) if (!strcmp(k, #name)) { strncpy(c->name, v, sizeof c->name - 1); continue; }
#define DBL(name) if (!strcmp(k, #name)) { c->p.name = strtod(v, NULL); continue; }
        STR(fasta) STR(model) STR(out) STR(slow5) STR(fasta_out) STR(paf) STR(sam) STR(trans_count) STR(svb_out) STR(svb_in) STR(meth_freq)
        DBL(digitisation) DBL(sample_rate) DBL(bps) DBL(range) DBL(offset_mean) DBL(offset_std)
        DBL(median_before_mean) DBL(median_before_std) DBL(dwell_mean) DBL(dwell_std)
        if (!strcmp(k, "flags")) { c->flags = (uint32_t)strtoul(v, NULL, 0); continue; }
        if (!strcmp(k, "amp_noise")) { c->amp_noise = strtof(v, NULL); continue; }
        if (!strcmp(k, "seed")) { c->seed = atol(v); continue; }
        if (!strcmp(k, "threads")) { c->threads = atol(v); continue; }
        if (!strcmp(k, "batch")) { c->batch = atol(v); continue; }
        if (!strcmp(k, "nreads")) { c->nreads = atol(v); continue; }
        if (!strcmp(k, "rlen")) { c->rlen = atol(v); continue; }
        if (!strcmp(k, "time_s")) { c->time_s = strtod(v, NULL); continue; }
        if (!strcmp(k, "device")) { c->device = atol(v); continue; }
        if (!strcmp(k, "mode")) { c->exact = !strcmp(v, "exact"); continue; }
        fprintf(stderr, "ref_harness: unknown key '%s'\n", k); exit(2);
    }
    fclose(fp);
}

/* f5c-format text table -> model_t[]; the same "%f" conversions the
 * reference's reader applies (src/model.c:101-102) */
static uint32_t load_table(const char *path, model_t *m, int meth) {
    FILE *fp = fopen(path, "r");
    if (!fp) { perror(path); exit(2); }
    char line[512], kmer[32];
    uint32_t k = 0, n = 0;
    while (fgets(line, sizeof line, fp)) {
        if (line[0] == '#') { int v; if (sscanf(line, "#k\t%d", &v) == 1) k = (uint32_t)v; continue; }
        if (!strncmp(line, "kmer", 4) || line[0] == '\n') continue;
        if (sscanf(line, "%12s\t%f\t%f", kmer, &m[n].level_mean, &m[n].level_stdv) != 3) { fprintf(stderr, "bad model line\n"); exit(2); }
        n++;
    }
    fclose(fp);
    uint32_t want = 1u << (2 * k);
    if (meth) { want = 1; for (uint32_t i = 0; i < k; i++) want *= 5; }      /* rows in file order, src/model.c:100-120 */
    if (!k || n != want) { fprintf(stderr, "bad model file (k=%u, n=%u)\n", k, n); exit(2); }
    return k;
}

static void seed_workers(core_t *core) {
    const int T = core->opt.num_thread;
    const uint32_t nk = core->num_kmer;
    const profile_t p = core->profile;
    core->ref_pos = malloc(T * sizeof(int64_t));
    core->rand_strand = malloc(T * sizeof(int64_t));
    core->rand_time = malloc(T * sizeof(nrng_t *));
    core->rand_rlen = malloc(T * sizeof(grng_t *));
    core->rand_offset = malloc(T * sizeof(nrng_t *));
    core->rand_median_before = malloc(T * sizeof(nrng_t *));
    core->kmer_gen = malloc(T * sizeof(nrng_t **));
    const model_t *m = core->opt.meth_freq ? core->cpgmodel : core->model;      /* src/sim.c:231-236 */
    core->rand_meth = core->opt.meth_freq ? malloc(T * sizeof(int64_t)) : NULL;
    int64_t s = core->opt.seed;
    for (int t = 0; t < T; t++, s += nk + 10) {
        core->ref_pos[t] = s;
        core->rand_strand[t] = s + 1;
        core->rand_time[t] = init_nrng(s + 2, p.dwell_mean, p.dwell_std);
        core->rand_rlen[t] = init_grng(s + 3, 2.0, core->opt.rlen / 2);
        core->rand_offset[t] = init_nrng(s + 4, p.offset_mean, p.offset_std);
        core->rand_median_before[t] = init_nrng(s + 5, p.median_before_mean, p.median_before_std);
        core->kmer_gen[t] = malloc(nk * sizeof(nrng_t *));
        for (uint32_t j = 0; j < nk; j++)
            core->kmer_gen[t][j] = init_nrng(s + j, m[j].level_mean, m[j].level_stdv * core->opt.amp_noise);
        if (core->rand_meth) core->rand_meth[t] = s + 6;                          /* src/sim.c:252-254 */
    }
}

static void put(FILE *fp, const void *p, size_t n) { if (n && fwrite(p, 1, n, fp) != n) { perror("fwrite"); exit(2); } }

#endif
