/*
 * sqg_oracle.c -- CPU restatement of squigulator's per-read signal path.
 *
 * TEST INFRASTRUCTURE ONLY (see sqg_oracle.h).  Plain C99, built with
 * -O2 -std=c99 -ffp-contract=off like the reference (Makefile:4), so that
 * x*s+m is two roundings and libm's log/cos/sqrt are the host's.
 *
 * Parity pins (DESIGN.md "Oracle"): (1) every model-independent column of
 * the reference's own test/ goldens (tests/test_oracle_goldens.py); (2) the
 * reference's gensig.c/genread.c compiled where they lie and driven by
 * oracle/ref_harness.c, compared sample-for-sample on synthetic pore models
 * (tests/test_oracle_vs_ref.py + committed vectors in tests/golden/refvec).
 */
#define _XOPEN_SOURCE 700
#include "sqg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define LCG_M 2147483647LL
#define LCG_A 16807LL
#define LCG_Q 127773LL   /* M / A */
#define LCG_R 2836LL     /* M % A */

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "[oracle] out of memory (%zu bytes)\n", n); abort(); }
    return p;
}
static void *xrealloc(void *q, size_t n) {
    void *p = realloc(q, n ? n : 1);
    if (!p) { fprintf(stderr, "[oracle] out of memory (%zu bytes)\n", n); abort(); }
    return p;
}

/* ------------------------------------------------------------------ */
/* L1: PRNG and distributions                                          */
/* ------------------------------------------------------------------ */

/* src/rand.h:79-85.  MINSTD by Schrage's method.  The stored state is the
 * UNCORRECTED Schrage value (may be <= 0); the returned uniform uses the
 * corrected one.  C99 '%' and '/' truncate toward zero, which is what makes
 * the uncorrected carry-over equivalent to canonical MINSTD. */
double orc_rng(int64_t *xp) {
    const int64_t x = *xp;
    const int64_t nx = LCG_A * (x % LCG_Q) - LCG_R * (x / LCG_Q);
    const int64_t pos = nx > 0 ? nx : nx + LCG_M;
    *xp = nx;
    return (double)pos / 2147483647;
}

/* src/rand.h:87-94.  Box-Muller, cosine branch only, pi truncated to
 * 3.14159265; the re-draw loops on exact 0.0 are kept for fidelity. */
double orc_nrng(orc_norm_t *r) {
    double u = 0.0, t = 0.0;
    while (u == 0.0) u = orc_rng(&r->x);
    while (t == 0.0) t = 2.0 * 3.14159265 * orc_rng(&r->x);
    const double z = sqrt(-2.0 * log(u)) * cos(t);
    return (z * r->s) + r->m;
}

/* src/rand.h:96-102.  Erlang-k: the loop bound compares int i with double a */
double orc_grng(orc_gamma_t *r) {
    double acc = 0.0;
    for (int i = 0; i < r->a; i++) acc += -log(1 - orc_rng(&r->x));
    return acc * r->b;
}

/* src/seq.h:14-27 (IUPAC-tolerant 2-bit code; unknown -> 0 with a warning) */
uint32_t orc_base_rank(char b) {
    switch (b) {
    case 'A': case 'a': case 'R': case 'W': case 'M': case 'D': case 'H': case 'V': return 0;
    case 'C': case 'c': case 'Y': case 'B': return 1;
    case 'G': case 'g': case 'S': case 'K': return 2;
    case 'T': case 't': case 'U': return 3;
    default: return 0;
    }
}

/* src/seq.h:31-42: first base is the most significant digit */
uint32_t orc_kmer_rank(const char *s, uint32_t k) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < k; i++) r += orc_base_rank(s[k - 1 - i]) << (2 * i);
    return r;
}

/* src/seq.h:45-74: the 5-letter alphabet of the methylation models, A C G M T = 0..4 (upper case only; anything else
 * -> 0 with a warning); first base is the most significant base-5 digit */
static uint32_t meth_base_rank(char b) {
    switch (b) {
    case 'C': return 1;
    case 'G': return 2;
    case 'M': return 3;
    case 'T': return 4;
    default: return 0;
    }
}
uint32_t orc_meth_kmer_rank(const char *s, uint32_t k) {
    uint32_t p = 1, r = 0;
    for (uint32_t i = 0; i < k; i++) { r += meth_base_rank(s[k - i - 1]) * p; p *= 5; }
    return r;
}

/* src/seq.h:78-112: complement maps anything outside ACGTacgt to 'T' */
char *orc_revcomp(const char *f) {
    const size_t n = strlen(f);
    char *r = (char *)xmalloc(n + 1);
    for (size_t i = 0; i < n; i++) {
        char c = f[n - 1 - i], o;
        switch (c) {
        case 'A': case 'a': o = 'T'; break;
        case 'C': case 'c': o = 'G'; break;
        case 'G': case 'g': o = 'C'; break;
        case 'T': case 't': o = 'A'; break;
        default: o = 'T'; break;
        }
        r[i] = o;
    }
    r[n] = '\0';
    return r;
}

/* ------------------------------------------------------------------ */
/* model + genome loaders                                              */
/* ------------------------------------------------------------------ */

/* src/model.c:40-142.  '#k\t<k>' header is mandatory; data lines are parsed
 * with sscanf("%12s\t%f\t%f") exactly as the reference does. */
uint32_t orc_read_model(const char *path, orc_kmer_t **out) {
    FILE *fp = fopen(path, "r");
    if (!fp) return 0;
    uint32_t k = 0, num = 0, cap = 0, n = 0;
    orc_kmer_t *m = NULL;
    char *line = NULL; size_t lcap = 0;
    while (getline(&line, &lcap, fp) != -1) {
        if (line[0] == '#' || line[0] == '\n' || line[0] == '\r' || strncmp(line, "kmer\t", 5) == 0) {
            char key[1000]; int val = 0;
            if (sscanf(line, "%999s\t%d", key, &val) == 2 && strcmp(key, "#k") == 0) {
                if (val <= 0 || val > 9) { free(m); free(line); fclose(fp); return 0; }
                k = (uint32_t)val; num = 1u << (2 * k);
            }
            continue;
        }
        if (!k) { free(m); free(line); fclose(fp); return 0; }
        if (n == cap) { cap = cap ? cap * 2 : 4096; m = (orc_kmer_t *)xrealloc(m, cap * sizeof *m); }
        char kmer[16];
        if (sscanf(line, "%12s\t%f\t%f", kmer, &m[n].level_mean, &m[n].level_stdv) != 3 ||
            strlen(kmer) != k) { free(m); free(line); fclose(fp); return 0; }
        n++;
        if (n > num) { free(m); free(line); fclose(fp); return 0; }
    }
    free(line); fclose(fp);
    if (n != num) { free(m); return 0; }
    *out = m;
    return k;
}

/* src/ref.c:54-117 via kseq: record name = header up to first whitespace,
 * sequence = concatenation of the following lines.  Plain-text FASTA only. */
orc_ref_t *orc_ref_load(const char *fasta) {
    FILE *fp = fopen(fasta, "r");
    if (!fp) return NULL;
    orc_ref_t *ref = (orc_ref_t *)calloc(1, sizeof *ref);
    int cap = 0;
    char *line = NULL; size_t lcap = 0; ssize_t len;
    char *cur = NULL; size_t cur_n = 0, cur_cap = 0;
    while ((len = getline(&line, &lcap, fp)) != -1) {
        while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = '\0';
        if (line[0] == '>') {
            if (ref->num_ref > 0) {
                cur[cur_n] = '\0';
                ref->seqs[ref->num_ref - 1] = cur; ref->lengths[ref->num_ref - 1] = (int32_t)cur_n;
                ref->sum += (int64_t)cur_n;
            }
            if (ref->num_ref == cap) {
                cap = cap ? cap * 2 : 16;
                ref->names = (char **)xrealloc(ref->names, cap * sizeof(char *));
                ref->seqs = (char **)xrealloc(ref->seqs, cap * sizeof(char *));
                ref->lengths = (int32_t *)xrealloc(ref->lengths, cap * sizeof(int32_t));
            }
            size_t e = 1;
            while (line[e] && line[e] != ' ' && line[e] != '\t') e++;
            char *nm = (char *)xmalloc(e);
            memcpy(nm, line + 1, e - 1); nm[e - 1] = '\0';
            ref->names[ref->num_ref++] = nm;
            cur_cap = 1 << 16; cur = (char *)xmalloc(cur_cap); cur_n = 0;
        } else if (ref->num_ref > 0) {
            if (cur_n + (size_t)len + 1 > cur_cap) {
                while (cur_n + (size_t)len + 1 > cur_cap) cur_cap *= 2;
                cur = (char *)xrealloc(cur, cur_cap);
            }
            memcpy(cur + cur_n, line, (size_t)len); cur_n += (size_t)len;
        }
    }
    if (ref->num_ref > 0) {
        cur[cur_n] = '\0';
        ref->seqs[ref->num_ref - 1] = cur; ref->lengths[ref->num_ref - 1] = (int32_t)cur_n;
        ref->sum += (int64_t)cur_n;
    }
    free(line); fclose(fp);
    return ref;
}

/* src/ref.c:145-273: rows sorted by count descending with a stable merge
 * sort, then a float cumulative distribution of count/sum accumulated in
 * double. */
int orc_ref_load_trans_count(orc_ref_t *ref, const char *tsv) {
    FILE *fp = fopen(tsv, "r");
    if (!fp) return -1;
    int n = 0, cap = 0; char **nm = NULL; int32_t *ct = NULL;
    char *line = NULL; size_t lcap = 0;
    while (getline(&line, &lcap, fp) != -1) {
        if (line[0] == '#') continue;
        char *name = strtok(line, "\t");
        char *count = strtok(NULL, "\t");
        if (!name || !count) { fclose(fp); free(line); return -1; }
        if (n == cap) { cap = cap ? cap * 2 : 256; nm = (char **)xrealloc(nm, cap * sizeof *nm); ct = (int32_t *)xrealloc(ct, cap * sizeof *ct); }
        nm[n] = strdup(name); ct[n] = atoi(count); n++;
    }
    free(line); fclose(fp);
    /* stable insertion sort, descending by count */
    for (int i = 1; i < n; i++) {
        char *kn = nm[i]; int32_t kc = ct[i]; int j = i - 1;
        while (j >= 0 && ct[j] < kc) { nm[j + 1] = nm[j]; ct[j + 1] = ct[j]; j--; }
        nm[j + 1] = kn; ct[j + 1] = kc;
    }
    ref->trans_n = n;
    ref->trans_csum = (float *)xmalloc(n * sizeof(float));
    ref->trans_idx = (int32_t *)xmalloc(n * sizeof(int32_t));
    double total = 0, cum = 0;
    for (int i = 0; i < n; i++) total += ct[i];
    int rc = 0;
    for (int i = 0; i < n; i++) {
        cum += ct[i] / total;
        ref->trans_csum[i] = cum;
        int j = -1;
        for (int q = 0; q < ref->num_ref; q++) if (strcmp(ref->names[q], nm[i]) == 0) { j = q; break; }
        if (j < 0) rc = -2;
        ref->trans_idx[i] = j;
    }
    for (int i = 0; i < n; i++) free(nm[i]);
    free(nm); free(ct);
    return rc;
}

void orc_ref_free(orc_ref_t *ref) {
    if (!ref) return;
    for (int i = 0; i < ref->num_ref; i++) { free(ref->names[i]); free(ref->seqs[i]); }
    free(ref->names); free(ref->seqs); free(ref->lengths);
    free(ref->trans_csum); free(ref->trans_idx);
    if (ref->meth) { for (int i = 0; i < ref->num_ref; i++) free(ref->meth[i]); free(ref->meth); }
    free(ref);
}

/* src/ref.c:291-361: tab-separated [contig, 0-based position, frequency in 0..1]; the position must hold a C; the frequency is
 * kept as round(255*freq) in a per-contig byte array allocated on the contig's first line.  '#' lines are skipped.
 * Returns 0, or -(1-based line number) where the reference prints an error and exits. */
int orc_ref_load_meth_freq(orc_ref_t *ref, const char *tsv) {
    FILE *fp = fopen(tsv, "r");
    if (!fp) return -1000000000;
    ref->meth = (uint8_t **)calloc((size_t)ref->num_ref, sizeof(uint8_t *));
    char *line = NULL; size_t cap = 0; ssize_t len;
    int ln = 0, rc = 0;
    while ((len = getline(&line, &cap, fp)) != -1) {
        ln++;
        if (line[0] == '#') continue;
        char *name = strtok(line, "\t"), *pos_s = strtok(NULL, "\t"), *fr_s = strtok(NULL, "\t");
        if (!name || !pos_s || !fr_s) { rc = -ln; break; }
        int idx = -1;
        for (int i = 0; i < ref->num_ref; i++) if (!strcmp(ref->names[i], name)) { idx = i; break; }
        if (idx < 0) { rc = -ln; break; }
        const int32_t pos = atoi(pos_s);
        if (pos < 0 || pos >= ref->lengths[idx]) { rc = -ln; break; }
        const char b = ref->seqs[idx][pos];
        if (!(b == 'C' || b == 'c')) { rc = -ln; break; }
        const float freq = atof(fr_s);
        if (freq < 0 || freq > 1) { rc = -ln; break; }
        if (!ref->meth[idx]) ref->meth[idx] = (uint8_t *)calloc((size_t)ref->lengths[idx], 1);
        ref->meth[idx][pos] = (uint8_t)roundf(freq * 255);
    }
    free(line); fclose(fp);
    return rc;
}

/* ------------------------------------------------------------------ */
/* core: seeding layout                                                */
/* ------------------------------------------------------------------ */

/* src/sim.c:215-258.  Per worker, with s the worker's base seed:
 *   ref_pos=s  strand=s+1  time=s+2  rlen=s+3  offset=s+4  median=s+5
 *   meth=s+6   kmer_gen[j]=s+j  (j over all k-mers; overlaps the scalars)
 * and s advances by num_kmer+10 per worker.  nrng_t.s of a k-mer stream is
 * the FLOAT product level_stdv*amp_noise widened to double; the gamma scale
 * is the INTEGER quotient rlen/2. */
orc_core_t *orc_core_new(const orc_profile_t *p, uint32_t flags, float amp_noise,
                         uint32_t kmer_size, const orc_kmer_t *model,
                         int64_t seed, int32_t num_workers, int32_t rlen) {
    orc_core_t *c = (orc_core_t *)calloc(1, sizeof *c);
    c->prof = *p; c->flags = flags; c->amp_noise = amp_noise;
    c->kmer_size = kmer_size; c->num_kmer = 1u << (2 * kmer_size);
    if (flags & ORC_METH) { c->num_kmer = 1; for (uint32_t i = 0; i < kmer_size; i++) c->num_kmer *= 5; }   /* (uint32_t)pow(5,k), src/sim.c:325 */
    c->seed = seed; c->num_workers = num_workers; c->rlen = rlen;
    c->model = (orc_kmer_t *)xmalloc(c->num_kmer * sizeof(orc_kmer_t));
    memcpy(c->model, model, c->num_kmer * sizeof(orc_kmer_t));
    c->kmer_s = (double *)xmalloc(c->num_kmer * sizeof(double));
    for (uint32_t j = 0; j < c->num_kmer; j++) {
        float sd = model[j].level_stdv * amp_noise;
        c->kmer_s[j] = sd;
    }
    c->workers = (orc_worker_t *)calloc((size_t)num_workers, sizeof(orc_worker_t));
    int64_t s = seed;
    for (int t = 0; t < num_workers; t++) {
        orc_worker_t *w = &c->workers[t];
        w->pos_x = s;
        w->strand_x = s + 1;
        w->dwell = (orc_norm_t){p->dwell_mean, p->dwell_std, s + 2};
        w->rlen = (orc_gamma_t){2.0, (double)(rlen / 2), s + 3};
        w->offset = (orc_norm_t){p->offset_mean, p->offset_std, s + 4};
        w->median = (orc_norm_t){p->median_before_mean, p->median_before_std, s + 5};
        w->meth_x = s + 6;
        w->kmer_x = (int64_t *)xmalloc(c->num_kmer * sizeof(int64_t));
        for (uint32_t j = 0; j < c->num_kmer; j++) w->kmer_x[j] = s + j;
        s += (int64_t)c->num_kmer + 10;
    }
    return c;
}

void orc_core_free(orc_core_t *c) {
    if (!c) return;
    for (int t = 0; t < c->num_workers; t++) free(c->workers[t].kmer_x);
    free(c->workers); free(c->model); free(c->kmer_s); free(c);
}

/* ------------------------------------------------------------------ */
/* L2: signal generation                                               */
/* ------------------------------------------------------------------ */

typedef struct {
    int16_t *raw; int64_t n, cap;
    int32_t *ss; int64_t ss_n, ss_cap; int want_ss;
} sigbuf_t;

/* (int16_t)double as gcc/x86-64 lowers it for src/gensig.c:270: cvttsd2si to
 * a 32-bit integer (0x80000000 when out of range / NaN), then the low 16
 * bits. */
static inline int16_t dbl_to_i16(double v) {
    int32_t t;
    if (v > -2147483649.0 && v < 2147483648.0) t = (int32_t)v; else t = INT32_MIN;
    return (int16_t)(uint16_t)((uint32_t)t & 0xffffu);
}

/* src/gensig.c:226-288: the event/sample double loop */
static void emit_events(orc_core_t *c, sigbuf_t *b, double offset, const char *read, int32_t len, int tid) {
    const orc_profile_t *p = &c->prof;
    const uint32_t k = c->kmer_size;
    orc_worker_t *w = &c->workers[tid];
    const int ideal = (c->flags & ORC_IDEAL) != 0;
    const int ideal_time = (c->flags & ORC_IDEAL_TIME) != 0;
    const int ideal_amp = (c->flags & ORC_IDEAL_AMP) != 0;
    int sps = (int)p->dwell_mean;
    int64_t n_ev = (int64_t)len - k + 1;
    if (len < (int32_t)k) { n_ev = 5; read = "ACGTACGTACGT"; }       /* gensig.c:242-245 */
    for (int i = 0; i < n_ev; i++) {
        const uint32_t rank = (c->flags & ORC_METH) ? orc_meth_kmer_rank(read + i, k) : orc_kmer_rank(read + i, k);   /* gensig.c:250-253 */
        if (!(ideal || ideal_time)) {                                /* gensig.c:254-257 */
            sps = round(orc_nrng(&w->dwell));
            sps = sps < 1 ? -sps + 1 : sps;
        }
        for (int j = 0; j < sps; j++) {
            if (b->n == b->cap) { b->cap *= 2; b->raw = (int16_t *)xrealloc(b->raw, b->cap * sizeof(int16_t)); }
            float s;
            if (ideal || ideal_amp) {
                s = c->model[rank].level_mean;
            } else {                                                 /* gensig.c:268 */
                orc_norm_t g = {c->model[rank].level_mean, c->kmer_s[rank], w->kmer_x[rank]};
                s = orc_nrng(&g);
                w->kmer_x[rank] = g.x;
            }
            b->raw[b->n++] = dbl_to_i16(s * (p->digitisation) / (p->range) - (offset));
        }
        if (b->want_ss) {                                            /* gensig.c:273-281 */
            if (b->ss_n == b->ss_cap) { b->ss_cap *= 2; b->ss = (int32_t *)xrealloc(b->ss, b->ss_cap * sizeof(int32_t)); }
            b->ss[b->ss_n++] = sps >= 0 ? sps : 0;
        }
    }
}

#define POLYA_LEN 158                                            /* genread.c:37: 158 x 'A' */
static const char ADAPTOR_DNA[] = "GGCGTCTGCTTGGGTGTTTAACCTTTTTTTTTTAATGTACTTCGTTCAGTTACGTATTGCT"; /* genread.c:38 */
static const char ADAPTOR_RNA[] = "TGATGATGAGGGATAGACGATGGTTGTTTCTGTTGGTGCTGATATTGCTTTTTTTTTTTTTATGATGCAAGATACGCAC"; /* genread.c:39 */
static const char STALL_DNA[] = "TTTTTTTTTTTTTTTTTTAATCAA";       /* genread.c:110 */
static const char STALL_RNA[] = "AAAAAGAAAAAACCCCCCCCCCCCCCCCCC"; /* genread.c:87  */

/* src/gensig.c:293-356 with src/genread.c:71-123 folded in */
int16_t *orc_gen_sig(orc_core_t *c, const char *read, int32_t len,
                     double *offset, double *median_before, int64_t *len_raw_signal,
                     int tid, int32_t **ss, int64_t *ss_n) {
    const orc_profile_t *p = &c->prof;
    orc_worker_t *w = &c->workers[tid];
    const int rna = (c->flags & ORC_RNA) != 0;
    const int prefix = (c->flags & ORC_PREFIX) != 0;
    const int64_t n_est = len < (int32_t)c->kmer_size ? 1 : (int64_t)len - c->kmer_size + 1;

    sigbuf_t b;
    b.n = 0; b.cap = n_est * (int)p->dwell_mean + 2000;               /* gensig.c:305-309 */
    b.raw = (int16_t *)xmalloc(b.cap * sizeof(int16_t));
    b.want_ss = ss != NULL; b.ss_n = 0; b.ss_cap = 1000;
    b.ss = b.want_ss ? (int32_t *)xmalloc(b.ss_cap * sizeof(int32_t)) : NULL;

    if (c->flags & ORC_IDEAL) {                                      /* gensig.c:311-317 */
        *offset = p->offset_mean;
        *median_before = p->median_before_mean;
    } else {
        *offset = orc_nrng(&w->offset);
        *median_before = orc_nrng(&w->median);
    }

    char *tmp = NULL;
    if (prefix) {                                                    /* genread.c:95-123 */
        if (rna) {
            const int pa = POLYA_LEN, ad = (int)strlen(ADAPTOR_RNA);
            tmp = (char *)xmalloc((size_t)len + pa + ad + 1);
            memcpy(tmp, read, (size_t)len);
            memset(tmp + len, 'A', (size_t)pa);
            memcpy(tmp + len + pa, ADAPTOR_RNA, (size_t)ad);
            len += pa + ad;
        } else {
            const int st = (int)strlen(STALL_DNA), ad = (int)strlen(ADAPTOR_DNA);
            tmp = (char *)xmalloc((size_t)len + st + ad + 1);
            memcpy(tmp, STALL_DNA, (size_t)st);
            memcpy(tmp + st, ADAPTOR_DNA, (size_t)ad);
            memcpy(tmp + st + ad, read, (size_t)len);
            len += st + ad;
        }
        tmp[len] = '\0';
        read = tmp;
    }

    emit_events(c, &b, *offset, read, len, tid);

    if (prefix && rna) {                                             /* genread.c:71-93 */
        const int st = (int)(b.n - (int64_t)(strlen(ADAPTOR_RNA) * (int)p->dwell_mean));
        const int end = (int)b.n;
        const int16_t shift = dbl_to_i16(30 * (p->digitisation) / (p->range));
        for (int i = st; i < end; i++) {
            if (i >= 0) b.raw[i] = (int16_t)(uint16_t)((uint32_t)((int)b.raw[i] - (int)shift) & 0xffffu);
        }
        emit_events(c, &b, *offset, STALL_RNA, (int32_t)strlen(STALL_RNA), tid);
    }
    free(tmp);

    if (rna) {                                                       /* gensig.c:348-354 */
        for (int64_t i = 0; i < b.n / 2; i++) {
            int16_t t = b.raw[i]; b.raw[i] = b.raw[b.n - 1 - i]; b.raw[b.n - 1 - i] = t;
        }
    }
    *len_raw_signal = b.n;
    if (ss) { *ss = b.ss; *ss_n = b.ss_n; }
    return b.raw;
}

/* ------------------------------------------------------------------ */
/* feeder: read sampling                                               */
/* ------------------------------------------------------------------ */

/* src/genread.c:125-146: <200 nt -> -1; N -> base drawn from a FRESH LCG
 * seeded 100 on every call; more than 10% N -> reject with the count */
static int32_t screen_read(char *seq, int32_t len) {
    if (len < 200) return -1;
    int64_t r = 100;
    int nc = 0;
    for (int i = 0; i < len; i++) {
        if (seq[i] == 'N') {
            nc++;
            int v = round(orc_rng(&r) * 3);
            seq[i] = v == 0 ? 'A' : v == 1 ? 'C' : v == 2 ? 'G' : 'T';
            if (nc > 0.1 * len) return nc;
        }
    }
    return 0;
}

/* src/genread.c:149-177: copy at most len bases starting at pos (clipped by
 * the contig's terminating NUL) and screen them */
static char *cut_read(const orc_ref_t *ref, int len, int idx, int32_t pos, int32_t *rlen) {
    if (len < 0) len = 0;
    char *seq = (char *)xmalloc((size_t)len + 1);
    const char *src = ref->seqs[idx] + pos;
    int n = 0;
    while (n < len && src[n]) { seq[n] = src[n]; n++; }
    seq[n] = '\0';
    *rlen = n;
    if (screen_read(seq, n) != 0) { free(seq); return NULL; }
    return seq;
}

/* src/genread.c:179-194 */
static int pick_contig(orc_core_t *c, const orc_ref_t *ref, int tid, int32_t *gap) {
    int64_t at = round(orc_rng(&c->workers[tid].pos_x) * ref->sum);
    int64_t s = 0; int i;
    for (i = 0; i < ref->num_ref; i++) {
        s += ref->lengths[i];
        if (s >= at) { *gap = (int32_t)(at - s); break; }
    }
    return i;
}

/* src/genread.c:196-200: '+' when round(u) is 1 */
static char pick_strand(orc_core_t *c, int tid) {
    int64_t v = round(orc_rng(&c->workers[tid].strand_x));
    return v ? '+' : '-';
}

/* src/genread.c:207-241: every CpG of the read's span on the (forward) reference -- both bases upper case, both inside the
 * read -- takes one draw from the worker's rand_meth stream, whatever the outcome; the C becomes 'M' when
 * (int)(u*254) <= the position's frequency byte.  On the '-' strand the CpG sits at rlen-i-2 of the reverse complement.
 * Contigs without any line in the frequency file take no draws. */
static void methylate(orc_core_t *c, const orc_ref_t *ref, int idx, int32_t ref_len, int32_t ref_pos, int32_t rlen, char strand,
                      char *seq, int tid) {
    if (!ref->meth || !ref->meth[idx]) return;
    const char *g = ref->seqs[idx];
    for (int i = 0; i < rlen; i++) {
        if (ref_pos + i + 1 < ref_len && i + 1 < rlen && g[ref_pos + i] == 'C' && g[ref_pos + i + 1] == 'G') {
            const int methr = orc_rng(&c->workers[tid].meth_x) * 254;
            if (methr <= ref->meth[idx][ref_pos + i]) seq[strand == '-' ? rlen - i - 2 : i] = 'M';
        }
    }
}

/* src/genread.c:243-281 */
static char *sample_dna(orc_core_t *c, const orc_ref_t *ref, int tid, int32_t *ref_idx,
                        int32_t *ref_len, int32_t *ref_pos, int32_t *rlen, char *strand) {
    char *seq;
    for (;;) {
        int len = orc_grng(&c->workers[tid].rlen);
        int32_t gap = 0;
        int idx = pick_contig(c, ref, tid, &gap);
        *ref_idx = idx;
        *ref_pos = gap + ref->lengths[idx];
        *ref_len = ref->lengths[idx];
        *strand = pick_strand(c, tid);
        seq = cut_read(ref, len, idx, *ref_pos, rlen);
        if (seq) break;
    }
    if (*strand == '-') {
        char *r = orc_revcomp(seq);
        free(seq); seq = r;
    }
    if (c->flags & ORC_METH) methylate(c, ref, *ref_idx, *ref_len, *ref_pos, *rlen, *strand, seq, tid);   /* genread.c:276-278 */
    return seq;
}

/* src/genread.c:283-300: uniform over transcripts, or by the abundance CDF;
 * note the uniform is narrowed to float before the comparison */
static int pick_transcript(orc_core_t *c, const orc_ref_t *ref, int tid) {
    if (ref->trans_n == 0) return (int)round(orc_rng(&c->workers[tid].pos_x) * (ref->num_ref - 1));
    float r = orc_rng(&c->workers[tid].pos_x);
    for (int i = 0; i < ref->trans_n; i++) if (r <= ref->trans_csum[i]) return ref->trans_idx[i];
    return 0;
}

/* src/genread.c:311-355 */
static char *sample_rna(orc_core_t *c, const orc_ref_t *ref, int tid, int cdna, int32_t *ref_idx,
                        int32_t *ref_len, int32_t *ref_pos, int32_t *rlen, char *strand) {
    char *seq;
    for (;;) {
        int idx = pick_transcript(c, ref, tid);
        *ref_idx = idx;
        int len = ref->lengths[idx];
        *ref_pos = 0;
        if (c->flags & ORC_TRANS_TRUNC) {                            /* genread.c:303-309 */
            double frac = orc_grng(&c->workers[tid].rlen) / (double)(c->rlen);
            int tl = frac * len;
            tl = tl > len ? len : tl;
            *ref_pos = ref->lengths[idx] - tl;
            len = tl;
        }
        *ref_len = len;
        *strand = cdna ? pick_strand(c, tid) : '+';
        seq = cut_read(ref, len, idx, *ref_pos, rlen);
        if (seq) break;
    }
    if (cdna && *strand == '-') {
        char *r = orc_revcomp(seq);
        free(seq); seq = r;
    }
    return seq;
}

/* src/genread.c:358-370 */
char *orc_gen_read(orc_core_t *c, const orc_ref_t *ref, int tid, int32_t *ref_idx,
                   int32_t *ref_len, int32_t *ref_pos, int32_t *rlen, char *strand) {
    if (c->flags & ORC_RNA) return sample_rna(c, ref, tid, 0, ref_idx, ref_len, ref_pos, rlen, strand);
    if (c->flags & ORC_CDNA) return sample_rna(c, ref, tid, 1, ref_idx, ref_len, ref_pos, rlen, strand);
    return sample_dna(c, ref, tid, ref_idx, ref_len, ref_pos, rlen, strand);
}

/* ------------------------------------------------------------------ */
/* batch driver                                                        */
/* ------------------------------------------------------------------ */

/* src/thread.c:80-99 (static partition, step = ceil(n_rec/T)) and :122-125
 * (-t1 runs everything on tid 0).  Work stealing (thread.c:22-40,64-66) is
 * non-deterministic in the reference and never fires when T==1 or T>=n_rec. */
int32_t orc_worker_of(int32_t i, int32_t n_rec, int32_t T) {
    if (T <= 1) return 0;
    int32_t step = (n_rec + T - 1) / T;
    return i / step;
}

typedef struct {
    orc_core_t *c; const orc_ref_t *ref; orc_batch_t *b;
    const char *const *seqs; const int32_t *lens;
    int want_ss; int32_t n_rec;
    int32_t next_worker; pthread_mutex_t mu;
} job_t;

/* src/sim.c:514-563: one read on worker tid */
static void one_read(job_t *j, int32_t i, int tid) {
    orc_core_t *c = j->c;
    orc_read_t *r = &j->b->reads[i];
    r->tid = tid;
    r->read_number = c->total_reads + i;
    if (j->seqs) {
        r->rlen = j->lens[i];
        r->seq = (char *)xmalloc((size_t)r->rlen + 1);
        memcpy(r->seq, j->seqs[i], (size_t)r->rlen); r->seq[r->rlen] = '\0';
        r->ref_idx = -1; r->ref_len = r->rlen; r->ref_pos_st = 0; r->strand = '+';
    } else if (c->flags & ORC_FULL_CONTIG) {                         /* sim.c:542-548 */
        int32_t q = (int32_t)(c->total_reads + i);
        r->ref_idx = q; r->rlen = j->ref->lengths[q]; r->ref_len = r->rlen;
        r->seq = strdup(j->ref->seqs[q]); r->strand = '+'; r->ref_pos_st = 0;
    } else {
        r->seq = orc_gen_read(c, j->ref, tid, &r->ref_idx, &r->ref_len, &r->ref_pos_st, &r->rlen, &r->strand);
    }
    r->ref_pos_end = r->ref_pos_st + r->rlen;
    r->raw_signal = orc_gen_sig(c, r->seq, r->rlen, &r->offset, &r->median_before, &r->len_raw_signal,
                                tid, j->want_ss ? &r->ss : NULL, &r->ss_n);
}

static void *worker_main(void *arg) {
    job_t *j = (job_t *)arg;
    const int32_t T = j->c->num_workers;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int32_t t = j->next_worker++;
        pthread_mutex_unlock(&j->mu);
        if (t >= T) break;
        const int64_t step = ((int64_t)j->n_rec + T - 1) / T;
        const int64_t lo = (int64_t)t * step;
        const int64_t hi = lo + step < j->n_rec ? lo + step : j->n_rec;
        for (int64_t i = lo; i < hi; i++) one_read(j, (int32_t)i, t);
    }
    return NULL;
}

static orc_batch_t *run_batch(orc_core_t *c, const orc_ref_t *ref, int32_t n_rec,
                              const char *const *seqs, const int32_t *lens, int want_ss, int nthreads) {
    orc_batch_t *b = (orc_batch_t *)calloc(1, sizeof *b);
    b->n = n_rec;
    b->reads = (orc_read_t *)calloc((size_t)(n_rec > 0 ? n_rec : 1), sizeof(orc_read_t));
    job_t j = {c, ref, b, seqs, lens, want_ss, n_rec, 0, PTHREAD_MUTEX_INITIALIZER};
    const int32_t T = c->num_workers;
    if (nthreads <= 1 || T == 1) {
        /* virtual workers in turn; each worker handles its own reads in index order */
        if (T == 1) { for (int32_t i = 0; i < n_rec; i++) one_read(&j, i, 0); }
        else worker_main(&j);
    } else {
        pthread_t th[256];
        if (nthreads > 256) nthreads = 256;
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, worker_main, &j);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    /* start_time (src/sim.c:602): the running sample count.  Sequential at
     * -t1; racy otherwise in the reference -- here defined as the prefix sum in
     * read-index order. */
    for (int32_t i = 0; i < n_rec; i++) {
        b->reads[i].start_time = c->n_samples;
        c->n_samples += b->reads[i].len_raw_signal;
    }
    c->total_reads += n_rec;                                         /* sim.c:653 */
    return b;
}

orc_batch_t *orc_batch_run(orc_core_t *c, const orc_ref_t *ref, int32_t n_rec, int want_ss, int nthreads) {
    return run_batch(c, ref, n_rec, NULL, NULL, want_ss, nthreads);
}

orc_batch_t *orc_batch_run_seqs(orc_core_t *c, int32_t n_rec, const char *const *seqs,
                                const int32_t *lens, int want_ss, int nthreads) {
    return run_batch(c, NULL, n_rec, seqs, lens, want_ss, nthreads);
}

orc_batch_t *orc_batch_run_assigned(orc_core_t *c, int32_t n_rec, const char *const *seqs,
                                    const int32_t *lens, const int32_t *workers, int want_ss) {
    orc_batch_t *b = (orc_batch_t *)calloc(1, sizeof *b);
    b->n = n_rec;
    b->reads = (orc_read_t *)calloc((size_t)(n_rec > 0 ? n_rec : 1), sizeof(orc_read_t));
    job_t j = {c, NULL, b, seqs, lens, want_ss, n_rec, 0, PTHREAD_MUTEX_INITIALIZER};
    for (int32_t i = 0; i < n_rec; i++) one_read(&j, i, workers[i]);   /* index order == per-worker order */
    for (int32_t i = 0; i < n_rec; i++) {
        b->reads[i].start_time = c->n_samples;
        c->n_samples += b->reads[i].len_raw_signal;
    }
    c->total_reads += n_rec;
    return b;
}

void orc_batch_free(orc_batch_t *b) {
    if (!b) return;
    for (int32_t i = 0; i < b->n; i++) { free(b->reads[i].seq); free(b->reads[i].raw_signal); free(b->reads[i].ss); }
    free(b->reads); free(b);
}

/* ------------------------------------------------------------------ */
/* svb-zd (slow5lib's signal compression), "next" row                   */
/* ------------------------------------------------------------------ */

/* streamvbyte.h: __slow5_streamvbyte_max_compressedbytes = key bytes + 4 per value; + the count word of
 * slow5_press.c:1040 */
size_t orc_svb_zd_bound(int64_t n) {
    return sizeof(uint32_t) + (size_t)((n + 3) / 4) + 4 * (size_t)n;
}

/* slow5_press.c:1055-1087 -> streamvbyte_zigzag.c (delta, zig-zag) -> streamvbyte_encode.c (scalar coder;
 * the SSSE3/NEON coders produce the same bytes) */
size_t orc_svb_zd(const int16_t *sig, int64_t n, uint8_t *out) {
    const uint32_t count = (uint32_t)n;
    memcpy(out, &count, sizeof count);                               /* slow5_press.c:1047 */
    uint8_t *key = out + sizeof count;
    uint8_t *data = key + (count + 3) / 4;
    int32_t prev = 0;
    uint8_t kb = 0;
    for (uint32_t i = 0; i < count; i++) {
        const int32_t v = (int32_t)sig[i] - prev;                    /* zigzag_delta_encode */
        prev = sig[i];
        const uint32_t z = ((uint32_t)v + (uint32_t)v) ^ (uint32_t)(v >> 31);
        const unsigned code = z < (1u << 8) ? 0 : z < (1u << 16) ? 1 : z < (1u << 24) ? 2 : 3;
        for (unsigned b = 0; b <= code; b++) *data++ = (uint8_t)(z >> (8 * b));    /* little endian */
        kb |= (uint8_t)(code << (2 * (i & 3)));
        if ((i & 3) == 3) { *key++ = kb; kb = 0; }
    }
    if (count & 3) *key = kb;                                        /* the last, partial key byte */
    return (size_t)(data - out);
}
