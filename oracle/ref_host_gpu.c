/*
 * ref_host_gpu.c -- the drop-in, compiled: the REFERENCE's own host code (src/genread.c, src/ref.c, src/format.c, src/gensig.c's
 * record / header helpers, slow5lib -- built from the sources where they lie under /root/reference, see oracle/Makefile) with
 * process_db() (src/sim.c:622-627 -> work_db -> work_per_single_read, src/sim.c:514-618) replaced by process_db_gpu() below, which
 * makes every signal of a batch through the C ABI of include/sqg.h (libsqg_hip.so).  INTEGRATION.md quotes init_sqg() and
 * process_db_gpu() from this file: they are the binding a maintainer would add to src/sim.c.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it contains no reference code: it includes the reference's headers at build time
 * and calls its functions (gen_read, paf_str, sam_str, set_record_*_fields, slow5_encode, slow5_write_bytes ...).  The reference's
 * driver (src/sim.c) cannot be linked -- src/model.c needs the absent src/model.h -- so main() here does what sim_main does around
 * the batch loop (src/sim.c:1065-1075): fill a core_t from a configuration file (ref_common.h), open the outputs, and per batch
 * process_db_gpu() + the output loop of output_db (src/sim.c:630-656).  gen_sig() is never called.
 *
 * tests/test_dropin.py: the SLOW5 / BLOW5, FASTA, PAF and SAM files this program writes are `cmp`-identical to those
 * oracle/_ref/ref_harness (the reference's own gen_sig) writes from the same configuration.
 *
 * usage: ref_host_gpu <config-file>     (the key=value file ref_harness takes; + device=, mode=exact|certified)
 */
#define _XOPEN_SOURCE 700
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>

#include "sq.h"      /* core_t, db_t, opt_t, profile_t, model_t, SQ_* flags */
#include "format.h"  /* aln_t, init_aln, paf_str, sam_str, sam_hdr_wr */
#include "error.h"

#include "sqg.h"     /* -I <this repo>/include, link -lsqg_hip */

/* entry points of the reference this program keeps using (src/genread.c, src/gensig.c, src/ref.c) */
char *gen_read(core_t *core, char **ref_id, int32_t *ref_len, int32_t *ref_pos, int32_t *rlen, char *c, int8_t rna, int tid);
void set_header_attributes(slow5_file_t *sp, int8_t rna, int8_t r10, double sample_frequency);
void set_header_aux_fields(slow5_file_t *sp, int8_t ont_friendly);
void set_record_primary_fields(profile_t *profile, slow5_rec_t *rec, char *read_id, double offset, int64_t len_raw_signal, int16_t *raw_signal);
void set_record_aux_fields(slow5_rec_t *rec, slow5_file_t *sp, double median_before, int32_t read_number, uint64_t start_time, int8_t ont_friendly);
void load_meth_freq(const char *meth_freq, ref_t *ref);        /* src/ref.c:291 */

#include "ref_common.h"   /* cfg_t, parse_cfg, load_table, seed_workers */

/* ===================== the binding (what a maintainer adds to src/sim.c) ===================== */

static sqg_ctx_t *g_sqg;               /* created at the end of init_core() (src/sim.c:262-360), destroyed in free_core() */

static void init_sqg(core_t *core, int device, uint32_t mode) {
    sqg_cfg_t cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SQG_ABI_VERSION;
    memcpy(&cfg.profile, &core->profile, sizeof cfg.profile);     /* profile_t: the same 10 doubles in the same order (src/sq.h:47-58) */
    cfg.flags = core->opt.flag & (SQ_RNA | SQ_IDEAL | SQ_IDEAL_TIME | SQ_IDEAL_AMP | SQ_PREFIX);   /* same bit values (include/sqg.h) */
    cfg.amp_noise = core->opt.amp_noise;
    cfg.kmer_size = core->kmer_size;
    cfg.model = (const sqg_kmer_t *)core->model;                  /* model_t == {float level_mean, level_stdv} (src/sq.h:61-68) */
    if (core->opt.meth_freq) {                                    /* --meth-freq: the 5-letter table, 5^k rows (src/sim.c:296-326) */
        cfg.flags |= SQG_METH;
        cfg.model = (const sqg_kmer_t *)core->cpgmodel;
    }
    cfg.seed = core->opt.seed;
    cfg.num_workers = core->opt.num_thread;                       /* -t: the streams are seeded per worker as init_rand does (src/sim.c:238-257) */
    cfg.worker_lo = 0; cfg.worker_hi = core->opt.num_thread;
    cfg.device = device; cfg.mode = mode;
    const int rc = sqg_create(&cfg, &g_sqg);
    if (rc) { ERROR("sqg_create: %s", sqg_strerror(rc)); exit(EXIT_FAILURE); }
}

typedef struct { char *rid, *seq, strand; int32_t ref_len, pos_st, rlen; } read_meta_t;

/* replaces process_db() (src/sim.c:622-627): the reads are sampled on the host by the reference's own gen_read(), every signal of
 * the batch is generated on the GPU, the records are encoded by the reference's own slow5lib / format.c */
static void process_db_gpu(core_t *core, db_t *db) {
    const int n = db->n_rec, T = core->opt.num_thread;
    const opt_t opt = core->opt;
    const int8_t rna = opt.flag & SQ_RNA ? 1 : 0;
    read_meta_t *m = (read_meta_t *)calloc((size_t)n, sizeof *m);
    int64_t *off = (int64_t *)malloc(((size_t)n + 1) * sizeof *off);
    MALLOC_CHK(m); MALLOC_CHK(off);
    off[0] = 0;
    for (int i = 0; i < n; i++) {                   /* a worker's reads in index order: the order its streams are drawn in (src/thread.c:80-99) */
        const int tid = sqg_worker_of(i, n, T);
        if (opt.flag & SQ_FULL_CONTIG) {            /* src/sim.c:544-550 */
            m[i].rid = core->ref->ref_names[core->total_reads + i]; m[i].rlen = core->ref->ref_lengths[core->total_reads + i];
            m[i].seq = core->ref->ref_seq[core->total_reads + i]; m[i].strand = '+'; m[i].pos_st = 0;
        } else {
            m[i].strand = '+';
            m[i].seq = gen_read(core, &m[i].rid, &m[i].ref_len, &m[i].pos_st, &m[i].rlen, &m[i].strand, rna, tid);
        }
        off[i + 1] = off[i] + m[i].rlen;
    }
    char *blob = (char *)malloc((size_t)off[n] + 1);
    MALLOC_CHK(blob);
    for (int i = 0; i < n; i++) memcpy(blob + off[i], m[i].seq, (size_t)m[i].rlen);

    sqg_batch_t *b; sqg_result_t res;
    int rc = sqg_submit(g_sqg, n, blob, off, NULL /* the reference's static partition over -t */, &b, &res);
    if (rc) { ERROR("sqg_submit: %s (%s)", sqg_strerror(rc), sqg_last_error(g_sqg)); exit(EXIT_FAILURE); }
    int16_t *sig = (int16_t *)malloc((size_t)(res.n_samples > 0 ? res.n_samples : 1) * sizeof *sig);
    MALLOC_CHK(sig);
    if ((rc = sqg_fetch_signal(g_sqg, b, sig))) { ERROR("sqg_fetch_signal: %s", sqg_strerror(rc)); exit(EXIT_FAILURE); }
    int32_t *ss = NULL;
    if (core->fp_paf || core->fp_sam) {             /* aln->ss (src/gensig.c:273-281): one dwell per event, prefix and stall events included */
        ss = (int32_t *)malloc((size_t)(res.n_events > 0 ? res.n_events : 1) * sizeof *ss);
        MALLOC_CHK(ss);
        if ((rc = sqg_fetch_dwell(g_sqg, b, ss))) { ERROR("sqg_fetch_dwell: %s", sqg_strerror(rc)); exit(EXIT_FAILURE); }
    }

    for (int i = 0; i < n; i++) {                   /* the tail of work_per_single_read, src/sim.c:557-617, with gen_sig()'s outputs taken from the batch */
        const int64_t len_raw_signal = res.sig_off[i + 1] - res.sig_off[i];
        const int32_t rlen = m[i].rlen, ref_pos_st = m[i].pos_st, ref_pos_end = m[i].pos_st + m[i].rlen;
        assert(len_raw_signal > 0);
        int16_t *raw_signal = (int16_t *)malloc((size_t)len_raw_signal * sizeof *raw_signal);   /* ownership passes to the slow5 record (src/gensig.c:171-183) */
        MALLOC_CHK(raw_signal);
        memcpy(raw_signal, sig + res.sig_off[i], (size_t)len_raw_signal * sizeof *raw_signal);
        char *read_id = (char *)malloc(10000);
        MALLOC_CHK(read_id);
        if (opt.flag & SQ_ONT) sprintf(read_id, "00000000-0000-0000-0000-%012d", (int)(core->total_reads + i + 1));
        else sprintf(read_id, "S1_%ld!%s!%d!%d!%c", (long)(core->total_reads + i + 1), m[i].rid, ref_pos_st, ref_pos_end, m[i].strand);
        if (core->fp_fasta) {
            db->fasta[i] = (char *)malloc(strlen(read_id) + strlen(m[i].seq) + 10);
            MALLOC_CHK(db->fasta[i]);
            sprintf(db->fasta[i], ">%s\n%s\n", read_id, m[i].seq);
        }
        if (core->fp_paf || core->fp_sam) {
            aln_t *aln = init_aln();
            const int64_t n_ev = res.ev_off[i + 1] - res.ev_off[i], n_kmer = rlen - core->kmer_size + 1;
            free(aln->ss);                          /* the batch's dwell slice in place of the array gen_sig() grows */
            aln->ss = (int32_t *)malloc((size_t)(n_ev > 0 ? n_ev : 1) * sizeof(int32_t));
            MALLOC_CHK(aln->ss);
            memcpy(aln->ss, ss + res.ev_off[i], (size_t)n_ev * sizeof(int32_t));
            aln->ss_n = aln->ss_c = n_ev;
            aln->sig_start = 0; aln->sig_end = len_raw_signal;    /* src/gensig.c:247,284 */
            aln->read_id = read_id; aln->len_raw_signal = len_raw_signal; aln->strand = m[i].strand;
            aln->si_st_ref = rna ? ref_pos_end - core->kmer_size + 1 : ref_pos_st;
            aln->si_end_ref = rna ? ref_pos_st : ref_pos_end - core->kmer_size + 1;
            if (opt.flag & SQ_PAF_REF) {
                aln->tid = m[i].rid;
                aln->tlen = !(opt.flag & SQ_FULL_CONTIG) ? m[i].ref_len - core->kmer_size + 1 : n_kmer;
                aln->t_st = aln->si_st_ref; aln->t_end = aln->si_end_ref;
            } else {
                aln->tid = read_id; aln->tlen = n_kmer;
                aln->t_st = rna ? n_kmer : 0; aln->t_end = rna ? 0 : n_kmer;
            }
            if (core->fp_paf) db->paf[i] = paf_str(aln);
            if (core->fp_sam) db->sam[i] = sam_str(aln, m[i].seq, m[i].rid, ref_pos_st);
            free_aln(aln);
        }
        const int64_t start_time = core->n_samples;             /* (src/sim.c:602: the fetch-add, in read order here) */
        core->n_samples += len_raw_signal;
        slow5_rec_t *rec = slow5_rec_init();
        set_record_primary_fields(&core->profile, rec, read_id, res.offset[i], len_raw_signal, raw_signal);
        set_record_aux_fields(rec, core->sp, res.median_before[i], (int32_t)(core->total_reads + i), (uint64_t)start_time, opt.flag & SQ_ONT ? 1 : 0);
        if (core->sp && slow5_encode(&db->mem_records[i], &db->mem_bytes[i], rec, core->sp) < 0) { ERROR("%s", "Error encoding record"); exit(EXIT_FAILURE); }
        if (!(opt.flag & SQ_FULL_CONTIG)) free(m[i].seq);
        slow5_rec_free(rec);                        /* frees raw_signal and read_id */
    }
    sqg_batch_free(g_sqg, b);
    free(sig); free(ss); free(blob); free(off); free(m);
}

/* ===================== what sim_main does around it (src/sim.c:1010-1080), restated ===================== */

int main(int argc, char **argv) {
    if (argc != 2) { fprintf(stderr, "usage: %s <config>\n", argv[0]); return 2; }
    cfg_t cfg; parse_cfg(argv[1], &cfg);
    set_log_level(LOG_ERR);

    core_t *core = calloc(1, sizeof *core);
    core->opt.rlen = (int32_t)cfg.rlen; core->opt.seed = cfg.seed; core->opt.flag = cfg.flags;
    core->opt.num_thread = (int32_t)cfg.threads; core->opt.batch_size = (int32_t)cfg.batch;
    core->opt.amp_noise = cfg.amp_noise;
    core->profile = cfg.p;
    core->model = malloc(sizeof(model_t) * MAX_NUM_KMER);
    if (cfg.meth_freq[0]) {
        core->opt.meth_freq = cfg.meth_freq;
        core->cpgmodel = malloc(sizeof(model_t) * MAX_NUM_KMER_METH);
        core->kmer_size = load_table(cfg.model, core->cpgmodel, 1);
        core->num_kmer = (uint32_t)pow(5, core->kmer_size);
    } else {
        core->kmer_size = load_table(cfg.model, core->model, 0);
        core->num_kmer = 1u << (2 * core->kmer_size);
    }
    seed_workers(core);                              /* gen_read() keeps drawing from core->ref_pos / rand_strand / rand_rlen / rand_meth */
    core->ref = load_ref(cfg.fasta);
    if (cfg.meth_freq[0]) load_meth_freq(cfg.meth_freq, core->ref);
    if (cfg.trans_count[0]) load_trans_count(cfg.trans_count, core->ref);

    const int8_t rna = (cfg.flags & SQ_RNA) ? 1 : 0, ont = (cfg.flags & SQ_ONT) ? 1 : 0;
    core->fp_fasta = cfg.fasta_out[0] ? fopen(cfg.fasta_out, "w") : NULL;
    core->fp_paf = cfg.paf[0] ? fopen(cfg.paf, "w") : NULL;
    core->fp_sam = cfg.sam[0] ? fopen(cfg.sam, "w") : NULL;
    if (core->fp_sam) sam_hdr_wr(core->fp_sam, core->ref);
    if (cfg.slow5[0]) {
        core->sp = slow5_open(cfg.slow5, "w");
        if (!core->sp) { fprintf(stderr, "cannot open %s\n", cfg.slow5); return 2; }
        set_header_attributes(core->sp, rna, (cfg.flags & SQ_R10) ? 1 : 0, cfg.p.sample_rate);
        set_header_aux_fields(core->sp, ont);
        if (slow5_hdr_write(core->sp) < 0) return 2;
    } else { fprintf(stderr, "ref_host_gpu: slow5= is required (the record helpers need the open file's header)\n"); return 2; }

    init_sqg(core, (int)cfg.device, cfg.exact ? SQG_MODE_EXACT : SQG_MODE_CERTIFIED);

    long n = cfg.nreads;
    if (cfg.flags & SQ_FULL_CONTIG) n = core->ref->num_ref;
    long done = 0;
    while (done < n) {                               /* src/sim.c:1065-1075 */
        const long nb = n - done < cfg.batch ? n - done : cfg.batch;
        db_t db;
        memset(&db, 0, sizeof db);
        db.n_rec = (int32_t)nb; db.capacity_rec = (int32_t)nb;
        db.mem_records = calloc((size_t)nb, sizeof *db.mem_records);
        db.mem_bytes = calloc((size_t)nb, sizeof *db.mem_bytes);
        if (core->fp_fasta) db.fasta = calloc((size_t)nb, sizeof *db.fasta);
        if (core->fp_paf) db.paf = calloc((size_t)nb, sizeof *db.paf);
        if (core->fp_sam) db.sam = calloc((size_t)nb, sizeof *db.sam);

        process_db_gpu(core, &db);

        for (long i = 0; i < nb; i++) {              /* output_db, src/sim.c:630-656 */
            if (slow5_write_bytes(db.mem_records[i], db.mem_bytes[i], core->sp) < 0) { fprintf(stderr, "Error writing record!\n"); return 2; }
            if (core->fp_fasta) { fputs(db.fasta[i], core->fp_fasta); free(db.fasta[i]); }
            if (core->fp_paf) { fputs(db.paf[i], core->fp_paf); free(db.paf[i]); }
            if (core->fp_sam) { fputs(db.sam[i], core->fp_sam); free(db.sam[i]); }
            free(db.mem_records[i]);
        }
        free(db.mem_records); free(db.mem_bytes); free(db.fasta); free(db.paf); free(db.sam);
        core->total_reads += nb;
        done += nb;
    }
    sqg_destroy(g_sqg);
    slow5_close(core->sp);
    if (core->fp_fasta) fclose(core->fp_fasta);
    if (core->fp_paf) fclose(core->fp_paf);
    if (core->fp_sam) fclose(core->fp_sam);
    return 0;
}
