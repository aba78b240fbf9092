/*
 * ref_harness.c -- drives the REFERENCE's own per-read path, compiled from
 * the sources where they lie under /root/reference (see oracle/Makefile).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it contains no reference
 * code.  It includes the reference's headers at build time, fills a core_t,
 * and calls the reference's gen_read() (src/genread.c:358) and gen_sig()
 * (src/gensig.c:346) -- i.e. the real rand.h / seq.h / gensig.c / genread.c
 * arithmetic -- so the oracle restatement can be checked sample for sample.
 *
 * What is NOT the reference here: the reference's driver (src/sim.c) cannot be
 * linked because src/model.c needs the absent src/model.h; so this harness
 * (a) reads the pore model from an f5c-format text file itself, and
 * (b) seeds the per-worker streams the way src/sim.c:238-257 does.  (b) is
 * pinned independently by the reference's goldens: with slow5=/fasta_out=/paf=
 * this harness writes the same files scripts/test.sh diffs, through the
 * reference's own slow5lib and format.c.
 *
 * usage: ref_harness <config-file>     (key=value lines, see parse_cfg)
 */
#define _XOPEN_SOURCE 700
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>

#include "sq.h"      /* core_t, opt_t, profile_t, model_t, SQ_* flags */
#include "format.h"  /* aln_t, init_aln, paf_str, sam_str, sam_hdr_wr */
#include "error.h"

/* entry points of the reference's path (definitions in src/genread.c, src/gensig.c) */
char *gen_read(core_t *core, char **ref_id, int32_t *ref_len, int32_t *ref_pos, int32_t *rlen, char *c, int8_t rna, int tid);
int16_t *gen_sig(core_t *core, const char *read, int32_t len, double *offset, double *median_before, int64_t *len_raw_signal, int8_t rna, int tid, aln_t *aln);
void set_header_attributes(slow5_file_t *sp, int8_t rna, int8_t r10, double sample_frequency);
void set_header_aux_fields(slow5_file_t *sp, int8_t ont_friendly);
void set_record_primary_fields(profile_t *profile, slow5_rec_t *rec, char *read_id, double offset, int64_t len_raw_signal, int16_t *raw_signal);
void set_record_aux_fields(slow5_rec_t *rec, slow5_file_t *sp, double median_before, int32_t read_number, uint64_t start_time, int8_t ont_friendly);
void load_meth_freq(const char *meth_freq, ref_t *ref);        /* src/ref.c:291 */

#include "ref_common.h"   /* cfg_t, parse_cfg, load_table, seed_workers, put, now_s */

int main(int argc, char **argv) {
    if (argc != 2) { fprintf(stderr, "usage: %s <config>\n", argv[0]); return 2; }
    cfg_t cfg; parse_cfg(argv[1], &cfg);
    set_log_level(LOG_ERR);
    const double t_start = now_s();

    if (cfg.svb_in[0]) {         /* compress arbitrary int16 arrays with the library's svb-zd coder, nothing else */
        FILE *fi = fopen(cfg.svb_in, "rb"), *fo = fopen(cfg.svb_out, "wb");
        if (!fi || !fo) { perror("svb_in/svb_out"); return 2; }
        int32_t na = 0;
        if (fread(&na, 4, 1, fi) != 1) return 2;
        for (int32_t a = 0; a < na; a++) {
            int64_t len = 0;
            if (fread(&len, 8, 1, fi) != 1) return 2;
            int16_t *buf = malloc((size_t)(len > 0 ? len : 1) * sizeof *buf);
            if (len && fread(buf, 2, (size_t)len, fi) != (size_t)len) return 2;
            size_t nb = 0;
            void *z = slow5_ptr_compress_solo(SLOW5_COMPRESS_SVB_ZD, buf, (size_t)len * sizeof *buf, &nb);
            if (!z) { fprintf(stderr, "svb-zd failed\n"); return 2; }
            int64_t nb64 = (int64_t)nb;
            put(fo, &nb64, 8); put(fo, z, nb);
            free(z); free(buf);
        }
        fclose(fi); fclose(fo);
        return 0;
    }

    core_t *core = calloc(1, sizeof *core);
    core->opt.rlen = (int32_t)cfg.rlen; core->opt.seed = cfg.seed; core->opt.flag = cfg.flags;
    core->opt.num_thread = (int32_t)cfg.threads; core->opt.batch_size = (int32_t)cfg.batch;
    core->opt.amp_noise = cfg.amp_noise;
    core->profile = cfg.p;
    core->model = malloc(sizeof(model_t) * MAX_NUM_KMER);
    if (cfg.meth_freq[0]) {                                    /* src/sim.c:296-326 */
        core->opt.meth_freq = cfg.meth_freq;
        core->cpgmodel = malloc(sizeof(model_t) * MAX_NUM_KMER_METH);
        core->kmer_size = load_table(cfg.model, core->cpgmodel, 1);
        core->num_kmer = (uint32_t)pow(5, core->kmer_size);
    } else {
        core->kmer_size = load_table(cfg.model, core->model, 0);
        core->num_kmer = 1u << (2 * core->kmer_size);
    }
    seed_workers(core);
    core->ref = load_ref(cfg.fasta);
    if (cfg.meth_freq[0]) load_meth_freq(cfg.meth_freq, core->ref);          /* src/sim.c:339-341 */
    if (cfg.trans_count[0]) load_trans_count(cfg.trans_count, core->ref);

    const int8_t rna = (cfg.flags & SQ_RNA) ? 1 : 0, ont = (cfg.flags & SQ_ONT) ? 1 : 0;
    FILE *fout = cfg.out[0] ? fopen(cfg.out, "wb") : NULL;
    FILE *ffa = cfg.fasta_out[0] ? fopen(cfg.fasta_out, "w") : NULL;
    FILE *fsvb = cfg.svb_out[0] ? fopen(cfg.svb_out, "wb") : NULL;
    FILE *fpaf = cfg.paf[0] ? fopen(cfg.paf, "w") : NULL;
    FILE *fsam = cfg.sam[0] ? fopen(cfg.sam, "w") : NULL;
    if (fsam) sam_hdr_wr(fsam, core->ref);
    slow5_file_t *sp = NULL;
    if (cfg.slow5[0]) {
        sp = slow5_open(cfg.slow5, "w");
        if (!sp) { fprintf(stderr, "cannot open %s\n", cfg.slow5); return 2; }
        set_header_attributes(sp, rna, (cfg.flags & SQ_R10) ? 1 : 0, cfg.p.sample_rate);
        set_header_aux_fields(sp, ont);
        if (slow5_hdr_write(sp) < 0) return 2;
    }
    core->sp = sp;

    long n = cfg.nreads;
    if (cfg.time_s > 0) n = 2000000000L;                      /* timed mode: until the clock says stop */
    if (cfg.flags & SQ_FULL_CONTIG) n = core->ref->num_ref;
    if (fout) { int32_t n32 = (int32_t)n; put(fout, "SQGREF1", 8); put(fout, &n32, 4); }

    const int T = core->opt.num_thread;
    long done = 0, timed_reads = 0;
    const double t_loop = now_s();
    int stop = 0;
    while (done < n && !stop) {
        const long nb = n - done < cfg.batch ? n - done : cfg.batch;
        const long step = T > 1 ? (nb + T - 1) / T : nb;      /* static partition, src/thread.c:80 */
        for (long i = 0; i < nb; i++) {
            const int tid = T > 1 ? (int)(i / step) : 0;
            /* NB: reads of one worker are generated in index order, workers one after
             * another -- equal to the threaded run whenever no work is stolen */
            char *rid = NULL, *seq = NULL, strand = '+';
            int32_t rlen = 0, pos_st = 0, ref_len = 0;
            aln_t *aln = init_aln();
            if (cfg.flags & SQ_FULL_CONTIG) {
                rid = core->ref->ref_names[done + i]; rlen = core->ref->ref_lengths[done + i];
                seq = core->ref->ref_seq[done + i];
            } else {
                seq = gen_read(core, &rid, &ref_len, &pos_st, &rlen, &strand, rna, tid);
            }
            double offset = 0, median = 0; int64_t len = 0;
            int16_t *sig = gen_sig(core, seq, rlen, &offset, &median, &len, rna, tid, aln);
            const int64_t start_time = core->n_samples;
            core->n_samples += len;

            char read_id[10000];
            if (ont) sprintf(read_id, "00000000-0000-0000-0000-%012d", (int)(done + i + 1));
            else sprintf(read_id, "S1_%ld!%s!%d!%d!%c", done + i + 1, rid, pos_st, pos_st + rlen, strand);

            if (fout) {
                int32_t hdr[6] = {tid, -1, ref_len, pos_st, rlen, (int32_t)strand};
                for (int q = 0; q < core->ref->num_ref; q++) if (core->ref->ref_names[q] == rid) hdr[1] = q;
                put(fout, hdr, sizeof hdr);
                put(fout, &offset, 8); put(fout, &median, 8); put(fout, &len, 8); put(fout, &start_time, 8);
                int64_t ssn = aln->ss_n; put(fout, &ssn, 8);
                put(fout, seq, (size_t)rlen);
                put(fout, sig, (size_t)len * 2);
                put(fout, aln->ss, (size_t)ssn * 4);
            }
            if (fsvb) {      /* the library's signal compression as slow5_rec_to_mem applies it (slow5_press.c:317,1055) */
                size_t nb_svb = 0;
                void *z = slow5_ptr_compress_solo(SLOW5_COMPRESS_SVB_ZD, sig, (size_t)len * sizeof *sig, &nb_svb);
                if (!z) { fprintf(stderr, "svb-zd failed\n"); return 2; }
                int64_t nb64 = (int64_t)nb_svb;
                put(fsvb, &nb64, 8); put(fsvb, z, nb_svb);
                free(z);
            }
            if (ffa) fprintf(ffa, ">%s\n%s\n", read_id, seq);
            if (fpaf || fsam) {
                const int64_t nk = rlen - core->kmer_size + 1;
                aln->read_id = read_id; aln->len_raw_signal = len; aln->strand = strand;
                aln->si_st_ref = rna ? pos_st + rlen - core->kmer_size + 1 : pos_st;
                aln->si_end_ref = rna ? pos_st : pos_st + rlen - core->kmer_size + 1;
                if (cfg.flags & SQ_PAF_REF) {
                    aln->tid = rid;
                    aln->tlen = !(cfg.flags & SQ_FULL_CONTIG) ? ref_len - core->kmer_size + 1 : nk;
                    aln->t_st = aln->si_st_ref; aln->t_end = aln->si_end_ref;
                } else {
                    aln->tid = read_id; aln->tlen = nk;
                    aln->t_st = rna ? nk : 0; aln->t_end = rna ? 0 : nk;
                }
                if (fpaf) { char *s = paf_str(aln); fputs(s, fpaf); free(s); }
                if (fsam) { char *s = sam_str(aln, seq, rid, pos_st); fputs(s, fsam); free(s); }
            }
            if (sp) {
                slow5_rec_t *rec = slow5_rec_init();
                char *idcopy = strdup(read_id);
                set_record_primary_fields(&core->profile, rec, idcopy, offset, len, sig);
                set_record_aux_fields(rec, sp, median, (int32_t)(done + i), (uint64_t)start_time, ont);
                if (slow5_write(rec, sp) < 0) { fprintf(stderr, "slow5_write failed\n"); return 2; }
                slow5_rec_free(rec);           /* frees sig and idcopy */
            } else {
                free(sig);
            }
            free_aln(aln);
            if (!(cfg.flags & SQ_FULL_CONTIG)) free(seq);
            timed_reads++;
            if (cfg.time_s > 0 && now_s() - t_loop >= cfg.time_s) { stop = 1; break; }
        }
        core->total_reads += nb;
        done += nb;
    }
    if (sp) slow5_close(sp);
    if (cfg.time_s > 0) {
        const double t_end = now_s();
        printf("SQGTIME samples=%lld reads=%ld loop_seconds=%.6f total_seconds=%.6f\n", (long long)core->n_samples, timed_reads,
               t_end - t_loop, t_end - t_start);
    }
    if (fout) fclose(fout);
    if (ffa) fclose(ffa);
    if (fsvb) fclose(fsvb);
    if (fpaf) fclose(fpaf);
    if (fsam) fclose(fsam);
    return 0;
}
