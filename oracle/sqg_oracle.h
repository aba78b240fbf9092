/*
 * sqg_oracle.h -- CPU restatement of squigulator's per-read signal path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product (squigulator_amd/csrc, include/sqg.h) never links or calls it.
 *
 * Every function cites the reference file:line it restates (paths are under
 * the upstream tree, e.g. src/gensig.c:226).  Parity pins are listed in
 * DESIGN.md section "Oracle".
 */
#ifndef SQG_ORACLE_H
#define SQG_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* option bits: same values as the reference's opt_t.flag (src/sq.h:32-42) */
#define ORC_RNA         0x001
#define ORC_FULL_CONTIG 0x002
#define ORC_IDEAL       0x004
#define ORC_IDEAL_TIME  0x008
#define ORC_IDEAL_AMP   0x010
#define ORC_PREFIX      0x020
#define ORC_R10         0x040
#define ORC_PAF_REF     0x080
#define ORC_TRANS_TRUNC 0x100
#define ORC_CDNA        0x200
#define ORC_ONT         0x400
/* not an opt_t.flag bit: the reference keys methylation on opt.meth_freq != NULL (src/sim.c:231,297; src/gensig.c:231,251).
 * Set: the pore model is the 5-letter (A C G M T) table of 5^k rows, ranks are src/seq.h:45-74, a worker's seeds advance
 * by 5^k + 10, and gen_read methylates CpGs from the contig's frequency array (src/genread.c:207-241,276-277) */
#define ORC_METH        0x1000

/* profile_t, src/sq.h:47-58 */
typedef struct {
    double digitisation;
    double sample_rate;
    double bps;
    double range;
    double offset_mean;
    double offset_std;
    double median_before_mean;
    double median_before_std;
    double dwell_mean;
    double dwell_std;
} orc_profile_t;

/* model_t, src/sq.h:61-68 */
typedef struct {
    float level_mean;
    float level_stdv;
} orc_kmer_t;

/* nrng_t / grng_t, src/rand.h:40-50 */
typedef struct { double m, s; int64_t x; } orc_norm_t;
typedef struct { double a, b; int64_t x; } orc_gamma_t;

/* one virtual worker = one reference thread id; the stream set of
 * src/sq.h:89-100 as seeded by src/sim.c:238-257 */
typedef struct {
    int64_t pos_x;        /* core->ref_pos[tid]      */
    int64_t strand_x;     /* core->rand_strand[tid]  */
    int64_t meth_x;       /* core->rand_meth[tid] */
    orc_norm_t dwell;     /* core->rand_time[tid]    */
    orc_gamma_t rlen;     /* core->rand_rlen[tid]    */
    orc_norm_t offset;    /* core->rand_offset[tid]  */
    orc_norm_t median;    /* core->rand_median_before[tid] */
    int64_t *kmer_x;      /* core->kmer_gen[tid][rank]->x, raw Schrage state */
} orc_worker_t;

/* reference genome, src/ref.h:12-20 (+ transcript abundance table) */
typedef struct {
    int32_t num_ref;
    int64_t sum;
    char **names;
    char **seqs;
    int32_t *lengths;
    int32_t trans_n;      /* 0 when no --trans-count table */
    float *trans_csum;
    int32_t *trans_idx;
    uint8_t **meth;       /* ref->ref_meth: per contig NULL or round(255*freq) per base (src/ref.c:291-361); NULL: no --meth-freq */
} orc_ref_t;

typedef struct {
    orc_profile_t prof;
    uint32_t flags;
    float amp_noise;
    uint32_t kmer_size;
    uint32_t num_kmer;
    orc_kmer_t *model;    /* num_kmer entries */
    double *kmer_s;       /* nrng_t.s per k-mer: (double)(float)(level_stdv*amp_noise) */
    int64_t seed;
    int32_t num_workers;  /* -t */
    int32_t rlen;         /* -r */
    orc_worker_t *workers;
    int64_t n_samples;    /* core->n_samples: running start_time */
    int64_t total_reads;  /* core->total_reads */
} orc_core_t;

/* one simulated read, the fields work_per_single_read hands to the writers
 * (src/sim.c:514-618) */
typedef struct {
    int32_t tid;
    int32_t ref_idx;
    int32_t ref_len;
    int32_t ref_pos_st;
    int32_t ref_pos_end;
    int32_t rlen;
    char strand;
    char *seq;            /* the read as gen_read returned it */
    double offset;
    double median_before;
    int64_t len_raw_signal;
    int16_t *raw_signal;
    int64_t start_time;
    int64_t read_number;  /* core->total_reads + i */
    int64_t ss_n;         /* per-event dwell (aln->ss), incl. prefix/stall events */
    int32_t *ss;
} orc_read_t;

typedef struct {
    int32_t n;
    orc_read_t *reads;
} orc_batch_t;

/* ---- L1 primitives ---- */
double   orc_rng(int64_t *xp);                       /* src/rand.h:79-85  */
double   orc_nrng(orc_norm_t *r);                    /* src/rand.h:87-94  */
double   orc_grng(orc_gamma_t *r);                   /* src/rand.h:96-102 */
uint32_t orc_base_rank(char base);                   /* src/seq.h:14-27   */
uint32_t orc_kmer_rank(const char *s, uint32_t k);   /* src/seq.h:31-42   */
char    *orc_revcomp(const char *f);                 /* src/seq.h:78-112  */

/* ---- model / genome loaders ---- */
/* f5c-format text model, src/model.c:40-142. Returns k (0 on error). */
uint32_t orc_read_model(const char *path, orc_kmer_t **out);
orc_ref_t *orc_ref_load(const char *fasta);          /* src/ref.c:54-117 (plain text FASTA only) */
int      orc_ref_load_trans_count(orc_ref_t *ref, const char *tsv); /* src/ref.c:206-273 */
int      orc_ref_load_meth_freq(orc_ref_t *ref, const char *tsv);   /* src/ref.c:291-361; 0 ok, <0: the line the reference would exit on */
uint32_t orc_meth_kmer_rank(const char *s, uint32_t k);             /* src/seq.h:45-74 */
void     orc_ref_free(orc_ref_t *ref);

/* ---- core ---- */
orc_core_t *orc_core_new(const orc_profile_t *p, uint32_t flags, float amp_noise,
                         uint32_t kmer_size, const orc_kmer_t *model,
                         int64_t seed, int32_t num_workers, int32_t rlen); /* src/sim.c:215-258 */
void     orc_core_free(orc_core_t *c);

/* ---- the hot path: gen_sig, src/gensig.c:226-356 (+ src/genread.c:71-123) ----
 * Returns a malloc'd int16 array; *ss (optional) receives a malloc'd per-event
 * dwell array as aln->ss would hold. */
int16_t *orc_gen_sig(orc_core_t *c, const char *read, int32_t len,
                     double *offset, double *median_before, int64_t *len_raw_signal,
                     int tid, int32_t **ss, int64_t *ss_n);

/* ---- feeder: gen_read, src/genread.c:125-370 ---- */
char *orc_gen_read(orc_core_t *c, const orc_ref_t *ref, int tid,
                   int32_t *ref_idx, int32_t *ref_len, int32_t *ref_pos,
                   int32_t *rlen, char *strand);

/* ---- batch driver: process_db/work_db with the static partition of
 * src/thread.c:73-131 and no work stealing (deterministic regimes -t1 and
 * -t T -K T).  nthreads>1 runs virtual workers on host
 * threads; results are independent of nthreads. ---- */
orc_batch_t *orc_batch_run(orc_core_t *c, const orc_ref_t *ref, int32_t n_rec,
                           int want_ss, int nthreads);
/* as above but with caller-provided reads (seqs[i] of length lens[i]);
 * used to drive the GPU path and the oracle from the same sequences */
orc_batch_t *orc_batch_run_seqs(orc_core_t *c, int32_t n_rec, const char *const *seqs,
                                const int32_t *lens, int want_ss, int nthreads);
/* as orc_batch_run_seqs but with an explicit worker id per read (reads of one worker are
 * processed in index order); used by the multi-GPU sharding tests */
orc_batch_t *orc_batch_run_assigned(orc_core_t *c, int32_t n_rec, const char *const *seqs,
                                    const int32_t *lens, const int32_t *workers, int want_ss);
void     orc_batch_free(orc_batch_t *b);

/* ---- "next" row: the signal compression slow5lib applies to raw_signal in BLOW5 (svb-zd) ----
 * slow5lib/src/slow5_press.c:1055-1087 (ptr_compress_svb_zd): int16 -> int32, zig-zag of the delta to the
 * previous sample (first: to 0; thirdparty/streamvbyte/src/streamvbyte_zigzag.c), then StreamVByte
 * (thirdparty/streamvbyte/src/streamvbyte_encode.c): uint32 count | ceil(count/4) key bytes (2 bits per value =
 * bytes-1, first value in the low bits) | the values' 1-4 little-endian bytes.
 * orc_svb_zd_bound: bytes the output can need; orc_svb_zd: encodes, returns the bytes written. */
size_t   orc_svb_zd_bound(int64_t n_samples);
size_t   orc_svb_zd(const int16_t *sig, int64_t n_samples, uint8_t *out);

/* worker id of read i in a batch of n_rec under -t T (src/thread.c:80-99,122-125) */
int32_t  orc_worker_of(int32_t i, int32_t n_rec, int32_t T);

#ifdef __cplusplus
}
#endif
#endif
