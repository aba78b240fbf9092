/*
 * sqg.h -- C ABI of the MI355X-native per-read signal generator.
 *
 * Drop-in seam (reference file:line under the upstream tree):
 *   the batch-level replacement of
 *       process_db(core, db)                         src/sim.c:622-627
 *         -> work_db(core, db, work_per_single_read) src/thread.c:119-131
 *            -> gen_sig(core, read, len, &offset, &median_before,
 *                       &len_raw_signal, rna, tid, aln)   src/gensig.c:346
 *   A GPU wants a batch, not a read, so the unit handed over is one batch of
 *   reads exactly as gen_read() returned them (src/genread.c:358, i.e. after
 *   strand/revcomp and N substitution), plus the worker (tid) each read runs
 *   on.  Results are what gen_sig returns per read: the int16 raw signal, its
 *   length, `offset`, `median_before`, and -- for PAF/SAM -- the per-event
 *   dwell array aln->ss (src/gensig.c:273-281).
 *
 * Determinism contract: output is a pure function of (cfg, the sequence of
 * batches).  It equals the reference run with `-t T -K K --seed S` in the two
 * regimes where the reference itself is deterministic: T==1, and T>=batch size
 * (one read per worker per batch, no work stealing).  For
 * 1<T<K the reference's static partition (src/thread.c:80-99) without stealing
 * is used.
 *
 * Plain C, no torch types.  All functions return 0 on success or a negative
 * SQG_E* code; nothing here calls exit() (the reference's ERROR()+exit paths,
 * src/error.h:75-111, become return codes).
 */
#ifndef SQG_H
#define SQG_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQG_ABI_VERSION 2

/* option bits -- identical values to opt_t.flag, src/sq.h:32-42 */
#define SQG_RNA         0x001u
#define SQG_IDEAL       0x004u
#define SQG_IDEAL_TIME  0x008u
#define SQG_IDEAL_AMP   0x010u
#define SQG_PREFIX      0x020u
#define SQG_R10         0x040u   /* header of the BLOW5 writer: sequencing_kit (src/gensig.c:104-109)                 */
#define SQG_ONT         0x400u   /* ... and the ONT-friendly extra field end_reason (src/gensig.c:160-168)            */
/* not an opt_t.flag bit: the reference keys CpG methylation on opt.meth_freq != NULL (src/sim.c:231,297; src/gensig.c:231,251).
 * Set: cfg.model is the 5-letter (A C G M T) table of 5^k rows (core->cpgmodel), k-mer ranks are the base-5 numbers of
 * src/seq.h:45-74, and a worker's seeds advance by 5^k + 10 (src/sim.c:325).  Reads handed to sqg_batch_stage carry 'M' where
 * gen_read methylated a CpG; the device sampler does it itself once sqg_genome_set_meth has been called. */
#define SQG_METH        0x1000u
/* not an opt_t.flag bit either, and no effect on any result: the few-worker stream hand-out normally relies on the lanes of one
 * LDS atomic being served in ascending lane order -- verified for gfx950 offline (2e8 fetch-adds, tools/README.md), re-checked
 * on the device at sqg_create in the production shape (4096-entry table, 16-bit addends, four wavefronts per CU) and sampled in
 * every batch (the first events of every slice against order-free prefix sums; a mismatch fails the batch).  Set: the context
 * uses the order-free kernels (claim protocol, lane masks) from the start -- slower, no such dependence. */
#define SQG_ORDER_FREE  0x2000u

/* error codes */
#define SQG_OK            0
#define SQG_EINVAL       -1   /* bad argument / unsupported configuration     */
#define SQG_ENOMEM       -2   /* host or device allocation failed             */
#define SQG_EDEVICE      -3   /* HIP runtime error (see sqg_last_error)       */
#define SQG_ESEQUENCE    -4   /* batches must be run in the order staged      */
#define SQG_ENODEVICE    -5   /* no usable gfx950 device / HIP not available  */
#define SQG_EOVERFLOW    -6   /* a read would exceed UINT32_MAX samples (src/sim.c:559-562) */
#define SQG_EIO          -7   /* the BLOW5 writer could not open / write its file (errno text: stderr / sqg_blow5_last_error) */

/* arithmetic mode of the sample kernel */
#define SQG_MODE_EXACT      0 /* every sample through the FP64 path                      */
#define SQG_MODE_CERTIFIED  1 /* fp32 fast path + error-bounded test, FP64 for the rest  */

/* profile_t, src/sq.h:47-58 -- same fields, same order */
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
} sqg_profile_t;

/* model_t, src/sq.h:61-68 -- one pore-model row per k-mer rank */
typedef struct {
    float level_mean;
    float level_stdv;
} sqg_kmer_t;

typedef struct {
    uint32_t abi_version;       /* SQG_ABI_VERSION                                           */
    sqg_profile_t profile;      /* core->profile                                             */
    uint32_t flags;             /* SQG_* bits of core->opt.flag                              */
    float amp_noise;            /* core->opt.amp_noise                                       */
    uint32_t kmer_size;         /* core->kmer_size, 1..9                                     */
    const sqg_kmer_t *model;    /* core->model, 4^k rows (host memory, copied)               */
    int64_t seed;               /* core->opt.seed (must be != 0 as in the reference CLI)     */
    int32_t num_workers;        /* T = core->opt.num_thread: virtual workers in the job      */
    int32_t worker_lo;          /* this context owns workers [worker_lo, worker_hi):         */
    int32_t worker_hi;          /*   0,T on one GPU; a shard of them per GPU otherwise       */
    int32_t device;             /* HIP device ordinal                                        */
    uint32_t mode;              /* SQG_MODE_*                                                */
} sqg_cfg_t;

typedef struct sqg_ctx sqg_ctx_t;
typedef struct sqg_batch sqg_batch_t;

/* per-batch results; host arrays are owned by the batch and stay valid until
 * sqg_batch_free(); device pointers (and sqg_fetch_*) stay valid until TWO more
 * batches have been run on the same context: outputs live in two context-owned
 * sets of HBM slabs used alternately (sqg_batch_run is asynchronous and batches
 * can be queued back to back; sqg_batch_wait waits for that batch only) */
typedef struct {
    int32_t n_reads;
    int64_t n_events;           /* k-mer events incl. prefix/stall events                    */
    int64_t n_samples;          /* int16 samples written                                     */
    int64_t n_bases;            /* sequence bytes read by the kernels                        */
    const int64_t *sig_off;     /* [n_reads+1] exclusive scan of len_raw_signal (host)       */
    const int64_t *ev_off;      /* [n_reads+1] exclusive scan of per-read event counts (host)*/
    const double *offset;       /* [n_reads] slow5 `offset` per read (host)                  */
    const double *median_before;/* [n_reads] (host)                                          */
    const int16_t *d_signal;    /* device: n_samples int16, read i at sig_off[i]             */
    const uint16_t *d_dwell;    /* device: n_events samples-per-event (aln->ss order)        */
} sqg_result_t;

/* kernel timings of the batch last waited for, from hipEvents on the context's stream (see sqg_set_phase_timing) */
typedef struct {
    float dwell_ms;             /* stand-alone k_dwell (0 when the draws are made inside k_events) */
    float events_ms;            /* k_events (dwell draws, k-mer ranks, stream hand-out)      */
    float samples_ms;           /* k_scan + k_samples_lean + k_samples<generic> + k_fixup*   */
    float lean_ms;              /* k_samples_lean alone: the dominant, roofline-priced kernel (0 if not launched) */
    float total_ms;             /* first launch to last completion                           */
    int64_t fallback_samples;   /* CERTIFIED mode: samples recomputed on the FP64 path; -1: not known any more (the batch was
                                 * waited for after two later batches had been run: the counters live with the output slabs) */
    int32_t carried_first_pass; /* 1: this batch's launch sequence (events_ms) also held the first event pass of the batch behind it */
    int32_t first_pass_ran_ahead; /* 1: this batch's own first event pass ran inside its predecessor's sequence (not in events_ms)  */
} sqg_timing_t;

int  sqg_create(const sqg_cfg_t *cfg, sqg_ctx_t **out);
void sqg_destroy(sqg_ctx_t *ctx);
const char *sqg_last_error(const sqg_ctx_t *ctx);   /* "" if none */
const char *sqg_strerror(int code);
int  sqg_device_count(void);                        /* <0: HIP unusable */
/* "source_hash=<16 hex digits>;dev=<0|1>": the sources the library was built from (squigulator_amd/build.py) and whether it is
 * the development build, the only one that reads A/B and test knobs from the environment (tools/README.md) */
const char *sqg_build_info(void);

/* Stage one batch: sequences are the reads exactly as gen_read() returns them.
 *   seqs     concatenated read bytes (ASCII; IUPAC handled as src/seq.h:14-27)
 *   seq_off  [n_reads+1] byte offsets into seqs
 *   worker   [n_reads] global worker id (tid) of each read, or NULL for the
 *            reference's static partition of this batch over num_workers
 * Host-side per-read draws (offset, median_before: src/gensig.c:311-317) are
 * made here, in staging order; sequences are uploaded to HBM. */
int  sqg_batch_stage(sqg_ctx_t *ctx, int32_t n_reads, const char *seqs,
                     const int64_t *seq_off, const int32_t *worker, sqg_batch_t **out);
/* Launch the kernels for a staged batch (asynchronous on the context stream).
 * Batches must be run in the order they were staged.
 * If the NEXT batch is already staged when this one is run (few workers, k > 6: the bucketed hand-out), its first event pass --
 * the dwell draws of src/gensig.c:254-257, which need nothing but the staged reads -- is launched inside this batch's sequence,
 * sharing the GPU with this batch's stream hand-out (one is bound by the VALU, the other by memory).  A host that wants it stages
 * batch i+1 before it runs batch i; results, order, and what stays valid for how long do not change.  Two things do, and a host may
 * notice them: (1) sqg_batch_free of a batch that was staged, whose first pass ran ahead this way, and that is freed WITHOUT having been
 * run waits for the context's stream (the pass may still be reading the batch's memory); (2) sqg_get_timing attributes that pass to
 * the batch whose launch sequence carried it: events_ms of batch i holds batch i+1's first pass, and batch i+1's own events_ms does
 * not (sqg_timing_t.carried_first_pass / .first_pass_ran_ahead say which batches that applies to).  The pass is best effort: if a
 * buffer it needs cannot be had, the batch behind simply counts for itself.
 *
 * Placement calibration: the first twelve batches of a context with >= 2^24 events each are also used to place the event records -- the scatter
 * pass is timed (one host synchronisation per such batch) on up to three more allocations per buffer set and the fastest is kept; results
 * are not affected.  A host that times its first batches sees it; run a dozen batches before measuring.
 */
int  sqg_batch_run(sqg_ctx_t *ctx, sqg_batch_t *b);
/* Block until the batch has finished; fills *res. */
int  sqg_batch_wait(sqg_ctx_t *ctx, sqg_batch_t *b, sqg_result_t *res);
/* Copy results of the most recently run batch to host memory. */
int  sqg_fetch_signal(sqg_ctx_t *ctx, sqg_batch_t *b, int16_t *dst /* n_samples */);
int  sqg_fetch_dwell(sqg_ctx_t *ctx, sqg_batch_t *b, int32_t *dst /* n_events, as aln->ss */);
/* Waits for the batch's own kernels only.  A staged batch may be freed without having been run: the batches staged
 * after it then run in order as usual, but its reads' share of the workers' streams stays spent. */
void sqg_batch_free(sqg_ctx_t *ctx, sqg_batch_t *b);
int  sqg_get_timing(sqg_ctx_t *ctx, sqg_timing_t *t);
/* The phase boundaries behind sqg_timing_t are hipEvents recorded between the kernels -- barrier packets that cost the GPU
 * 3-8 us each (1.2 % of a 16384-read step, 5 % of a 1000-read one).  every = 1 (default): each batch carries them; every = n:
 * the batches whose run index is a multiple of n do, the others report 0 ms in every field (fallback_samples is always filled
 * in); every = 0: none.  (No counterpart in the reference: its only timings are realtime() around process_db, src/sim.c:575-583.) */
int  sqg_set_phase_timing(sqg_ctx_t *ctx, int every);
/* Host threads (the calling one included) that share the per-read `offset` / `median_before` draws of sqg_batch_stage /
 * sqg_batch_sample (host libm: src/gensig.c:315-316).  n = 0 (default): automatic -- four from 8192 reads per batch on, never more
 * than the CPUs the process may use (affinity mask, cgroup quota); n in 1..64: exactly n.  A host that drives one context per GPU
 * divides its CPU budget by the number of contexts (the reference's counterpart is -t, src/thread.c:73-116).  The draws are the
 * same doubles for every n.  Returns the number of threads the context's last staging call used (0: none yet), < 0 on error. */
int  sqg_set_stage_threads(sqg_ctx_t *ctx, int n);

/* Convenience: stage + run + wait in one call (one process_db()). */
int  sqg_submit(sqg_ctx_t *ctx, int32_t n_reads, const char *seqs, const int64_t *seq_off,
                const int32_t *worker, sqg_batch_t **out, sqg_result_t *res);

/* Worker id the reference's scheduler gives read i of a batch of n_rec reads
 * under -t T when no work is stolen (src/thread.c:80-99,122-125). */
int32_t sqg_worker_of(int32_t i, int32_t n_rec, int32_t T);

/* ---- next row: the signal compression of BLOW5 records, on the device ----
 * For every read of the batch, the bytes slow5lib's ptr_compress_svb_zd (slow5lib/src/slow5_press.c:1055-1087,
 * reached from slow5_rec_to_mem -> slow5_ptr_compress for the raw_signal field when the file's signal method is
 * svb-zd) would produce from its raw_signal: uint32 count | StreamVByte keys | data of the zig-zag deltas.
 * A host that writes BLOW5 copies read i's bytes [svb_off[i], svb_off[i+1]) instead of the 2-byte samples:
 * ~1.3 B/sample over PCIe instead of 2, and no svb encode on the CPU.  Call after sqg_batch_run; blocks. */
typedef struct {
    int64_t n_bytes;            /* total encoded bytes of the batch                              */
    const int64_t *svb_off;     /* [n_reads+1] byte offsets (host, owned by the batch)           */
    const uint8_t *d_svb;       /* device: the encodings, valid until the next sqg_batch_compress */
} sqg_svb_t;
int  sqg_batch_compress(sqg_ctx_t *ctx, sqg_batch_t *b, sqg_svb_t *out);
int  sqg_fetch_svb(sqg_ctx_t *ctx, sqg_batch_t *b, uint8_t *dst /* n_bytes */);

/* ---- native BLOW5 writer: the slow5_encode / slow5_write_bytes half of work_per_single_read and output_db
 * (src/sim.c:604-640), without slow5lib.  The file `squigulator -o x.blow5` writes: zlib record compression, svb-zd signal
 * compression (slow5lib/src/slow5.c:421-423), header of set_header_attributes / set_header_aux_fields
 * (src/gensig.c:40-169), records laid out as slow5_rec_to_mem does (slow5lib/src/slow5.c:3928-4072), "5WOLB" at the end.
 * The signal field of a record is the svb-zd encoding sqg_batch_compress made on the device; framing and zlib (one deflate
 * stream per record, as slow5lib's) run on `threads` host threads (<= 0: up to 16).  read_number and start_time continue
 * over the calls in read order (src/sim.c:602).  Pure host code: usable without a GPU when the encodings come from elsewhere. */
typedef struct sqg_blow5 sqg_blow5_t;
/* flags of sqg_blow5_open only -- the record compression MODE.  Default (bit clear): one deflate stream per record with slow5lib's
 * parameters, on the host's threads: the file is `cmp`-identical to the reference's (28 MB/s per thread is what bounds it).
 * SQG_BLOW5_STORED: the same records, each in a zlib stream of STORED blocks (RFC 1951 BTYPE 00: 78 01 | {final, LEN, ~LEN, <= 65535
 * bytes}* | Adler-32) -- a valid BLOW5 file that slow5lib (any inflate) reads back to the same records, field for field and sample for
 * sample, in other bytes (1.3 per sample instead of 0.97); sqg_blow5_write_batch then takes the records framed on the device
 * (sqg_batch_blow5_records) and the host's part is one PCIe copy and one pwrite(), made behind the caller while the next batch is fetched.  (A deflate block with the fixed Huffman
 * code would be larger, not smaller: svb-zd bytes are nearly uniform, and that code spends 8-9 bits on a literal.) */
#define SQG_BLOW5_STORED 0x10000u
/* ... on n files instead of one (with SQG_BLOW5_STORED; n <= 255): `path` x.blow5 names x.0.blow5 ... x.<n-1>.blow5, each a BLOW5 file of its own
 * (header, records, end marker); every batch's reads are dealt out to them in n contiguous ranges, read_number and start_time stay the
 * job's.  What bounds the stored mode is the file: one file of a tmpfs takes 6.9 GB/s from any number of writers, n files n times that
 * (tools/io_probe.cpp) -- the way a run that writes many BLOW5 files (as the sequencers do) scales its sink.  sqg_blow5_close reports the
 * files' sizes added up. */
#define SQG_BLOW5_SHARDS(n) (((uint32_t)(n) & 0xffu) << 24)
int  sqg_blow5_open(const char *path, const sqg_profile_t *profile, uint32_t flags /* SQG_RNA | SQG_R10 | SQG_ONT | SQG_BLOW5_STORED | SQG_BLOW5_SHARDS(n) */,
                    int32_t threads, sqg_blow5_t **out);
/* The batch's records as SQG_BLOW5_STORED writes them, framed on the device around its svb-zd encodings (compressing the batch first
 * if need be) and copied to pinned host memory of the context: *records (valid until the next-but-one call on this context), *n_bytes, and --
 * rec_off may be NULL -- [n_reads+1] offsets of the records in it.  read_number0 / start_time0: records and samples written before
 * this batch (src/sim.c:602).  Read ids of at most 4096 bytes. */
int  sqg_batch_blow5_records(sqg_ctx_t *ctx, sqg_batch_t *b, const sqg_profile_t *profile, uint32_t flags, const char *read_ids,
                             const int64_t *id_off, int64_t read_number0, uint64_t start_time0,
                             const uint8_t **records, int64_t *n_bytes, const int64_t **rec_off);
/*   read_ids/id_off [n+1]: the reads' ids, concatenated (src/sim.c:564-570); offset, median_before [n]; sig_off [n+1]: samples
 *   per read as differences (sqg_result_t.sig_off); svb/svb_off [n+1]: the encodings (sqg_fetch_svb / sqg_svb_t.svb_off) */
int  sqg_blow5_write(sqg_blow5_t *w, int32_t n, const char *read_ids, const int64_t *id_off, const double *offset,
                     const double *median_before, const int64_t *sig_off, const uint8_t *svb, const int64_t *svb_off);
/* the same for a batch that has been run: waits for it, compresses on the device, fetches and writes */
int  sqg_blow5_write_batch(sqg_blow5_t *w, sqg_ctx_t *ctx, sqg_batch_t *b, const char *read_ids, const int64_t *id_off);
int  sqg_blow5_close(sqg_blow5_t *w, int64_t *n_bytes /* may be NULL: the file's size */);
const char *sqg_blow5_last_error(const sqg_blow5_t *w);

/* ---- next row: device-resident genome + read sampling on the device ----
 * sqg_genome_load keeps the reference sequences (ref_t after load_ref, src/ref.c:54-117) in HBM;
 * sqg_batch_sample then does what the host loop `gen_read(core, ...)` (src/genread.c:358-370, called from
 * work_per_single_read, src/sim.c:550) does for every read of a batch -- with the workers' ref_pos / rand_strand /
 * rand_rlen streams (src/sim.c:238-247) living on the device -- and stages the batch exactly as sqg_batch_stage
 * would have staged those reads: no sequence bytes cross PCIe. */
#define SQG_SAMPLE_DNA    0u   /* gen_read_dna, src/genread.c:243-281                              */
#define SQG_SAMPLE_RNA    1u   /* whole transcripts, '+' strand, src/genread.c:311-355             */
#define SQG_SAMPLE_CDNA   2u   /* transcripts with a strand draw (--cdna)                          */
#define SQG_SAMPLE_TRUNC  4u   /* --trans-trunc, src/genread.c:303-309                             */
#define SQG_SAMPLE_FULL   8u   /* --full-contigs (src/sim.c:543-549): read i of the job IS contig i, '+', as loaded;
                                  gen_read is not called, no sampler stream moves; more reads than contigs: SQG_EINVAL */
typedef struct {
    int32_t n_contigs;
    const char *seqs;           /* contigs back to back, as loaded (no terminators)                 */
    const int64_t *contig_off;  /* [n_contigs+1]                                                    */
    int32_t rlen;               /* -r: mean read length (opt.rlen)                                  */
    uint32_t mode;              /* SQG_SAMPLE_* bits                                                */
    int32_t n_trans;            /* --trans-count table (src/ref.c:206-273), 0 if none               */
    const float *trans_csum;    /* [n_trans] cumulative abundance                                   */
    const int32_t *trans_idx;   /* [n_trans] contig of each entry                                   */
} sqg_genome_t;
typedef struct {                /* what gen_read returned, per read (host arrays owned by the batch) */
    const int32_t *ref_idx;     /* contig                                                           */
    const int32_t *ref_len;     /* *ref_len                                                         */
    const int32_t *ref_pos;     /* *ref_pos (0-based start on the forward strand)                   */
    const int32_t *rlen;        /* bases in the read                                                */
    const char *strand;         /* '+' / '-'                                                        */
    const int64_t *seq_off;     /* [n_reads+1] offsets of the reads in the sqg_fetch_reads() array  */
} sqg_sample_t;
int  sqg_genome_load(sqg_ctx_t *ctx, const sqg_genome_t *g);
/* the same with g->seqs in DEVICE memory (a reference already resident in HBM -- e.g. decompressed or synthesised there --
 * is copied device to device; everything else in *g is host memory as above) */
int  sqg_genome_load_device(sqg_ctx_t *ctx, const sqg_genome_t *g);
/* --meth-freq (contexts created with SQG_METH), after sqg_genome_load: `freq` holds one byte per base of the loaded genome,
 * contigs back to back -- round(255 * frequency) at the C of a CpG, 0 elsewhere, as load_meth_freq builds ref->ref_meth
 * (src/ref.c:291-361); contig_has[i] != 0: contig i has at least one line in the frequency file (ref->ref_meth[i] != NULL:
 * only its reads take draws from rand_meth).  sqg_batch_sample* then methylate as gen_read_dna does
 * (methylate_dna, src/genread.c:207-241,276-278): the reads carry 'M'. */
int  sqg_genome_set_meth(sqg_ctx_t *ctx, const uint8_t *freq, const uint8_t *contig_has);
int  sqg_batch_sample(sqg_ctx_t *ctx, int32_t n_reads, const int32_t *worker, sqg_batch_t **out, sqg_sample_t *info);
/* the sampled reads as gen_read returned them (after N substitution and revcomp), for the FASTA/SAM writers */
int  sqg_fetch_reads(sqg_ctx_t *ctx, sqg_batch_t *b, char *dst /* seq_off[n_reads] bytes */);

/* ---- SURVEY.md section 8e, "strict -t 1" over several GPUs: range sharding ----
 * A job is normally sharded by worker (worker_lo/worker_hi): no exchange at all.  With fewer workers than GPUs -- the
 * reference's reproducible regime is -t 1 -- every GPU's context instead owns ALL workers and generates a contiguous
 * range [lo, hi) of each batch's reads.  What a worker's streams need to know about the reads generated elsewhere:
 *   scalar streams (offset, median_before, time; the sampler's three): a fixed number of draws per read / event /
 *     gen_read attempt.  sqg_batch_sample_range samples the whole batch on every GPU, stages [lo, hi) and moves the
 *     streams over the rest; sqg_skip_reads does the latter for reads staged with sqg_batch_stage (call it for the
 *     reads before the range, stage, call it for the reads after it; seq_len = bases of each read, worker = its id);
 *   k-mer streams: how many samples the other ranges draw from each (worker, k-mer) stream -- the path's one
 *     exchange step.  sqg_batch_run_begin leaves this range's counts on the device ([num_workers][4^k] uint32, complete
 *     when it returns, valid until the next begin; d_before / d_after must be complete when sqg_batch_run_end is called); the caller all-gathers them over the GPUs in range order (RCCL) and passes
 *     sqg_batch_run_end the element-wise sums over the earlier (d_before) and the later (d_after) ranges.  A batch
 *     may draw at most 2^32-1 samples from one stream over all ranges (k > 6: 3.2e9).
 * sqg_set_range_mode(ctx, 1) must be on when such batches are staged (no staged batch may be pending);
 * sqg_batch_run still works and equals begin + end(NULL, NULL): the whole batch is then this context's. */
int  sqg_set_range_mode(sqg_ctx_t *ctx, int on);
int  sqg_skip_reads(sqg_ctx_t *ctx, int32_t n_reads, const int64_t *seq_len, const int32_t *worker);
int  sqg_batch_sample_range(sqg_ctx_t *ctx, int32_t n_reads, const int32_t *worker, int32_t lo, int32_t hi,
                            sqg_batch_t **out, sqg_sample_t *info);
int  sqg_batch_run_begin(sqg_ctx_t *ctx, sqg_batch_t *b, const uint32_t **d_counts);
int  sqg_batch_run_end(sqg_ctx_t *ctx, sqg_batch_t *b, const uint32_t *d_before, const uint32_t *d_after);

/* Page-locked host memory for the sqg_fetch_* destinations: the D2H copy of a batch's signal (2 B/sample, or ~1.3
 * with sqg_batch_compress) is what bounds a host that consumes the output (Measurement); into pinned
 * memory it runs at the link rate instead of through a staging buffer.  Plain malloc'd destinations keep working. */
void *sqg_host_alloc(size_t bytes);
void  sqg_host_free(void *p);

/* HBM streaming-store probe used by bench.py to state the measured write
 * ceiling next to the 8 TB/s spec figure: writes `bytes` of int16 `iters`
 * times and returns the average milliseconds per pass. */
int  sqg_probe_store_bandwidth(sqg_ctx_t *ctx, size_t bytes, int iters, float *ms_per_pass);

/* The few-worker paths hand the k-mer streams out with LDS atomics whose lanes are served in lane order (Kernels);
 * sqg_create measures that on the device and falls back to order-free kernels when it does not hold.  This entry repeats the
 * measurement at any size (`workgroups` wavefronts of `rounds` rounds of 1024 fetch-adds each): *mismatches <- fetch-adds whose
 * result differs from the serial one, *in_use <- 1 when the context's kernels rely on the property. */
int  sqg_probe_lds_order(sqg_ctx_t *ctx, int workgroups, int rounds, unsigned int *mismatches, int *in_use);

#ifdef __cplusplus
}
#endif
#endif
