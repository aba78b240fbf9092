/* examples/process_db_gpu.c -- the batch loop a C host writes against include/sqg.h.
 *
 * Plain C99, no HIP or torch headers: this is what replaces process_db() (src/sim.c:622-627) in the reference
 * (INTEGRATION.md has the same code embedded in the reference's own types).  It simulates `n` reads from a FASTA-less
 * toy genome with the device-side sampler, runs the signal path, compresses the signals with svb-zd on the device and
 * prints one line per read.  Build:  gcc -std=c99 -Iinclude examples/process_db_gpu.c -Lsquigulator_amd/csrc -lsqg_hip
 * (tests/test_abi.py compiles it with -fsyntax-only on every run and links it when a GPU is present). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sqg.h"

static void die(sqg_ctx_t *ctx, const char *what, int rc) {
    fprintf(stderr, "%s: %s (%s)\n", what, sqg_strerror(rc), ctx ? sqg_last_error(ctx) : "");
    exit(EXIT_FAILURE);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    const int k = 6, nk = 1 << (2 * k);

    /* -x dna-r9-prom (src/sim.c:55-150); a toy pore model: level_mean in [60,140), level_stdv in [1,4) */
    sqg_cfg_t cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = SQG_ABI_VERSION;
    const sqg_profile_t r9 = {2048, 4000, 450, 748.5801, -237.4102, 14.1575, 214.2890337, 18.0127916, 9.0, 4.0};
    cfg.profile = r9;
    cfg.amp_noise = 1.0f;
    cfg.kmer_size = k;
    sqg_kmer_t *model = malloc(sizeof *model * nk);
    for (int j = 0; j < nk; j++) { model[j].level_mean = 60.0f + (float)(j * 37 % 5120) / 64.0f; model[j].level_stdv = 1.0f + (float)(j * 11 % 192) / 64.0f; }
    cfg.model = model;
    cfg.seed = 42;
    cfg.num_workers = n; cfg.worker_lo = 0; cfg.worker_hi = n;      /* -t n -K n: one read per worker per batch */
    cfg.device = 0;
    cfg.mode = SQG_MODE_CERTIFIED;

    sqg_ctx_t *ctx = NULL;
    int rc = sqg_create(&cfg, &ctx);
    if (rc) die(NULL, "sqg_create", rc);
    /* this host never asks for sqg_get_timing: no phase events between the kernels (a few microseconds of idle GPU each) */
    if ((rc = sqg_set_phase_timing(ctx, 0))) die(ctx, "sqg_set_phase_timing", rc);

    /* a 50-kb toy contig kept on the device; reads are drawn there as gen_read() would (src/genread.c:243-281) */
    const int glen = 50000;
    char *genome = malloc(glen);
    unsigned s = 12345;
    for (int i = 0; i < glen; i++) { s = s * 1103515245u + 12345u; genome[i] = "ACGT"[(s >> 16) & 3]; }
    const int64_t contig_off[2] = {0, glen};
    sqg_genome_t g = {1, genome, contig_off, 2000, SQG_SAMPLE_DNA, 0, NULL, NULL};
    if ((rc = sqg_genome_load(ctx, &g))) die(ctx, "sqg_genome_load", rc);

    sqg_batch_t *b = NULL;
    sqg_sample_t smp;
    sqg_result_t res;
    if ((rc = sqg_batch_sample(ctx, n, NULL, &b, &smp))) die(ctx, "sqg_batch_sample", rc);
    if ((rc = sqg_batch_run(ctx, b))) die(ctx, "sqg_batch_run", rc);
    if ((rc = sqg_batch_wait(ctx, b, &res))) die(ctx, "sqg_batch_wait", rc);

    int16_t *sig = malloc(sizeof *sig * (size_t)(res.n_samples > 0 ? res.n_samples : 1));
    if ((rc = sqg_fetch_signal(ctx, b, sig))) die(ctx, "sqg_fetch_signal", rc);
    sqg_svb_t z;
    if ((rc = sqg_batch_compress(ctx, b, &z))) die(ctx, "sqg_batch_compress", rc);

    for (int i = 0; i < res.n_reads; i++) {
        const int64_t len = res.sig_off[i + 1] - res.sig_off[i];
        printf("read %d\tcontig %d:%d%c\t%d nt\t%lld samples\toffset %.4f\tfirst %d\tsvb-zd %lld bytes\n", i, smp.ref_idx[i],
               smp.ref_pos[i], smp.strand[i], smp.rlen[i], (long long)len, res.offset[i], len ? sig[res.sig_off[i]] : 0,
               (long long)(z.svb_off[i + 1] - z.svb_off[i]));
    }
    sqg_batch_free(ctx, b);
    sqg_destroy(ctx);
    free(sig); free(genome); free(model);
    return 0;
}
