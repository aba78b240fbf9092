/* oracle/ref_blow5_dump.c -- TEST INFRASTRUCTURE ONLY.  Reads a SLOW5 / BLOW5 file through the REFERENCE's own slow5lib
 * (compiled where it lies under /root/reference by oracle/Makefile `ref`) and prints every header attribute and every
 * record -- primary fields, the auxiliary fields squigulator writes (src/gensig.c:131-223), and the signal, sample for sample or as
 * length + FNV-1a hash -- as text.  Two files that hold the same records dump identically, whatever their record compression
 * (the product's stored-block zlib streams against the reference's deflate).
 *   usage: ref_blow5_dump FILE [hash]                                                                       */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <slow5/slow5.h>

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s FILE [hash]\n", argv[0]); return 2; }
    const int hash_only = argc > 2 && strcmp(argv[2], "hash") == 0;
    slow5_file_t *sp = slow5_open(argv[1], "r");
    if (!sp) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    static const char *attrs[] = {"asic_id", "exp_start_time", "experiment_type", "flow_cell_id", "run_id", "sample_frequency", "sequencing_kit"};
    for (size_t i = 0; i < sizeof attrs / sizeof attrs[0]; i++) {
        const char *v = slow5_hdr_get(attrs[i], 0, sp->header);
        printf("@%s\t%s\n", attrs[i], v ? v : "(none)");
    }
    printf("num_read_groups\t%u\n", (unsigned)sp->header->num_read_groups);
    slow5_rec_t *rec = NULL;
    int ret;
    long n = 0;
    while ((ret = slow5_get_next(&rec, sp)) >= 0) {
        int err = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0, e6 = 0;
        uint64_t chl = 0;
        const char *ch = slow5_aux_get_string(rec, "channel_number", &chl, &err);
        const double med = slow5_aux_get_double(rec, "median_before", &e2);
        const int32_t rn = slow5_aux_get_int32(rec, "read_number", &e3);
        const uint8_t mux = slow5_aux_get_uint8(rec, "start_mux", &e4);
        const uint64_t st = slow5_aux_get_uint64(rec, "start_time", &e5);
        const uint8_t er = slow5_aux_get_enum(rec, "end_reason", &e6);
        printf("%s\t%u\t%a\t%a\t%a\t%a\t%" PRIu64 "\t%s\t%a\t%d\t%u\t%" PRIu64 "\t%s", rec->read_id, (unsigned)rec->read_group, rec->digitisation, rec->offset,
               rec->range, rec->sampling_rate, rec->len_raw_signal, err ? "(none)" : ch, e2 ? 0.0 : med, e3 ? -1 : rn, e4 ? 255u : (unsigned)mux, e5 ? 0 : st,
               e6 ? "-" : "");
        if (!e6) printf("%u", (unsigned)er);
        if (hash_only) {
            uint64_t h = 1469598103934665603ULL;
            for (uint64_t i = 0; i < rec->len_raw_signal; i++) {
                const uint16_t v = (uint16_t)rec->raw_signal[i];
                h = (h ^ (v & 0xffu)) * 1099511628211ULL;
                h = (h ^ (v >> 8)) * 1099511628211ULL;
            }
            printf("\tfnv1a:%016" PRIx64 "\n", h);
        } else {
            putchar('\t');
            for (uint64_t i = 0; i < rec->len_raw_signal; i++) printf(i ? ",%d" : "%d", (int)rec->raw_signal[i]);
            putchar('\n');
        }
        n++;
    }
    slow5_rec_free(rec);
    if (ret != SLOW5_ERR_EOF) { fprintf(stderr, "slow5_get_next failed after %ld records: %d\n", n, ret); slow5_close(sp); return 1; }
    printf("records\t%ld\n", n);
    slow5_close(sp);
    return 0;
}
