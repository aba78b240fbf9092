// k_blow5.h -- BLOW5 records framed on the device: slow5_rec_to_mem's layout around the svb-zd bytes, inside a zlib stream of STORED blocks
// Part of the device code of the per-read signal path; included through sqg_kernels.h (see there for the overview).
//
// The reference compresses a record inside the worker (src/sim.c:607 -> slow5lib/src/slow5.c:3815-4075 -> slow5_press.c:794: deflate,
// 28 MB/s per host thread): end to end that is what bounds a host that writes BLOW5.  A zlib stream need not compress: RFC 1951
// "stored" blocks (BTYPE 00) carry the bytes as they are, and any inflate -- slow5lib's included -- reads them.  So the record
//   u64 compressed size | 78 01 | { 00/01 | LEN u16 | ~LEN u16 | <= 65535 bytes }* | adler32 (big endian)
// is written here, per read, straight from the device's svb-zd encoding (k_svb.h): the file holds the same records as the reference's
// (field for field, sample for sample: tests read it back through the reference's own slow5lib) in other bytes, 1.3 instead of 0.97
// per sample, and the host's part is one copy and one write.  The raw record (slow5.c:3928-4072; src/gensig.c:171-223):
//   u16 len(read_id) | read_id | u32 read_group | f64 digitisation | f64 offset | f64 range | f64 sampling_rate |
//   u64 bytes of the compressed signal | those bytes | u64 1 | "0" | f64 median_before | i32 read_number | u8 start_mux |
//   u64 start_time [| u8 end_reason]
#pragma once

#define B5_ID_MAX 4096                    // read ids longer than this: the host-zlib mode (the reference's ids are ~40-80 bytes)
#define B5_BLOCK 65535u                   // bytes per stored block
#define B5_ADLER 65521u

struct Blow5Params {
    const uint8_t* svb;                   // the batch's svb-zd encodings (k_svb_encode) ...
    const long long* svb_off;             // [n+1] ... and their offsets
    const long long* sig_off;             // [n+1] samples before each read of the batch (start_time)
    const uint8_t* ids;                   // read ids, back to back
    const long long* id_off;              // [n+1]
    const double* offset;                 // [n]
    const double* median;                 // [n]
    const long long* rec_off;             // [n+1] where each record goes in `out`
    uint8_t* out;
    double digitisation, range, sample_rate;
    long long read_number0;               // read_number of the batch's first read
    unsigned long long start_time0;       // samples written before this batch
    int ont;                              // the end_reason field (--ont-friendly)
    int n;
};

// bytes of a record whose raw form has R bytes (host and device)
__host__ __device__ static inline unsigned long long b5_stored_size(unsigned long long R) {
    const unsigned long long nblk = R ? (R + B5_BLOCK - 1) / B5_BLOCK : 1;
    return 8 + 2 + 5 * nblk + R + 4;
}

// grid: reads, 256 threads
__global__ __launch_bounds__(256) void k_blow5_frame(const Blow5Params P) {
    __shared__ uint8_t hdr[2 + B5_ID_MAX + 4 + 32 + 8];
    __shared__ uint8_t trl[32];
    __shared__ unsigned long long red_a[4], red_b[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    if (i >= P.n) return;
    const unsigned long long idl = (unsigned long long)(P.id_off[i + 1] - P.id_off[i]);
    const unsigned long long S = (unsigned long long)(P.svb_off[i + 1] - P.svb_off[i]);
    const unsigned long long hl = 2 + idl + 4 + 32 + 8, tl = 8 + 1 + 8 + 4 + 1 + 8 + (P.ont ? 1 : 0);
    const unsigned long long R = hl + S + tl;
    const unsigned long long nblk = (R + B5_BLOCK - 1) / B5_BLOCK;
    uint8_t* const out = P.out + P.rec_off[i];
    auto put = [&](uint8_t* dst, const void* src, int nb) { const uint8_t* q = (const uint8_t*)src; for (int j = 0; j < nb; j++) dst[j] = q[j]; };
    for (unsigned long long j = tid; j < idl; j += 256) hdr[2 + j] = P.ids[P.id_off[i] + (long long)j];
    if (tid == 0) {
        const uint16_t l16 = (uint16_t)idl; const uint32_t rg = 0;
        put(hdr, &l16, 2);
        uint8_t* h = hdr + 2 + idl;
        put(h, &rg, 4); put(h + 4, &P.digitisation, 8); put(h + 12, &P.offset[i], 8); put(h + 20, &P.range, 8); put(h + 28, &P.sample_rate, 8);
        put(h + 36, &S, 8);
        const unsigned long long one = 1; const uint8_t ch = '0', mux = 0, er = 0;
        const int32_t rn = (int32_t)(P.read_number0 + i);
        const unsigned long long st = P.start_time0 + (unsigned long long)P.sig_off[i];
        put(trl, &one, 8); trl[8] = ch; put(trl + 9, &P.median[i], 8); put(trl + 17, &rn, 4); trl[21] = mux; put(trl + 22, &st, 8);
        if (P.ont) trl[30] = er;
        const unsigned long long csize = 2 + 5 * nblk + R + 4;
        put(out, &csize, 8);
        out[8] = 0x78; out[9] = 0x01;                              // CMF: deflate, 32 KiB window; FLG: no dictionary, level 0, check bits
    }
    __syncthreads();
    // the block headers
    for (unsigned long long bk = tid; bk < nblk; bk += 256) {
        const unsigned long long r0 = bk * B5_BLOCK;
        const uint32_t len = (uint32_t)((R - r0 < B5_BLOCK) ? R - r0 : B5_BLOCK);
        uint8_t* q = out + 10 + bk * (B5_BLOCK + 5);
        q[0] = bk + 1 == nblk ? 1 : 0;
        q[1] = (uint8_t)len; q[2] = (uint8_t)(len >> 8); q[3] = (uint8_t)~len; q[4] = (uint8_t)(~len >> 8);
    }
    // the bytes, 16 per thread and step, and their Adler-32 sums: A = 1 + sum d, B = R + sum (R - r) d  (mod 65521)
    const uint8_t* const svb = P.svb + P.svb_off[i];
    unsigned long long sa = 0, sb = 0;
    for (unsigned long long r0 = (unsigned long long)tid * 16; r0 < R; r0 += 256 * 16) {
        const unsigned long long r1 = (r0 + 16 < R) ? r0 + 16 : R;
        uint8_t v[16];
        const bool inner = r0 >= hl && r1 <= hl + S && r1 - r0 == 16;
        if (inner) __builtin_memcpy(v, svb + (r0 - hl), 16);
        else for (unsigned long long r = r0; r < r1; r++) v[r - r0] = r < hl ? hdr[r] : r < hl + S ? svb[r - hl] : trl[r - hl - S];
        const unsigned long long p0 = 10 + 5 * (r0 / B5_BLOCK + 1) + r0;
        if (inner && r0 / B5_BLOCK == (r1 - 1) / B5_BLOCK) __builtin_memcpy(out + p0, v, 16);
        else for (unsigned long long r = r0; r < r1; r++) out[10 + 5 * (r / B5_BLOCK + 1) + r] = v[r - r0];
        unsigned long long a16 = 0, b16 = 0;                        // (16 bytes: a16 <= 4080, b16 <= 16 * 255 * 16)
#pragma unroll
        for (int j = 0; j < 16; j++) { const unsigned long long d = (r0 + j < r1) ? v[j] : 0; a16 += d; b16 += d * (unsigned long long)(16 - j); }
        // sum (R - r) d over the run = (R - r0 - 16) * a16 + b16
        sa += a16;
        sb = (sb + ((R - r0 - 16 + B5_ADLER) % B5_ADLER) * a16 + b16) % B5_ADLER;     // (R - r0 - 16 may be negative by < 16 for the last run: + 65521 first)
    }
    sa %= B5_ADLER;
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
    if ((tid & 63) == 0) { red_a[tid >> 6] = sa; red_b[tid >> 6] = sb; }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long A = (1 + red_a[0] + red_a[1] + red_a[2] + red_a[3]) % B5_ADLER;
        const unsigned long long B = (R % B5_ADLER + red_b[0] + red_b[1] + red_b[2] + red_b[3]) % B5_ADLER;
        const uint32_t ad = (uint32_t)(B << 16 | A);
        uint8_t* q = out + 10 + 5 * nblk + R;
        q[0] = (uint8_t)(ad >> 24); q[1] = (uint8_t)(ad >> 16); q[2] = (uint8_t)(ad >> 8); q[3] = (uint8_t)ad;
    }
}
