// h_blow5.h -- native BLOW5 writer: header, record framing, zlib record compression (sqg_blow5_*)
// Host side of include/sqg.h; included by sqg_hip.hip (one translation unit with the kernels), in the order listed there.
// Nothing here touches the device or the context's internals: the batch is reached through the public entry points only.
//
// The other half of work_per_single_read (src/sim.c:604-640: slow5_encode per read, slow5_write_bytes in output_db): the file
// the reference writes through slow5lib for `-o x.blow5` -- zlib record compression, svb-zd signal compression
// (slow5lib/src/slow5.c:421-423) -- written without slow5lib.  The signal field of a record is the svb-zd encoding
// sqg_batch_compress made on the device; this file frames it:
//   file    = magic "BLOW5\1" | version 0.2.0 | record method (1: zlib) | num_read_groups u32 | signal method (1: svb-zd) |
//             zeros up to byte 64 | header size u32 | header text | records | "5WOLB"     (slow5_hdr_to_mem, slow5.c:948-1158)
//   header  = "@attr\tvalue\n" sorted by attribute (set_header_attributes, src/gensig.c:40-129), the type line and the
//             column line with the auxiliary fields of set_header_aux_fields (src/gensig.c:131-169)
//   record  = u64 compressed size | zlib(deflate level default, window 15, memLevel 8, one stream per record) of:
//             u16 len(read_id) | read_id | u32 read_group | f64 digitisation | f64 offset | f64 range | f64 sampling_rate |
//             u64 bytes of the compressed signal | those bytes | u64 1 | "0" (channel_number) | f64 median_before |
//             i32 read_number | u8 start_mux | u64 start_time [| u8 end_reason]    (slow5_rec_to_mem, slow5.c:3928-4072;
//             set_record_primary_fields / set_record_aux_fields, src/gensig.c:171-223)
// SQG_BLOW5_STORED in the flags of sqg_blow5_open: the same records, each in a zlib stream of STORED blocks (RFC 1951, BTYPE 00) instead of
// deflate's output -- a valid BLOW5 file with the reference's records that any slow5lib reads, 1.3 instead of 0.97 bytes per sample, not
// the reference's bytes -- framed on the device (k_blow5.h, sqg_batch_blow5_records) and written behind the caller's back while the next batch's
// records cross PCIe: the sink at the speed of the file system for hosts that do not need `cmp`-identity (zlib at 28 MB/s per thread is what bounds the default mode).
#pragma once
#include "h_cpus.h"

#include <zlib.h>
#include <unistd.h>

struct sqg_blow5 {
    FILE* fp = nullptr;
    sqg_profile_t profile{};
    uint32_t flags = 0;
    int threads = 1;
    long long n_reads = 0;               // records written: the next read_number
    unsigned long long n_samples = 0;    // samples written: the next start_time (core->n_samples, src/sim.c:602)
    unsigned long long n_bytes = 0;      // file bytes so far
    bool stored = false;                 // SQG_BLOW5_STORED: the records' zlib streams are stored blocks (valid BLOW5, not the reference's bytes)
    std::thread bg;                      // stored mode: the write of the previous batch's records, running behind the caller
    int bg_bad = 0;                      // ... and whether it failed (read after the join)
    bool failed = false;                 // a write came up short: records of an unfinished batch are on disk, nothing more is written
    std::string err;
};

extern "C" const char* sqg_blow5_last_error(const sqg_blow5_t* w) { return w ? w->err.c_str() : ""; }

static std::string blow5_header(const sqg_profile_t& p, uint32_t flags) {
    const bool rna = flags & SQG_RNA, r10 = flags & SQG_R10, ont = flags & SQG_ONT;
    const char* kit = rna ? (r10 ? "sqk-rna004" : "sqk-rna002") : (r10 ? "sqk-lsk114" : "sqk-lsk109");
    char freq[64];
    snprintf(freq, sizeof freq, "%d", (int)p.sample_rate);             // src/gensig.c:122-123
    std::string t;
    // slow5lib writes the attributes in sorted order (slow5_get_hdr_keys)
    t += "@asic_id\tasic_id_0\n";
    t += "@exp_start_time\t2022-07-20T00:00:00Z\n";
    t += std::string("@experiment_type\t") + (rna ? "rna" : "genomic_dna") + "\n";
    t += "@flow_cell_id\tFAN00000\n";
    t += "@run_id\trun_0\n";
    t += std::string("@sample_frequency\t") + freq + "\n";
    t += std::string("@sequencing_kit\t") + kit + "\n";
    t += "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\tchar*\tdouble\tint32_t\tuint8_t\tuint64_t";
    if (ont) t += "\tenum{unknown,partial,mux_change,unblock_mux_change,data_service_unblock_mux_change,signal_positive,signal_negative}";
    t += "\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\tchannel_number\tmedian_before\tread_number\tstart_mux\tstart_time";
    if (ont) t += "\tend_reason";
    t += "\n";
    std::string h("BLOW5\1", 6);
    const unsigned char fixed[] = {0, 2, 0, /*record: zlib*/ 1, /*num_read_groups*/ 1, 0, 0, 0, /*signal: svb-zd*/ 1};
    h.append(reinterpret_cast<const char*>(fixed), sizeof fixed);
    h.resize(64, '\0');
    const uint32_t hs = (uint32_t)t.size();
    h.append(reinterpret_cast<const char*>(&hs), 4);
    return h + t;
}

extern "C" int sqg_blow5_open(const char* path, const sqg_profile_t* profile, uint32_t flags, int32_t threads, sqg_blow5_t** out) {
    if (!path || !profile || !out) return SQG_EINVAL;
    *out = nullptr;
    if (!(profile->sample_rate > 0) || profile->sample_rate > 1000000000.0) return SQG_EINVAL;     // src/gensig.c:117-120
    sqg_blow5* w = new (std::nothrow) sqg_blow5();
    if (!w) return SQG_ENOMEM;
    w->fp = fopen(path, "wb");
    if (!w->fp) {                                                   // (no writer to ask: the reason goes to stderr, the code says I/O)
        fprintf(stderr, "[sqg] sqg_blow5_open: cannot open %s for writing: %s\n", path, strerror(errno));
        delete w;
        return SQG_EIO;
    }
    w->profile = *profile; w->flags = flags; w->stored = (flags & SQG_BLOW5_STORED) != 0;
    w->threads = threads > 0 ? threads : std::min(16, usable_cpus());
    const std::string h = blow5_header(*profile, flags);
    if (fwrite(h.data(), 1, h.size(), w->fp) != h.size()) {
        fprintf(stderr, "[sqg] sqg_blow5_open: cannot write the header of %s: %s\n", path, strerror(errno));
        fclose(w->fp); delete w;
        return SQG_EIO;
    }
    w->n_bytes = h.size();
    *out = w;
    return SQG_OK;
}

// one record, compressed, appended to `dst` (size prefix included); zs: the thread's deflate stream
static bool blow5_record(z_stream& zs, std::vector<uint8_t>& raw, std::vector<uint8_t>& dst, const sqg_blow5* w, const char* id, size_t id_len,
                         double offset, double median_before, const uint8_t* svb, uint64_t svb_bytes, int32_t read_number, uint64_t start_time) {
    raw.clear();
    auto put = [&](const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; raw.insert(raw.end(), q, q + n); };
    const uint16_t idl = (uint16_t)id_len;
    const uint32_t rg = 0;
    put(&idl, 2); put(id, id_len); put(&rg, 4);
    put(&w->profile.digitisation, 8); put(&offset, 8); put(&w->profile.range, 8); put(&w->profile.sample_rate, 8);
    put(&svb_bytes, 8); put(svb, (size_t)svb_bytes);
    const uint64_t one = 1; const char ch = '0'; const uint8_t mux = 0;
    put(&one, 8); put(&ch, 1);                                          // channel_number = "0"
    put(&median_before, 8); put(&read_number, 4); put(&mux, 1); put(&start_time, 8);
    if (w->flags & SQG_ONT) { const uint8_t end_reason = 0; put(&end_reason, 1); }
    if (deflateReset(&zs) != Z_OK) return false;
    const size_t at = dst.size();
    const uLong bound = deflateBound(&zs, (uLong)raw.size());
    dst.resize(at + 8 + bound);
    zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size();
    zs.next_out = dst.data() + at + 8; zs.avail_out = (uInt)bound;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) return false;
    const uint64_t csize = bound - zs.avail_out;
    memcpy(dst.data() + at, &csize, 8);
    dst.resize(at + 8 + csize);
    return true;
}

// RFC 1950 / 1951 by hand: 78 01 | stored blocks of at most 65535 bytes | Adler-32.  The same bytes k_blow5_frame writes on the device.
static void blow5_record_stored(std::vector<uint8_t>& raw, std::vector<uint8_t>& dst, const sqg_blow5* w, const char* id, size_t id_len,
                                double offset, double median_before, const uint8_t* svb, uint64_t svb_bytes, int32_t read_number, uint64_t start_time) {
    raw.clear();
    auto put = [&](const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; raw.insert(raw.end(), q, q + n); };
    const uint16_t idl = (uint16_t)id_len;
    const uint32_t rg = 0;
    put(&idl, 2); put(id, id_len); put(&rg, 4);
    put(&w->profile.digitisation, 8); put(&offset, 8); put(&w->profile.range, 8); put(&w->profile.sample_rate, 8);
    put(&svb_bytes, 8); put(svb, (size_t)svb_bytes);
    const uint64_t one = 1; const char ch = '0'; const uint8_t mux = 0;
    put(&one, 8); put(&ch, 1);
    put(&median_before, 8); put(&read_number, 4); put(&mux, 1); put(&start_time, 8);
    if (w->flags & SQG_ONT) { const uint8_t end_reason = 0; put(&end_reason, 1); }
    const size_t R = raw.size(), nb = (R + 65534) / 65535;
    const uint64_t csize = 2 + 5 * nb + R + 4;
    size_t at = dst.size();
    dst.resize(at + 8 + csize);
    uint8_t* o = dst.data() + at;
    memcpy(o, &csize, 8); o[8] = 0x78; o[9] = 0x01; o += 10;
    uint32_t a = 1, b2 = 0;
    for (size_t k = 0; k < nb; k++) {
        const size_t r0 = k * 65535, len = std::min<size_t>(65535, R - r0);
        o[0] = k + 1 == nb ? 1 : 0; o[1] = (uint8_t)len; o[2] = (uint8_t)(len >> 8); o[3] = (uint8_t)~len; o[4] = (uint8_t)(~len >> 8);
        memcpy(o + 5, raw.data() + r0, len);
        o += 5 + len;
    }
    for (size_t r = 0; r < R; ) {                                  // Adler-32, 5552 bytes between the reductions (zlib's NMAX)
        const size_t e = std::min(R, r + 5552);
        for (; r < e; r++) { a += raw[r]; b2 += a; }
        a %= 65521u; b2 %= 65521u;
    }
    const uint32_t ad = b2 << 16 | a;
    o[0] = (uint8_t)(ad >> 24); o[1] = (uint8_t)(ad >> 16); o[2] = (uint8_t)(ad >> 8); o[3] = (uint8_t)ad;
}

// `n` bytes at offset `base` of the file: ONE stream of pwrite()s.  Measured on the GPU boxes (tools/io_probe.cpp, 350 MB appended to
// a file in /dev/shm): one pwrite 6.9 GB/s, sixteen side by side 6.3 (they serialise on the inode's lock), sixteen threads storing
// through a shared mapping 3.3 (a page fault per 4 KiB) -- a single file takes what one writer gives it.
static bool blow5_copy_out(const int fd, const uint8_t* data, const size_t n, const off_t base) {
    size_t lo = 0;
    while (lo < n) {
        const ssize_t k = pwrite(fd, data + lo, std::min<size_t>(n - lo, (size_t)1 << 30), base + (off_t)lo);
        if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; }
        lo += (size_t)k;
    }
    return true;
}
// the previous batch's write (stored mode, sqg_blow5_write_batch) has finished; false: it failed
static bool blow5_drain(sqg_blow5* w) {
    if (w->bg.joinable()) w->bg.join();
    return w->bg_bad == 0;
}
// `n` bytes at the end of the file.  Stored mode: the bytes are many (1.3 per sample) and nothing is left to do to them, so the write itself
// is what takes the time; `async`: it runs behind the caller -- `data` stays valid until the next call's drain -- so that
// the next batch's PCIe copy overlaps it.  The other mode goes through the FILE.
static bool blow5_append(sqg_blow5* w, const uint8_t* data, size_t n, bool async = false) {
    if (!w->stored) return fwrite(data, 1, n, w->fp) == n;
    if (!blow5_drain(w)) return false;
    if (fflush(w->fp) != 0) return false;
    const int fd = fileno(w->fp);
    const off_t base = (off_t)w->n_bytes;
    if (!async) return blow5_copy_out(fd, data, n, base);
    w->bg = std::thread([w, fd, data, n, base] { if (!blow5_copy_out(fd, data, n, base)) w->bg_bad = 1; });
    return true;
}

extern "C" int sqg_blow5_write(sqg_blow5_t* w, int32_t n, const char* read_ids, const int64_t* id_off, const double* offset,
                               const double* median_before, const int64_t* sig_off, const uint8_t* svb, const int64_t* svb_off) {
    if (!w || !w->fp || n < 0) return SQG_EINVAL;
    if (!blow5_drain(w)) { w->failed = true; w->err = "sqg_blow5_write: the previous batch's write failed: the file is incomplete"; }
    if (w->failed) { w->err = "sqg_blow5_write: the writer failed earlier (the file is incomplete): close it"; return SQG_EIO; }
    if (n == 0) return SQG_OK;
    if (!read_ids || !id_off || !offset || !median_before || !sig_off || !svb || !svb_off) { w->err = "sqg_blow5_write: null argument"; return SQG_EINVAL; }
    for (int i = 0; i < n; i++) {
        const int64_t il = id_off[i + 1] - id_off[i];
        if (il < 0 || il > 65535 || svb_off[i + 1] < svb_off[i] || sig_off[i + 1] < sig_off[i]) { w->err = "sqg_blow5_write: bad offsets"; return SQG_EINVAL; }
        // one deflate call per record: the raw record (encoding + ~100 B of fields + the id) and its deflateBound must fit zlib's 32-bit
        // avail_in / avail_out (deflateBound(n) <= n + n/1000 + 64 KiB for these parameters)
        if (svb_off[i + 1] - svb_off[i] > 0xf0000000LL - il) { w->err = "sqg_blow5_write: record too large"; return SQG_EOVERFLOW; }
    }
    // start_time of read i = samples of every read before it (src/sim.c:602, in read order)
    std::vector<unsigned long long> start((size_t)n);
    unsigned long long run = w->n_samples;
    for (int i = 0; i < n; i++) { start[(size_t)i] = run; run += (unsigned long long)(sig_off[i + 1] - sig_off[i]); }
    const int nth = std::max(1, std::min(w->threads, n));
    std::vector<std::vector<uint8_t>> outs((size_t)nth);
    std::vector<int> bad((size_t)nth, 0);
    auto work = [&](int t) {
        const int lo = (int)((long long)n * t / nth), hi = (int)((long long)n * (t + 1) / nth);
        std::vector<uint8_t> raw;
        std::vector<uint8_t>& dst = outs[(size_t)t];
        dst.reserve((size_t)(svb_off[hi] - svb_off[lo]) + (size_t)(hi - lo) * 160);
        if (w->stored) {
            for (int i = lo; i < hi; i++)
                blow5_record_stored(raw, dst, w, read_ids + id_off[i], (size_t)(id_off[i + 1] - id_off[i]), offset[i], median_before[i],
                                    svb + svb_off[i], (uint64_t)(svb_off[i + 1] - svb_off[i]), (int32_t)(w->n_reads + i), start[(size_t)i]);
            return;
        }
        z_stream zs; memset(&zs, 0, sizeof zs);
        // zlib_init_deflate, slow5lib/src/slow5_press.c:789-800
        if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, MAX_WBITS, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad[(size_t)t] = 1; return; }
        for (int i = lo; i < hi; i++)
            if (!blow5_record(zs, raw, dst, w, read_ids + id_off[i], (size_t)(id_off[i + 1] - id_off[i]), offset[i], median_before[i],
                              svb + svb_off[i], (uint64_t)(svb_off[i + 1] - svb_off[i]), (int32_t)(w->n_reads + i), start[(size_t)i])) { bad[(size_t)t] = 1; break; }
        deflateEnd(&zs);
    };
    if (nth == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nth; t++) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    for (int t = 0; t < nth; t++) if (bad[(size_t)t]) { w->err = "sqg_blow5_write: zlib failed"; return SQG_EINVAL; }   // (nothing written yet)
    for (int t = 0; t < nth; t++) {
        if (!blow5_append(w, outs[(size_t)t].data(), outs[(size_t)t].size())) {
            // part of the batch's records is on disk: a retry would duplicate them.  The writer is dead from here on.
            w->failed = true;
            w->err = std::string("sqg_blow5_write: short write (") + strerror(errno) + "): the file is incomplete";
            return SQG_EIO;
        }
        w->n_bytes += outs[(size_t)t].size();
    }
    w->n_reads += n;
    w->n_samples = run;
    return SQG_OK;
}

// the batch's records: svb-zd on the device (sqg_batch_compress), ~1.3 B/sample over PCIe, framing and zlib on the host threads
extern "C" int sqg_blow5_write_batch(sqg_blow5_t* w, sqg_ctx_t* c, sqg_batch_t* b, const char* read_ids, const int64_t* id_off) {
    if (!w || !c || !b) return SQG_EINVAL;
    if (w->failed) { w->err = "sqg_blow5_write_batch: the writer failed earlier (the file is incomplete): close it"; return SQG_EIO; }
    sqg_result_t res;
    int rc = sqg_batch_wait(c, b, &res);
    if (rc) { w->err = sqg_last_error(c); return rc; }
    if (w->stored) {
        // the records come framed from the device (sqg_batch_blow5_records): one copy over PCIe, one write
        const uint8_t* recs = nullptr; int64_t nb = 0;
        unsigned long long run = w->n_samples;
        for (int i = 0; i < res.n_reads; i++) run += (unsigned long long)(res.sig_off[i + 1] - res.sig_off[i]);
        if ((rc = sqg_batch_blow5_records(c, b, &w->profile, w->flags, read_ids, id_off, w->n_reads, w->n_samples, &recs, &nb, nullptr))) { w->err = sqg_last_error(c); return rc; }
        if (nb > 0 && !blow5_append(w, recs, (size_t)nb, /*async=*/true)) {
            w->failed = true;
            w->err = std::string("sqg_blow5_write_batch: short write (") + strerror(errno) + "): the file is incomplete";
            return SQG_EIO;
        }
        w->n_bytes += (unsigned long long)nb; w->n_reads += res.n_reads; w->n_samples = run;
        return SQG_OK;
    }
    sqg_svb_t sv;
    if ((rc = sqg_batch_compress(c, b, &sv))) { w->err = sqg_last_error(c); return rc; }
    uint8_t* host = (uint8_t*)sqg_host_alloc((size_t)std::max<int64_t>(sv.n_bytes, 1));
    if (!host) { w->err = "sqg_blow5_write_batch: no pinned host memory for the batch's encodings"; return SQG_ENOMEM; }
    rc = sqg_fetch_svb(c, b, host);
    if (rc == SQG_OK) rc = sqg_blow5_write(w, res.n_reads, read_ids, id_off, res.offset, res.median_before, res.sig_off, host, sv.svb_off);
    else w->err = sqg_last_error(c);
    sqg_host_free(host);
    return rc;
}

extern "C" int sqg_blow5_close(sqg_blow5_t* w, int64_t* n_bytes) {
    if (!w) return SQG_EINVAL;
    if (!blow5_drain(w)) w->failed = true;                            // (stored mode: the last batch's records may still be on their way)
    int rc = w->failed ? SQG_EIO : SQG_OK;                            // (a failed writer leaves no end marker: the file is not a valid BLOW5)
    if (w->fp) {
        if (!w->failed) {
            if (!blow5_append(w, reinterpret_cast<const uint8_t*>("5WOLB"), 5)) rc = SQG_EIO;   // slow5_eof_fwrite, slow5.c:4206
            else w->n_bytes += 5;
        }
        if (fclose(w->fp) != 0) rc = SQG_EIO;
    }
    if (n_bytes) *n_bytes = (int64_t)w->n_bytes;
    delete w;
    return rc;
}
