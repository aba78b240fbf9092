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

#ifndef B5_ID_MAX
#define B5_ID_MAX 4096                    // (k_blow5.h's: the framing kernel's LDS header; repeated for the CPU backend, which compiles this file without the kernels)
#endif
struct sqg_blow5 {
    struct Shard { FILE* fp = nullptr; unsigned long long n_bytes = 0; };
    std::vector<Shard> sh;               // the files: one, or -- SQG_BLOW5_SHARDS(n), stored mode -- n of them (path.0.blow5 ...), each a BLOW5 file of its own
    FILE* fp = nullptr;                  // (= sh[0].fp)
    sqg_profile_t profile{};
    uint32_t flags = 0;
    int threads = 1;
    long long n_reads = 0;               // records written: the next read_number
    unsigned long long n_samples = 0;    // samples written: the next start_time (core->n_samples, src/sim.c:602)
    unsigned long long n_bytes = 0;      // file bytes so far (one file: = sh[0].n_bytes while it is being written; all files at close)
    bool stored = false;                 // SQG_BLOW5_STORED: the records' zlib streams are stored blocks (valid BLOW5, not the reference's bytes)
    std::thread bg;                      // stored mode: the write of the previous batch's records, running behind the caller
    int bg_bad = 0;                      // ... and whether it failed (read after the join)
    bool failed = false;                 // a write came up short: records of an unfinished batch are on disk, nothing more is written
    std::string err;
    sqg_ctx* bound = nullptr;            // the context whose pinned records the background write reads (registered there: sqg_destroy and the next filler of that buffer drain it)
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
    w->profile = *profile; w->flags = flags; w->stored = (flags & SQG_BLOW5_STORED) != 0;
    w->threads = threads > 0 ? threads : std::min(16, usable_cpus());
    const int ns = std::max(1, (int)(flags >> 24));
    if (ns > 1 && !w->stored) { delete w; return SQG_EINVAL; }     // (several files: the stored-block mode only)
    const std::string h = blow5_header(*profile, flags);
    auto fail = [&](const char* what, const std::string& pth) {
        fprintf(stderr, "[sqg] sqg_blow5_open: cannot %s %s: %s\n", what, pth.c_str(), strerror(errno));   // (no writer to ask: the reason goes to stderr, the code says I/O)
        for (auto& q : w->sh) if (q.fp) fclose(q.fp);
        delete w;
        return SQG_EIO;
    };
    for (int i = 0; i < ns; i++) {
        std::string pth = path;
        if (ns > 1) {                                               // x.blow5 -> x.<i>.blow5
            const size_t dot = pth.rfind(".blow5");
            const std::string tag = "." + std::to_string(i);
            if (dot != std::string::npos && dot + 6 == pth.size()) pth.insert(dot, tag); else pth += tag;
        }
        sqg_blow5::Shard q;
        q.fp = fopen(pth.c_str(), "wb");
        if (!q.fp) return fail("open for writing", pth);
        w->sh.push_back(q);
        if (fwrite(h.data(), 1, h.size(), q.fp) != h.size()) return fail("write the header of", pth);
        w->sh.back().n_bytes = h.size();
    }
    w->fp = w->sh[0].fp;
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
// (appended to `dst`, or -- at_ptr != null -- written there: 8 + 2 + 5 ceil(R / 65535) + R + 4 bytes for a raw record of R bytes)
static void blow5_record_stored(std::vector<uint8_t>& raw, std::vector<uint8_t>& dst, const sqg_blow5* w, const char* id, size_t id_len,
                                double offset, double median_before, const uint8_t* svb, uint64_t svb_bytes, int32_t read_number, uint64_t start_time,
                                uint8_t* at_ptr = nullptr) {
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
    uint8_t* o = at_ptr;
    if (!o) {
        const size_t at = dst.size();
        dst.resize(at + 8 + csize);
        o = dst.data() + at;
    }
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
// what a context calls before it refills (or frees) the pinned buffer this writer's background write reads
static void blow5_ctx_hook(void* w_, bool unbind) {
    sqg_blow5* w = static_cast<sqg_blow5*>(w_);
    if (!w) return;
    if (!blow5_drain(w)) w->failed = true;
    if (unbind && w->bound) { w->bound->b5_reader = nullptr; w->bound->b5_reader_drain = nullptr; w->bound->b5_reader_buf = -1; w->bound = nullptr; }
}
static void blow5_unbind(sqg_blow5* w) {
    if (w->bound && w->bound->b5_reader == w) { w->bound->b5_reader = nullptr; w->bound->b5_reader_drain = nullptr; w->bound->b5_reader_buf = -1; }
    w->bound = nullptr;
}
// stored mode: the batch's records -- record i at data + ro[i] - ro[0] -- dealt out to the files by ranges of reads, one stream of pwrite()s
// per file, side by side (different files do not share a lock: 11.6 / 22 / 37 GB/s into 2 / 4 / 8 files of a tmpfs, tools/io_probe.cpp)
static bool blow5_append_records(sqg_blow5* w, const uint8_t* data, const int64_t* ro, const int n, const bool async) {
    if (!blow5_drain(w)) return false;
    const int ns = (int)w->sh.size();
    struct Job { int fd; const uint8_t* p; size_t n; off_t base; };
    std::vector<Job> jobs;
    for (int q = 0; q < ns; q++) {
        const int lo = (int)((long long)n * q / ns), hi = (int)((long long)n * (q + 1) / ns);
        const size_t nb = (size_t)(ro[hi] - ro[lo]);
        if (!nb) continue;
        if (fflush(w->sh[(size_t)q].fp) != 0) return false;
        jobs.push_back(Job{fileno(w->sh[(size_t)q].fp), data + (ro[lo] - ro[0]), nb, (off_t)w->sh[(size_t)q].n_bytes});
        w->sh[(size_t)q].n_bytes += nb;
    }
    auto run = [w, jobs]() {
        std::vector<int> bad(jobs.size(), 0);
        std::vector<std::thread> th;
        for (size_t j = 1; j < jobs.size(); j++) {
            auto one = [&jobs, &bad, j] { if (!blow5_copy_out(jobs[j].fd, jobs[j].p, jobs[j].n, jobs[j].base)) bad[j] = 1; };
            try { th.emplace_back(one); } catch (const std::system_error&) { one(); }       // (no thread to be had: this one does it)
        }
        if (!jobs.empty() && !blow5_copy_out(jobs[0].fd, jobs[0].p, jobs[0].n, jobs[0].base)) bad[0] = 1;
        for (auto& t : th) t.join();
        for (int x : bad) if (x) w->bg_bad = 1;
    };
    if (async) {
        try { w->bg = std::thread(run); return true; } catch (const std::system_error&) {}  // (no thread to be had: the write happens here and now)
    }
    run();
    return w->bg_bad == 0;
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
    if (w->stored) {
        // the records' places are known before they are framed (a stored record's size is a function of its fields' sizes)
        const int tl = 30 + ((w->flags & SQG_ONT) ? 1 : 0);
        std::vector<int64_t> ro((size_t)n + 1, 0);
        for (int i = 0; i < n; i++) {
            const unsigned long long R = (unsigned long long)(2 + (id_off[i + 1] - id_off[i]) + 4 + 32 + 8) + (unsigned long long)(svb_off[i + 1] - svb_off[i]) + (unsigned long long)tl;
            ro[(size_t)i + 1] = ro[(size_t)i] + (int64_t)(8 + 2 + 5 * ((R + 65534) / 65535) + R + 4);
        }
        std::vector<uint8_t> all((size_t)ro[(size_t)n]);
        auto frame = [&](int t) {
            const int lo = (int)((long long)n * t / nth), hi = (int)((long long)n * (t + 1) / nth);
            std::vector<uint8_t> raw, none;
            for (int i = lo; i < hi; i++)
                blow5_record_stored(raw, none, w, read_ids + id_off[i], (size_t)(id_off[i + 1] - id_off[i]), offset[i], median_before[i],
                                    svb + svb_off[i], (uint64_t)(svb_off[i + 1] - svb_off[i]), (int32_t)(w->n_reads + i), start[(size_t)i], all.data() + ro[(size_t)i]);
        };
        if (nth == 1) frame(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < nth; t++) { try { th.emplace_back(frame, t); } catch (const std::system_error&) { frame(t); } }
            for (auto& t : th) t.join();
        }
        if (!blow5_append_records(w, all.data(), ro.data(), n, /*async=*/false)) {
            w->failed = true;
            w->err = std::string("sqg_blow5_write: short write (") + strerror(errno) + "): the file is incomplete";
            return SQG_EIO;
        }
        w->n_bytes += (unsigned long long)ro[(size_t)n]; w->n_reads += n; w->n_samples = run;
        return SQG_OK;
    }
    std::vector<std::vector<uint8_t>> outs((size_t)nth);
    std::vector<int> bad((size_t)nth, 0);
    auto work = [&](int t) {
        const int lo = (int)((long long)n * t / nth), hi = (int)((long long)n * (t + 1) / nth);
        std::vector<uint8_t> raw;
        std::vector<uint8_t>& dst = outs[(size_t)t];
        dst.reserve((size_t)(svb_off[hi] - svb_off[lo]) + (size_t)(hi - lo) * 160);
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
        for (int t = 0; t < nth; t++) { try { th.emplace_back(work, t); } catch (const std::system_error&) { work(t); } }
        for (auto& t : th) t.join();
    }
    for (int t = 0; t < nth; t++) if (bad[(size_t)t]) { w->err = "sqg_blow5_write: zlib failed"; return SQG_EINVAL; }   // (nothing written yet)
    for (int t = 0; t < nth; t++) {
        if (fwrite(outs[(size_t)t].data(), 1, outs[(size_t)t].size(), w->fp) != outs[(size_t)t].size()) {
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
    bool device_frames = w->stored;
    if (device_frames && read_ids && id_off)
        for (int i = 0; i < res.n_reads; i++) if (id_off[i + 1] - id_off[i] > B5_ID_MAX) { device_frames = false; break; }
    // (a read id longer than the framing kernel's LDS header, 4096 bytes: this batch is framed on the host below -- sqg_blow5_write takes ids
    // up to 65535 bytes in stored mode as well -- instead of failing mid-file)
    if (device_frames) {
        // the records come framed from the device (sqg_batch_blow5_records): one copy over PCIe, one write
        if (w->bound && w->bound != c) { blow5_ctx_hook(w, false); blow5_unbind(w); }      // (the writer moves to another context: nothing of the old one is read any more)
        const uint8_t* recs = nullptr; int64_t nb = 0; const int64_t* ro = nullptr;
        unsigned long long run = w->n_samples;
        for (int i = 0; i < res.n_reads; i++) run += (unsigned long long)(res.sig_off[i + 1] - res.sig_off[i]);
        if ((rc = sqg_batch_blow5_records(c, b, &w->profile, w->flags, read_ids, id_off, w->n_reads, w->n_samples, &recs, &nb, &ro))) { w->err = sqg_last_error(c); return rc; }
        if (nb > 0 && !blow5_append_records(w, recs, ro, res.n_reads, /*async=*/true)) {
            w->failed = true;
            w->err = std::string("sqg_blow5_write_batch: short write (") + strerror(errno) + "): the file is incomplete";
            return SQG_EIO;
        }
        if (nb > 0) { w->bound = c; c->b5_reader = w; c->b5_reader_buf = c->b5_flip; c->b5_reader_drain = &blow5_ctx_hook; }
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
    blow5_unbind(w);
    int rc = w->failed ? SQG_EIO : SQG_OK;                            // (a failed writer leaves no end marker: the file is not a valid BLOW5)
    unsigned long long total = 0;
    for (size_t q = 0; q < w->sh.size(); q++) {
        sqg_blow5::Shard& f = w->sh[q];
        if (!f.fp) continue;
        if (!w->stored) f.n_bytes = w->n_bytes;                      // (one file, written through the FILE)
        if (!w->failed) {
            // slow5_eof_fwrite, slow5.c:4206 (stored mode: the records went out with pwrite(): the marker goes behind them the same way)
            const bool ok = w->stored ? (fflush(f.fp) == 0 && blow5_copy_out(fileno(f.fp), reinterpret_cast<const uint8_t*>("5WOLB"), 5, (off_t)f.n_bytes))
                                      : fwrite("5WOLB", 1, 5, f.fp) == 5;
            if (!ok) rc = SQG_EIO; else f.n_bytes += 5;
        }
        if (fclose(f.fp) != 0) rc = SQG_EIO;
        total += f.n_bytes;
    }
    w->n_bytes = total;
    if (n_bytes) *n_bytes = (int64_t)w->n_bytes;
    delete w;
    return rc;
}
