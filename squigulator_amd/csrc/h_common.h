// h_common.h -- context and batch objects, error macros, the host-side Lehmer / Box-Muller restatement, shared constants
// Host side of include/sqg.h; included by sqg_hip.hip (one translation unit with the kernels), in the order listed there.
#pragma once

// Development knobs -- A/B kernels, forced code paths, fault injection, timing-only ablations that produce wrong results -- are read
// from the environment ONLY by the -DSQG_DEV build (libsqg_hip_dev.so: tests that have to force a path, tools/).  The release
// library (libsqg_hip.so) does not contain their names: a host program's environment cannot steer it (tests/test_release_build.py).
// SQG_VERBOSE, SQG_DEBUG_SYNC and SQG_STAGE_TIMING (diagnostics on stderr, results untouched) stay in both.
#if defined(SQG_DEV)
#define SQG_DEV_ENV(name) getenv(name)
#else
#define SQG_DEV_ENV(name) (static_cast<const char*>(nullptr))
#endif
static inline int dev_env_int(const char* v, int dflt) { return v ? atoi(v) : dflt; }

// sha256 (16 hex digits) over the sources this library was built from (squigulator_amd/build.py: source_hash()), stamped by the build;
// build.py finds it by its marker without loading the library, bench.py reads it through sqg_build_info()
#ifndef SQG_SOURCE_HASH
#define SQG_SOURCE_HASH "unstamped"
#endif
extern "C" const char sqg_source_hash_marker[] = "SQG_SOURCE_HASH=" SQG_SOURCE_HASH ";";
#if defined(SQG_DEV)
static const char kBuildInfo[] = "source_hash=" SQG_SOURCE_HASH ";dev=1";
#else
static const char kBuildInfo[] = "source_hash=" SQG_SOURCE_HASH ";dev=0";
#endif
extern "C" const char* sqg_build_info(void) { return kBuildInfo; }

#include "h_cpus.h"      // usable_cpus()

static uint32_t lcg_pow(uint32_t base, unsigned long long e) {
    uint32_t r = 1, b = base;
    while (e) { if (e & 1) r = lcg_mul(r, b); b = lcg_mul(b, b); e >>= 1; }
    return r;
}

// A few helper threads that live as long as the context (staging shares the per-read libm draws of a batch with them): starting a
// thread costs 0.4 ms in the containers this runs in -- a third of the work it would take over -- waking a sleeping one less.
struct HostPool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, done_cv;
    std::vector<std::function<void()>> jobs;       // not yet taken
    int pending = 0;                               // taken or not, not yet finished
    bool stop = false;
    void worker() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || !jobs.empty(); });
            if (stop) return;
            std::function<void()> job = std::move(jobs.back());
            jobs.pop_back();
            lk.unlock();
            job();
            lk.lock();
            if (--pending == 0) done_cv.notify_all();
        }
    }
    // hands `js` to the helpers (started on first use, as many as the largest request so far) and returns; wait() blocks until they are done
    void post(std::vector<std::function<void()>> js) {
        std::unique_lock<std::mutex> lk(m);
        while (th.size() < js.size()) th.emplace_back([this] { worker(); });
        pending += (int)js.size();
        for (auto& j : js) jobs.push_back(std::move(j));
        lk.unlock();
        cv.notify_all();
    }
    void wait() { std::unique_lock<std::mutex> lk(m); done_cv.wait(lk, [&] { return pending == 0; }); }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct sqg_ctx {
    HostPool pool_threads;
    struct DrawAhead* draw_ahead = nullptr;        // few workers: their per-read scalar draws, made ahead of staging by a thread of its own (below)
    sqg_cfg_t cfg;
    int k = 0, num_kmer = 0, T = 0, wlo = 0, whi = 0, nw = 0;
    hipStream_t stream = nullptr;
    uint32_t* d_rows = nullptr;
    float2* d_model = nullptr;
    uint32_t* d_pow = nullptr;
    std::vector<uint32_t> h_pow;                   // the jump tables on the host (staging: a worker's time stream moves past a read's events)
    // a^(2n) from three table look-ups (k_common.h: lcg_jump2) -- square-and-multiply took 100 ns of staging's 140 per read
    uint32_t jump2(unsigned long long n) const {
        if (n >> 32) return lcg_pow(lcg_mul(LCG_A, LCG_A), n);
        uint32_t r = h_pow[2 * POW_N + (size_t)(n & (POW_N - 1))];
        const uint32_t hi = (uint32_t)(n >> 10) & (POW_N - 1), hi2 = (uint32_t)(n >> 20);
        if (hi) r = lcg_mul(r, h_pow[3 * POW_N + hi]);
        if (hi2) r = lcg_mul(r, h_pow[4 * POW_N + hi2]);
        return r;
    }
    unsigned int* d_err = nullptr;                 // the read sampler's error word (sqg_batch_sample* read and clear it synchronously);
                                                   // the kernels of a batch report into the batch's own word
    unsigned long long fix_tickets = 0;            // k_fixup launches so far: the tag of the launch's list entries
    unsigned long long scan_tickets = 0;           // k_scan launches so far: every launch gets a ticket of its own
    // Everything a batch's kernels write lives in one of two SLOTS (batch seq & 1): a batch's results stay valid while
    // the next one runs (sqg_batch_wait / sqg_fetch_* of batch i do not wait for batch i+1), and with SQG_OVERLAP=1 the
    // event kernels of batch i+1 (stream) run while the sample kernels of batch i (stream2) are still busy.
    struct Slot {
        int16_t* d_sig = nullptr; size_t sig_cap = 0;
        long long* d_sigoff = nullptr; size_t reads_cap = 0;
        FixEntry* d_fix = nullptr; size_t fix_cap = 0;
        unsigned int* d_fix_count = nullptr;       // [0] fix-up entries, [1] slow tiles, [2] entries that went through the lean kernel's lists
        FixEntry* d_fix_sh = nullptr; size_t fix_sh_cap = 0;   // the lean kernel's FIX_SHARDS lists (entries in all; sized by the batch)
        unsigned int fix_sh_per = 0;               // ... entries per list
        unsigned int* d_fix_sh_count = nullptr;    // ... their counters, FIX_SHARD_STRIDE words apart
        uint2* d_evrec = nullptr; size_t evrec_cap = 0;
        int* d_slow = nullptr; size_t slow_cap = 0;
        ItemDesc* d_items = nullptr; size_t items_cap = 0;          // [n_stiles] work items of the lean kernel (k_items)
        uint32_t* d_part = nullptr; size_t part_cap = 0;            // [n_events] bucketed events (k_part.h).  Per slot: with one partition
                                                                    // (k <= 6) the sample kernels -- the generic one and the fix-ups run on
                                                                    // fix_stream, next to the following batch's event pass -- read rank and dwell from it
        uint32_t* d_lbase = nullptr; size_t lbase_cap = 0;          // [n_links][PART_MAX] first slot of every (link, partition): written by the scatter
                                                                    // pass, read by the sample kernels (SigParams.evrec32)
        int* d_tile_link = nullptr; size_t tile_link_cap = 0;       // [n_tiles] the link of every 64-event tile (same)
        uint32_t* d_part_state = nullptr; size_t part_state_cap = 0;   // [n_events] k > 6, split chains (k_part.h): stream state at each
                                                                    // bucketed event; read by the sample kernels like evrec
        // placement calibration (h_run.h, place_calibrate): the scatter pass of the slot's previous batch between two events, how often `evrec` was
        // allocated anew, the last measurement
        hipEvent_t cal_a = nullptr, cal_b = nullptr; bool cal_pending = false; long long cal_events = 0; int cal_tries = 0; float cal_ps = 0.f;
        uint2* cal_prev = nullptr; float cal_prev_ps = 0.f;   // the allocation before the candidate, until the candidate has been measured
        unsigned long long gen = 0;                // bumped whenever a batch starts writing the slot's buffers
        hipEvent_t done = nullptr;                 // recorded after the slot's last kernel (fix-ups included)
        hipEvent_t sampled = nullptr;              // recorded on stream2 after the slot's sample kernels, before the fix-ups
    } slot[2];
    // What a batch's FIRST event pass writes -- the dwells, the first sample of every 64-event tile, the reads' sample totals -- lives in
    // one of THREE sets (batch run index % 3): the first pass of batch i+1 may run inside the launch sequence of batch i (k_part_hand_count,
    // h_run.h: precount), while batch i-1's dwells -- sqg_fetch_dwell, sqg_result_t.d_dwell -- are still promised to the host.
    struct CountSet {
        uint16_t* d_dwell = nullptr; size_t dwell_cap = 0;
        uint32_t* d_tile_so = nullptr; size_t tile_cap = 0;
        unsigned long long* d_seglen = nullptr; size_t seglen_cap = 0;
        size_t seglen_dirty = 0;      // reads whose seglen words may be non-zero (what a batch leaves behind unless its k_fixup zeroes them)
        unsigned long long gen = 0;   // bumped whenever a batch starts writing the set
    } cset[3];
    int num_cu = 256;                              // compute units of the device
    unsigned int* d_zero = nullptr;                // one word that is always zero ("no slices": k_part_hand_count as a counting-only launch, development builds)
    std::deque<sqg_batch*> staged_q;               // staged, not yet run, in staging order (the batch behind the one being run: precount)
    hipStream_t stream2 = nullptr;                 // the sample kernels (k_samples_lean, generic); == stream unless SQG_OVERLAP=1
    hipStream_t fix_stream = nullptr;              // the FP64 fix-ups of batch i (two small kernels) run next to k_events of batch i+1
    unsigned long long* d_scan_part = nullptr; size_t scan_part_cap = 0;   // k_scan: {ticket, total} per workgroup
    uint32_t* d_link_rows = nullptr; size_t link_rows_cap = 0;   // split chains: one row per link of the running batch
    // k > 6, split chains (k_part.h), buffers of the running batch
    uint32_t* d_pcnt[2] = {nullptr, nullptr}; size_t pcnt_cap[2] = {0, 0};   // [n_part][n_links] counts, then offsets; one per run-index parity: the first
                                                                 // pass of batch i+1 (precount) fills its own while batch i's is still in use, and
                                                                 // either can be made larger before a batch's first launch without losing the other
    uint32_t* d_slice = nullptr; size_t slice_cap = 0;           // {slice_lo, slice_hi}[max_slices], pfirst[n_pairs + 1], pstart, ptotal [n_pairs] (k_part.h)
    uint32_t* d_phist = nullptr; size_t phist_cap = 0;           // [max_slices][PART_SUB]
    double row_bound = 0;                          // k > 6: upper bound of any sample count held in d_rows
    bool range_mode = false;                       // range sharding (sqg_set_range_mode): every batch is cut into links and run in two phases
    uint32_t* d_xcounts = nullptr; size_t xcounts_cap = 0;       // [nw][num_kmer] samples the running batch draws per stream (sqg_batch_run_begin)
    // device block, pinned offsets and events of freed batches, kept for the next sqg_batch_stage / sqg_batch_sample
    struct Recycled { uint8_t* d_block; size_t block_bytes; long long* h_sigoff; long long* h_sigoff_dev; size_t h_n; hipEvent_t ev[8];
                      uint8_t* h_meta; size_t h_meta_bytes; hipEvent_t ev_staged; };
    std::vector<Recycled> pool;
    hipStream_t stage_stream = nullptr;            // uploads and the staging kernels (k_sample, k_copy_reads, k_fill_tiles): a host
                                                   // can stage batch i+1 while batch i runs
    std::vector<uint32_t> time_c;          // canonical time-stream state per local worker
    std::vector<long long> off_x, med_x;   // raw Schrage states (as the reference keeps them)
    unsigned long long next_stage = 0, next_run = 0, compress_seq = 0;
    unsigned long long runs = 0;                   // batches run so far: a batch's slot is its run index & 1
    unsigned int* d_mid_done = nullptr;            // k_part_mid: offsets workgroups that have finished (the last one resets it)
    int stage_threads = 0;                         // sqg_set_stage_threads: host threads that share a batch's per-read libm draws (0: automatic)
    int stage_threads_last = 0;                    // ... and how many the last staging call used
    int phase_timing_every = 1;                    // sqg_set_phase_timing: the batches whose run index is a multiple carry the phase events (0: none)
    std::set<unsigned long long> abandoned;        // staged batches that were freed without having been run
    sqg_timing_t timing = {0, 0, 0, 0, 0, 0, 0, 0};
    bool use_dwell_stream = true, use_kmer_streams = true;
    float delta_x = 0.f;                   // certified mode: swept |x_fast - x_exact| bound incl. margin
    float delta_x_measured = 0.f;
    double amp_floor = 0, amp_ceil = 0;    // min/max over k-mers of m*kd -/+ 7|sd*kd| (ADC value range before the offset)
    float thr_all = -1.f;                  // lean-kernel acceptance threshold (0.5 - largest eps over the table)
    int lean_epl = 4;                      // events per lane of the lean kernel (work item = 64*lean_epl events)
    double dwell_hi = 1;                   // hard upper bound of a dwell draw
    bool force_fix = false;
    bool lds_ordered = false;             // k_lds_order_check passed on this device: k_part_hand_ord hands the streams out (k_part.h)
    uint8_t* d_genome = nullptr;                                // resident reference (sqg_genome_load)
    uint32_t* d_nprefix = nullptr;                              // ... and its 'N's per 64-base block, summed (k_nprefix_*)
    long long* d_contig_off = nullptr; long long* d_cum = nullptr;
    float* d_trans_csum = nullptr; int* d_trans_idx = nullptr;
    uint32_t* d_samp = nullptr;                                 // [nw][3] sampler stream states: ref_pos, rand_strand, rand_rlen
    uint8_t* d_meth = nullptr; uint8_t* d_meth_has = nullptr;   // --meth-freq (sqg_genome_set_meth): frequency byte per base, flag per contig
    uint32_t* d_meth_st = nullptr;                              // [nw] rand_meth stream states
    GenomeParams genome{};
    uint32_t* h_samp = nullptr; size_t h_samp_cap = 0;          // pinned: the sampler streams' snapshot and the error word of a sqg_batch_sample call
    uint8_t* d_samp_scratch = nullptr; size_t samp_scratch_cap = 0;   // sqg_batch_sample: records, chain lists, attempt slots
    bool genome_loaded = false;
    std::vector<long long> h_contig_off;                        // host copy of the contig offsets
    long long full_next = 0;                                    // --full-contigs: the next contig to hand out (core->total_reads)
    double samp_ratio = 1.1;                                    // attempts per accepted read seen so far (long chains)
    uint8_t* d_svb = nullptr; size_t svb_cap = 0;               // svb-zd encodings of the last compressed batch
    long long* d_svb_size = nullptr; size_t svb_size_cap = 0;   // per read
    long long* d_svb_off = nullptr; size_t svb_off_cap = 0;
    uint8_t* d_b5meta = nullptr; size_t b5meta_cap = 0;        // sqg_batch_blow5_records: ids, offsets, per-read doubles (uploaded per call)
    uint8_t* d_b5out = nullptr; size_t b5out_cap = 0;          // ... the framed records on the device
    uint8_t* h_b5out[2] = {nullptr, nullptr}; size_t h_b5out_cap[2] = {0, 0};   // ... and in pinned host memory (what the call returns), two of
    int b5_flip = 0;                                           //     them used alternately: a result stays valid over the NEXT call (a writer copies it out behind that call)
    uint8_t* h_b5meta = nullptr; size_t h_b5meta_cap = 0;      // ... pinned staging of the upload
    std::vector<int64_t> b5_rec_off;
    // a stored-mode writer whose background write still reads h_b5out[b5_reader_buf] (h_blow5.h registers itself here): drained before that
    // buffer is filled again by anybody, and before the context goes (the thread read memory the writer did not own)
    // placement calibration: the scatter pass runs in one of several modes (825 ... 1130 us per 32768 10-kb reads) for the life of a slot's `evrec` allocation
    // (profiles/r05_summary.md); the first large batches of a context time the pass itself on three more allocations per slot and keep the best
    int cal_runs_left = 12;
    void* b5_reader = nullptr; int b5_reader_buf = -1; void (*b5_reader_drain)(void* writer, bool unbind) = nullptr;
    hipStream_t b5_stream = nullptr;                           // the records' upload, framing kernel and copy back: a stream of their own (not behind the next batch's kernels)
    std::string err;
};

#define SQG_HRES_LL ((FIX_SHARDS + 8) / 2)          /* long longs at the end of a batch's mapped host block: 4 + FIX_SHARDS 32-bit words, rounded up */
struct sqg_batch {
    unsigned long long seq = 0;
    int n = 0;
    long long n_events = 0, n_bases = 0, n_samples = 0;
    int n_chains = 0;
    std::vector<long long> ev_off, sig_off;
    std::vector<double> offset, median;
    std::vector<unsigned long long> seglen_host;   // only when dwell is constant
    uint8_t* d_block = nullptr;          // the batch's one device allocation; the pointers below point into it
    size_t block_bytes = 0, h_n = 0;     // its size; entries of h_sigoff
    uint8_t* d_bases = nullptr;
    ReadDesc* d_reads = nullptr;
    int* d_blk_read = nullptr;
    int* d_chain_off = nullptr;
    int* d_chain_reads = nullptr;
    int* d_chain_order = nullptr;
    bool split = false;                  // the worker chains are cut into links (d_chain_off describes the links)
    int n_wchains = 0;                   // workers with reads in this batch
    int* d_wlink_off = nullptr;          // [n_wchains+1] links of each worker chain
    int* d_wlink_worker = nullptr;       // [n_wchains]
    long long max_wchain_ev = 0;         // events of the longest worker chain
    bool part = false;                   // k > 6, split: the hand-out runs over bucketed events (k_part.h)
    bool pieces = false;                 // the links are runs of pieces of reads (k_part_events), not of whole reads
    bool split_reads = false;            // ... and some read is cut into several pieces
    int4* d_pieces = nullptr;            // [n_pieces] {read, first event, end event, -}
    uint32_t* d_piece_total = nullptr;   // [n_pieces] samples of each piece (written by the first event pass)
    int n_pieces = 0;
    bool one = false;                    // ... with ONE partition (k <= 6): the events stay in chain order, no counting and no scatter pass
    uint32_t* d_link_slot = nullptr;     // one: [n_chains] first slot of every link in part[]
    uint32_t* d_wchain_total = nullptr;  // one: [n_wchains] events of every worker chain
    uint32_t slice_len = 0;              // events per slice of a (worker chain, partition) (k_part.h)
    long long max_slices = 0;            // bound on their number (the device counts them)
    int* d_link_q = nullptr;             // [n_chains] the worker chain of every link (k_part.h)
    int* d_tile_read = nullptr;
    int* d_stile_read = nullptr;
    long long n_tiles = 0, n_stiles = 0;
    long long* h_sigoff = nullptr;   // pinned, device-mapped: k_scan writes it directly
    long long* h_sigoff_dev = nullptr;   // its device-side address
    // (the same allocation ends with SQG_HRES_LL words for k_fixup's report: error word, fix-up counts -- SigParams.host_res)
    long long n_bases_total = 0;         // bytes in d_bases
    std::vector<long long> h_base_off;   // per read: its segment 0 in d_bases
    std::vector<int32_t> s_ref_idx, s_ref_len, s_ref_pos, s_rlen;   // sqg_batch_sample: what gen_read returned
    std::vector<char> s_strand;
    std::vector<long long> s_src;                                   // where each sampled read starts in the resident genome
    std::vector<long long> s_seq_off, s_read_at;                    // offsets of the reads in sqg_fetch_reads / in d_bases
    long long* h_svboff = nullptr;       // pinned, device-mapped: offsets of the svb-zd encodings (sqg_batch_compress)
    long long n_svb = -1;
    unsigned long long compress_seq = 0;
    bool untimed = false;                          // no phase events in this batch (sqg_set_phase_timing): its timings read 0
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // kernel-phase boundaries; [7]: the event side is done
    uint8_t* h_meta = nullptr; size_t h_meta_bytes = 0;   // pinned: the host-built arrays of the batch (descriptors, chain lists), uploaded in one copy
    hipEvent_t ev_staged = nullptr;      // recorded on the staging stream after the batch's last staging operation
    unsigned int* d_err = nullptr;       // the batch's own device error word (in d_block; zeroed by the meta upload): batches queued back
                                         // to back never see each other's errors
    int wait_rc = 0;                     // what the first sqg_batch_wait returned (latched)
    unsigned long long slot_gen = 0;     // generation of the slot when this batch took it (results are stale once it differs)
    int slot = 0;                        // which of the context's two buffer sets this batch runs in
    int cset = 0;                        // ... and which of the three sets of first-pass outputs (run index % 3)
    unsigned long long cset_gen = 0;     // generation of that set when this batch took it
    bool precounted = false;             // its first event pass was run inside the launch sequence of the batch before it (into cset)
    int pre_slot = -1;                   // ... which, with one partition, wrote part[] of this slot
    bool carried_precount = false;       // this batch's launch sequence carried its successor's first event pass (k_part_hand_count)
    unsigned long long run_idx = 0;      // how many batches had been run before this one
    bool ran = false, waited = false, lean_timed = false, dwell_timed = false, fixup_launched = false;
    bool staged = false;                 // staging completed: the batch holds a place in the run order
    bool begun = false, other_fresh = false;   // sqg_batch_run_begin has run; the other slot had never held a batch then
};

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return e_ == hipErrorOutOfMemory ? SQG_ENOMEM : SQG_EDEVICE;                       \
        }                                                                                      \
    } while (0)

// the reference's rng()/nrng() on the host, for the two per-read scalar draws that are
// RETURNED as doubles (offset, median_before; src/gensig.c:311-317): made with the host's libm
// so they are the very doubles the CPU reference produces on this machine.
static double host_rng(long long* xp) {            // src/rand.h:79-85
    const long long x = *xp;
    const long long nx = 16807LL * (x % 127773LL) - 2836LL * (x / 127773LL);
    *xp = nx;
    return (double)(nx > 0 ? nx : nx + 2147483647LL) / 2147483647;
}
static double host_nrng(double m, double s, long long* xp) {   // src/rand.h:87-94
    double u = 0.0, t = 0.0;
    while (u == 0.0) u = host_rng(xp);
    while (t == 0.0) t = 2.0 * 3.14159265 * host_rng(xp);
    const double z = std::sqrt(-2.0 * std::log(u)) * std::cos(t);
    return (z * s) + m;
}

// The per-read scalar draws (`offset`, `median_before`: src/gensig.c:315-316) are host libm work -- they are printed values, so they
// are made with the host's log / sqrt / cos -- and they depend on nothing but the worker's two streams: draw i of a worker is known
// before its i-th read is.  With few workers (`-t 1`: the regime of the bench) a thread of the context therefore keeps up to CAP
// draws per worker ready; staging takes what is there and draws the rest itself.  100 us of a 1000-read batch's 330 us of staging
// (the reference's default -K), nothing a 16384-read batch notices.  A ring is valid only for the stream states it was started
// from: whoever moves a worker's streams another way (sqg_skip_reads, a failed staging that puts them back) makes the next take() start over.
struct DrawAhead {
    static constexpr size_t CAP = 16384, CHUNK = 256;           // (CAP a power of two)
    struct Ring {
        std::vector<double> off, med;                             // draw p at index p & (CAP - 1)
        std::vector<long long> offx, medx;                        // ... and the streams' raw states behind it
        long long x_off_head = 0, x_med_head = 0, x_off_tail = 0, x_med_tail = 0;   // the states in front of draw `head` / `filled`
        size_t head = 0, filled = 0;
        unsigned long long epoch = 0;
        bool active = false;
    };
    std::vector<Ring> rings;
    std::mutex m;
    std::condition_variable cv;
    std::thread th;
    bool stop = false, started = false;
    double off_mean, off_std, med_mean, med_std;
    DrawAhead(int nw, const sqg_profile_t& p) : rings((size_t)nw), off_mean(p.offset_mean), off_std(p.offset_std), med_mean(p.median_before_mean), med_std(p.median_before_std) {}
    ~DrawAhead() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    void producer() {
        std::unique_lock<std::mutex> lk(m);
        double o[CHUNK], d[CHUNK]; long long ox[CHUNK], dx[CHUNK];
        for (;;) {
            Ring* pick = nullptr;
            for (auto& R : rings) if (R.active && R.filled - R.head + CHUNK <= CAP && (!pick || R.filled - R.head < pick->filled - pick->head)) pick = &R;
            if (stop) return;
            if (!pick) { cv.wait(lk); continue; }
            const unsigned long long ep = pick->epoch;
            long long xo = pick->x_off_tail, xm = pick->x_med_tail;
            lk.unlock();
            for (size_t i = 0; i < CHUNK; i++) { o[i] = host_nrng(off_mean, off_std, &xo); ox[i] = xo; d[i] = host_nrng(med_mean, med_std, &xm); dx[i] = xm; }
            lk.lock();
            if (pick->epoch != ep) continue;                      // (the worker's streams went elsewhere meanwhile)
            for (size_t i = 0; i < CHUNK; i++) {
                const size_t at = (pick->filled + i) & (CAP - 1);
                pick->off[at] = o[i]; pick->med[at] = d[i]; pick->offx[at] = ox[i]; pick->medx[at] = dx[i];
            }
            pick->filled += CHUNK; pick->x_off_tail = xo; pick->x_med_tail = xm;
        }
    }
    void restart(Ring& R, long long x_off, long long x_med) {    // (locked)
        if (R.off.empty()) { R.off.resize(CAP); R.med.resize(CAP); R.offx.resize(CAP); R.medx.resize(CAP); }
        R.head = R.filled = 0; R.x_off_head = R.x_off_tail = x_off; R.x_med_head = R.x_med_tail = x_med; R.epoch++; R.active = true;
    }
    // up to `want` draws of worker w whose streams stand at (x_off, x_med): values to off_out / med_out, the states behind them to *x_off / *x_med
    size_t take(int w, size_t want, long long* x_off, long long* x_med, double* off_out, double* med_out) {
        size_t n = 0;
        {
            std::lock_guard<std::mutex> lk(m);
            if (!started) { started = true; th = std::thread([this] { producer(); }); }
            Ring& R = rings[(size_t)w];
            if (!R.active || R.x_off_head != *x_off || R.x_med_head != *x_med) restart(R, *x_off, *x_med);
            else {
                n = std::min(want, R.filled - R.head);
                for (size_t i = 0; i < n; i++) { const size_t at = (R.head + i) & (CAP - 1); off_out[i] = R.off[at]; med_out[i] = R.med[at]; }
                if (n) {
                    const size_t last = (R.head + n - 1) & (CAP - 1);
                    *x_off = R.x_off_head = R.offx[last]; *x_med = R.x_med_head = R.medx[last];
                    R.head += n;
                }
            }
        }
        cv.notify_one();
        return n;
    }
    // the worker's streams stand at (x_off, x_med) now: what was made for other states is dropped
    void rebase(int w, long long x_off, long long x_med) {
        {
            std::lock_guard<std::mutex> lk(m);
            Ring& R = rings[(size_t)w];
            if (R.active && R.x_off_head == x_off && R.x_med_head == x_med) return;
            restart(R, x_off, x_med);
        }
        cv.notify_one();
    }
};

static uint32_t canon(long long s) {
    s %= (long long)LCG_M;
    if (s < 0) s += LCG_M;
    return (uint32_t)s;
}

// staged batches are run in staging order; those freed without a run are stepped over
static void skip_abandoned(sqg_ctx* c) {
    while (!c->abandoned.empty() && *c->abandoned.begin() == c->next_run) { c->abandoned.erase(c->abandoned.begin()); c->next_run++; }
}

static int ensure(sqg_ctx* c, void** p, size_t* cap, size_t need, size_t elem) {
    if (need <= *cap) return SQG_OK;
    size_t ncap = std::max(need + need / 4, *cap + *cap / 2);     // slack: batches of similar size never re-allocate
    if (*p) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream2)); HIPCHK(c, hipStreamSynchronize(c->fix_stream)); HIPCHK(c, hipFree(*p)); *p = nullptr; *cap = 0; }
    HIPCHK(c, hipMalloc(p, ncap * elem));
    *cap = ncap;
    return SQG_OK;
}

static const char kStallRna[] = "AAAAAGAAAAAACCCCCCCCCCCCCCCCCC";                  // src/genread.c:87
static const char kStallDna[] = "TTTTTTTTTTTTTTTTTTAATCAA";                       // src/genread.c:110
static const char kAdaptorDna[] = "GGCGTCTGCTTGGGTGTTTAACCTTTTTTTTTTAATGTACTTCGTTCAGTTACGTATTGCT";  // src/genread.c:38
static const char kAdaptorRna[] = "TGATGATGAGGGATAGACGATGGTTGTTTCTGTTGGTGCTGATATTGCTTTTTTTTTTTTTATGATGCAAGATACGCAC";  // src/genread.c:39
static const int kPolyA = 158;                                                   // src/genread.c:37
static const char kShortHack[] = "ACGTACGTACGTA";   // src/gensig.c:242-245: "ACGTACGTACGT" + its NUL (rank 0)

// debugging aid: SQG_DEBUG_SYNC=1 synchronises after every launch and names the kernel that faulted
static int dbg_sync(sqg_ctx* c, const char* what) {
    static const bool on = getenv("SQG_DEBUG_SYNC") != nullptr;
    if (!on) return SQG_OK;
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream2);
    if (e == hipSuccess) e = hipStreamSynchronize(c->fix_stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { c->err = std::string(what) + ": " + hipGetErrorString(e); fprintf(stderr, "[sqg] %s\n", c->err.c_str()); return SQG_EDEVICE; }
    fprintf(stderr, "[sqg] %s ok\n", what);
    return SQG_OK;
}

// phase 0: the whole run; 1: up to the per-stream sample counts of a split batch (sqg_batch_run_begin); 2: the rest
