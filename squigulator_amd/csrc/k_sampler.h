// k_sampler.h -- gen_read on the device-resident genome: k_init_sampler, k_sample, k_copy_reads
// Part of the device code of the per-read signal path; included through sqg_kernels.h (see there for the overview).
#pragma once

// ---- read sampler on the device-resident genome ("next" row) ----------
// gen_read, src/genread.c:125-370: per worker the streams ref_pos (seed s), rand_strand (s+1) and rand_rlen
// (s+3; Erlang-2 with scale rlen/2, the INTEGER quotient) of src/sim.c:238-247, consumed in read order.
// GenomeParams.flags holds the SQG_SAMPLE_* bits of include/sqg.h

struct GenomeParams {
    const uint8_t* seq;          // contigs back to back (no terminators)
    const long long* contig_off; // [n_contigs+1]
    const long long* cum;        // [n_contigs] inclusive prefix sums of the contig lengths (src/genread.c:181-191)
    const float* trans_csum;     // --trans-count: cumulative abundances (float, src/ref.c:206-273), or null
    const int* trans_idx;        // ... and the contig of each entry
    long long sum;               // ref->sum
    double grng_b;               // (double)(rlen / 2)
    int n_contigs, n_trans, rlen, flags;
    const uint32_t* nprefix;     // [sum / 64 + 2] 'N's in seq[0, 64 i): a candidate read's count without reading the read (k_nprefix_*)
    const uint8_t* meth;         // --meth-freq (sqg_genome_set_meth): round(255*freq) per base, laid out like seq; or null
    const uint8_t* meth_has;     // [n_contigs] the contig has an array at all (ref->ref_meth[i] != NULL, src/genread.c:208)
};

struct SampleRec {               // what gen_read returns, per read
    long long src;               // offset of the read's first base in GenomeParams.seq (forward strand)
    int ref_idx, ref_pos, rlen;  // contig, 0-based start, bases copied
    int strand;                  // '+' or '-'
    int n_N;                     // 'N's substituted (src/genread.c:132-138)
    int ref_len;                 // *ref_len of gen_read: the contig's length (DNA) / the transcript part used (RNA)
};

__global__ void k_init_sampler(uint32_t* __restrict__ st, long long seed, int worker_lo, int nw, int num_kmer) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    const long long s = seed + (long long)(w + worker_lo) * ((long long)num_kmer + 10);
    const long long add[3] = {0, 1, 3};                              // ref_pos, rand_strand, rand_rlen
    for (int j = 0; j < 3; j++) {
        long long v = (s + add[j]) % (long long)LCG_M;
        if (v < 0) v += LCG_M;
        st[3 * w + j] = (uint32_t)v;
    }
}

// rng() of src/rand.h:79-85 on a canonical state
__device__ static inline double samp_rng(uint32_t& c) { c = lcg_mul(c, LCG_A); return lcg_uniform(c); }

// exact count of bytes equal to 'N' in p[0..n), by the 64 lanes of a wavefront together (8 bytes per lane per step)
__device__ static inline int count_N(const uint8_t* __restrict__ p, int n, int lane) {
    int cnt = 0;
    const int n8 = n & ~7;
    for (int i = lane * 8; i < n8; i += 512) {
        unsigned long long v;
        __builtin_memcpy(&v, p + i, 8);
        const unsigned long long x = v ^ 0x4e4e4e4e4e4e4e4eull;        // zero byte <=> 'N'
        const unsigned long long t = ~(((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x | 0x7f7f7f7f7f7f7f7full);
        cnt += __popcll(t);
    }
    if (lane < n - n8) cnt += p[n8 + lane] == 'N';
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    return cnt;
}

// the same count from the genome's table of 'N's per 64-base block: two look-ups and the two ragged ends
__device__ static inline int count_N_at(const GenomeParams& G, long long src, int n, int lane) {
    const long long a = src, b = src + n, ba = (a + 63) >> 6, bb = b >> 6;
    if (!G.nprefix || ba >= bb) return count_N(G.seq + src, n, lane);
    return (int)(G.nprefix[bb] - G.nprefix[ba]) + count_N(G.seq + a, (int)((ba << 6) - a), lane) + count_N(G.seq + (bb << 6), (int)(b - (bb << 6)), lane);
}

// the table: 'N's of every 64-base block (one thread each), then their running sum (one workgroup, 16384 blocks per round)
__global__ __launch_bounds__(256) void k_nprefix_count(const uint8_t* __restrict__ seq, long long total, long long n_blocks, uint32_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_blocks) return;
    const uint8_t* p = seq + (i << 6);
    const int n = (int)min(64LL, total - (i << 6));
    int cnt = 0;
    if (n == 64) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            unsigned long long v;
            __builtin_memcpy(&v, p + 8 * q, 8);
            const unsigned long long x = v ^ 0x4e4e4e4e4e4e4e4eull;        // zero byte <=> 'N'
            cnt += __popcll(~(((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x | 0x7f7f7f7f7f7f7f7full));
        }
    } else for (int q = 0; q < n; q++) cnt += p[q] == 'N';
    out[i + 1] = (uint32_t)cnt;                                       // (out[0] = 0: set by the scan)
}
__global__ __launch_bounds__(1024) void k_nprefix_scan(uint32_t* __restrict__ t, long long n_blocks) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) { carry = 0; t[0] = 0; }
    __syncthreads();
    for (long long base = 0; base < n_blocks; base += 16384) {
        uint32_t v[16], run = 0;
        const long long i0 = base + (long long)tid * 16;
#pragma unroll
        for (int q = 0; q < 16; q++) { v[q] = i0 + q < n_blocks ? t[i0 + q + 1] : 0u; run += v[q]; }
        const uint32_t incl = (uint32_t)wave_incl_scan_dpp((int)run);
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        uint32_t at = carry + incl - run;
        for (int w = 0; w < wid; w++) at += wsum[w];
#pragma unroll
        for (int q = 0; q < 16; q++) { at += v[q]; if (i0 + q < n_blocks) t[i0 + q + 1] = at; }
        __syncthreads();
        if (tid == 1023) carry = at;
        __syncthreads();
    }
}

// a^n mod M
__device__ static inline uint32_t lcg_pow_a(unsigned long long n) {
    uint32_t r = 1, b = LCG_A;
    while (n) { if (n & 1) r = lcg_mul(r, b); b = lcg_mul(b, b); n >>= 1; }
    return r;
}

// draws one attempt takes from (ref_pos, rand_strand, rand_rlen): fixed per sampler variant, whatever the outcome
__device__ static inline void samp_draws(int flags, int& dp, int& ds, int& dl) {
    if (flags & (SQG_SAMPLE_RNA | SQG_SAMPLE_CDNA)) { dp = 1; ds = (flags & SQG_SAMPLE_CDNA) ? 1 : 0; dl = (flags & SQG_SAMPLE_TRUNC) ? 2 : 0; }
    else { dp = 1; ds = 1; dl = 2; }
}

// one attempt of gen_read (src/genread.c:125-370) by a wavefront: every lane makes the same draws, the lanes share the
// scan of the candidate for 'N's.  true: accepted, rec filled
__device__ static inline bool samp_attempt(const GenomeParams& G, uint32_t& c_pos, uint32_t& c_strand, uint32_t& c_len, int lane, SampleRec& rec) {
    int idx, pos, len, strand = '+';
    if (G.flags & (SQG_SAMPLE_RNA | SQG_SAMPLE_CDNA)) {
        // src/genread.c:283-300: uniform over transcripts, or by the abundance CDF (uniform narrowed to float)
        if (G.n_trans == 0) idx = (int)round(samp_rng(c_pos) * (G.n_contigs - 1));
        else {
            const float r = (float)samp_rng(c_pos);
            idx = 0;
            for (int i = 0; i < G.n_trans; i++) if (r <= G.trans_csum[i]) { idx = G.trans_idx[i]; break; }
        }
        const int clen = (int)(G.contig_off[idx + 1] - G.contig_off[idx]);
        len = clen; pos = 0;
        if (G.flags & SQG_SAMPLE_TRUNC) {                     // src/genread.c:303-309
            double acc = 0.0;
            acc += -log(1 - samp_rng(c_len));
            acc += -log(1 - samp_rng(c_len));
            const double frac = (acc * G.grng_b) / (double)G.rlen;
            int tl = (int)(frac * clen);
            tl = tl > clen ? clen : tl;
            pos = clen - tl; len = tl;
        }
        if (G.flags & SQG_SAMPLE_CDNA) strand = ((long long)round(samp_rng(c_strand))) ? '+' : '-';
    } else {
        // src/genread.c:243-281
        double acc = 0.0;                                     // grng, src/rand.h:96-102 (Erlang-2)
        acc += -log(1 - samp_rng(c_len));
        acc += -log(1 - samp_rng(c_len));
        len = (int)(acc * G.grng_b);
        const long long at = (long long)round(samp_rng(c_pos) * (double)G.sum);   // src/genread.c:181
        idx = 0;
        while (idx < G.n_contigs - 1 && G.cum[idx] < at) idx++;
        pos = (int)(at - G.cum[idx]) + (int)(G.contig_off[idx + 1] - G.contig_off[idx]);
        strand = ((long long)round(samp_rng(c_strand))) ? '+' : '-';            // src/genread.c:196-200
    }
    if (len < 0) len = 0;
    const int clen = (int)(G.contig_off[idx + 1] - G.contig_off[idx]);
    const int n = min(len, clen - pos);                       // src/genread.c:149-177: clipped at the contig's end
    if (n < 200) return false;                                // src/genread.c:126
    const long long src = G.contig_off[idx] + pos;
    const int nN = count_N_at(G, src, n, lane);
    if ((double)nN > 0.1 * (double)n) return false;           // src/genread.c:139-142
    rec.src = src; rec.ref_idx = idx; rec.ref_pos = pos; rec.rlen = n; rec.strand = strand; rec.n_N = nN;
    rec.ref_len = (G.flags & (SQG_SAMPLE_RNA | SQG_SAMPLE_CDNA)) ? len : clen;
    return true;
}

// attempts until one is accepted (the reference has no limit; 100000 rejections in a row flag an unusable genome)
__device__ static inline void samp_read(const GenomeParams& G, uint32_t& c_pos, uint32_t& c_strand, uint32_t& c_len, int lane, SampleRec& rec,
                                        unsigned int* err, long long* n_attempts) {
    for (int attempt = 0;; attempt++) {
        if (attempt > 100000) { atomicOr(err, 16u); rec.src = 0; rec.ref_idx = 0; rec.ref_pos = 0; rec.rlen = 0; rec.strand = '+'; rec.n_N = 0; rec.ref_len = 0; return; }
        if (n_attempts) ++*n_attempts;
        if (samp_attempt(G, c_pos, c_strand, c_len, lane, rec)) return;
    }
}

// one wavefront per worker chain (that worker's reads of the batch, in order)
__global__ __launch_bounds__(64) void k_sample(const GenomeParams G, uint32_t* __restrict__ st, const int* __restrict__ chain_off,
                                               const int* __restrict__ chain_reads, const int* __restrict__ chain_worker,
                                               int n_chains, SampleRec* __restrict__ out, unsigned int* __restrict__ err) {
    const int ch = blockIdx.x, lane = threadIdx.x;
    if (ch >= n_chains) return;
    const int w = chain_worker[ch];
    uint32_t c_pos = st[3 * w], c_strand = st[3 * w + 1], c_len = st[3 * w + 2];
    for (int ci = chain_off[ch]; ci < chain_off[ch + 1]; ci++) {
        SampleRec rec;
        samp_read(G, c_pos, c_strand, c_len, lane, rec, err, nullptr);
        if (lane == 0) out[chain_reads[ci]] = rec;
    }
    if (lane == 0) { st[3 * w] = c_pos; st[3 * w + 1] = c_strand; st[3 * w + 2] = c_len; }
}

// Long chains (few workers, many reads).  An attempt takes a fixed number of draws from each stream, so attempt a of a
// chain is a function of a alone: the attempts are evaluated concurrently (k_sample_try, one wavefront each) and the
// chain's reads are the accepted ones in attempt order (k_sample_pick).  att_off[ch]..att_off[ch+1]: the chain's
// attempt slots; should they hold fewer acceptable reads than the chain needs, k_sample_pick goes on one by one.
__global__ __launch_bounds__(256) void k_sample_try(const GenomeParams G, const uint32_t* __restrict__ st, const int* __restrict__ chain_worker,
                                                    const long long* __restrict__ att_off, SampleRec* __restrict__ try_rec, unsigned char* __restrict__ try_ok) {
    const int ch = blockIdx.y, lane = threadIdx.x & 63;
    const long long a = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= att_off[ch + 1] - att_off[ch]) return;
    const int w = chain_worker[ch];
    int dp, ds, dl;
    samp_draws(G.flags, dp, ds, dl);
    uint32_t c_pos = lcg_mul(st[3 * w], lcg_pow_a((unsigned long long)a * dp));
    uint32_t c_strand = lcg_mul(st[3 * w + 1], lcg_pow_a((unsigned long long)a * ds));
    uint32_t c_len = lcg_mul(st[3 * w + 2], lcg_pow_a((unsigned long long)a * dl));
    SampleRec rec;
    const bool ok = samp_attempt(G, c_pos, c_strand, c_len, lane, rec);
    if (lane == 0) { try_ok[att_off[ch] + a] = ok ? 1 : 0; if (ok) try_rec[att_off[ch] + a] = rec; }
}

__global__ __launch_bounds__(256) void k_sample_pick(const GenomeParams G, uint32_t* __restrict__ st, const int* __restrict__ chain_off,
                                                     const int* __restrict__ chain_reads, const int* __restrict__ chain_worker,
                                                     const long long* __restrict__ att_off, const SampleRec* __restrict__ try_rec,
                                                     const unsigned char* __restrict__ try_ok, SampleRec* __restrict__ out,
                                                     long long* __restrict__ att_used, unsigned int* __restrict__ err) {
    __shared__ int wcnt[4];
    __shared__ long long s_used;
    const int ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = chain_off[ch + 1] - chain_off[ch];
    const long long o = att_off[ch], A = att_off[ch + 1] - o;
    const int* reads = chain_reads + chain_off[ch];
    if (tid == 0) s_used = 0;
    int found = 0;                                                    // accepted so far (the same in every thread)
    for (long long a0 = 0; a0 < A && found < m; a0 += 256) {
        const long long a = a0 + tid;
        const bool ok = a < A && try_ok[o + a];
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) wcnt[wid] = __popcll(bal);
        __syncthreads();
        int before = found;
        for (int w2 = 0; w2 < wid; w2++) before += wcnt[w2];
        before += __popcll(bal & ((1ull << lane) - 1));
        if (ok && before < m) {
            out[reads[before]] = try_rec[o + a];
            if (before == m - 1) s_used = a + 1;
        }
        found += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    __syncthreads();
    if (wid != 0) return;
    const int w = chain_worker[ch];
    int dp, ds, dl;
    samp_draws(G.flags, dp, ds, dl);
    long long used = found >= m ? s_used : A;
    uint32_t c_pos = lcg_mul(st[3 * w], lcg_pow_a((unsigned long long)used * dp));
    uint32_t c_strand = lcg_mul(st[3 * w + 1], lcg_pow_a((unsigned long long)used * ds));
    uint32_t c_len = lcg_mul(st[3 * w + 2], lcg_pow_a((unsigned long long)used * dl));
    for (int i = found; i < m; i++) {                                 // the slots did not hold enough: one by one from here
        SampleRec rec;
        samp_read(G, c_pos, c_strand, c_len, lane, rec, err, &used);
        if (lane == 0) out[reads[i]] = rec;
    }
    if (lane == 0) { st[3 * w] = c_pos; st[3 * w + 1] = c_strand; st[3 * w + 2] = c_len; att_used[ch] = used; }
}

// ---- CpG methylation of the sampled reads (methylate_dna, src/genread.c:207-241) ------------------------------------
// Every CpG of a read's span on the forward reference (both bases upper case and inside the read) takes ONE draw from the
// worker's rand_meth stream, whatever the outcome, in position order; reads of a worker in read order.  So a read's first
// draw is addressed by the CpGs of the worker's earlier reads: count (k_meth_count), scan per worker chain (k_meth_scan),
// then k_copy_reads takes the j-th CpG's draw as state * a^(j+1).
__device__ static inline bool meth_is_cpg(const uint8_t* __restrict__ src, int i, int n, int ref_pos, int ref_len) {
    return i + 1 < n && ref_pos + i + 1 < ref_len && src[i] == 'C' && src[i + 1] == 'G';
}

// one workgroup per read of the batch: CpGs that draw
__global__ __launch_bounds__(256) void k_meth_count(const GenomeParams G, const SampleRec* __restrict__ recs, int n_reads, int* __restrict__ cnt) {
    __shared__ int wsum[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    if (r >= n_reads) return;
    const SampleRec rec = recs[r];
    int c = 0;
    if (G.meth_has[rec.ref_idx]) {
        const uint8_t* src = G.seq + rec.src;
        for (int i = tid; i < rec.rlen; i += 256) c += meth_is_cpg(src, i, rec.rlen, rec.ref_pos, rec.ref_len) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((tid & 63) == 0) wsum[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) cnt[r] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one workgroup per worker chain: mstate[read] = the worker's rand_meth state before the read's first draw; the worker's
// stream moves past the chain
__global__ __launch_bounds__(256) void k_meth_scan(const int* __restrict__ chain_off, const int* __restrict__ chain_reads,
                                                   const int* __restrict__ chain_worker, const int* __restrict__ cnt,
                                                   uint32_t* __restrict__ st, uint32_t* __restrict__ mstate) {
    __shared__ long long wsum[4];
    __shared__ long long base_sh;
    const int ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = chain_off[ch], hi = chain_off[ch + 1];
    const int w = chain_worker[ch];
    const uint32_t s0 = st[w];
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int i0 = lo; i0 < hi; i0 += 256) {
        const int i = i0 + tid;
        const long long v = i < hi ? (long long)cnt[chain_reads[i]] : 0;
        long long x = v;
        for (int o = 1; o < 64; o <<= 1) { const long long y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) wsum[wid] = x;
        __syncthreads();
        long long before = base_sh, total = 0;
        for (int q = 0; q < 4; q++) { if (q < wid) before += wsum[q]; total += wsum[q]; }
        if (i < hi) mstate[chain_reads[i]] = lcg_mul(s0, lcg_pow_a((unsigned long long)(before + x - v)));
        __syncthreads();
        if (tid == 0) base_sh += total;
        __syncthreads();
    }
    if (tid == 0) st[w] = lcg_mul(s0, lcg_pow_a((unsigned long long)base_sh));
}

__device__ static const char kd_stall_dna[] = "TTTTTTTTTTTTTTTTTTAATCAA";                       // src/genread.c:110
__device__ static const char kd_adaptor_dna[] = "GGCGTCTGCTTGGGTGTTTAACCTTTTTTTTTTAATGTACTTCGTTCAGTTACGTATTGCT";  // src/genread.c:38
__device__ static const char kd_adaptor_rna[] = "TGATGATGAGGGATAGACGATGGTTGTTTCTGTTGGTGCTGATATTGCTTTTTTTTTTTTTATGATGCAAGATACGCAC";  // src/genread.c:39
__device__ static const char kd_stall_rna[] = "AAAAAGAAAAAACCCCCCCCCCCCCCCCCC";                  // src/genread.c:87

// one workgroup per read: the sampled slice of the genome -> the batch's base buffer, exactly as gen_read returns
// it ('N' -> a base from a FRESH state-100 stream per read, src/genread.c:132-138; '-' -> revcomp, src/seq.h:78-112)
// with the prefix / stall attached as src/genread.c:95-123 does.  read_at: where the read starts in segment 0.
// mstate (optional): per read the worker's rand_meth state before the read's first CpG draw -> methylated Cs become 'M'
__global__ __launch_bounds__(256) void k_copy_reads(const GenomeParams G, const SampleRec* __restrict__ recs, const ReadDesc* __restrict__ reads,
                                                    uint8_t* __restrict__ bases, int n_reads, int rna, int prefix,
                                                    const uint32_t* __restrict__ mstate) {
    __shared__ int wcnt[4];
    __shared__ int carry;
    __shared__ uint8_t comp[256];
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const SampleRec rec = recs[r];
    const ReadDesc rd = reads[r];
    uint8_t* dst = bases + rd.base_off;
    const int n = rec.rlen;
    int read_at = 0;
    if (prefix) {
        if (rna) {                                                    // read + polyA(158) + adaptor
            for (int i = tid; i < 158; i += 256) dst[n + i] = 'A';
            for (int i = tid; i < (int)sizeof(kd_adaptor_rna) - 1; i += 256) dst[n + 158 + i] = (uint8_t)kd_adaptor_rna[i];
        } else {                                                      // stall + adaptor + read
            const int st = (int)sizeof(kd_stall_dna) - 1, ad = (int)sizeof(kd_adaptor_dna) - 1;
            for (int i = tid; i < st; i += 256) dst[i] = (uint8_t)kd_stall_dna[i];
            for (int i = tid; i < ad; i += 256) dst[st + i] = (uint8_t)kd_adaptor_dna[i];
            read_at = st + ad;
        }
    }
    for (int i = tid; i < rd.len1; i += 256) dst[rd.len0 + i] = (uint8_t)kd_stall_rna[i];
    const uint8_t* src = G.seq + rec.src;
    const bool rev = rec.strand == '-';
    if (tid == 0) carry = 0;
    {   // complement table of src/seq.h:78-112 (anything that is not ACGT/acgt -> 'T')
        uint8_t o;
        switch (tid) {
        case 'A': case 'a': o = 'T'; break;
        case 'C': case 'c': o = 'G'; break;
        case 'G': case 'g': o = 'C'; break;
        case 'T': case 't': o = 'A'; break;
        default: o = 'T'; break;
        }
        comp[tid] = o;
    }
    __syncthreads();
    int i_from = 0;
    if (!rec.n_N) {
        // no 'N' to substitute (almost every read): eight bases per thread and step, the reverse strand through the table and a
        // byte swap; the last n % 8 bases take the byte path below
        const int n8 = n & ~7;
        for (int i = tid * 8; i < n8; i += 2048) {
            unsigned long long v;
            __builtin_memcpy(&v, src + i, 8);
            if (rev) {
                unsigned long long w = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) w = (w << 8) | comp[(uint32_t)(v >> (8 * q)) & 0xffu];   // complemented, last base first
                __builtin_memcpy(dst + read_at + n - 8 - i, &w, 8);
            } else __builtin_memcpy(dst + read_at + i, &v, 8);
        }
        i_from = n8;
    }
    for (int i0 = i_from; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        uint8_t c = i < n ? src[i] : (uint8_t)'A';
        if (rec.n_N) {                                                // ordinal of every 'N' in forward order
            const bool isN = i < n && c == 'N';
            const unsigned long long m = __ballot(isN);
            if (lane == 0) wcnt[wid] = __popcll(m);
            __syncthreads();
            int before = carry;
            for (int w2 = 0; w2 < wid; w2++) before += wcnt[w2];
            const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            if (isN) {
                const int j = before + __popcll(m & ((1ull << lane) - 1)); // 0-based; draw j+1 of the state-100 stream
                const uint32_t cst = lcg_mul(100u, lcg_pow_a((unsigned long long)j + 1));
                const int v = (int)round(lcg_uniform(cst) * 3);
                c = v == 0 ? 'A' : v == 1 ? 'C' : v == 2 ? 'G' : 'T';
            }
            __syncthreads();
            if (tid == 0) carry += tot;
            __syncthreads();
        }
        if (i < n) {
            if (rev) dst[read_at + n - 1 - i] = comp[c];
            else dst[read_at + i] = c;
        }
    }
    if (mstate && G.meth && G.meth_has[rec.ref_idx]) {               // methylate_dna, src/genread.c:207-241
        __syncthreads();                                              // the read is in place (same workgroup wrote it)
        const uint32_t ms = mstate[r];
        const uint8_t* mf = G.meth + rec.src;
        if (tid == 0) carry = 0;
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += 256) {
            const int i = i0 + tid;
            const bool cg = i < n && meth_is_cpg(src, i, n, rec.ref_pos, rec.ref_len);
            const unsigned long long m = __ballot(cg);
            if (lane == 0) wcnt[wid] = __popcll(m);
            __syncthreads();
            int before = carry;
            for (int w2 = 0; w2 < wid; w2++) before += wcnt[w2];
            const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            if (cg) {
                const int j = before + __popcll(m & ((1ull << lane) - 1));     // the read's j-th CpG takes draw j+1
                const uint32_t cst = lcg_mul(ms, lcg_pow_a((unsigned long long)j + 1));
                const int methr = (int)(lcg_uniform(cst) * 254);
                if (methr <= (int)mf[i]) dst[read_at + (rev ? n - i - 2 : i)] = 'M';
            }
            __syncthreads();
            if (tid == 0) carry += tot;
            __syncthreads();
        }
    }
}

