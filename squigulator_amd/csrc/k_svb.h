// k_svb.h -- slow5lib's svb-zd signal compression on the device: k_svb_size, k_svb_scan, k_svb_encode
// Part of the device code of the per-read signal path; included through sqg_kernels.h (see there for the overview).
#pragma once

// ---- svb-zd: slow5lib's signal compression, per read ("next" row) -----
// slow5lib/src/slow5_press.c:1055-1087: int16 -> zig-zag of the delta to the previous sample (first: to 0) ->
// StreamVByte: uint32 count | ceil(count/4) key bytes (2 bits per value = bytes-1, first value in the low
// bits) | the values' 1-4 little-endian bytes.  One quad of samples (= one key byte) per thread.
// One quad: samples 4q..4q+3 arrive in one 8-byte load; the predecessor of the quad's first sample is the last
// sample of the lane to the left (DPP), the wavefront's first lane fetches it.  `full`: the quad has 4 samples.
__device__ static inline void svb_quad(const int16_t* __restrict__ sig, long long n, long long q, int lane,
                                       uint32_t z[4], uint32_t& key, uint32_t& nbytes) {
    int32_t v[5];
    const long long i0 = 4 * q;
    if (i0 + 4 <= n) {
        unsigned long long w;
        __builtin_memcpy(&w, sig + i0, 8);                                       // 2-byte aligned 8-byte load
        v[1] = (int16_t)(w & 0xffff); v[2] = (int16_t)((w >> 16) & 0xffff); v[3] = (int16_t)((w >> 32) & 0xffff); v[4] = (int16_t)(w >> 48);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) v[j + 1] = (i0 + j < n) ? (int32_t)sig[i0 + j] : 0;
    }
    // all lanes of the wavefront call this together with consecutive q (inactive quads carry zeros)
    v[0] = __builtin_amdgcn_update_dpp(0, v[4], 0x138, 0xf, 0xf, false);        // wave_shr:1 -> lane-1's last sample
    if (lane == 0) v[0] = (i0 > 0 && i0 - 1 < n) ? (int32_t)sig[i0 - 1] : 0;
    key = 0; nbytes = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int32_t d = v[j + 1] - v[j];
        z[j] = ((uint32_t)d + (uint32_t)d) ^ (uint32_t)(d >> 31);               // streamvbyte_zigzag.c
        const uint32_t code = z[j] < (1u << 8) ? 0u : z[j] < (1u << 16) ? 1u : z[j] < (1u << 24) ? 2u : 3u;
        if (i0 + j < n) { key |= code << (2 * j); nbytes += code + 1; }
    }
}

// bytes each read's encoding takes: 4 + ceil(n/4) + data bytes
__global__ __launch_bounds__(256) void k_svb_size(const int16_t* __restrict__ sig, const long long* __restrict__ sig_off,
                                                  int n_reads, long long* __restrict__ size) {
    __shared__ unsigned long long wsum[4];
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const long long n = sig_off[r + 1] - sig_off[r], nq = (n + 3) / 4;
    const int16_t* s = sig + sig_off[r];
    const int lane = threadIdx.x & 63;
    unsigned long long sum = 0;
    for (long long q0 = 0; q0 < nq; q0 += 256) {                                // whole wavefronts stay together (DPP)
        const long long q = q0 + threadIdx.x;
        uint32_t z[4], key, nb;
        svb_quad(s, n, q, lane, z, key, nb);
        sum += nb;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    if (lane == 0) wsum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) size[r] = 4 + nq + (long long)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

// exclusive scan of the per-read sizes (single workgroup), also through the pinned host mapping
__global__ __launch_bounds__(1024) void k_svb_scan(const long long* __restrict__ size, int n, long long* __restrict__ off, long long* __restrict__ host_off) {
    __shared__ long long wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int per = (n + 1023) / 1024;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    long long v = 0;
    for (int i = lo; i < hi; i++) v += size[i];
    long long x = v;
    for (int o = 1; o < 64; o <<= 1) { long long y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    long long run = x - v;
    for (int w = 0; w < wid; w++) run += wsum[w];
    for (int i = lo; i < hi; i++) { off[i] = run; host_off[i] = run; run += size[i]; }
    if (tid == 1023) { off[n] = run; host_off[n] = run; }
}

__global__ __launch_bounds__(256) void k_svb_encode(const int16_t* __restrict__ sig, const long long* __restrict__ sig_off, int n_reads,
                                                    const long long* __restrict__ svb_off, uint8_t* __restrict__ out) {
    __shared__ uint32_t wsum[2][4];
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long n = sig_off[r + 1] - sig_off[r], nq = (n + 3) / 4;
    const int16_t* s = sig + sig_off[r];
    uint8_t* o = out + svb_off[r];
    if (tid < 4) o[tid] = (uint8_t)((uint32_t)n >> (8 * tid));          // slow5_press.c:1047: the count word
    uint8_t* keys = o + 4;
    uint8_t* data = keys + nq;
    long long base = 0;                                                  // data bytes of the chunks before this one
    int buf = 0;
    for (long long q0 = 0; q0 < nq; q0 += 256, buf ^= 1) {
        const long long q = q0 + tid;
        uint32_t z[4], key, nb;
        svb_quad(s, n, q, lane, z, key, nb);                             // quads past the end carry zeros
        const int incl = wave_incl_scan_dpp((int)nb);
        if (lane == 63) wsum[buf][wid] = (uint32_t)incl;
        __syncthreads();                                                 // one barrier per chunk: the sums are double-buffered
        uint32_t woff = 0, tot = 0;
        for (int w = 0; w < 4; w++) { const uint32_t x = wsum[buf][w]; if (w < wid) woff += x; tot += x; }
        if (q < nq) {
            keys[q] = (uint8_t)key;
            uint8_t* d = data + base + woff + (uint32_t)incl - nb;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (4 * q + j < n) {
                    const uint32_t code = (key >> (2 * j)) & 3u;
                    if (code == 0) d[0] = (uint8_t)z[j];
                    else {                                                // little endian, one or two stores
                        const uint16_t lo = (uint16_t)z[j];
                        __builtin_memcpy(d, &lo, 2);
                        if (code >= 2) d[2] = (uint8_t)(z[j] >> 16);
                        if (code >= 3) d[3] = (uint8_t)(z[j] >> 24);
                    }
                    d += code + 1;
                }
            }
        }
        base += tot;
    }
}

