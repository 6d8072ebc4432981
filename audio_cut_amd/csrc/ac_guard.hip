// Quiet-guard / cut-refinement kernels (cutting/refine.py, pure_vocal_pause_detector.py:1020-1094).
// All window sums add non-negative float64 terms only (no running subtraction), so windows of exact
// zeros stay exact zeros and the first-minimum tie rule of np.argmin is preserved.
#include <math.h>

#include "ac_common.h"

// -------------------------------------------------------------------------------------------------
// Inclusive scan of 256 per-thread doubles across the workgroup (4 waves).  Returns the EXCLUSIVE
// prefix for this thread; *total receives the block sum.  smem: 4 doubles.
// -------------------------------------------------------------------------------------------------
__device__ inline double block_excl_scan_256(double v, double* smem4, double* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(inc, off, AC_WAVE);
        if (lane >= off) inc += t;
    }
    __syncthreads();
    if (lane == 63) smem4[w] = inc;
    __syncthreads();
    double base = 0.0;
    for (int q = 0; q < w; ++q) base += smem4[q];
    *total = smem4[0] + smem4[1] + smem4[2] + smem4[3];
    // exclusive value = inclusive value of the previous lane (never "inc - v": that is not exact)
    const double prev = __shfl_up(inc, 1, AC_WAVE);
    return base + (lane > 0 ? prev : 0.0);
}

// =================================================================================================
// Moving mean of x^2 over `win` samples ('same' alignment: window [i - win/2, i - win/2 + win)),
// float64, as dB.  van Herk / Gil-Werman decomposition with blocks of `win` samples: a window that
// starts at offset j of block b is suffix_b[j] + prefix_{b+1}[j-1]; both scans live in LDS, every
// sample is read from HBM twice (second time from L2) and every output written once.
// =================================================================================================
#define MS_MAX_WIN 8192
#define MS_PT 32   // MS_MAX_WIN / 256

__global__ __launch_bounds__(256) void k_moving_meansq_db(const float* __restrict__ x, int64_t n, int win,
                                                          double* __restrict__ out) {
    extern __shared__ double s_buf[];          // [0,win): suffix sums of block A ; [win,2win): prefix sums of block B
    __shared__ double s_tmp[4];
    double* s_suf = s_buf;
    double* s_pre = s_buf + win;
    // workgroup b serves the window starts s in [(b-1)*win, b*win): block A = [blk0, blk0+win), B = the next one
    const int64_t blk0 = ((int64_t)blockIdx.x - 1) * (int64_t)win;
    const int per = (win + 255) / 256;
    double loc[MS_PT];
    double total;
    // ---- suffix sums of block A: thread t owns the chunk of rank 255-t, so that a forward exclusive
    //      scan over threads yields the sum of all LATER chunks (no subtraction anywhere)
    {
        const int e0 = (255 - (int)threadIdx.x) * per;
        double tsum = 0.0;
        for (int q = per - 1; q >= 0; --q) {
            const int e = e0 + q;
            double v = 0.0;
            if (e < win) { const int64_t g = blk0 + e; if (g >= 0 && g < n) { const double t = (double)x[g]; v = t * t; } }
            tsum += v;
            loc[q] = tsum;                      // sum of elements e0+q .. e0+per-1
        }
        const double later = block_excl_scan_256(tsum, s_tmp, &total);
        for (int q = 0; q < per; ++q) { const int e = e0 + q; if (e < win) s_suf[e] = loc[q] + later; }
    }
    // ---- prefix sums of block B
    {
        const int e0 = (int)threadIdx.x * per;
        double tsum = 0.0;
        for (int q = 0; q < per; ++q) {
            const int e = e0 + q;
            double v = 0.0;
            if (e < win) { const int64_t g = blk0 + win + e; if (g >= 0 && g < n) { const double t = (double)x[g]; v = t * t; } }
            tsum += v;
            loc[q] = tsum;
        }
        const double earlier = block_excl_scan_256(tsum, s_tmp, &total);
        for (int q = 0; q < per; ++q) { const int e = e0 + q; if (e < win) s_pre[e] = loc[q] + earlier; }
    }
    __syncthreads();
    // ---- outputs: start s = blk0 + j  ->  i = s + win/2
    const double inv = 1.0 / (double)win;
    const int half = win / 2;
    for (int j = threadIdx.x; j < win; j += 256) {
        const int64_t i = blk0 + j + half;
        if (i < 0 || i >= n) continue;
        double s = s_suf[j];
        if (j > 0) s += s_pre[j - 1];
        const double ms = s * inv;
        out[i] = 20.0 * log10(sqrt(ms + 1e-12) + 1e-12);
    }
}

extern "C" int ac_moving_meansq_db_f64(ac_ctx* ctx, const float* x, int64_t n, int win, double* db_out, void* stream) {
    AC_REQUIRE(ctx && x && db_out, "null pointer");
    AC_REQUIRE(n > 0 && win >= 1 && win <= MS_MAX_WIN, "win must be in [1, 8192]");
    // last window start is n - 1 - win/2 (>= -win/2); it belongs to workgroup floor(start / win) + 1
    const int64_t s_max = n - 1 - win / 2;
    const int64_t b_max = (s_max + win) / win;            // s_max + win >= 0
    AC_REQUIRE(b_max + 1 < (1LL << 31), "too many blocks");
    const size_t lds = (size_t)2 * win * sizeof(double);
    if (lds > 64 * 1024)
        AC_CHECK_HIP(hipFuncSetAttribute((const void*)k_moving_meansq_db, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_moving_meansq_db, dim3((unsigned)(b_max + 1)), dim3(256), lds, (hipStream_t)stream, x, n, win, db_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// next_out[i] = smallest j >= i with db[j] <= floor, else -1.   Three passes over blocks of 4096.
// =================================================================================================
#define NQ_BLK 4096
#define NQ_INF 0x7fffffffffffffffLL

extern "C" int64_t ac_next_leq_scratch(int64_t n) { return 2 * ((n + NQ_BLK - 1) / NQ_BLK) + 2; }

__global__ __launch_bounds__(256) void k_nq_first(const double* __restrict__ db, int64_t n, double floor_db, int64_t* __restrict__ first) {
    __shared__ long long s_m[4];
    const int64_t base = (int64_t)blockIdx.x * NQ_BLK;
    long long m = NQ_INF;
    for (int i = threadIdx.x; i < NQ_BLK; i += 256) {
        const int64_t g = base + i;
        if (g < n && db[g] <= floor_db) { m = g; break; }   // indices increase with i for a fixed thread
    }
    for (int off = 32; off > 0; off >>= 1) { const long long o = __shfl_down(m, off, AC_WAVE); m = o < m ? o : m; }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long r = s_m[0];
        for (int w = 1; w < 4; ++w) r = s_m[w] < r ? s_m[w] : r;
        first[blockIdx.x] = r;
    }
}

// carry[b] = min(first[b+1 ..]) : suffix-min over the (small) block array, single workgroup.
__global__ __launch_bounds__(256) void k_nq_carry(const int64_t* __restrict__ first, int64_t nb, int64_t* __restrict__ carry) {
    __shared__ long long s_t[256];
    const int64_t per = (nb + 255) / 256;
    const int64_t a = (int64_t)threadIdx.x * per, b = min(nb, a + per);
    long long m = NQ_INF;
    for (int64_t i = b - 1; i >= a; --i) m = first[i] < m ? first[i] : m;
    s_t[threadIdx.x] = m;
    __syncthreads();
    long long later = NQ_INF;                  // min over threads > threadIdx.x
    for (int t = threadIdx.x + 1; t < 256; ++t) later = s_t[t] < later ? s_t[t] : later;
    long long run = later;
    for (int64_t i = b - 1; i >= a; --i) {
        carry[i] = run;
        run = first[i] < run ? first[i] : run;
    }
}

__global__ __launch_bounds__(256) void k_nq_final(const double* __restrict__ db, int64_t n, double floor_db,
                                                  const int64_t* __restrict__ carry, int64_t* __restrict__ next_out) {
    __shared__ long long s_t[256];
    const int64_t base = (int64_t)blockIdx.x * NQ_BLK;
    const int per = NQ_BLK / 256;              // 16 consecutive elements per thread
    const int64_t a = base + (int64_t)threadIdx.x * per;
    long long loc[NQ_BLK / 256];
    long long m = NQ_INF;
    for (int q = per - 1; q >= 0; --q) {
        const int64_t g = a + q;
        if (g < n && db[g] <= floor_db) m = g;
        loc[q] = m;
    }
    s_t[threadIdx.x] = m;
    __syncthreads();
    long long later = carry[blockIdx.x];
    for (int t = threadIdx.x + 1; t < 256; ++t) later = s_t[t] < later ? s_t[t] : later;
    for (int q = 0; q < per; ++q) {
        const int64_t g = a + q;
        if (g < n) {
            const long long r = loc[q] < later ? loc[q] : later;
            next_out[g] = (r == NQ_INF) ? -1 : r;
        }
    }
}

extern "C" int ac_next_leq_scan(ac_ctx* ctx, const double* db, int64_t n, double floor_db, int64_t* next_out,
                                int64_t* scratch, void* stream) {
    AC_REQUIRE(ctx && db && next_out && scratch, "null pointer");
    AC_REQUIRE(n > 0, "n must be positive");
    const int64_t nb = (n + NQ_BLK - 1) / NQ_BLK;
    AC_REQUIRE(nb < (1LL << 31), "too many blocks");
    int64_t* first = scratch;
    int64_t* carry = scratch + nb + 1;
    hipLaunchKernelGGL(k_nq_first, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, db, n, floor_db, first);
    AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_nq_carry, dim3(1), dim3(256), 0, (hipStream_t)stream, first, nb, carry);
    AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_nq_final, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, db, n, floor_db, carry, next_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// First argmin over k windows of a float64 series.  One workgroup per window.
// =================================================================================================
__device__ inline void argmin_combine(double& v, long long& i, double ov, long long oi) {
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__device__ inline void block_argmin_256(double& v, long long& i, double* s_v, long long* s_i) {
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(v, off, AC_WAVE);
        const long long oi = __shfl_down(i, off, AC_WAVE);
        argmin_combine(v, i, ov, oi);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = v; s_i[threadIdx.x >> 6] = i; }
    __syncthreads();
    v = s_v[0]; i = s_i[0];
    for (int w = 1; w < 4; ++w) argmin_combine(v, i, s_v[w], s_i[w]);
}

__global__ __launch_bounds__(256) void k_window_argmin(const double* __restrict__ db, int64_t n, const int64_t* __restrict__ start,
                                                       const int64_t* __restrict__ len, int64_t* __restrict__ arg_out,
                                                       double* __restrict__ val_out) {
    __shared__ double s_v[4];
    __shared__ long long s_i[4];
    const int q = blockIdx.x;
    const int64_t a = start[q];
    const int64_t b = min(n, a + len[q]);
    double v = INFINITY; long long bi = NQ_INF;
    for (int64_t g = a + threadIdx.x; g < b; g += 256) {
        const double d = db[g];
        // NaN never wins (np.argmin would return the first NaN; the dB series has no NaN by construction)
        if (d < v) { v = d; bi = g; }
    }
    block_argmin_256(v, bi, s_v, s_i);
    if (threadIdx.x == 0) {
        if (bi == NQ_INF) { bi = a; v = (a < n) ? db[a] : NAN; }
        arg_out[q] = bi;
        val_out[2 * q] = (a < n) ? db[a] : NAN;
        val_out[2 * q + 1] = v;
    }
}

extern "C" int ac_window_argmin_f64(ac_ctx* ctx, const double* db, int64_t n, const int64_t* start, const int64_t* len, int k,
                                    int64_t* arg_out, double* val_out, void* stream) {
    AC_REQUIRE(ctx && db && start && len && arg_out && val_out, "null pointer");
    AC_REQUIRE(n > 0 && k > 0, "sizes must be positive");
    hipLaunchKernelGGL(k_window_argmin, dim3(k), dim3(256), 0, (hipStream_t)stream, db, n, start, len, arg_out, val_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// Nearest zero crossing (refine.py:72-110).  One wave per query; positions pos in [lo, hi] examine
// the pair (x[pos-1], x[pos]); float64 position (numpy<2 scalar promotion), first minimum wins.
// =================================================================================================
__global__ __launch_bounds__(64) void k_zero_cross(const float* __restrict__ x, int64_t n, const int64_t* __restrict__ idx,
                                                   int half, double* __restrict__ pos_out) {
    const int q = blockIdx.x;
    const int64_t c = idx[q];
    if (c <= 0 || c >= n) { if (threadIdx.x == 0) pos_out[q] = NAN; return; }
    const int64_t lo = max((int64_t)1, c - half), hi = min(n - 1, c + half);
    if (hi <= lo) { if (threadIdx.x == 0) pos_out[q] = NAN; return; }
    double best_d = INFINITY, best_z = NAN;
    long long best_p = NQ_INF;
    for (int64_t p = lo + threadIdx.x; p <= hi; p += 64) {
        const float l = x[p - 1], r = x[p];
        double z;
        if (l == 0.0f) z = (double)(p - 1);
        else if (r == 0.0f) z = (double)p;
        else if (l * r < 0.0f) {                         // float32 product, like the reference's scalar loop
            const float al = fabsf(l), den = al + fabsf(r);
            const float frac = (den > 1e-12f) ? al / den : 0.5f;   // float32 division
            z = (double)(p - 1) + (double)frac;
        } else continue;
        const double d = fabs(z - (double)c);
        if (d < best_d) { best_d = d; best_z = z; best_p = p; }     // strict <: earliest position wins within a lane
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_down(best_d, off, AC_WAVE);
        const double oz = __shfl_down(best_z, off, AC_WAVE);
        const long long op = __shfl_down(best_p, off, AC_WAVE);
        if (od < best_d || (od == best_d && op < best_p)) { best_d = od; best_z = oz; best_p = op; }
    }
    if (threadIdx.x == 0) pos_out[q] = (best_p == NQ_INF) ? NAN : best_z;
}

extern "C" int ac_zero_cross_nearest(ac_ctx* ctx, const float* x, int64_t n, const int64_t* idx, int half, int k,
                                     double* pos_out, void* stream) {
    AC_REQUIRE(ctx && x && idx && pos_out, "null pointer");
    AC_REQUIRE(n > 0 && k > 0 && half >= 1, "sizes must be positive");
    hipLaunchKernelGGL(k_zero_cross, dim3(k), dim3(64), 0, (hipStream_t)stream, x, n, idx, half, pos_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// Slow quiet guard (refine.py:113-157).  segment = x[idx : min(n, idx+span)); if |segment| <= win the
// level is the raw segment, else the edge-padded 'valid' window mean of float32 squares (float64
// accumulation) -> sqrt(. + eps) ; dB = 20 log10(level + eps) ; first argmin.
// One workgroup per query.  Each thread owns a run of consecutive windows: the samples common to
// all of its windows are summed once, the rest are short non-negative head/tail sums.
// =================================================================================================
#define QG_RUN_MAX 128

__global__ __launch_bounds__(256) void k_quiet_guard_slow(const float* __restrict__ x, int64_t n, const int64_t* __restrict__ idx,
                                                          int span, int win, int64_t* __restrict__ arg_out,
                                                          double* __restrict__ val_out) {
    extern __shared__ float s_sq[];            // float32 squares of the edge-padded segment (seg + win - 1)
    __shared__ double s_v[4];
    __shared__ long long s_i[4];
    __shared__ double s_db0;
    const int q = blockIdx.x;
    int64_t c = idx[q];
    if (c < 0) c = 0;
    const int64_t end = min(n, c + (int64_t)span);
    if (end <= c + 1) { if (threadIdx.x == 0) { arg_out[q] = -1; val_out[2 * q] = NAN; val_out[2 * q + 1] = NAN; } return; }
    const int seg = (int)(end - c);
    double v = INFINITY; long long bi = NQ_INF;
    if (seg <= win) {
        // level = raw samples (can be negative -> log10 of a negative is NaN in the reference; keep IEEE behaviour)
        for (int i = threadIdx.x; i < seg; i += 256) {
            const double d = 20.0 * log10((double)x[c + i] + 1e-12);
            if (i == 0) s_db0 = d;
            if (d < v) { v = d; bi = i; }
        }
    } else {
        const int plen = seg + win - 1;
        const float last = x[c + seg - 1];
        for (int i = threadIdx.x; i < plen; i += 256) {
            const float t = (i < seg) ? x[c + i] : last;
            s_sq[i] = t * t;                    // float32 product, as `padded * padded` on float32 input
        }
        __syncthreads();
        const double inv = 1.0 / (double)win;
        const int per = (seg + 255) / 256;      // windows per thread (<= QG_RUN_MAX by the host-side check)
        const int w0 = threadIdx.x * per;
        const int w1 = min(seg, w0 + per);
        if (w0 < w1) {
            // windows [w, w+win) for w in [w0, w1): common core [w1-1, w0+win)
            // the additions stay one dependent float64 chain in index order (bit-identical sums); only the LDS reads are taken eight
            // at a time in front of it - one read, one wait, one add per iteration left the ~3 500-term chain latency-bound on LDS
            // (0.2-3.5 ms per launch for a few dozen queries)
            double core = 0.0;
            int ci = w1 - 1;
            const int ce = w0 + win;
            for (; ci + 8 <= ce; ci += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = s_sq[ci + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) core += (double)t[u] * inv;
            }
            for (; ci < ce; ++ci) core += (double)s_sq[ci] * inv;
            double head[QG_RUN_MAX];            // head[j] = sum of sq[w0+j .. w1-2]
            double acc = 0.0;
            for (int j = (w1 - 1 - w0) - 1; j >= 0; --j) { acc += (double)s_sq[w0 + j] * inv; head[j] = acc; }
            double tail = 0.0;                  // sum of sq[w0+win .. w+win-1]
            for (int w = w0; w < w1; ++w) {
                if (w > w0) tail += (double)s_sq[w + win - 1] * inv;
                const double hsum = (w < w1 - 1) ? head[w - w0] : 0.0;
                const double level = sqrt(hsum + core + tail + 1e-12);
                const double d = 20.0 * log10(level + 1e-12);
                if (w == 0) s_db0 = d;
                if (d < v) { v = d; bi = w; }
            }
        }
    }
    block_argmin_256(v, bi, s_v, s_i);
    if (threadIdx.x == 0) {
        arg_out[q] = (bi == NQ_INF) ? 0 : bi;
        val_out[2 * q] = s_db0;
        val_out[2 * q + 1] = (bi == NQ_INF) ? s_db0 : v;
    }
}

extern "C" int ac_quiet_guard_slow(ac_ctx* ctx, const float* x, int64_t n, const int64_t* idx, int span, int win, int k,
                                   int64_t* arg_out, double* val_out, void* stream) {
    AC_REQUIRE(ctx && x && idx && arg_out && val_out, "null pointer");
    AC_REQUIRE(n > 0 && k > 0 && span >= 1 && win >= 1, "sizes must be positive");
    AC_REQUIRE((span + 255) / 256 <= QG_RUN_MAX, "span too large (<= 32768)");
    const size_t lds = ((size_t)span + win) * sizeof(float);
    AC_REQUIRE(lds <= 150 * 1024, "span + win too large for LDS");
    if (lds > 64 * 1024)
        AC_CHECK_HIP(hipFuncSetAttribute((const void*)k_quiet_guard_slow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_quiet_guard_slow, dim3(k), dim3(256), lds, (hipStream_t)stream, x, n, idx, span, win, arg_out, val_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// Precise cut points of pauses (pure_vocal_pause_detector.py:1047-1078).
// envelope = sqrt(max(np.convolve(seg^2, ones/W, 'same'), 1e-12)) in float32; argmin over the
// segment, then the same over the look-ahead x[cut : cut+guard).  np.convolve swaps operands when
// the signal is shorter than the kernel: the 'same' output then has length W (handled below).
// One workgroup per pause; window sums from an LDS float64 prefix table built per 4096-sample tile
// would need subtraction, so each thread again sums a run: core + short head/tail (non-negative).
// =================================================================================================
#define PC_RUN 16

__device__ void seg_env_argmin(const float* __restrict__ x, int64_t a, int m, int W, double& best, long long& best_i,
                               int tile_first = 0, int tile_stride = 1) {
    // per-thread partial argmin over outputs [0, max(m, W)) of the 'same' convolution of
    // x[a : a+m)^2 (float32 squares) with ones(W)/W; envelope compared as float32 like the reference.
    // output i sums samples j in [i + off - W + 1, i + off] clipped to [0, m).
    const int out_len = m >= W ? m : W;
    const int off = ((m < W ? m : W) - 1) / 2;          // (min(m, W) - 1) // 2
    const float invf = 1.0f / (float)W;
    best = INFINITY; best_i = NQ_INF;
    // tiles of 256 * PC_RUN outputs; a caller that splits a segment over workgroups takes tiles tile_first, + tile_stride, ...
    for (int tile = tile_first * 256 * PC_RUN; tile < out_len; tile += tile_stride * 256 * PC_RUN) {
        const int o0 = tile + (int)threadIdx.x * PC_RUN;
        const int o1 = min(out_len, o0 + PC_RUN);
        if (o0 >= o1) continue;
        // samples common to every window of the run: j in [o1-1+off-W+1, o0+off]
        const int core_lo = o1 - 1 + off - W + 1, core_hi = o0 + off;
        // one dependent float64 chain in index order (bit-identical sums); the loads of eight terms are issued together in front of it
        double core = 0.0;
        int j = max(core_lo, 0);
        const int je = min(core_hi, m - 1);
        for (; j + 7 <= je; j += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = x[a + j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) core += (double)(t[u] * t[u]);
        }
        for (; j <= je; ++j) { const float s = x[a + j]; core += (double)(s * s); }
        double low[PC_RUN];                         // low[i-o0] = sum over j in [i+off-W+1, core_lo)
        double acc = 0.0;
        for (int i = o1 - 2; i >= o0; --i) {
            const int j = i + off - W + 1;          // lowest sample of window i (not in window i+1)
            if (j >= 0 && j < m) { const float s = x[a + j]; acc += (double)(s * s); }
            low[i - o0] = acc;
        }
        double hi_acc = 0.0;                        // sum over j in (o0+off, i+off]
        for (int i = o0; i < o1; ++i) {
            if (i > o0) { const int j = i + off; if (j >= 0 && j < m) { const float s = x[a + j]; hi_acc += (double)(s * s); } }
            const double lowsum = (i < o1 - 1) ? low[i - o0] : 0.0;
            const float conv = (float)((lowsum + core + hi_acc) * (double)invf);
            const float env = sqrtf(fmaxf(conv, 1e-12f));
            const double e = (double)env;
            if (e < best) { best = e; best_i = i; }
        }
    }
}

// Phase 1 of a pause's cut point: the argmin of the envelope over the whole pause, split over PC_SPLIT workgroups (a long pause
// used to keep one CU busy for milliseconds while the rest of the chip idled).  The envelope is a non-negative float32 and the
// rule is "first minimum", so (float bits << 32 | output index) under an unsigned 64-bit atomicMin IS the argmin.
#define PC_SPLIT 64
__global__ __launch_bounds__(256) void k_pause_argmin(const float* __restrict__ x, const int64_t* __restrict__ pa,
                                                      const int64_t* __restrict__ pb, int win, unsigned long long* __restrict__ key,
                                                      unsigned long long* __restrict__ zeros /* aux_out[2 q], zeroed */) {
    __shared__ double s_v[4];
    __shared__ long long s_i[4];
    __shared__ int s_zero[4];
    const int q = blockIdx.x;
    const int64_t a = pa[q], b = pb[q];
    const int m = (int)(b - a);
    if (m <= 1) return;
    const int out_len = m >= win ? m : win;
    if ((int)blockIdx.y * 256 * PC_RUN >= out_len) return;
    double v; long long bi;
    seg_env_argmin(x, a, m, win, v, bi, (int)blockIdx.y, PC_SPLIT);
    block_argmin_256(v, bi, s_v, s_i);
    if (threadIdx.x == 0 && bi != NQ_INF) {
        const float e = (float)v;                                   // the envelope values are float32 (exactly representable)
        atomicMin(&key[q], ((unsigned long long)__float_as_uint(e) << 32) | (unsigned long long)(unsigned)bi);
    }
    // exact zeros of the segment (the percentile floor test), over the same tiles
    int z = 0;
    for (int tile = (int)blockIdx.y * 256 * PC_RUN; tile < m; tile += PC_SPLIT * 256 * PC_RUN) {
        const int hi = min(m, tile + 256 * PC_RUN);
        for (int i = tile + (int)threadIdx.x; i < hi; i += 256) z += (x[a + i] == 0.0f);
    }
    for (int off = 32; off > 0; off >>= 1) z += __shfl_down(z, off, AC_WAVE);
    if ((threadIdx.x & 63) == 0) s_zero[threadIdx.x >> 6] = z;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int zt = s_zero[0] + s_zero[1] + s_zero[2] + s_zero[3];
        if (zt) atomicAdd(&zeros[2 * q], (unsigned long long)zt);
    }
}

__global__ __launch_bounds__(256) void k_pause_cut(const float* __restrict__ x, int64_t n, const int64_t* __restrict__ pa,
                                                   const int64_t* __restrict__ pb, int win, int guard,
                                                   int64_t* __restrict__ cut_out, int64_t* __restrict__ aux_out) {
    __shared__ double s_v[4];
    __shared__ long long s_i[4];
    __shared__ long long s_cut;
    __shared__ int s_zero[4];
    const int q = blockIdx.x;
    const int64_t a = pa[q], b = pb[q];
    const int m = (int)(b - a);
    if (m <= 1) { if (threadIdx.x == 0) { cut_out[q] = -1; aux_out[2 * q] = 0; aux_out[2 * q + 1] = 0; } return; }
    double v; long long bi;
    const unsigned long long key = (unsigned long long)cut_out[q];          // phase 1's packed (envelope, index); all ones = no finite value
    __syncthreads();                                                        // everyone has read the key before thread 0 overwrites it
    bi = (key == ~0ULL) ? NQ_INF : (long long)(key & 0xFFFFFFFFULL);
    long long cut = a + (bi == NQ_INF ? 0 : bi);
    if (guard > 0) {
        const int64_t g_end = min(n, (int64_t)cut + guard);
        const int gm = (int)(g_end - cut);
        if (gm > 0) {
            __syncthreads();
            seg_env_argmin(x, cut, gm, win, v, bi);
            block_argmin_256(v, bi, s_v, s_i);
            long long c2 = cut + (bi == NQ_INF ? 0 : bi);
            if (c2 > g_end - 1) c2 = g_end - 1;
            cut = c2;
        }
    }
    if (threadIdx.x == 0) {              // aux_out[2 q] (zeros in the segment) was accumulated by k_pause_argmin
        cut_out[q] = cut;
        aux_out[2 * q + 1] = (x[cut] != 0.0f) ? 1 : 0;
    }
    (void)s_cut; (void)s_zero;
}

extern "C" int ac_pause_cut_points(ac_ctx* ctx, const float* x, int64_t n, const int64_t* a, const int64_t* b, int k, int win,
                                   int guard, int64_t* cut_out, int64_t* aux_out, void* stream) {
    AC_REQUIRE(ctx && x && a && b && cut_out && aux_out, "null pointer");
    AC_REQUIRE(n > 0 && k > 0 && win >= 2 && guard >= 0, "sizes must be positive");
    // cut_out doubles as the phase-1 key array: all ones, atomicMin'ed by k_pause_argmin, consumed and overwritten by k_pause_cut
    AC_CHECK_HIP(hipMemsetAsync(cut_out, 0xFF, (size_t)k * sizeof(int64_t), (hipStream_t)stream));
    AC_CHECK_HIP(hipMemsetAsync(aux_out, 0, (size_t)k * 2 * sizeof(int64_t), (hipStream_t)stream));
    hipLaunchKernelGGL(k_pause_argmin, dim3(k, PC_SPLIT), dim3(256), 0, (hipStream_t)stream, x, a, b, win, (unsigned long long*)cut_out,
                       (unsigned long long*)aux_out);
    hipLaunchKernelGGL(k_pause_cut, dim3(k), dim3(256), 0, (hipStream_t)stream, x, n, a, b, win, guard, cut_out, aux_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// Post-path boundary policy.
// (1) framed RMS of every segment of a cut list in ONE launch (`_classify_segments_vocal_presence`,
//     seamless_splitter.py:2335-2342: librosa.feature.rms(y=segment, frame 2205, hop 882), i.e. centred frames that
//     see zeros outside their own segment).  One wave per frame; the segment of a frame is found by binary search in
//     the per-segment frame offsets.
// (2) local-valley search around each boundary (`_refine_boundaries_local_valley`, :2646-2661): float64 moving mean of
//     x^2 over `win` samples ('valid'), sqrt(. + 1e-12), 20 log10(. + 1e-12); the value at the boundary and the first
//     minimum of the +-radius window.  One workgroup per boundary, 1024 outputs per LDS tile.
// =================================================================================================
__global__ __launch_bounds__(256) void k_segment_frame_rms(const float* __restrict__ x, const int64_t* __restrict__ seg_start,
                                                           const int64_t* __restrict__ seg_end, const int64_t* __restrict__ frame_off,
                                                           int n_seg, int frame, int hop, int center, float* __restrict__ out,
                                                           int64_t n_frames) {
    const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= n_frames) return;
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = n_seg - 1;                      // last segment with frame_off[s] <= f
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (frame_off[mid] <= f) lo = mid; else hi = mid - 1; }
    const int64_t a = seg_start[lo], b = seg_end[lo];
    const int64_t c0 = a + (f - frame_off[lo]) * (int64_t)hop - (center ? frame / 2 : 0);
    double acc = 0.0;
    for (int i = lane; i < frame; i += 64) {
        const int64_t g = c0 + i;
        if (g >= a && g < b) { const double v = (double)x[g]; acc += v * v; }
    }
    acc = wave_sum_f64(acc);
    if (lane == 0) out[f] = (float)sqrt(acc / (double)frame);
}

extern "C" int ac_segment_frame_rms(ac_ctx* ctx, const float* x, int64_t n, const int64_t* seg_start, const int64_t* seg_end,
                                     const int64_t* frame_off, int n_seg, int frame, int hop, int center, float* out,
                                     int64_t n_frames, void* stream) {
    AC_REQUIRE(ctx && x && seg_start && seg_end && frame_off && out, "null pointer");
    AC_REQUIRE(n > 0 && n_seg > 0 && frame > 0 && hop > 0 && n_frames > 0 && n_frames < (1LL << 33), "sizes must be positive");
    hipLaunchKernelGGL(k_segment_frame_rms, dim3((unsigned)((n_frames + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, seg_start,
                       seg_end, frame_off, n_seg, frame, hop, center, out, n_frames);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

#define LV_TILE 1024
#define NQ_INF_I 0x7fffffffffffffffLL
#define LV_MAX_WIN 2048

// grid (boundary, tile of LV_TILE outputs): a +-500 ms search is 43 tiles, which one workgroup per boundary used to walk alone
__global__ __launch_bounds__(256) void k_local_valley(const float* __restrict__ x, int64_t n, const int64_t* __restrict__ centers,
                                                      int radius, int win, int n_tiles, double* __restrict__ orig_db,
                                                      double* __restrict__ part_v, int64_t* __restrict__ part_i) {
    __shared__ double s_sq[LV_TILE + LV_MAX_WIN];
    __shared__ double s_bv[4];
    __shared__ long long s_bi[4];
    const int k = blockIdx.x, tile = blockIdx.y;
    const int64_t c = centers[k];
    const int64_t a = c - radius < 0 ? 0 : c - radius;
    const int64_t b = c + radius > n ? n : c + radius;
    const int64_t len = b - a;
    double best = INFINITY; long long besti = NQ_INF_I;
    const int64_t m = len - win + 1;                 // 'valid' outputs (<= 0: `if segment.size <= win: continue`)
    const int64_t t0 = (int64_t)tile * LV_TILE;
    if (len > win && t0 < m) {
        int64_t o = c - a - win / 2;
        o = o < 0 ? 0 : (o > m - 1 ? m - 1 : o);
        const double inv = 1.0 / (double)win;
        const int cnt = (int)((m - t0) < LV_TILE ? (m - t0) : LV_TILE);
        for (int i = threadIdx.x; i < cnt + win - 1; i += 256) { const double v = (double)x[a + t0 + i]; s_sq[i] = v * v; }
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += 256) {
            double acc = 0.0;
            for (int j = 0; j < win; ++j) acc += s_sq[i + j] * inv;      // np.convolve(sq, ones / win): products, then the sum
            const double rms = sqrt(acc + 1e-12);
            const double db = 20.0 * log10(rms + 1e-12);
            const long long gi = (long long)(t0 + i);
            if (gi == o) orig_db[k] = db;
            if (db < best) { best = db; besti = gi; }                    // ascending i per thread: first minimum kept
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off, AC_WAVE); const long long oi = __shfl_down(besti, off, AC_WAVE);
        if (ov < best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_bv[threadIdx.x >> 6] = best; s_bi[threadIdx.x >> 6] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_bv[w] < best || (s_bv[w] == best && s_bi[w] < besti)) { best = s_bv[w]; besti = s_bi[w]; }
        part_v[(size_t)k * n_tiles + tile] = best; part_i[(size_t)k * n_tiles + tile] = besti;
    }
}

// first minimum over a boundary's tiles (tiles are in index order: a strict `<` keeps the earliest)
__global__ __launch_bounds__(64) void k_local_valley_pick(const int64_t* __restrict__ centers, int n_k, int64_t n, int radius, int win, int n_tiles,
                                                          const double* __restrict__ part_v, const int64_t* __restrict__ part_i,
                                                          double* __restrict__ orig_db, double* __restrict__ min_db,
                                                          int64_t* __restrict__ min_idx) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= n_k) return;
    const int64_t c = centers[k];
    const int64_t a = c - radius < 0 ? 0 : c - radius;
    const int64_t b = c + radius > n ? n : c + radius;
    if (b - a <= win) { orig_db[k] = 0.0; min_db[k] = 0.0; min_idx[k] = -1; return; }
    double best = INFINITY; long long besti = NQ_INF_I;
    for (int t = 0; t < n_tiles; ++t) {
        const double v = part_v[(size_t)k * n_tiles + t]; const long long i = part_i[(size_t)k * n_tiles + t];
        if (v < best || (v == best && i < besti)) { best = v; besti = i; }
    }
    min_db[k] = best; min_idx[k] = besti;
}

extern "C" int ac_local_valley_tiles(int radius, int win) { const int m = 2 * radius - win + 1; return m <= 0 ? 1 : (m + LV_TILE - 1) / LV_TILE; }

extern "C" int ac_local_valley(ac_ctx* ctx, const float* x, int64_t n, const int64_t* centers, int k, int radius, int win,
                                double* orig_db, double* min_db, int64_t* min_idx, double* part_v, int64_t* part_i, void* stream) {
    AC_REQUIRE(ctx && x && centers && orig_db && min_db && min_idx && part_v && part_i, "null pointer");
    AC_REQUIRE(n > 0 && k > 0 && radius > 0 && win > 0 && win <= LV_MAX_WIN, "0 < win <= 2048, radius > 0");
    const int n_tiles = ac_local_valley_tiles(radius, win);
    AC_REQUIRE(n_tiles <= 65535, "radius too large");
    hipLaunchKernelGGL(k_local_valley, dim3((unsigned)k, (unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, x, n, centers, radius, win,
                       n_tiles, orig_db, part_v, part_i);
    hipLaunchKernelGGL(k_local_valley_pick, dim3((unsigned)((k + 63) / 64)), dim3(64), 0, (hipStream_t)stream, centers, k, n, radius,
                       win, n_tiles, part_v, part_i, orig_db, min_db, min_idx);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// (3) per-segment sum of squares (float64) and peak |x| for the weak-tail rule (`_merge_short_weak_human_tails_...`,
//     seamless_splitter.py:2179-2196) and the short-segment branch of the classifier (:2349-2358).  One workgroup per
//     (segment, sixteenth), fixed-order tree reduction; the host adds the 16 partials of a segment in order (deterministic).
#define SSP_PARTS 16
__global__ __launch_bounds__(256) void k_segment_sumsq_peak(const float* __restrict__ x, const int64_t* __restrict__ seg_start,
                                                            const int64_t* __restrict__ seg_end, double* __restrict__ part_sumsq,
                                                            float* __restrict__ part_peak) {
    // grid (segment, part): part p owns the p-th contiguous sixteenth of the segment; the host adds the parts in order
    __shared__ double s_s[4];
    __shared__ float s_p[4];
    const int s = blockIdx.x, part = blockIdx.y;
    const int64_t a = seg_start[s], b = seg_end[s];
    const int64_t chunk = (b - a + SSP_PARTS - 1) / SSP_PARTS;
    const int64_t lo = a + part * chunk, hi = (lo + chunk < b) ? lo + chunk : b;
    double acc = 0.0; float pk = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { const float v = x[i]; acc += (double)v * (double)v; pk = fmaxf(pk, fabsf(v)); }
    acc = wave_sum_f64(acc);
    pk = wave_max_f32(pk);
    if ((threadIdx.x & 63) == 0) { s_s[threadIdx.x >> 6] = acc; s_p[threadIdx.x >> 6] = pk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part_sumsq[(int64_t)s * SSP_PARTS + part] = (s_s[0] + s_s[1]) + (s_s[2] + s_s[3]);
        part_peak[(int64_t)s * SSP_PARTS + part] = fmaxf(fmaxf(s_p[0], s_p[1]), fmaxf(s_p[2], s_p[3]));
    }
}

extern "C" int ac_segment_sumsq_peak(ac_ctx* ctx, const float* x, int64_t n, const int64_t* seg_start, const int64_t* seg_end,
                                      int n_seg, double* sumsq, float* peak, void* stream) {
    AC_REQUIRE(ctx && x && seg_start && seg_end && sumsq && peak, "null pointer");
    AC_REQUIRE(n > 0 && n_seg > 0, "sizes must be positive");
    hipLaunchKernelGGL(k_segment_sumsq_peak, dim3((unsigned)n_seg, SSP_PARTS), dim3(256), 0, (hipStream_t)stream, x, seg_start, seg_end, sumsq, peak);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
