// Framewise feature kernels: framed RMS, STFT-2048 -> flatness / mel-128, onset strength, tempogram.
// gfx950: 256-thread workgroups (4 waves of 64); LDS-staged frames; float64 accumulation.
#include <math.h>

#include "ac_common.h"

// =================================================================================================
// Framed RMS.  One workgroup computes FPB consecutive frames from one LDS-resident span of the
// signal (the frames overlap 2-5x, so the span is read from HBM once instead of once per frame).
// =================================================================================================
__global__ __launch_bounds__(256) void k_frame_rms(const float* __restrict__ x, int64_t n, int frame, int hop,
                                                   int pad, int fpb, float* __restrict__ out, int64_t n_frames) {
    extern __shared__ float s_sq[];  // the span of samples shared by this block's frames
    const int64_t f0 = (int64_t)blockIdx.x * fpb;
    const int nf = (int)min((int64_t)fpb, n_frames - f0);
    const int64_t span0 = f0 * hop - pad;                 // first sample of the span (may be < 0)
    const int span = (nf - 1) * hop + frame;
    for (int i = threadIdx.x; i < span; i += 256) {
        const int64_t g = span0 + i;
        s_sq[i] = (g >= 0 && g < n) ? x[g] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int f = w; f < nf; f += 4) {
        const float* p = s_sq + f * hop;
        double acc = 0.0;
        for (int i = lane; i < frame; i += 64) { const double v = (double)p[i]; acc += v * v; }
        acc = wave_sum_f64(acc);
        if (lane == 0) out[f0 + f] = (float)sqrt(acc / (double)frame);
    }
}

extern "C" int ac_frame_rms(ac_ctx* ctx, const float* x, int64_t n, int frame, int hop, int center, float* out,
                            int64_t n_frames, void* stream) {
    AC_REQUIRE(ctx && x && out, "null pointer");
    AC_REQUIRE(n > 0 && frame > 0 && hop > 0 && n_frames > 0, "sizes must be positive");
    const int pad = center ? frame / 2 : 0;
    const int64_t expect = 1 + (n + 2 * (int64_t)pad - frame) / hop;
    AC_REQUIRE(n + 2 * (int64_t)pad >= frame && n_frames == expect, "n_frames != 1 + (n + 2*pad - frame)/hop");
    AC_REQUIRE(frame <= 15 * 1024, "frame too large for the LDS span");
    // frames per block: as many as fit a 64 KiB span (16 at most), so the overlap is read from HBM once
    int fpb = (16 * 1024 - frame) / hop + 1;
    fpb = fpb < 1 ? 1 : (fpb > 16 ? 16 : fpb);
    const size_t lds = ((size_t)(fpb - 1) * hop + frame) * sizeof(float);
    const int64_t blocks = (n_frames + fpb - 1) / fpb;
    AC_REQUIRE(blocks < (1LL << 31), "too many frames");
    hipLaunchKernelGGL(k_frame_rms, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, n, frame, hop, pad, fpb, out, n_frames);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// Several (frame, hop) configurations of one wave in one pass: a block stages the samples around AC_RMSM_SPAN frame centres once
// and its waves walk every configuration's frames centred in that range (summation exactly as k_frame_rms: bit-identical series).
#define AC_RMSM_SPAN 8192
struct RmsMultiCfg {
    int n_cfg, reach;                     // reach = samples staged on either side of the block's centre range
    int frame[AC_RMS_MULTI_MAX], hop[AC_RMS_MULTI_MAX];
    float* out[AC_RMS_MULTI_MAX];
    int64_t n_frames[AC_RMS_MULTI_MAX];
};

__global__ __launch_bounds__(256) void k_frame_rms_multi(const float* __restrict__ x, int64_t n, RmsMultiCfg cfg) {
    extern __shared__ float s_sq[];
    const int64_t c0 = (int64_t)blockIdx.x * AC_RMSM_SPAN;            // frame centres [c0, c0 + SPAN) belong to this block
    const int64_t span0 = c0 - cfg.reach;
    const int span = AC_RMSM_SPAN + 2 * cfg.reach;
    for (int i = threadIdx.x; i < span; i += 256) {
        const int64_t g = span0 + i;
        s_sq[i] = (g >= 0 && g < n) ? x[g] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c = 0; c < cfg.n_cfg; ++c) {
        const int frame = cfg.frame[c], hop = cfg.hop[c], pad = frame / 2;
        const int64_t f_lo = (c0 + hop - 1) / hop;
        int64_t f_hi = (c0 + AC_RMSM_SPAN + hop - 1) / hop;
        f_hi = f_hi < cfg.n_frames[c] ? f_hi : cfg.n_frames[c];
        for (int64_t f = f_lo + w; f < f_hi; f += 4) {
            const float* p = s_sq + (f * hop - pad - span0);
            double acc = 0.0;
            for (int i = lane; i < frame; i += 64) { const double v = (double)p[i]; acc += v * v; }
            acc = wave_sum_f64(acc);
            if (lane == 0) cfg.out[c][f] = (float)sqrt(acc / (double)frame);
        }
    }
}

extern "C" int ac_frame_rms_multi(ac_ctx* ctx, const float* x, int64_t n, int n_cfg, const int* frame, const int* hop,
                                  float* const* out, const int64_t* n_frames, void* stream) {
    AC_REQUIRE(ctx && x && frame && hop && out && n_frames, "null pointer");
    AC_REQUIRE(n > 0 && n_cfg >= 1 && n_cfg <= AC_RMS_MULTI_MAX, "1 <= n_cfg <= AC_RMS_MULTI_MAX");
    RmsMultiCfg cfg;
    cfg.n_cfg = n_cfg; cfg.reach = 0;
    int64_t last_centre = 0;
    for (int c = 0; c < n_cfg; ++c) {
        AC_REQUIRE(frame[c] > 0 && hop[c] > 0 && out[c] && n_frames[c] > 0, "sizes must be positive");
        const int pad = frame[c] / 2;
        AC_REQUIRE(n + 2 * (int64_t)pad >= frame[c] && n_frames[c] == 1 + (n + 2 * (int64_t)pad - frame[c]) / hop[c], "n_frames != 1 + (n + 2*pad - frame)/hop");
        AC_REQUIRE(frame[c] <= 8192, "frame too large for the LDS span");
        cfg.frame[c] = frame[c]; cfg.hop[c] = hop[c]; cfg.out[c] = out[c]; cfg.n_frames[c] = n_frames[c];
        const int reach = pad > frame[c] - pad ? pad : frame[c] - pad;
        cfg.reach = reach > cfg.reach ? reach : cfg.reach;
        const int64_t lc = (n_frames[c] - 1) * hop[c];
        last_centre = lc > last_centre ? lc : last_centre;
    }
    for (int c = n_cfg; c < AC_RMS_MULTI_MAX; ++c) { cfg.frame[c] = 1; cfg.hop[c] = 1; cfg.out[c] = nullptr; cfg.n_frames[c] = 0; }
    const int64_t blocks = last_centre / AC_RMSM_SPAN + 1;
    AC_REQUIRE(blocks < (1LL << 31), "too many frames");
    const size_t lds = ((size_t)AC_RMSM_SPAN + 2 * (size_t)cfg.reach) * sizeof(float);
    hipLaunchKernelGGL(k_frame_rms_multi, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, n, cfg);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// STFT-2048 (float64, like librosa.stft which multiplies the float64 Hann into the frames before
// the FFT and only then rounds to complex64) -> power -> flatness / mel-128.
// One workgroup per frame; real FFT via a 1024-point complex Stockham radix-4 FFT in LDS.
// =================================================================================================
__device__ inline double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// 1024-point complex forward FFT, 256 threads, 5 radix-4 Stockham passes; result in `a` (ping-pong with `b`).
__device__ inline double2* fft1024_f64(double2* a, double2* b, const double2* __restrict__ tw2048) {
    const int j = threadIdx.x;            // butterfly index, 0..255
    int Ns = 1;
#pragma unroll
    for (int pass = 0; pass < 5; ++pass) {
        const int k = j & (Ns - 1);
        double2 v0 = a[j], v1 = a[j + 256], v2 = a[j + 512], v3 = a[j + 768];
        // twiddle exp(-2*pi*i*k*m/(4*Ns)) = tw2048[k*m*(2048/(4*Ns))]
        const int stride = 512 / Ns;      // 2048 / (4*Ns)
        if (Ns > 1) {
            v1 = cmul(v1, tw2048[k * stride]);
            const int i2 = 2 * k * stride, i3 = 3 * k * stride;   // < 2048*3/4 ; tw table holds k < 1024: fold
            double2 t2 = tw2048[i2 & 1023]; if (i2 & 1024) { t2.x = -t2.x; t2.y = -t2.y; }
            double2 t3 = tw2048[i3 & 1023]; if (i3 & 1024) { t3.x = -t3.x; t3.y = -t3.y; }
            v2 = cmul(v2, t2);
            v3 = cmul(v3, t3);
        }
        // radix-4 butterfly (forward: -i rotation)
        const double2 s02 = make_double2(v0.x + v2.x, v0.y + v2.y), d02 = make_double2(v0.x - v2.x, v0.y - v2.y);
        const double2 s13 = make_double2(v1.x + v3.x, v1.y + v3.y), d13 = make_double2(v1.x - v3.x, v1.y - v3.y);
        const int base = ((j - k) << 2) + k;
        b[base] = make_double2(s02.x + s13.x, s02.y + s13.y);
        b[base + Ns] = make_double2(d02.x + d13.y, d02.y - d13.x);
        b[base + 2 * Ns] = make_double2(s02.x - s13.x, s02.y - s13.y);
        b[base + 3 * Ns] = make_double2(d02.x - d13.y, d02.y + d13.x);
        __syncthreads();
        double2* t = a; a = b; b = t;
        Ns <<= 2;
    }
    return a;
}

__global__ __launch_bounds__(256) void k_stft2048(const float* __restrict__ x, int64_t n, int hop,
                                                  const int64_t* __restrict__ frame_center,
                                                  const int64_t* __restrict__ frame_lo,
                                                  const int64_t* __restrict__ frame_hi,
                                                  const double2* __restrict__ tw, const double* __restrict__ hann,
                                                  const float* __restrict__ mel_w, const int* __restrict__ mel_lo,
                                                  const int* __restrict__ mel_hi, float* __restrict__ flat_out,
                                                  float* __restrict__ mel_out) {
    __shared__ double2 s_a[1024];
    __shared__ double2 s_b[1024];
    __shared__ float s_p[1025];
    __shared__ double s_red[8];
    const int64_t f = blockIdx.x;
    const int64_t c = frame_center ? frame_center[f] : f * (int64_t)hop;
    const int64_t lo = frame_lo ? frame_lo[f] : 0;
    const int64_t hi = frame_hi ? frame_hi[f] : n;
    const int64_t s0 = c - 1024;
    // z[m] = w[2m] x[2m] + i w[2m+1] x[2m+1]
    for (int m = threadIdx.x; m < 1024; m += 256) {
        const int64_t g0 = s0 + 2 * m, g1 = g0 + 1;
        const double a0 = (g0 >= lo && g0 < hi) ? (double)x[g0] : 0.0;
        const double a1 = (g1 >= lo && g1 < hi) ? (double)x[g1] : 0.0;
        s_a[m] = make_double2(a0 * hann[2 * m], a1 * hann[2 * m + 1]);
    }
    __syncthreads();
    const double2* Z = fft1024_f64(s_a, s_b, tw);
    // untangle: X[k] = (Z[k] + conj(Z[N-k]))/2 - i W^k (Z[k] - conj(Z[N-k]))/2, N = 1024, W = exp(-2 pi i/2048)
    for (int k = threadIdx.x; k <= 1024; k += 256) {
        const double2 zk = Z[k & 1023];
        const double2 zn = Z[(1024 - k) & 1023];
        const double2 e = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));
        const double2 o = make_double2(0.5 * (zk.x - zn.x), 0.5 * (zk.y + zn.y));
        double2 w = (k < 1024) ? tw[k] : make_double2(-1.0, 0.0);
        // -i * w * o
        const double2 wo = cmul(w, o);
        const double re = e.x + wo.y, im = e.y - wo.x;
        // librosa stores complex64, then np.abs (hypotf) and **2 in float32
        const float re32 = (float)re, im32 = (float)im;
        const float mag = (float)sqrt((double)re32 * (double)re32 + (double)im32 * (double)im32);
        s_p[k] = mag * mag;
    }
    __syncthreads();
    if (flat_out) {
        double ls = 0.0, as = 0.0;
        for (int k = threadIdx.x; k <= 1024; k += 256) {
            const float st = fmaxf(1e-10f, s_p[k]);
            ls += log((double)st);
            as += (double)st;
        }
        const double lsum = block_sum_f64_256(ls, s_red);
        const double asum = block_sum_f64_256(as, s_red + 4);
        if (threadIdx.x == 0) {
            const float gmean = (float)exp(lsum / 1025.0);
            const float amean = (float)(asum / 1025.0);
            flat_out[f] = gmean / amean;
        }
    }
    if (mel_out) {
        // 2 threads per mel band: even/odd halves of the band's non-zero range
        const int m = threadIdx.x >> 1, half = threadIdx.x & 1;
        const int l = mel_lo[m], h = mel_hi[m];
        const float* wrow = mel_w + (size_t)m * 1025;
        double acc = 0.0;
        for (int k = l + half; k < h; k += 2) acc += (double)wrow[k] * (double)s_p[k];
        acc += __shfl_xor(acc, 1, AC_WAVE);
        if (half == 0) mel_out[f * 128 + m] = (float)acc;
    }
}

extern "C" int ac_stft2048_features(ac_ctx* ctx, const float* x, int64_t n, int hop, const int64_t* frame_center,
                                    const int64_t* frame_lo, const int64_t* frame_hi, float* flat_out, float* mel_out,
                                    int64_t n_frames, void* stream) {
    AC_REQUIRE(ctx && x, "null pointer");
    AC_REQUIRE(n > 0 && hop > 0 && n_frames > 0, "sizes must be positive");
    AC_REQUIRE(flat_out || mel_out, "at least one output");
    AC_REQUIRE((frame_lo == nullptr) == (frame_hi == nullptr), "frame_lo/frame_hi come together");
    if (!frame_center) AC_REQUIRE(n_frames == 1 + n / hop, "n_frames != 1 + n/hop");
    AC_REQUIRE(n_frames < (1LL << 31), "too many frames");
    hipLaunchKernelGGL(k_stft2048, dim3((unsigned)n_frames), dim3(256), 0, (hipStream_t)stream, x, n, hop, frame_center,
                       frame_lo, frame_hi, ctx->tw2048, ctx->hann2048, ctx->mel_w, ctx->mel_lo, ctx->mel_hi, flat_out, mel_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// Spectral centroid and low-third magnitude ratio per frame (the multi-feature detector branch):
//   centroid = sum_k f_k * float32(S_k / sum_k S_k)           (librosa.feature.spectral_centroid; `normalize` measures
//              the column in float64 and stores S / length back as float32)
//   ratio    = low / ((low + high) + 1e-10), low = sum S[:1025 // 3], high = the rest, float32
//              (`_calculate_harmonic_ratio_direct`, pure_vocal_pause_detector.py:936-957)
__global__ __launch_bounds__(256) void k_stft2048_spectral(const float* __restrict__ x, int64_t n, int hop, double sr,
                                                           const double2* __restrict__ tw, const double* __restrict__ hann,
                                                           double* __restrict__ centroid_out, float* __restrict__ ratio_out) {
    __shared__ double2 s_a[1024];
    __shared__ double2 s_b[1024];
    __shared__ float s_m[1025];
    __shared__ double s_red[12];
    const int64_t f = blockIdx.x;
    const int64_t s0 = f * (int64_t)hop - 1024;
    for (int m = threadIdx.x; m < 1024; m += 256) {
        const int64_t g0 = s0 + 2 * m, g1 = g0 + 1;
        const double a0 = (g0 >= 0 && g0 < n) ? (double)x[g0] : 0.0;
        const double a1 = (g1 >= 0 && g1 < n) ? (double)x[g1] : 0.0;
        s_a[m] = make_double2(a0 * hann[2 * m], a1 * hann[2 * m + 1]);
    }
    __syncthreads();
    const double2* Z = fft1024_f64(s_a, s_b, tw);
    for (int k = threadIdx.x; k <= 1024; k += 256) {
        const double2 zk = Z[k & 1023];
        const double2 zn = Z[(1024 - k) & 1023];
        const double2 e = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));
        const double2 o = make_double2(0.5 * (zk.x - zn.x), 0.5 * (zk.y + zn.y));
        double2 w = (k < 1024) ? tw[k] : make_double2(-1.0, 0.0);
        const double2 wo = cmul(w, o);
        const float re32 = (float)(e.x + wo.y), im32 = (float)(e.y - wo.x);     // complex64 storage, then np.abs
        s_m[k] = (float)sqrt((double)re32 * (double)re32 + (double)im32 * (double)im32);
    }
    __syncthreads();
    double tot = 0.0, low = 0.0;
    for (int k = threadIdx.x; k <= 1024; k += 256) { const double v = (double)s_m[k]; tot += v; if (k < 1025 / 3) low += v; }
    const double length = block_sum_f64_256(tot, s_red);
    const double lowsum = block_sum_f64_256(low, s_red + 4);
    const double len_eff = length < 1.17549435e-38 ? 1.0 : length;
    double c = 0.0;
    for (int k = threadIdx.x; k <= 1024; k += 256) {
        const float sn = (float)((double)s_m[k] / len_eff);
        c += ((double)k * sr / 2048.0) * (double)sn;
    }
    const double csum = block_sum_f64_256(c, s_red + 8);
    if (threadIdx.x == 0) {
        centroid_out[f] = csum;
        const float lo32 = (float)lowsum, hi32 = (float)(length - lowsum);
        ratio_out[f] = lo32 / ((lo32 + hi32) + 1e-10f);
    }
}

extern "C" int ac_stft2048_spectral(ac_ctx* ctx, const float* x, int64_t n, int hop, double sr, double* centroid_out, float* ratio_out,
                                     int64_t n_frames, void* stream) {
    AC_REQUIRE(ctx && x && centroid_out && ratio_out, "null pointer");
    AC_REQUIRE(n > 0 && hop > 0 && n_frames == 1 + n / hop && n_frames < (1LL << 31), "n_frames != 1 + n/hop");
    hipLaunchKernelGGL(k_stft2048_spectral, dim3((unsigned)n_frames), dim3(256), 0, (hipStream_t)stream, x, n, hop, sr, ctx->tw2048,
                       ctx->hann2048, centroid_out, ratio_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// Onset strength from mel power: dB (float32 like librosa.power_to_db), per-group top_db clip,
// lag-1 positive difference, mean / median over the 128 bands, librosa's left padding.
// =================================================================================================
__device__ inline float to_db(float p) { return (float)(10.0 * log10((double)fmaxf(1e-10f, p))); }

// monotone float <-> int key for atomicMax on floats of either sign
__device__ inline int f2key(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ inline float key2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

__global__ __launch_bounds__(256) void k_mel_group_max(const float* __restrict__ mel, const int64_t* __restrict__ group_start,
                                                       int* __restrict__ gmax_key) {
    // grid.x = blocks per group, grid.y = group
    const int g = blockIdx.y;
    const int64_t a = group_start[g] * 128, b = group_start[g + 1] * 128;
    float m = -INFINITY;
    for (int64_t i = a + (int64_t)blockIdx.x * 256 + threadIdx.x; i < b; i += (int64_t)gridDim.x * 256) m = fmaxf(m, to_db(mel[i]));
    m = wave_max_f32(m);
    if ((threadIdx.x & 63) == 0 && m > -INFINITY) atomicMax(&gmax_key[g], f2key(m));
}

__global__ __launch_bounds__(128) void k_onset_env(const float* __restrict__ mel, const int64_t* __restrict__ group_start,
                                                   int n_groups, int pad, int aggregate, const int* __restrict__ gmax_key,
                                                   float* __restrict__ env) {
    __shared__ float s_v[128];
    __shared__ double s_red[2];
    const int64_t j = blockIdx.x;     // output frame (global)
    // locate the group of frame j (n_groups is small: linear scan by every thread)
    int g = 0;
    while (g + 1 < n_groups && j >= group_start[g + 1]) ++g;
    const int64_t g0 = group_start[g];
    const int64_t local = j - g0;
    if (local < pad) {                // librosa left padding: zeros
        if (threadIdx.x == 0) env[j] = 0.f;
        return;
    }
    // env[local] = aggregate_m max(0, S[m, local-pad+1] - S[m, local-pad])
    const int64_t t1 = g0 + local - pad + 1, t0 = t1 - 1;
    const float floor_db = key2f(gmax_key[g]) - 80.0f;
    const int m = threadIdx.x;
    const float d1 = fmaxf(to_db(mel[t1 * 128 + m]), floor_db);
    const float d0 = fmaxf(to_db(mel[t0 * 128 + m]), floor_db);
    const float v = fmaxf(0.f, d1 - d0);
    if (aggregate == 0) {
        double s = wave_sum_f64((double)v);
        if ((m & 63) == 0) s_red[m >> 6] = s;
        __syncthreads();
        if (m == 0) env[j] = (float)((s_red[0] + s_red[1]) / 128.0);
    } else {
        s_v[m] = v;
        __syncthreads();
        int rank = 0;
        for (int q = 0; q < 128; ++q) {
            const float u = s_v[q];
            rank += (u < v) || (u == v && q < m);
        }
        __syncthreads();
        if (rank == 63) s_v[0] = v;   // safe: all reads of s_v finished at the barrier above
        if (rank == 64) s_v[1] = v;
        __syncthreads();
        if (m == 0) env[j] = (s_v[0] + s_v[1]) * 0.5f;
    }
}

extern "C" int ac_onset_strength(ac_ctx* ctx, const float* mel, int64_t n_frames, const int64_t* group_start, int n_groups,
                                 int hop, int aggregate, float* env_out, float* scratch, void* stream) {
    AC_REQUIRE(ctx && mel && group_start && env_out && scratch, "null pointer");
    AC_REQUIRE(n_frames > 0 && n_groups > 0 && hop > 0, "sizes must be positive");
    AC_REQUIRE(aggregate == 0 || aggregate == 1, "aggregate is 0 (mean) or 1 (median)");
    AC_REQUIRE(n_frames < (1LL << 31), "too many frames");
    const int pad = 1 + 2048 / (2 * hop);
    AC_CHECK_HIP(hipMemsetAsync(scratch, 0x80, (size_t)n_groups * sizeof(int), (hipStream_t)stream));  // key of a very negative float
    hipLaunchKernelGGL(k_mel_group_max, dim3(64, n_groups), dim3(256), 0, (hipStream_t)stream, mel, group_start, (int*)scratch);
    AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_onset_env, dim3((unsigned)n_frames), dim3(128), 0, (hipStream_t)stream, mel, group_start, n_groups, pad,
                       aggregate, (const int*)scratch, env_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// Tempogram reduction.  Frame t = Hann(win) * padded_env[t : t+win] (linear-ramp padding of win/2),
// autocorrelation over all lags (direct, float64, LDS), max-normalised; never written to HBM:
// accumulated into per-part lag sums and reduced to a per-frame argmax against the host's log-prior.
// =================================================================================================
#define TG_FRAMES_PER_PART 64
#define TG_MAX_WIN 1024

extern "C" int ac_tempogram_parts(int64_t n) { return (int)((n + TG_FRAMES_PER_PART - 1) / TG_FRAMES_PER_PART); }

__device__ inline double padded_env(const float* __restrict__ env, int64_t n, int p, int64_t i) {
    // np.pad(env, (p, p), mode="linear_ramp", end_values=0): index i in [0, n + 2p)
    // numpy builds each ramp with linspace(0, edge, p, endpoint=False) in float64 and stores it in the
    // array's dtype (float32): value = float32(k * (edge / p))
    if (i < p) return (double)(float)((double)i * ((double)env[0] / (double)p));
    if (i < p + n) return (double)env[i - p];
    const int64_t r = i - (p + n);            // 0 .. p-1, ramps down to 0 at the far end
    return (double)(float)((double)(p - 1 - r) * ((double)env[n - 1] / (double)p));
}

__global__ __launch_bounds__(256) void k_tempogram(const float* __restrict__ env, int64_t n, int win,
                                                   const double* __restrict__ logprior, double* __restrict__ part_sum,
                                                   int32_t* __restrict__ argmax_out) {
    __shared__ double s_y[TG_MAX_WIN];
    __shared__ double s_ac[TG_MAX_WIN];
    __shared__ double s_w[TG_MAX_WIN];
    __shared__ double s_red[4];
    __shared__ int s_redi[4];
    const int p = win / 2;
    const int64_t t0 = (int64_t)blockIdx.x * TG_FRAMES_PER_PART;
    const int64_t t1 = min(n, t0 + TG_FRAMES_PER_PART);
    // per-thread accumulators for the lags this thread owns (lag = threadIdx.x + 256*q)
    double acc[TG_MAX_WIN / 256] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < win; i += 256) s_w[i] = 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)win);
    for (int64_t t = t0; t < t1; ++t) {
        __syncthreads();
        for (int i = threadIdx.x; i < win; i += 256) s_y[i] = padded_env(env, n, p, t + i) * s_w[i];
        __syncthreads();
        double lmax = 0.0;
        for (int q = 0; q < TG_MAX_WIN / 256; ++q) {
            const int lag = threadIdx.x + 256 * q;
            if (lag < win) {
                // one dependent float64 chain in index order (bit-identical sums); the LDS reads of eight terms are issued together in
                // front of it - read, wait, multiply, add per term left the chain latency-bound on LDS (3.2 ms per launch)
                double a = 0.0;
                int i = 0;
                const int ie = win - lag;
                for (; i + 8 <= ie; i += 8) {
                    double u[8], v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { u[k] = s_y[i + k]; v[k] = s_y[i + k + lag]; }
#pragma unroll
                    for (int k = 0; k < 8; ++k) a += u[k] * v[k];
                }
                for (; i < ie; ++i) a += s_y[i] * s_y[i + lag];
                s_ac[lag] = a;
                lmax = fmax(lmax, fabs(a));
            }
        }
        // block max of |ac| (norm = inf)
        for (int off = 32; off > 0; off >>= 1) lmax = fmax(lmax, __shfl_down(lmax, off, AC_WAVE));
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = lmax;
        __syncthreads();
        double nrm = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
        if (nrm < 2.2250738585072014e-308) nrm = 1.0;   // librosa.util.normalize: tiny -> leave un-normalised
        // normalised values: accumulate for the mean, score for the argmax
        double best = -INFINITY;
        int best_lag = 0x7fffffff;
        for (int q = 0; q < TG_MAX_WIN / 256; ++q) {
            const int lag = threadIdx.x + 256 * q;
            if (lag < win) {
                const double v = s_ac[lag] / nrm;
                acc[q] += v;
                const double sc = log1p(1e6 * v) + logprior[lag];
                if (sc > best || (sc == best && lag < best_lag)) { best = sc; best_lag = lag; }
            }
        }
        if (argmax_out) {
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_down(best, off, AC_WAVE);
                const int ol = __shfl_down(best_lag, off, AC_WAVE);
                if (ob > best || (ob == best && ol < best_lag)) { best = ob; best_lag = ol; }
            }
            __syncthreads();
            if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = best; s_redi[threadIdx.x >> 6] = best_lag; }
            __syncthreads();
            if (threadIdx.x == 0) {
                double b = s_red[0]; int bl = s_redi[0];
                for (int w = 1; w < 4; ++w)
                    if (s_red[w] > b || (s_red[w] == b && s_redi[w] < bl)) { b = s_red[w]; bl = s_redi[w]; }
                argmax_out[t] = (bl == 0x7fffffff) ? 0 : bl;
            }
        }
    }
    for (int q = 0; q < TG_MAX_WIN / 256; ++q) {
        const int lag = threadIdx.x + 256 * q;
        if (lag < win) part_sum[(int64_t)blockIdx.x * win + lag] = acc[q];
    }
}

__global__ void k_tempogram_mean(const double* __restrict__ part_sum, int n_parts, int win, int64_t n, double* __restrict__ mean_out) {
    const int lag = blockIdx.x * blockDim.x + threadIdx.x;
    if (lag >= win) return;
    double s = 0.0;
    for (int pidx = 0; pidx < n_parts; ++pidx) s += part_sum[(int64_t)pidx * win + lag];   // fixed order: deterministic
    mean_out[lag] = s / (double)n;
}

extern "C" int ac_tempogram_reduce(ac_ctx* ctx, const float* env, int64_t n, int win, const double* logprior,
                                   double* mean_out, int32_t* argmax_out, double* scratch, void* stream) {
    AC_REQUIRE(ctx && env && logprior && mean_out && scratch, "null pointer");
    AC_REQUIRE(n > 0 && win >= 2 && win <= TG_MAX_WIN, "win must be in [2, 1024]");
    const int parts = ac_tempogram_parts(n);
    hipLaunchKernelGGL(k_tempogram, dim3(parts), dim3(256), 0, (hipStream_t)stream, env, n, win, logprior, scratch, argmax_out);
    AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_tempogram_mean, dim3((win + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, parts, win, n, mean_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// YIN fundamental frequency ("autocorrelation F0"): the deterministic first stage of librosa.pyin
// (pure_vocal_pause_detector.py:422-428) = librosa.yin.  Per centred frame of `frame_length` samples
// (W = frame_length/2): acf[tau] = sum_{j=1..W} y_j y_{j+tau}, E[tau] = sum_{j=tau+1..tau+W} y_j^2,
// d[tau] = E[0] + E[tau] - 2 acf[tau], cumulative-mean-normalised, first trough under the threshold
// (else the global minimum) with parabolic refinement.  One workgroup per frame, frame in LDS, float64;
// |acf| and |E| below 1e-6 are snapped to zero like librosa does on its float32 FFT result.
// =================================================================================================
#define YIN_MAX_FRAME 4096
#define NQ_INF_I 0x7fffffffffffffffLL
#define YIN_MAX_LAGS 2048

__global__ __launch_bounds__(256) void k_yin(const float* __restrict__ x, int64_t n, int frame_length, int hop,
                                             int min_period, int max_period, double threshold,
                                             double* __restrict__ period_out, double* __restrict__ cmnd_out) {
    // Precision follows librosa under the reference's pinned numpy (< 2): np.fft works in double, so the
    // autocorrelation is float64, while the windowed energies come from a float32 np.cumsum of the squared frame
    // (sequential float32 adds, reproduced here on one thread); yin = e[0] + e[tau] - 2 acf and everything after it
    // is float64 (tiny = float64's).
    extern __shared__ double s_dyn[];
    double* s_x = s_dyn;                               // frame_length
    double* s_yin = s_x + frame_length;                // max_period + 1
    double* s_cm = s_yin + (max_period + 1);           // max_period + 1 (cmnd, index by tau)
    float* s_cs = reinterpret_cast<float*>(s_cm + (max_period + 1));   // frame_length: inclusive float32 cumsum of x^2
    __shared__ double s_red[4];
    __shared__ double s_scan[4];
    __shared__ long long s_idx[4];
    const int64_t f = blockIdx.x;
    const int W = frame_length / 2;
    const int64_t s0 = f * (int64_t)hop - frame_length / 2;
    for (int i = threadIdx.x; i < frame_length; i += 256) {
        const int64_t g = s0 + i;
        s_x[i] = (g >= 0 && g < n) ? (double)x[g] : 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float run = 0.f;
        for (int i = 0; i < frame_length; ++i) { const float v = (float)s_x[i]; run = run + v * v; s_cs[i] = run; }
    }
    __syncthreads();
    // difference function for tau = 0 .. max_period: energy[tau] = cs[tau + W] - cs[tau] (float32), acf in float64
    float e0 = s_cs[W] - s_cs[0];
    if (fabsf(e0) < 1e-6f) e0 = 0.f;
    for (int tau = threadIdx.x; tau <= max_period; tau += 256) {
        double acf = 0.0;
        for (int j = 1; j <= W; ++j) acf += s_x[j] * s_x[j + tau];
        if (fabs(acf) < 1e-6) acf = 0.0;
        float e = s_cs[tau + W] - s_cs[tau];
        if (fabsf(e) < 1e-6f) e = 0.f;
        s_yin[tau] = (double)(e0 + e) - 2.0 * acf;     // float32 + float32, then float64 with the autocorrelation
    }
    __syncthreads();
    // cumulative mean over tau = 1 .. max_period (np.cumsum: sequential float64), cmnd[tau] = yin[tau] / (cm[tau] + tiny)
    if (threadIdx.x == 0) {
        double run = 0.0;
        for (int t = 1; t <= max_period; ++t) {
            run += s_yin[t];
            s_cm[t] = s_yin[t] / (run / (double)t + 2.2250738585072014e-308);
        }
    }
    __syncthreads();
    const int n_lags = max_period - min_period + 1;
    if (cmnd_out) {
        double* row = cmnd_out + f * (int64_t)n_lags;
        for (int i = threadIdx.x; i < n_lags; i += 256) row[i] = s_cm[min_period + i];
    }
    // first trough below the threshold, else the first global minimum (indices relative to min_period)
    long long first = NQ_INF_I;
    double gmin = INFINITY; long long gidx = NQ_INF_I;
    for (int i = threadIdx.x; i < n_lags; i += 256) {
        const double v = s_cm[min_period + i];
        const double left = s_cm[min_period + (i > 0 ? i - 1 : 0)];
        const double right = s_cm[min_period + (i < n_lags - 1 ? i + 1 : n_lags - 1)];
        bool trough = (i == 0) ? (n_lags > 1 && v < s_cm[min_period + 1]) : (v < left && v <= right);
        if (trough && v < threshold && i < first) first = i;
        if (v < gmin) { gmin = v; gidx = i; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const long long of = __shfl_down(first, off, AC_WAVE); first = of < first ? of : first;
        const double ov = __shfl_down(gmin, off, AC_WAVE); const long long oi = __shfl_down(gidx, off, AC_WAVE);
        if (ov < gmin || (ov == gmin && oi < gidx)) { gmin = ov; gidx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_idx[threadIdx.x >> 6] = first; s_red[threadIdx.x >> 6] = gmin; s_scan[threadIdx.x >> 6] = (double)gidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long fi = s_idx[0]; double gm = s_red[0]; long long gi = (long long)s_scan[0];
        for (int w = 1; w < 4; ++w) {
            fi = s_idx[w] < fi ? s_idx[w] : fi;
            const long long oi = (long long)s_scan[w];
            if (s_red[w] < gm || (s_red[w] == gm && oi < gi)) { gm = s_red[w]; gi = oi; }
        }
        const long long pick = (fi != NQ_INF_I) ? fi : gi;
        double shift = 0.0;
        if (pick > 0 && pick < n_lags - 1) {
            const double xm = s_cm[min_period + pick - 1], x0 = s_cm[min_period + pick], xp = s_cm[min_period + pick + 1];
            const double a = xp + xm - 2.0 * x0;
            const double b = (xp - xm) / 2.0;
            if (!(fabs(b) >= fabs(a))) shift = -b / a;
        }
        period_out[f] = (double)min_period + (double)pick + shift;
    }
}

extern "C" int ac_yin_f0(ac_ctx* ctx, const float* x, int64_t n, int frame_length, int hop, int min_period, int max_period,
                         double threshold, double* period_out, double* cmnd_out, int64_t n_frames, void* stream) {
    AC_REQUIRE(ctx && x && period_out, "null pointer");
    AC_REQUIRE(n > 0 && hop > 0 && frame_length >= 4 && frame_length <= YIN_MAX_FRAME && (frame_length % 2) == 0, "frame_length in [4, 4096], even");
    AC_REQUIRE(min_period >= 1 && max_period > min_period + 1 && max_period <= frame_length - frame_length / 2 - 1 && max_period < YIN_MAX_LAGS,
               "1 <= min_period < max_period <= frame_length/2 - 1");
    AC_REQUIRE(n_frames == 1 + n / hop && n_frames < (1LL << 31), "n_frames != 1 + n/hop");
    const size_t lds = ((size_t)2 * frame_length + 1 + 2 * ((size_t)max_period + 1)) * sizeof(double);
    if (lds > 64 * 1024)
        AC_CHECK_HIP(hipFuncSetAttribute((const void*)k_yin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_yin, dim3((unsigned)n_frames), dim3(256), lds, (hipStream_t)stream, x, n, frame_length, hop, min_period,
                       max_period, threshold, period_out, cmnd_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
