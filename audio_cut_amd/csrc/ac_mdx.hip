// MDX23 separator front/back end: windowing + STFT-6144 (hop 1024), iSTFT with overlap-add, stem
// assembly + effective-region overlap-add.  float32 like torch.stft / torch.istft in the reference.
// Internal spectrogram layout is [item][4][T=256][F=3072] (F fastest): the U-Net works T-major
// (its first op after the 1x1 conv is a transpose), so the layout makes every access coalesced.
#include <math.h>

#include "ac_common.h"

#define MDX_NFFT 6144
#define MDX_M 3072            // complex FFT size of the packed real transform
#define MDX_HOP 1024
#define MDX_T 256
#define MDX_F 3072
#define MDX_ITEM 261120       // hop * (T - 1)
#define MDX_TRIM 3072
#define MDX_GEN 254976        // ITEM - 2*TRIM

__device__ inline float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// MDX_NT_LOADS / MDX_PROBE (probe builds of tools/sharing_probe_*.py only): every global load of the FFT kernels bypasses the
// CU's vector L1 (`nt`: served by L2) / bit 2: the butterflies run without their twiddle factors (no table loads in the passes)
#ifndef MDX_NT_LOADS
#define MDX_NT_LOADS 0
#endif
#ifndef MDX_PROBE
#define MDX_PROBE 0
#endif
__device__ inline float mdx_ldf(const float* p) {
#if MDX_NT_LOADS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ inline float2 mdx_ld2(const float2* p) {
#if MDX_NT_LOADS
    const unsigned long long u = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(p));
    return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
#else
    return *p;
#endif
}

// exp(-2*pi*i*q/3072) from the half-circle table tw[k] = exp(-2*pi*i*k/6144), k < 3072
__device__ inline float2 tw3072(const float2* __restrict__ tw, int q) {
    int k = 2 * q;                       // 0 .. 6142
    if (k >= MDX_M) { const float2 t = mdx_ld2(tw + (k - MDX_M)); return make_float2(-t.x, -t.y); }
    return mdx_ld2(tw + k);
}

// 3072-point complex forward FFT in LDS (256 threads): five radix-4 Stockham passes then one radix-3.
__device__ float2* fft3072_f32(float2* a, float2* b, const float2* __restrict__ tw) {
    int Ns = 1;
    for (int pass = 0; pass < 5; ++pass) {
        for (int j = threadIdx.x; j < MDX_M / 4; j += 256) {
            const int k = j & (Ns - 1);
            float2 v0 = a[j], v1 = a[j + MDX_M / 4], v2 = a[j + MDX_M / 2], v3 = a[j + 3 * MDX_M / 4];
            if (Ns > 1 && !(MDX_PROBE & 2)) {
                const int q = k * (MDX_M / (4 * Ns));       // exp(-2 pi i k m / (4 Ns)) = W_3072^(q m)
                v1 = cmulf(v1, tw3072(tw, q));
                v2 = cmulf(v2, tw3072(tw, 2 * q));
                v3 = cmulf(v3, tw3072(tw, 3 * q));
            }
            const float2 s02 = make_float2(v0.x + v2.x, v0.y + v2.y), d02 = make_float2(v0.x - v2.x, v0.y - v2.y);
            const float2 s13 = make_float2(v1.x + v3.x, v1.y + v3.y), d13 = make_float2(v1.x - v3.x, v1.y - v3.y);
            const int base = ((j - k) << 2) + k;
            b[base] = make_float2(s02.x + s13.x, s02.y + s13.y);
            b[base + Ns] = make_float2(d02.x + d13.y, d02.y - d13.x);
            b[base + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
            b[base + 3 * Ns] = make_float2(d02.x - d13.y, d02.y + d13.x);
        }
        __syncthreads();
        float2* t = a; a = b; b = t;
        Ns <<= 2;
    }
    // radix-3 pass, Ns = 1024
    {
        const float c3 = -0.5f, s3 = -0.86602540378443864676f;   // exp(-2 pi i / 3) = c3 + i s3
        for (int j = threadIdx.x; j < MDX_M / 3; j += 256) {
            const int k = j;                                      // j < Ns = 1024
            float2 v0 = a[j], v1 = a[j + 1024], v2 = a[j + 2048];
            if (!(MDX_PROBE & 2)) {
                v1 = cmulf(v1, tw3072(tw, k));                    // exp(-2 pi i k / 3072)
                v2 = cmulf(v2, tw3072(tw, 2 * k));
            }
            const float2 s = make_float2(v1.x + v2.x, v1.y + v2.y);
            const float2 d = make_float2(v1.x - v2.x, v1.y - v2.y);
            const float2 m = make_float2(v0.x + c3 * s.x, v0.y + c3 * s.y);
            // X1 = m + (-i)(-s3) ... : X1 = v0 + w v1 + w^2 v2, w = c3 + i s3 ; X2 uses conj(w)
            const float2 r = make_float2(-s3 * d.y, s3 * d.x);    // i * s3 * d
            b[k] = make_float2(v0.x + s.x, v0.y + s.y);
            b[k + 1024] = make_float2(m.x + r.x, m.y + r.y);
            b[k + 2048] = make_float2(m.x - r.x, m.y - r.y);
        }
        __syncthreads();
        float2* t = a; a = b; b = t;
    }
    return a;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mdx_stft(const float* __restrict__ track, int64_t n,
                                                  const int64_t* __restrict__ chunk_start,
                                                  const int64_t* __restrict__ chunk_len,
                                                  const int32_t* __restrict__ win_index,
                                                  const float2* __restrict__ tw, const float* __restrict__ hann,
                                                  float* __restrict__ spec, float* __restrict__ spec_amax) {
    __shared__ float2 s_a[MDX_M];
    __shared__ float2 s_b[MDX_M];
    const int t = blockIdx.x;            // frame
    const int item = blockIdx.y;
    const int64_t cs = chunk_start[item];
    const int64_t cl = chunk_len[item];
    const int64_t woff = (int64_t)win_index[item] * MDX_GEN - MDX_TRIM;   // chunk-local index of item sample 0
    for (int m = threadIdx.x; m < MDX_M; m += 256) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * m + h;
            int jj = t * MDX_HOP - MDX_NFFT / 2 + i;            // index into the (unpadded) item
            if (jj < 0) jj = -jj;                               // torch.stft(center=True): reflect padding
            else if (jj >= MDX_ITEM) jj = 2 * (MDX_ITEM - 1) - jj;
            const int64_t q = woff + jj;                        // chunk-local sample
            const float s = (q >= 0 && q < cl) ? track[cs + q] : 0.f;
            v[h] = s * hann[i];
        }
        s_a[m] = make_float2(v[0], v[1]);
    }
    __syncthreads();
    const float2* Z = fft3072_f32(s_a, s_b, tw);
    float* out = spec + (size_t)item * 4 * MDX_T * MDX_F + (size_t)t * MDX_F;
    const size_t cstride = (size_t)MDX_T * MDX_F;
    float vmax = 0.f;
    for (int k = threadIdx.x; k < MDX_F; k += 256) {
        const float2 zk = Z[k];
        const float2 zn = Z[(MDX_M - k) % MDX_M];
        const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        const float2 o = make_float2(0.5f * (zk.x - zn.x), 0.5f * (zk.y + zn.y));
        const float2 wo = cmulf(tw[k], o);
        const float re = e.x + wo.y, im = e.y - wo.x;
        out[k] = re;                       // L.re
        out[cstride + k] = im;             // L.im
        out[2 * cstride + k] = re;         // R.re  (mono input duplicated to both channels, backends.py:269-270)
        out[3 * cstride + k] = im;         // R.im
        vmax = fmaxf(vmax, fmaxf(fabsf(re), fabsf(im)));
    }
    // max |spectrogram| per (item, frame): the first conv's time-local activation scale (ac_common.h)
    if (spec_amax) ac_amax_commit(vmax, spec_amax + (size_t)item * MDX_T + t);
}

extern "C" int ac_mdx_stft(ac_ctx* ctx, const float* track, int64_t n, const int64_t* chunk_start, const int64_t* chunk_len,
                           const int32_t* win_index, int n_items, float* spec_out, float* spec_amax, void* stream) {
    AC_REQUIRE(ctx && track && chunk_start && chunk_len && win_index && spec_out, "null pointer");
    AC_REQUIRE(n > 0 && n_items > 0 && n_items <= 65535, "n_items must be in [1, 65535]");
    hipLaunchKernelGGL(k_mdx_stft, dim3(MDX_T, n_items), dim3(256), 0, (hipStream_t)stream, track, n, chunk_start, chunk_len,
                       win_index, ctx->tw6144, ctx->hann6144, spec_out, spec_amax);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// ---------------------------------------------------------------------------------------------
// iSTFT stage 1: one workgroup per (item, channel, frame): Hermitian spectrum (top bin zero, imaginary
// part of DC ignored like a c2r transform) -> 6144 real samples * Hann -> scratch.
__global__ __launch_bounds__(256) void k_mdx_istft_frames(const float* __restrict__ spec, const float2* __restrict__ tw,
                                                          const float* __restrict__ hann, float* __restrict__ frames) {
    __shared__ float2 s_a[MDX_M];
    __shared__ float2 s_b[MDX_M];
    const int t = blockIdx.x, ch = blockIdx.y, item = blockIdx.z;
    const size_t cstride = (size_t)MDX_T * MDX_F;
    const float* re_p = spec + ((size_t)item * 4 + 2 * ch) * cstride + (size_t)t * MDX_F;
    const float* im_p = re_p + cstride;
    // Z[k] = E[k] + i O[k];  E = (X[k] + conj X[M-k])/2 ; O = conj(W^k) (X[k] - conj X[M-k])/2 ; load conj(Z) for the
    // inverse-by-forward trick
    for (int k = threadIdx.x; k < MDX_M; k += 256) {
        float2 xk = make_float2(mdx_ldf(re_p + k), k == 0 ? 0.f : mdx_ldf(im_p + k));
        float2 xm = (k == 0) ? make_float2(0.f, 0.f) : make_float2(mdx_ldf(re_p + (MDX_M - k)), mdx_ldf(im_p + (MDX_M - k)));   // X[M] = 0 (dropped bin)
        const float2 e = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y - xm.y));
        const float2 d = make_float2(0.5f * (xk.x - xm.x), 0.5f * (xk.y + xm.y));
        const float2 w = mdx_ld2(tw + k);
        const float2 o = cmulf(make_float2(w.x, -w.y), d);       // conj(W^k) * d
        // Z = e + i o = (e.x - o.y, e.y + o.x) ; store conj(Z)
        s_a[k] = make_float2(e.x - o.y, -(e.y + o.x));
    }
    __syncthreads();
    const float2* z = fft3072_f32(s_a, s_b, tw);
    const float inv = 1.0f / (float)MDX_M;
    float* out = frames + (((size_t)item * 2 + ch) * MDX_T + t) * MDX_NFFT;
    for (int m = threadIdx.x; m < MDX_M; m += 256) {
        const float2 v = z[m];                                   // conj(v)/M = z[m]
        const float x0 = v.x * inv, x1 = -v.y * inv;
        const float2 hw = mdx_ld2(reinterpret_cast<const float2*>(hann) + m);
        reinterpret_cast<float2*>(out)[m] = make_float2(x0 * hw.x, x1 * hw.y);
    }
}

// iSTFT stage 2: overlap-add the (<= 6) frames covering each sample, divide by the window envelope.
__global__ __launch_bounds__(256) void k_mdx_istft_ola(const float* __restrict__ frames, const float* __restrict__ env,
                                                       float* __restrict__ wave) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over item*2*ITEM
    const int64_t row = idx / MDX_ITEM;                                // item*2 + ch
    const int nn = (int)(idx - row * MDX_ITEM);
    const int p = nn + MDX_NFFT / 2;                                   // padded coordinate
    int t_lo = (p - (MDX_NFFT - 1) + MDX_HOP - 1) / MDX_HOP;           // ceil((p - 6143)/1024)
    if (p - (MDX_NFFT - 1) <= 0) t_lo = 0;
    int t_hi = p / MDX_HOP;
    if (t_hi > MDX_T - 1) t_hi = MDX_T - 1;
    const float* base = frames + (size_t)row * MDX_T * MDX_NFFT;
    float acc = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) acc += base[(size_t)t * MDX_NFFT + (p - t * MDX_HOP)];
    wave[idx] = acc / env[p];
}

extern "C" int ac_mdx_istft(ac_ctx* ctx, const float* spec, int n_items, float* wave_out, float* scratch, void* stream) {
    AC_REQUIRE(ctx && spec && wave_out && scratch, "null pointer");
    AC_REQUIRE(n_items > 0 && n_items <= 65535, "n_items must be in [1, 65535]");
    hipLaunchKernelGGL(k_mdx_istft_frames, dim3(MDX_T, 2, n_items), dim3(256), 0, (hipStream_t)stream, spec, ctx->tw6144,
                       ctx->hann6144, scratch);
    AC_LAUNCH_CHECK();
    const int64_t total = (int64_t)n_items * 2 * MDX_ITEM;             // multiple of 256
    hipLaunchKernelGGL(k_mdx_istft_ola, dim3((unsigned)(total / 256)), dim3(256), 0, (hipStream_t)stream, scratch,
                       ctx->ola_env6144, wave_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// ---------------------------------------------------------------------------------------------
// Stem assembly + uniform overlap-add of the effective regions, as a gather over track samples.
__global__ __launch_bounds__(256) void k_mdx_assemble_ola(const float* __restrict__ track, int64_t n, const float* __restrict__ wave,
                                                          const int64_t* __restrict__ chunk_start,
                                                          const int64_t* __restrict__ chunk_len,
                                                          const int64_t* __restrict__ eff_start,
                                                          const int64_t* __restrict__ eff_end,
                                                          const int32_t* __restrict__ item_base, int n_chunks,
                                                          float* __restrict__ vocal_out, float* __restrict__ inst_out) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    // chunks whose effective region contains g form a contiguous index range (both tables ascend)
    int lo = 0, hi = n_chunks;              // first chunk with eff_end > g
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (eff_end[mid] > g) hi = mid; else lo = mid + 1; }
    const int c_lo = lo;
    const float mix = track[g];
    float v_acc = 0.f, i_acc = 0.f, w_acc = 0.f;
    for (int c = c_lo; c < n_chunks && eff_start[c] <= g; ++c) {
        if (eff_end[c] <= g) continue;
        const int64_t q = g - chunk_start[c];
        if (q < 0 || q >= chunk_len[c]) continue;
        const int item = item_base[c] + (int)(q / MDX_GEN);
        const int pos = MDX_TRIM + (int)(q % MDX_GEN);
        const float w0 = wave[((size_t)item * 2 + 0) * MDX_ITEM + pos];
        const float w1 = wave[((size_t)item * 2 + 1) * MDX_ITEM + pos];
        const float vocal = (w0 + w1) * 0.5f;                // vocal.mean(axis=0) in float32
        const float inst = ((mix - w0) + (mix - w1)) * 0.5f; // (mix - vocal).mean(axis=0)
        v_acc += vocal;
        i_acc += inst;
        w_acc += 1.0f;
    }
    if (w_acc == 0.f) w_acc = 1.0f;
    vocal_out[g] = v_acc / w_acc;
    inst_out[g] = i_acc / w_acc;
}

extern "C" int ac_mdx_assemble_ola(ac_ctx* ctx, const float* track, int64_t n, const float* wave, const int64_t* chunk_start,
                                   const int64_t* chunk_len, const int64_t* eff_start, const int64_t* eff_end,
                                   const int32_t* item_base, int n_chunks, float* vocal_out, float* inst_out, void* stream) {
    AC_REQUIRE(ctx && track && wave && chunk_start && chunk_len && eff_start && eff_end && item_base && vocal_out && inst_out,
               "null pointer");
    AC_REQUIRE(n > 0 && n_chunks > 0, "sizes must be positive");
    const int64_t blocks = (n + 255) / 256;
    AC_REQUIRE(blocks < (1LL << 31), "track too long");
    hipLaunchKernelGGL(k_mdx_assemble_ola, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, track, n, wave, chunk_start,
                       chunk_len, eff_start, eff_end, item_base, n_chunks, vocal_out, inst_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// ---------------------------------------------------------------------------------------------
// Per-chunk mono vocal (what `backend.infer_chunk(...).vocal` is in the reference, backends.py:389-406):
// the chunked VAD consumes it before the overlap-add (enhanced_vocal_separator.py:412-417).
// out is the concatenation of all chunks' vocals; out_offset[c] = first element of chunk c.
__global__ __launch_bounds__(256) void k_mdx_chunk_vocal(const float* __restrict__ wave, const int64_t* __restrict__ chunk_len,
                                                         const int64_t* __restrict__ out_offset,
                                                         const int32_t* __restrict__ item_base, float* __restrict__ out) {
    const int c = blockIdx.y;
    const int64_t cl = chunk_len[c];
    float* dst = out + out_offset[c];
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < cl; q += (int64_t)gridDim.x * 256) {
        const int item = item_base[c] + (int)(q / MDX_GEN);
        const int pos = MDX_TRIM + (int)(q % MDX_GEN);
        const float w0 = wave[((size_t)item * 2 + 0) * MDX_ITEM + pos];
        const float w1 = wave[((size_t)item * 2 + 1) * MDX_ITEM + pos];
        dst[q] = (w0 + w1) * 0.5f;
    }
}

extern "C" int ac_mdx_chunk_vocal(ac_ctx* ctx, const float* wave, const int64_t* chunk_len, const int64_t* out_offset,
                                  const int32_t* item_base, int n_chunks, float* out, void* stream) {
    AC_REQUIRE(ctx && wave && chunk_len && out_offset && item_base && out, "null pointer");
    AC_REQUIRE(n_chunks > 0 && n_chunks <= 65535, "n_chunks must be in [1, 65535]");
    hipLaunchKernelGGL(k_mdx_chunk_vocal, dim3(256, n_chunks), dim3(256), 0, (hipStream_t)stream, wave, chunk_len, out_offset,
                       item_base, out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// ---------------------------------------------------------------------------------------------
// per-block partial sums of squares in float64 (enhanced_vocal_separator.py:490-501 energy ratios);
// the host adds the (<= 4096) partials in index order, so the result is deterministic.
__global__ __launch_bounds__(256) void k_sum_squares(const float* __restrict__ x, int64_t n, double* __restrict__ out) {
    __shared__ double s_red[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const double v = x[i]; acc += v * v; }
    const double tot = block_sum_f64_256(acc, s_red);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// The same sum for MANY short windows of one wave in one launch (the VPBD beat candidates' vocal-risk windows, beat_candidates.py:97-109: one
// `mean_square(vocal[a:b])` - a launch and a download - per candidate until ABI 6).  One workgroup per window, summed EXACTLY as k_sum_squares sums a
// slice with one partial (thread t adds x[a + t], x[a + t + 256], ... in float64, then the same block reduction), so out[w] is bit-identical to
// ac_sum_squares(x + a, b - a, &partial, 1): windows must be shorter than 8192 samples, where the host wrapper asks for exactly one partial.
__global__ __launch_bounds__(256) void k_window_sum_squares(const float* __restrict__ x, const int64_t* __restrict__ w_start,
                                                            const int64_t* __restrict__ w_end, double* __restrict__ out) {
    __shared__ double s_red[4];
    const int64_t a = w_start[blockIdx.x], b = w_end[blockIdx.x];
    double acc = 0.0;
    for (int64_t i = a + threadIdx.x; i < b; i += 256) { const double v = x[i]; acc += v * v; }
    const double tot = block_sum_f64_256(acc, s_red);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

extern "C" int ac_window_sum_squares(ac_ctx* ctx, const float* x, int64_t n, const int64_t* w_start, const int64_t* w_end, int n_windows,
                                     double* out, void* stream) {
    AC_REQUIRE(ctx && x && w_start && w_end && out, "null pointer");
    AC_REQUIRE(n > 0 && n_windows > 0, "sizes must be positive");      // window bounds live on the device: the caller keeps 0 <= start <= end <= n, end - start < 8192
    hipLaunchKernelGGL(k_window_sum_squares, dim3((unsigned)n_windows), dim3(256), 0, (hipStream_t)stream, x, w_start, w_end, out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_sum_squares(ac_ctx* ctx, const float* x, int64_t n, double* partials, int n_partials, void* stream) {
    AC_REQUIRE(ctx && x && partials, "null pointer");
    AC_REQUIRE(n > 0 && n_partials > 0 && n_partials <= 4096, "n_partials must be in [1, 4096]");
    hipLaunchKernelGGL(k_sum_squares, dim3(n_partials), dim3(256), 0, (hipStream_t)stream, x, n, partials);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
