// Silero VAD (v5, 16 kHz) on the GPU: the network behind `VocalPauseDetectorV2._detect_speech_timestamps`
// (`src/vocal_smart_splitter/core/vocal_pause_detector.py:175-296`, reached per chunk from `SileroChunkVAD.process_chunk`,
// `src/audio_cut/detectors/silero_chunk_vad.py:56-117`).  The reference runs it window by window on the CPU through ONNX
// Runtime; here all chunks of a track go through four launches:
//   ac_resample_poly_segments  every chunk's 44.1 kHz -> 16 kHz resampling (`:189`) in one launch, written into a layout
//                              zero-padded per chunk to the 4096-sample bucket (`:192-196`)
//   ac_silero_frontend         per window: 64-sample context + 512 samples, reflect pad, STFT-as-convolution (258 x 256 basis,
//                              stride 128 -> 4 frames), magnitude, four Conv1d + ReLU, and the input half of the LSTM gates
//                              (W_ih feat + b_ih + b_hh).  Windows are independent here: 8 per workgroup share every weight read.
//   ac_silero_lstm             the only sequential part: one workgroup per chunk walks its windows; thread g keeps row g of
//                              W_hh in registers, h lives in LDS (two barriers per step)
//   ac_silero_out              probability = sigmoid(w . relu(h) + b), one wave per window
// float32 FMAs throughout, fixed summation order (deterministic).  Weights arrive transposed by the host
// (audio_cut_amd/detectors/silero_vad.py) so that consecutive threads read consecutive floats.
#include <math.h>

#include "ac_common.h"

#define SV_WIN 512
#define SV_CTX 64
#define SV_IN (SV_CTX + SV_WIN)          // 576 samples into the network per window
#define SV_PAD 640                       // + 64 reflected
#define SV_NW 8                          // windows per front-end workgroup
#define SV_H 128

// ---------------------------------------------------------------------------------------------------------------------
// Segmented polyphase resampling: ac_resample_poly for S independent segments in one launch.  Output m of segment s is
// sum_q hfull[(m + n_pre_remove) * down - q * up] x[in_off[s] + q] over the segment's own samples only (zero extension at its
// edges, as scipy.signal.resample_poly on the segment alone; taps as polyphase rows, ac_common.h); written at out[out_off[s] + m], m < out_len[s].
__global__ __launch_bounds__(256) void k_resample_poly_seg(const float* __restrict__ x, const int64_t* __restrict__ in_off,
                                                           const int64_t* __restrict__ in_len, const int64_t* __restrict__ out_off,
                                                           const int64_t* __restrict__ out_len, int n_seg, int up, int down,
                                                           const float* __restrict__ hp, int tpp, int64_t n_pre_remove,
                                                           float* __restrict__ out, int64_t n_work) {
    // a wave walks AC_RS_PER_WAVE consecutive indices of the concatenation of the segments' (bucket-padded) outputs
    const int64_t g0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * AC_RS_PER_WAVE;
    for (int64_t g = g0; g < g0 + AC_RS_PER_WAVE && g < n_work; ++g) {
        // out_off is increasing and segment s owns [out_off[s], out_off[s] + out_len[s]): binary search over the starts
        int lo = 0, hi = n_seg - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (out_off[mid] <= g) lo = mid; else hi = mid - 1;
        }
        const int64_t m = g - out_off[lo];
        if (m >= out_len[lo]) continue;                                   // bucket padding: stays zero
        const float v = ac_polyphase_dot_wave(x + in_off[lo], in_len[lo], hp, up, tpp, (m + n_pre_remove) * (int64_t)down);
        if ((threadIdx.x & 63) == 0) out[g] = v;
    }
}

extern "C" int ac_resample_poly_segments(ac_ctx* ctx, const float* x, const int64_t* in_off, const int64_t* in_len,
                                         const int64_t* out_off, const int64_t* out_len, int n_seg, int up, int down,
                                         const float* h, int64_t hlen, int64_t n_pre_remove, float* out, int64_t n_out_total,
                                         void* stream) {
    AC_REQUIRE(ctx && x && in_off && in_len && out_off && out_len && h && out, "null pointer");
    AC_REQUIRE(n_seg > 0 && up > 0 && down > 0 && hlen > 0 && n_pre_remove >= 0 && n_out_total > 0, "sizes must be positive");
    AC_REQUIRE(hlen % up == 0 && hlen / up < (1LL << 31), "h is [up][hlen / up] polyphase rows");
    const int64_t blocks = (n_out_total + 4 * AC_RS_PER_WAVE - 1) / (4 * AC_RS_PER_WAVE);
    AC_REQUIRE(blocks < (1LL << 31), "output too long");
    hipLaunchKernelGGL(k_resample_poly_seg, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, in_off, in_len,
                       out_off, out_len, n_seg, up, down, h, (int)(hlen / up), n_pre_remove, out, n_out_total);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Front end.  LDS images keep the 8 windows of the workgroup innermost ([...][8]) so that one float4 pair feeds the 8 FMAs a
// thread issues per weight; every thread of a layer reads the same activation address (LDS broadcast) and its own weight
// column (coalesced: weights are stored [k][c_out]).
struct sv_weights {
    const float* basis_t;     // [256][258]   forward_basis_buffer transposed
    const float* c1; const float* b1;     // [129 * 3][128], [128]
    const float* c2; const float* b2;     // [128 * 3][64],  [64]
    const float* c3; const float* b3;     // [64 * 3][64],   [64]
    const float* c4; const float* b4;     // [64 * 3][128],  [128]
    const float* wih_t; const float* bsum;   // [128][512], [512] = bias_ih + bias_hh
};

__device__ inline void sv_fma8(float (&acc)[SV_NW], float w, const float* __restrict__ s) {
    const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
    acc[0] = fmaf(w, a.x, acc[0]); acc[1] = fmaf(w, a.y, acc[1]); acc[2] = fmaf(w, a.z, acc[2]); acc[3] = fmaf(w, a.w, acc[3]);
    acc[4] = fmaf(w, b.x, acc[4]); acc[5] = fmaf(w, b.y, acc[5]); acc[6] = fmaf(w, b.z, acc[6]); acc[7] = fmaf(w, b.w, acc[7]);
}

__global__ __launch_bounds__(256) void k_silero_frontend(const float* __restrict__ x16, const int64_t* __restrict__ win_start,
                                                         int n_win, sv_weights W, float* __restrict__ gates_x) {
    // arena (floats): X [640][8] = 5120 | RI [258][4][8] = 8256 | MAG [129][4][8] = 4128 | A1 [128][4][8] = 4096
    //                 A2 [64][2][8], A3 [64][8], A4 [128][8] reuse X
    __shared__ __attribute__((aligned(16))) float s_x[SV_PAD * SV_NW];
    __shared__ __attribute__((aligned(16))) float s_ri[258 * 4 * SV_NW];
    __shared__ __attribute__((aligned(16))) float s_mag[129 * 4 * SV_NW];
    __shared__ __attribute__((aligned(16))) float s_a1[128 * 4 * SV_NW];
    float* s_a2 = s_x;                       // [64][2][8] = 1024
    float* s_a3 = s_x + 1024;                // [64][8]    = 512
    float* s_a4 = s_x + 2048;                // [128][8]   = 1024
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * SV_NW;
    // ---- input: context | window | reflected tail; x16 is zero-padded per chunk, so a window never reads past its chunk
    for (int e = tid; e < SV_PAD * SV_NW; e += 256) {
        const int pos = e >> 3, wi = e & 7;
        const int w = w0 + wi;
        float v = 0.f;
        if (w < n_win) {
            const int64_t st = win_start[w];                       // index of the window's first new sample; < 0: -(index) - 1 of a chunk's first window
            const bool first = st < 0;
            const int64_t base = first ? -(st + 1) : st;
            const int src = pos < SV_IN ? pos : (2 * SV_IN - 2 - pos);          // F.pad(mode="reflect"): x[575 - 1 - j] for pos = 576 + j
            const int rel = src - SV_CTX;                                        // relative to the window's first new sample
            if (rel >= 0 || !first) v = x16[base + rel];                          // context of a chunk's first window is zeros
        }
        s_x[e] = v;
    }
    __syncthreads();
    // ---- STFT as a convolution: out[r][f] = sum_j basis[r][j] x[128 f + j]
    for (int r = tid; r < 258; r += 256) {
        float acc[4][SV_NW];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int i = 0; i < SV_NW; ++i) acc[f][i] = 0.f;
        for (int j = 0; j < 256; ++j) {
            const float w = W.basis_t[j * 258 + r];
#pragma unroll
            for (int f = 0; f < 4; ++f) sv_fma8(acc[f], w, &s_x[(128 * f + j) * SV_NW]);
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int i = 0; i < SV_NW; ++i) s_ri[(r * 4 + f) * SV_NW + i] = acc[f][i];
    }
    __syncthreads();
    for (int e = tid; e < 129 * 4 * SV_NW; e += 256) {
        const float re = s_ri[e], im = s_ri[129 * 4 * SV_NW + e];
        s_mag[e] = sqrtf(re * re + im * im);
    }
    __syncthreads();
    // ---- conv1: 129 -> 128, k 3, pad 1, stride 1, T 4 -> 4; thread (co = tid & 127, t pair = tid >> 7)
    {
        const int co = tid & 127, tp = tid >> 7;
        float acc[2][SV_NW];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < SV_NW; ++i) acc[t][i] = W.b1[co];
        for (int ci = 0; ci < 129; ++ci) {
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const float w = W.c1[(ci * 3 + tap) * 128 + co];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int tin = 2 * tp + t + tap - 1;
                    if (tin >= 0 && tin < 4) sv_fma8(acc[t], w, &s_mag[(ci * 4 + tin) * SV_NW]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < SV_NW; ++i) s_a1[(co * 4 + 2 * tp + t) * SV_NW + i] = fmaxf(acc[t][i], 0.f);
    }
    __syncthreads();
    // ---- conv2: 128 -> 64, stride 2, T 4 -> 2; thread (co = tid & 63, t = (tid >> 6) & 1, window half = tid >> 7)
    {
        const int co = tid & 63, t = (tid >> 6) & 1, wh = tid >> 7;
        float acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = W.b2[co];
        for (int ci = 0; ci < 128; ++ci) {
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const int tin = 2 * t + tap - 1;
                if (tin >= 0 && tin < 4) {
                    const float w = W.c2[(ci * 3 + tap) * 64 + co];
                    const float4 a = *reinterpret_cast<const float4*>(&s_a1[(ci * 4 + tin) * SV_NW + 4 * wh]);
                    acc[0] = fmaf(w, a.x, acc[0]); acc[1] = fmaf(w, a.y, acc[1]); acc[2] = fmaf(w, a.z, acc[2]); acc[3] = fmaf(w, a.w, acc[3]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s_a2[(co * 2 + t) * SV_NW + 4 * wh + i] = fmaxf(acc[i], 0.f);
    }
    __syncthreads();
    // ---- conv3: 64 -> 64, stride 2, T 2 -> 1 (input times tap - 1: taps 1, 2 hit t = 0, 1); thread (co, window pair)
    {
        const int co = tid & 63, wq = tid >> 6;
        float acc[2] = {W.b3[co], W.b3[co]};
        for (int ci = 0; ci < 64; ++ci) {
#pragma unroll
            for (int tap = 1; tap < 3; ++tap) {
                const float w = W.c3[(ci * 3 + tap) * 64 + co];
                const float2 a = *reinterpret_cast<const float2*>(&s_a2[(ci * 2 + tap - 1) * SV_NW + 2 * wq]);
                acc[0] = fmaf(w, a.x, acc[0]); acc[1] = fmaf(w, a.y, acc[1]);
            }
        }
        s_a3[co * SV_NW + 2 * wq] = fmaxf(acc[0], 0.f);
        s_a3[co * SV_NW + 2 * wq + 1] = fmaxf(acc[1], 0.f);
    }
    __syncthreads();
    // ---- conv4: 64 -> 128, stride 1, T 1 -> 1 (only the centre tap meets data); thread (co = tid & 127, window half)
    {
        const int co = tid & 127, wh = tid >> 7;
        float acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = W.b4[co];
        for (int ci = 0; ci < 64; ++ci) {
            const float w = W.c4[(ci * 3 + 1) * 128 + co];
            const float4 a = *reinterpret_cast<const float4*>(&s_a3[ci * SV_NW + 4 * wh]);
            acc[0] = fmaf(w, a.x, acc[0]); acc[1] = fmaf(w, a.y, acc[1]); acc[2] = fmaf(w, a.z, acc[2]); acc[3] = fmaf(w, a.w, acc[3]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s_a4[co * SV_NW + 4 * wh + i] = fmaxf(acc[i], 0.f);
    }
    __syncthreads();
    // ---- input half of the LSTM gates: gates_x[w][g] = bias_ih[g] + bias_hh[g] + sum_j weight_ih[g][j] feat[j]
    for (int g = tid; g < 4 * SV_H; g += 256) {
        float acc[SV_NW];
#pragma unroll
        for (int i = 0; i < SV_NW; ++i) acc[i] = W.bsum[g];
        for (int j = 0; j < SV_H; ++j) sv_fma8(acc, W.wih_t[j * 4 * SV_H + g], &s_a4[j * SV_NW]);
#pragma unroll
        for (int i = 0; i < SV_NW; ++i)
            if (w0 + i < n_win) gates_x[(size_t)(w0 + i) * 4 * SV_H + g] = acc[i];
    }
}

extern "C" int ac_silero_frontend(ac_ctx* ctx, const float* x16, const int64_t* win_start, int n_windows, const float* basis_t,
                                  const float* c1, const float* b1, const float* c2, const float* b2, const float* c3, const float* b3,
                                  const float* c4, const float* b4, const float* wih_t, const float* bias_sum, float* gates_x,
                                  void* stream) {
    AC_REQUIRE(ctx && x16 && win_start && basis_t && c1 && b1 && c2 && b2 && c3 && b3 && c4 && b4 && wih_t && bias_sum && gates_x, "null pointer");
    AC_REQUIRE(n_windows > 0, "n_windows must be positive");
    sv_weights W{basis_t, c1, b1, c2, b2, c3, b3, c4, b4, wih_t, bias_sum};
    hipLaunchKernelGGL(k_silero_frontend, dim3((unsigned)((n_windows + SV_NW - 1) / SV_NW)), dim3(256), 0, (hipStream_t)stream, x16,
                       win_start, n_windows, W, gates_x);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// LSTMCell over the windows of one chunk (PyTorch gate order i, f, g, o).  512 threads: thread g owns gate row g.
__global__ __launch_bounds__(512) void k_silero_lstm(const float* __restrict__ gates_x, const int* __restrict__ seg_first,
                                                     const int* __restrict__ seg_count, const float* __restrict__ whh_t,
                                                     float* __restrict__ h_out) {
    __shared__ __attribute__((aligned(16))) float s_h[SV_H];
    __shared__ float s_g[4 * SV_H];
    const int g = threadIdx.x;
    const int first = seg_first[blockIdx.x], count = seg_count[blockIdx.x];
    float wrow[SV_H];                                  // weight_hh[g][:]: 128 registers, loaded once (coalesced over g)
#pragma unroll
    for (int j = 0; j < SV_H; ++j) wrow[j] = whh_t[j * 4 * SV_H + g];
    float c = 0.f;
    if (g < SV_H) s_h[g] = 0.f;
    float gx = count > 0 ? gates_x[(size_t)first * 4 * SV_H + g] : 0.f;
    __syncthreads();
    for (int w = 0; w < count; ++w) {
        const float gx_next = (w + 1 < count) ? gates_x[(size_t)(first + w + 1) * 4 * SV_H + g] : 0.f;     // one step ahead
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < SV_H; j += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(&s_h[j]);
            acc = fmaf(wrow[j], hv.x, acc); acc = fmaf(wrow[j + 1], hv.y, acc);
            acc = fmaf(wrow[j + 2], hv.z, acc); acc = fmaf(wrow[j + 3], hv.w, acc);
        }
        s_g[g] = gx + acc;
        __syncthreads();
        if (g < SV_H) {
            const float ig = 1.f / (1.f + expf(-s_g[g])), fg = 1.f / (1.f + expf(-s_g[SV_H + g]));
            const float gg = tanhf(s_g[2 * SV_H + g]), og = 1.f / (1.f + expf(-s_g[3 * SV_H + g]));
            c = fg * c + ig * gg;
            const float hv = og * tanhf(c);
            s_h[g] = hv;
            h_out[(size_t)(first + w) * SV_H + g] = hv;
        }
        gx = gx_next;
        __syncthreads();
    }
}

extern "C" int ac_silero_lstm(ac_ctx* ctx, const float* gates_x, const int* seg_first_window, const int* seg_window_count, int n_seg,
                              const float* whh_t, float* h_out, void* stream) {
    AC_REQUIRE(ctx && gates_x && seg_first_window && seg_window_count && whh_t && h_out, "null pointer");
    AC_REQUIRE(n_seg > 0, "n_seg must be positive");
    hipLaunchKernelGGL(k_silero_lstm, dim3((unsigned)n_seg), dim3(512), 0, (hipStream_t)stream, gates_x, seg_first_window,
                       seg_window_count, whh_t, h_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// probability[w] = sigmoid(b + sum_j w_out[j] relu(h[w][j])): decoder = Dropout (identity), ReLU, Conv1d(128, 1, 1), Sigmoid
__global__ __launch_bounds__(256) void k_silero_out(const float* __restrict__ h, const float* __restrict__ w_out, float b_out,
                                                    int n_win, float* __restrict__ probs) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_win) return;
    const float* hw = h + (size_t)w * SV_H;
    // fixed order: lane j adds element j then j + 64, then a butterfly over the wave
    float acc = fmaf(w_out[lane], fmaxf(hw[lane], 0.f), 0.f);
    acc = fmaf(w_out[lane + 64], fmaxf(hw[lane + 64], 0.f), acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, AC_WAVE);
    if (lane == 0) probs[w] = 1.f / (1.f + expf(-(acc + b_out)));
}

extern "C" int ac_silero_out(ac_ctx* ctx, const float* h, const float* w_out, float b_out, int n_windows, float* probs, void* stream) {
    AC_REQUIRE(ctx && h && w_out && probs, "null pointer");
    AC_REQUIRE(n_windows > 0, "n_windows must be positive");
    hipLaunchKernelGGL(k_silero_out, dim3((unsigned)((n_windows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h, w_out, b_out, n_windows, probs);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
