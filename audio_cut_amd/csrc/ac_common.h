// Shared declarations for libaudiocut_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/audiocut_hip.h"

#define AC_WAVE 64

struct ac_ctx {
    int device;
    int n_cu;            // compute units of the device (persistent-grid sizing)
    // real-FFT 2048 (float64): tw2048[k] = exp(-2*pi*i*k/2048), k < 1024 ; hann2048 periodic (float64)
    double2* tw2048;
    double* hann2048;
    // mel filter bank (128 x 1025 float32, Slaney) as dense rows + [lo, hi) non-zero ranges
    float* mel_w;
    int* mel_lo;
    int* mel_hi;
    // real-FFT 6144 (float32): tw6144[k] = exp(-2*pi*i*k/6144), k < 3072 ; hann6144 periodic (float32)
    float2* tw6144;
    float* hann6144;
    float* ola_env6144;  // 1 / sum_t w^2 over the 256-frame lattice, length 267264 (padded coordinates)
};

void ac_set_error(const char* fmt, ...);

#define AC_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            ac_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return AC_E_HIP;                                                            \
        }                                                                               \
    } while (0)

#define AC_REQUIRE(cond, msg)                                   \
    do {                                                        \
        if (!(cond)) {                                          \
            ac_set_error("invalid argument: %s (%s)", msg, #cond); \
            return AC_E_INVALID;                                \
        }                                                       \
    } while (0)

#define AC_LAUNCH_CHECK()                                                       \
    do {                                                                        \
        hipError_t _e = hipGetLastError();                                      \
        if (_e != hipSuccess) {                                                 \
            ac_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return AC_E_HIP;                                                    \
        }                                                                       \
    } while (0)

// ---- wave / block reductions (64-wide) -------------------------------------------------------
__device__ inline double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, AC_WAVE);
    return v;
}
__device__ inline float wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, AC_WAVE));
    return v;
}

// ---- time-local activation scale of the split-f16 MFMA kernels ----------------------------------------------------
// x = xh + xl with xh = f16(x * s), xl = f16(x * s - xh): without s the low part is a float16 subnormal for |x| < 6e-2 and
// the pair carries an ABSOLUTE error floor of 2^-25 however small the tensor is (decays into silence lose their relative
// precision, which is where the quiet guard decides).  Every kernel that writes a tensor a split-f16 kernel reads therefore
// also reduces max|x| per (batch item, row of the time axis H) into `amax[item][H]` (ordered-bits atomicMax, one per wave
// and row); a reader takes the maximum over exactly the rows that enter one accumulation - the 10 patch rows of a 3x3 conv
// tile, the single time row of a TDF GEMM row, the two input rows of a 2x2 down-sampling pixel - scales by the power of two
// that puts it in [2^14, 2^15) and folds the inverse into its epilogue.  Power-of-two scaling is exact, so values whose
// low part is a normal float16 either way round identically; the floor becomes 2^-40 of the LOCAL maximum (the leakage that
// decays by orders of magnitude per frame next to a loud passage keeps its relative accuracy), nothing saturates below 2^127.
#define AC_AMAX_ROWS 1
// amax_item = amax + item * n_blocks (or NULL: no scaling); blocks b0..b1 inclusive, already clamped; all arguments wave-uniform
__device__ inline float ac_act_scale(const float* __restrict__ amax_item, int b0, int b1, float gain, float offs, float* inv) {
    float s = 1.f;
    *inv = 1.f;
    if (amax_item) {
        float a = 0.f;
        for (int i = b0; i <= b1; ++i) a = fmaxf(a, amax_item[i]);
        a = a * gain + offs;
        if (a > 0.f && a < 3.0e38f) {
            int e;
            (void)frexpf(a, &e);                    // a = m * 2^e, m in [0.5, 1)
            e = 15 - e;
            e = e < -120 ? -120 : (e > 120 ? 120 : e);
            s = ldexpf(1.f, e);
            *inv = ldexpf(1.f, -e);
        }
    }
    // wave-uniform by construction: keep the two factors in scalar registers
    *inv = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(*inv)));
    s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s)));
    return s;
}
// The 3x3 conv kernels stage ONE patch (10 rows) for the eight output rows of a tile.  Where the maxima of those rows lie within
// 2^AC_ROWX_SPREAD of each other (music: practically always) one common scale serves every output row with a floor of
// 2^-40 * 2^SPREAD = 2^-28 of the quietest row - below float32's own 2^-24.  Where they do not (the leakage of a burst decaying by
// decades per frame into digital silence) the tile takes the row-exact path: every patch row is staged with its OWN scale
// s_r = 2^ex_r, output row y accumulates at the scale of the loudest of ITS three input rows, S_y = 2^min(ex_y-1, ex_y, ex_y+1),
// and an activation fragment of row r is multiplied by the exact power of two S_y / s_r <= 1 (v_pk_mul_f16) on its way into the
// MFMA (factors below 2^-24, the smallest float16, are contributions below float32's resolution and become 0).
#define AC_EX_NONE 0x7fffffff
#define AC_ROWX_SPREAD 12
// log2 of the power-of-two scale of a row whose maximum is a (the scale puts a in [2^14, 2^15)); AC_EX_NONE: no usable maximum
__device__ inline int ac_row_ex(float a) {
    if (!(a > 0.f && a < 3.0e38f)) return AC_EX_NONE;
    int e;
    (void)frexpf(a, &e);
    e = 15 - e;
    return e < -120 ? -120 : (e > 120 ? 120 : e);
}
// factor that takes a fragment staged at 2^ex_row to the output row's scale 2^ex_out (ex_out <= ex_row)
__device__ inline float ac_rowx_factor(int ex_row, int ex_out) {
    if (ex_row == AC_EX_NONE || ex_out == AC_EX_NONE) return 1.f;      // an all-zero row: its values are zeros at any scale
    const int d = ex_row - ex_out;
    return d <= 24 ? ldexpf(1.f, -d) : 0.f;
}
// s_ex[r] = log2 scale of patch row r (LDS); output row ty reads patch rows ty, ty + 1, ty + 2.  AC_EX_NONE is the largest int,
// so the minimum skips empty rows.  Computed where they are used: the common path must not carry registers for this one.
__device__ inline int ac_rowx_out_ex(const int* s_ex, int ty) {
    int e = s_ex[ty] < s_ex[ty + 1] ? s_ex[ty] : s_ex[ty + 1];
    return e < s_ex[ty + 2] ? e : s_ex[ty + 2];
}
__device__ inline _Float16 ac_rowx_frag_factor(const int* s_ex, int ty, int dy) {
    return (_Float16)ac_rowx_factor(s_ex[ty + dy], ac_rowx_out_ex(s_ex, ty));
}
__device__ inline float ac_rowx_unscale(const int* s_ex, int ty) {
    const int e = ac_rowx_out_ex(s_ex, ty);
    return e == AC_EX_NONE ? 1.f : ldexpf(1.f, -e);
}
// the same for arguments that differ from lane to lane (a GEMM row's own time row)
__device__ inline float ac_act_scale_lane(const float* __restrict__ amax_item, int b0, int b1, float* inv) {
    float s = 1.f;
    *inv = 1.f;
    if (amax_item) {
        float a = 0.f;
        for (int i = b0; i <= b1; ++i) a = fmaxf(a, amax_item[i]);
        if (a > 0.f && a < 3.0e38f) {
            int e;
            (void)frexpf(a, &e);
            e = 15 - e;
            e = e < -120 ? -120 : (e > 120 ? 120 : e);
            s = ldexpf(1.f, e);
            *inv = ldexpf(1.f, -e);
        }
    }
    return s;
}
// m = max |v| over this thread's outputs (>= 0; NaNs never enter through fmaxf): wave reduce, one atomic per wave.
__device__ inline void ac_amax_commit(float m, float* __restrict__ slot) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, AC_WAVE));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(m));
}
// the same per 32-lane half (lanes 0-31 -> slot_of_this_half as seen by lane 0, lanes 32-63 as seen by lane 32)
__device__ inline void ac_amax_commit_halves(float m, float* __restrict__ slot_of_my_half) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, AC_WAVE));
    if ((threadIdx.x & 31) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(slot_of_my_half), __float_as_uint(m));
}
// lanes may belong to different blocks: one reduction + atomic per distinct block id present in the wave (1-3 in practice)
__device__ inline void ac_amax_commit_blocks(float m, int blk, float* __restrict__ slots) {
    unsigned long long todo = __ballot(1);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int b = __builtin_amdgcn_readlane(blk, leader);
        const bool mine = blk == b;
        float v = mine ? m : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, AC_WAVE));
        if ((int)(threadIdx.x & 63) == leader && v > 0.f) atomicMax(reinterpret_cast<unsigned*>(slots + b), __float_as_uint(v));
        todo &= ~__ballot(mine);
    }
}

// One 1 KB piece global -> LDS (16 B per lane, lane i lands at lds + 16 i), issued from inline assembly ON PURPOSE: with the
// builtin the compiler knows an LDS-DMA is in flight and then drains lgkmcnt to 0 in front of every use of a ds_read result for as
// long as it is pending (measured on the ISA: every fragment wait in the K loop was lgkmcnt(0)); hidden from it, its counted
// lgkmcnt(N) waits are exact again and fragment reads can stay in flight behind the MFMAs.  The wave's own s_waitcnt vmcnt(0)
// in front of the stage barrier is what orders the data (as before); M0 is not used by anything else in these kernels.
// (Shared by the conv kernels' weight stream and the TDF epilogue's residual prefetch.)
__device__ __forceinline__ void ac_lds_dma16(const void* g, void* lds_wave_base) {
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(l) : "memory", "m0");
}


// ---- rational-rate polyphase FIR (ac_resample_poly, ac_resample_poly_segments) ----------------------------------------------------
// y[m] = sum_q hfull[i - q * up] x[q], i = (m + n_pre_remove) * down, over the n samples of x only (zero extension).  The taps
// arrive as polyphase ROWS hp[p][t] = hfull[p + t * up] (p < up, t < tpp; rows zero-padded): output m uses row p = i % up against
// x[i / up - t].  ONE WAVE PER OUTPUT: lane l takes taps t = l, l + 64, ... (a coalesced sweep of the row and of the input run),
// float64 partial sums, butterfly reduction - every lane returns the sum.  (A thread per output made every tap load of a wave touch
// 64 different rows = 64 cache lines: 8.9 ms for the 32 chunks of a 4-min track at 527 taps per output.)
#define AC_RS_PER_WAVE 8
__device__ inline float ac_polyphase_dot_wave(const float* __restrict__ x, int64_t n, const float* __restrict__ hp, int up, int tpp, int64_t i) {
    const int lane = threadIdx.x & (AC_WAVE - 1);
    const int64_t j0 = i / up;
    const float* __restrict__ row = hp + (int64_t)(i - j0 * up) * tpp;
    int64_t t_lo = j0 - (n - 1);                       // q = j0 - t <= n - 1
    if (t_lo < 0) t_lo = 0;
    const int64_t t_hi = j0 < tpp - 1 ? j0 : tpp - 1;  // q >= 0
    double acc = 0.0;
    for (int64_t t = t_lo + lane; t <= t_hi; t += AC_WAVE) acc += (double)row[t] * (double)x[j0 - t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, AC_WAVE);
    return (float)acc;
}

// Block-wide sum of doubles for blockDim.x == 256 (4 waves); result valid in every thread.
__device__ inline double block_sum_f64_256(double v, double* smem4) {
    v = wave_sum_f64(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem4[w] = v;
    __syncthreads();
    return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}
