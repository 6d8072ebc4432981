// The TDF pair of a TFC-TDF block at the deep U-Net levels (frequency width F = 384 / 192 / 96, bottleneck F / 8 = 48 / 24 / 12),
// both layers and the residual in ONE kernel, in exact float32 on the f32 matrix instruction:
//   h[m][j] = relu(s1[c] * sum_f x[m][f] w1[j][f] + b1[c])                 j < Hd
//   y[m][n] = x[m][n] + relu(s2[c] * sum_j h[m][j] w2[n][j] + b2[c])       c = (m / T) % C
// (the MatMul / BatchNormalization / Relu / Add nodes of the graph run at separation/backends.py:358; oracle/separator.py:_tfc_tdf).
// These shapes are too narrow for ac_tdf_linear_f16x3's 96-column tiles and far too small to matter for the matrix-core budget
// (14.5 GFLOP per call at level 3, batch 32), so they use v_mfma_f32_16x16x4_f32 - float32 products, float32 accumulation, the
// same values as an fmaf chain - and stay HBM-bound: x is read twice (operand, residual: the second read hits L2) and y
// written once, both as 128-byte row segments; the 48-wide intermediate never leaves the CU.
//
// Workgroup = 4 waves, each wave owns 32 rows (two 16-row MFMA tiles) end to end, no workgroup barrier.
//   GEMM 1: A = x straight from global: lane (row r = lane & 15, group g = lane >> 4) loads the float4 x[r][16 kk + 4 g ..] and
//           feeds its four elements to four k-steps; the host packs w1 in the matching order (conv_pack.pack_tdf_small), so a
//           B fragment is one coalesced float4 per lane.  The k order inside a dot product is free.
//   h:      accumulator layout (row 4 g + i, column lane & 15) -> affine + ReLU -> wave-private LDS tile [32][stride] ->
//           read back in A layout (row lane & 15, k = 4 ks + g).
//   GEMM 2: B = w2 packed [n tile][k step][lane] (one dword per lane, zero padded in k), epilogue adds the residual.
#include "ac_common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;

#define TS_ROWS 32               // rows per wave
#define TS_HMAX 48               // widest bottleneck (level 3)
#define TS_HSTRIDE 50            // floats per row of the wave's h tile: even stride -> ds_read_b32 A fragments are conflict-free
#define TS_OSTRIDE 36            // floats per row of the wave's output tile (32 columns + 4): 16-byte aligned rows, conflict-free writes

template <int NT1>               // 16-column tiles of the bottleneck: 3 (Hd = 48), 2 (24, padded to 32), 1 (12, padded to 16)
__global__ __launch_bounds__(256) void k_tdf_small(const float* __restrict__ x, const float4* __restrict__ w1p,
                                                   const float* __restrict__ w2p, const float* __restrict__ s1,
                                                   const float* __restrict__ b1, const float* __restrict__ s2,
                                                   const float* __restrict__ b2, float* __restrict__ y, long long M, int F,
                                                   int Hd, int T, int C, float* __restrict__ out_amax) {
    __shared__ __attribute__((aligned(16))) float s_h[4][TS_ROWS * TS_HSTRIDE];   // h tile, then (h is in registers by then) the output tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const long long m0 = ((long long)blockIdx.x * 4 + wave) * TS_ROWS;
    if (m0 >= M) return;                                   // whole waves only: M % 32 == 0
    const int n_kk = F / 16;
    const int n_ks2 = (Hd + 3) / 4;                         // k-steps of GEMM 2
    float* sh = s_h[wave];

    // ---- GEMM 1: h = x w1^T --------------------------------------------------------------------------------------
    f32x4 acc1[2][NT1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) acc1[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* xa0 = x + (m0 + r) * (long long)F + 4 * g;
    const float* xa1 = xa0 + 16LL * F;
    for (int kk = 0; kk < n_kk; ++kk) {
        const float4 a0 = *reinterpret_cast<const float4*>(xa0 + 16 * kk);
        const float4 a1 = *reinterpret_cast<const float4*>(xa1 + 16 * kk);
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) {
            const float4 b = w1p[((size_t)nt * n_kk + kk) * 64 + lane];
            acc1[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc1[0][nt], 0, 0, 0);
            acc1[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc1[1][nt], 0, 0, 0);
            acc1[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc1[0][nt], 0, 0, 0);
            acc1[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc1[1][nt], 0, 0, 0);
            acc1[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc1[0][nt], 0, 0, 0);
            acc1[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc1[1][nt], 0, 0, 0);
            acc1[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc1[0][nt], 0, 0, 0);
            acc1[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc1[1][nt], 0, 0, 0);
        }
    }
    // ---- h -> LDS (affine + ReLU; padded columns j >= Hd hold relu(b1) of zero sums and meet zero weights in GEMM 2) ----
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = mt * 16 + 4 * g + i;
            const int c = (int)(((m0 + row) / T) % C);
            const float sc = s1[c], sf = b1[c];
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt) {
                const int j = nt * 16 + r;
                sh[row * TS_HSTRIDE + j] = (j < Hd) ? fmaxf(acc1[mt][nt][i] * sc + sf, 0.f) : 0.f;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();                       // LDS executes one wave's accesses in order; no other wave touches sh
    // A fragments of GEMM 2: h[row = mt * 16 + r][k = 4 ks + g]
    float ha[2][TS_HMAX / 4];
#pragma unroll
    for (int ks = 0; ks < TS_HMAX / 4; ++ks) {
        const int k = 4 * ks + g;
        ha[0][ks] = (ks < n_ks2) ? sh[r * TS_HSTRIDE + k] : 0.f;
        ha[1][ks] = (ks < n_ks2) ? sh[(16 + r) * TS_HSTRIDE + k] : 0.f;
    }
    // ---- GEMM 2 + epilogue, one 16-column tile at a time -------------------------------------------------------------
    int crow[2][4];
    float sc2[2][4], sf2[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = mt * 16 + 4 * g + i;
            crow[mt][i] = row;
            const int c = (int)(((m0 + row) / T) % C);
            sc2[mt][i] = s2[c]; sf2[mt][i] = b2[c];
        }
    // Two 16-column tiles at a time: relu(affine(acc)) goes through a wave-private LDS tile [32 rows][32 columns] and comes back as
    // one float4 per lane with 8 lanes per row, so the residual read and the store are full 128-byte row segments (the
    // accumulator layout itself would give 64-byte segments of single dwords: 1.4 TB/s).
    __builtin_amdgcn_wave_barrier();                       // the A fragments of GEMM 2 are in registers: the tile is free
    float* so = sh;
    float vmax[4] = {0.f, 0.f, 0.f, 0.f};                  // rows (lane >> 3) + 8 p of this lane
    const int orow = lane >> 3, oc4 = lane & 7;
    const int n_nt2 = F / 16;                               // even: F % 32 == 0
    for (int nt = 0; nt < n_nt2; nt += 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
            const float* wb = w2p + ((size_t)(nt + t) * n_ks2) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < TS_HMAX / 4; ++ks) {
                if (ks < n_ks2) {
                    const float b = wb[ks * 64];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[0][ks], b, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[1][ks], b, acc[1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    so[crow[mt][i] * TS_OSTRIDE + t * 16 + r] = fmaxf(acc[mt][i] * sc2[mt][i] + sf2[mt][i], 0.f);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = orow + 8 * p;
            const size_t o = (size_t)(m0 + row) * F + nt * 16 + 4 * oc4;
            const float4 t4 = *reinterpret_cast<const float4*>(&so[row * TS_OSTRIDE + 4 * oc4]);
            float4 v = *reinterpret_cast<const float4*>(x + o);
            v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
            vmax[p] = fmaxf(fmaxf(vmax[p], fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            *reinterpret_cast<float4*>(y + o) = v;
        }
        __builtin_amdgcn_wave_barrier();                   // the tile is rewritten by the next pair
    }
    if (out_amax) {          // max |y| per (item, time row): a row's columns sit in 8 consecutive lanes
        float* slots = out_amax + (m0 / ((long long)C * T)) * T;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float v = vmax[p];
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, AC_WAVE));
            if (oc4 == 0 && v > 0.f)
                atomicMax(reinterpret_cast<unsigned*>(slots + (int)((m0 + orow + 8 * p) % T)), __float_as_uint(v));
        }
    }
}

extern "C" int ac_tdf_small_fused(ac_ctx* ctx, const float* x, const void* w1_packed, const void* w2_packed, const float* scale1,
                                  const float* shift1, const float* scale2, const float* shift2, float* y, long long M, int F,
                                  int Hd, int T, int C, float* out_amax, void* stream) {
    AC_REQUIRE(ctx && x && w1_packed && w2_packed && scale1 && shift1 && scale2 && shift2 && y, "null pointer");
    AC_REQUIRE(M > 0 && M % TS_ROWS == 0, "M % 32 == 0");
    AC_REQUIRE(F > 0 && F % 32 == 0, "F % 32 == 0 (the epilogue walks pairs of 16-column tiles)");
    AC_REQUIRE(Hd > 0 && Hd <= TS_HMAX, "bottleneck width in [1, 48]");
    AC_REQUIRE(T > 0 && C > 0, "T, C > 0");
    AC_REQUIRE(!out_amax || ((long long)C * T) % TS_ROWS == 0, "amax needs (C * T) % 32 == 0 (a wave's 32 rows inside one item)");
    AC_REQUIRE(x != y, "in-place not supported (the residual is re-read)");
    const long long n_wave = M / TS_ROWS;
    const long long nblk = (n_wave + 3) / 4;
    AC_REQUIRE(nblk < (1LL << 31), "grid too large");
    dim3 grid((unsigned)nblk), block(256);
    hipStream_t st = (hipStream_t)stream;
    const float4* w1 = (const float4*)w1_packed;
    const float* w2 = (const float*)w2_packed;
    const int nt1 = (Hd + 15) / 16;
    if (nt1 == 3)      hipLaunchKernelGGL(k_tdf_small<3>, grid, block, 0, st, x, w1, w2, scale1, shift1, scale2, shift2, y, M, F, Hd, T, C, out_amax);
    else if (nt1 == 2) hipLaunchKernelGGL(k_tdf_small<2>, grid, block, 0, st, x, w1, w2, scale1, shift1, scale2, shift2, y, M, F, Hd, T, C, out_amax);
    else               hipLaunchKernelGGL(k_tdf_small<1>, grid, block, 0, st, x, w1, w2, scale1, shift1, scale2, shift2, y, M, F, Hd, T, C, out_amax);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
