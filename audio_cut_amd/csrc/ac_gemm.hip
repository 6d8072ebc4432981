// TDF (time-distributed fully connected) layers of the TFC-TDF U-Net on the 16-bit matrix cores:
//   y[m][n] = (resid ? resid[m][n] : 0) + relu(scale[c(m)] * sum_k x[m][k] * w[n][k] + shift[c(m)]),   c(m) = (m / T) % C
// x is the NCHW activation seen as rows (b, c, t) of F floats, w a bias-free Linear weight [N][K], scale/shift the
// eval-mode BatchNorm2d over channels that follows it, resid the block's residual input (second TDF layer only).
// Arithmetic: the same 3-term float16 split as ac_conv.hip (x = xh + xl, w = wh + wl, xh*wh + xh*wl + xl*wh with
// float32 accumulation in v_mfma_f32_16x16x32_f16) - float32-class products at the 16-bit MFMA rate.
//
// Workgroup: 256 threads = 2 x 2 waves, tile 128 rows x (32 * NT) columns, wave tile 64 x (16 * NT) (NT = 6 or 3).
// The 128 rows of a tile are 16 channels x 8 consecutive time rows of one item (tile row r = channel r >> 3, time r & 7):
// every row is its own contiguous F-vector, so the loads do not care, and a tile then touches exactly 8 entries of the
// per-time-row activation maxima (ac_common.h): GEMM row r is scaled by the maximum of ITS time row (rows of a GEMM are
// independent, so a per-row power of two folds exactly into the epilogue) and the tile commits 8 output maxima.
// K is walked in stages of 32: the x tile is loaded as full 128-byte row segments (float4 per lane), split to f16
// hi/lo once and staged in LDS as [row][k] with an 96-byte row stride (conflict-free ds_read_b128 A fragments); the
// weights arrive pre-split and pre-packed in B-fragment order (conv_pack.pack_linear) and go global -> LDS by DMA
// (global_load_lds_dwordx4, double-buffered, no registers).  The
// next stage's global loads are issued before the current stage's MFMAs.  Epilogue: accumulators -> per-wave LDS
// strip -> affine + ReLU (+ residual) -> 384-byte (192-byte for NT = 3) contiguous row stores.
#include "ac_common.h"
#include <stdlib.h>
#ifndef GM_RESID_AHEAD
#define GM_RESID_AHEAD 0            // 1: all four residual strips of the epilogue requested at once (registers + hidden LDS-DMA): built and measured in round 4, bit-identical and 2 % SLOWER on the same box (profiles/r04j: 3.68 -> 3.75 ms at level 0) - the epilogue is not load-latency-bound; kept switchable
#endif
#ifndef AC_PROBES
#define AC_PROBES 0                   // 1: work-order overrides from the environment (tools/tdf_order_probe.py); never in the product build
#endif

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define GM_BM 128
#define GM_BK 32
#define GM_ASTRIDE 48            // f16 per staged x row: 32 used + 16 pad -> 96 bytes (conflict-free for ds_read_b128's four 16-lane groups; 80 was 2-way)
#define GM_MT 4                  // 16-row tiles per wave
#define GM_A_ITERS ((GM_BM * (GM_BK / 4)) / 256)     // float4 loads per thread per stage (4)

__device__ inline unsigned short gm_f16_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }

template <int NT, bool RESID>
__global__ __launch_bounds__(256, 2) void k_tdf_linear_f16x3(const float* __restrict__ x, const f16x8* __restrict__ wpk,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ resid, float* __restrict__ y,
                                                             int n_mblk, int N, int K, int T, int C, float w_unscale,
                                                             const float* __restrict__ in_amax, float* __restrict__ out_amax,
                                                             int ord_g, int ord_r) {
    constexpr int BN = 32 * NT;                       // columns per workgroup
    constexpr int BFRAGS = 2 * (BN / 16) * 64;        // 16-byte weight fragments per stage (hi, lo)
    constexpr int B_ITERS = BFRAGS / 256;             // 6 (NT = 6) or 3
    constexpr int OSTRIDE = 16 * NT + 4;              // floats per row of a wave's output strip
    constexpr int A_BYTES = 2 * GM_BM * GM_ASTRIDE * 2;
    constexpr int O_BYTES = 4 * 16 * OSTRIDE * 4;
    // the weight fragments are double-buffered and filled by LDS-DMA one stage ahead (they are stored in LDS order already)
    // epilogue: the residual rows of strips 2 and 3 are brought in by LDS-DMA behind the four output strips (R_BYTES per wave)
    constexpr int R_BYTES = (RESID && GM_RESID_AHEAD) ? 2 * (16 * (16 * NT) * 4) : 0;
    constexpr int K_ARENA = A_BYTES + 2 * BFRAGS * 16, E_ARENA = O_BYTES + 4 * R_BYTES;
    constexpr int ARENA = K_ARENA > E_ARENA ? K_ARENA : E_ARENA;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[ARENA];
    unsigned short* s_hi = reinterpret_cast<unsigned short*>(s_raw);
    unsigned short* s_lo = s_hi + GM_BM * GM_ASTRIDE;
    f16x8* s_b0 = reinterpret_cast<f16x8*>(s_raw + A_BYTES);
    f16x8* s_b1 = s_b0 + BFRAGS;
    float* s_out = reinterpret_cast<float*>(s_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_nblk = N / BN;
    // XCD-aware work order (see ac_conv.hip): XCD `L & 7` walks a contiguous strip of the order below, its ~64 resident workgroups
    // are consecutive in it.  The walk goes over super-groups of ord_r row tiles: inside one, ord_g column blocks of a row tile are
    // neighbours (they share the x tile in L2), then the next row tile, and only then the next ord_g column blocks - so the
    // resident workgroups hold ord_g column blocks' weights between them, not all of them (w9_tdf_order picks the two).
    int wi = blockIdx.x;
    if ((gridDim.x & 7) == 0) wi = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int per = ord_r * n_nblk;
    const int chunk = wi / per, i_in = wi - chunk * per;
    const int nb_hi = i_in / (ord_g * ord_r), j_in = i_in - nb_hi * (ord_g * ord_r);
    const int nb = nb_hi * ord_g + j_in % ord_g, mb = chunk * ord_r + j_in / ord_g;
    if (mb >= n_mblk) return;
    const int n0 = nb * BN;
    const int n_stage = K / GM_BK;
    // tile -> (item, group of 16 channels, block of 8 time rows); time blocks of one channel group are neighbours in the walk
    const int n_blk = T / 8, n_cg = C / 16;
    const int tb = mb % n_blk, cg = (mb / n_blk) % n_cg, item = mb / (n_blk * n_cg);
    const size_t m_base = ((size_t)item * C + (size_t)cg * 16) * T + (size_t)tb * 8;                 // global row of tile row 0
    auto grow = [&](int r) -> size_t { return m_base + (size_t)(r >> 3) * T + (r & 7); };            // global row of tile row r
    // time-local power-of-two activation scale (ac_common.h): GEMM rows are independent, so every tile row is scaled by the
    // maximum of its OWN time row (tile row r is time row tb * 8 + (r & 7)); the eight scale / inverse pairs and the eight
    // output maxima of the tile live in LDS
    __shared__ float s_scale[8], s_inv[8];
    __shared__ unsigned s_tmax[8];
    if (tid < 8) {
        float inv;
        s_scale[tid] = ac_act_scale_lane(in_amax ? in_amax + (size_t)item * T : nullptr, tb * 8 + tid, tb * 8 + tid, &inv);
        s_inv[tid] = inv;
        s_tmax[tid] = 0u;
    }

    f32x4 acc[GM_MT][NT];
#pragma unroll
    for (int m = 0; m < GM_MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging coordinates: float4 e covers row e >> 3, k-quad e & 7 (8 lanes = one 128-byte row segment)
    const float* a_ptr[GM_A_ITERS];
    int a_off[GM_A_ITERS];
#pragma unroll
    for (int i = 0; i < GM_A_ITERS; ++i) {
        const int e = tid + 256 * i;
        const int row = e >> 3, kq = e & 7;
        a_ptr[i] = x + grow(row) * (size_t)K + 4 * kq;
        a_off[i] = row * GM_ASTRIDE + 4 * kq;
    }
    const f16x8* wbase = wpk + (size_t)nb * n_stage * BFRAGS;

    float4 pre_a[GM_A_ITERS];
    auto prefetch = [&](int s) {
        f16x8* dst = (s & 1) ? s_b1 : s_b0;
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {            // BFRAGS / 64 wave-instructions of 1 KB, B_ITERS per wave
            const int inst = wave + 4 * i;
            __builtin_amdgcn_global_load_lds(wbase + (size_t)s * BFRAGS + inst * 64 + lane, dst + inst * 64, 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < GM_A_ITERS; ++i) pre_a[i] = *reinterpret_cast<const float4*>(a_ptr[i] + (size_t)s * GM_BK);
    };
    prefetch(0);                         // stage 0 is on its way while the eight scales (a dependent global load) arrive
    __syncthreads();
    float a_scale[GM_A_ITERS];
#pragma unroll
    for (int i = 0; i < GM_A_ITERS; ++i) a_scale[i] = s_scale[((tid + 256 * i) >> 3) & 7];

    const int frag_row = lane & 15, frag_k = 8 * (lane >> 4);
    for (int s = 0; s < n_stage; ++s) {
        __syncthreads();                 // previous stage fully consumed
        const f16x8* s_b = (s & 1) ? s_b1 : s_b0;
#pragma unroll
        for (int i = 0; i < GM_A_ITERS; ++i) {
            const float v[4] = {pre_a[i].x, pre_a[i].y, pre_a[i].z, pre_a[i].w};
            unsigned short h[4], l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float c = fminf(fmaxf(v[q] * a_scale[i], -65504.f), 65504.f);
                const _Float16 hv = (_Float16)c;
                h[q] = gm_f16_bits(hv);
                l[q] = gm_f16_bits((_Float16)(c - (float)hv));
            }
            *reinterpret_cast<uint2*>(&s_hi[a_off[i]]) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
            *reinterpret_cast<uint2*>(&s_lo[a_off[i]]) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
        }
        __builtin_amdgcn_s_waitcnt(0);   // this stage's weight fragments have landed (issued before the x loads just consumed)
        __syncthreads();
        if (s + 1 < n_stage) prefetch(s + 1);
        f16x8 ah[GM_MT], al[GM_MT];
#pragma unroll
        for (int m = 0; m < GM_MT; ++m) {
            const int off = (wm * 64 + m * 16 + frag_row) * GM_ASTRIDE + frag_k;
            ah[m] = *reinterpret_cast<const f16x8*>(&s_hi[off]);
            al[m] = *reinterpret_cast<const f16x8*>(&s_lo[off]);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const f16x8 bh = s_b[(0 * (BN / 16) + wn * NT + n) * 64 + lane];
            const f16x8 bl = s_b[(1 * (BN / 16) + wn * NT + n) * 64 + lane];
#pragma unroll
            for (int m = 0; m < GM_MT; ++m) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh, acc[m][n], 0, 0, 0);
            }
        }
    }
    // ---- epilogue: D[row = (lane >> 4) * 4 + r][col = lane & 15] -> per-wave LDS strip [16 rows][16 NT cols] -> row stores.
    // The strips are private to a wave and LDS executes a wave's accesses in order, so only the first hand-over (stage
    // buffers -> strips) needs a workgroup barrier; the residual rows of strip m + 1 are fetched while strip m is written.
    const int g = lane >> 4, px = lane & 15;
    float* so = s_out + wave * 16 * OSTRIDE;
    constexpr int ROW_F4 = 4 * NT;                    // float4 per output row of the wave tile
    constexpr int E_ITERS = 16 * ROW_F4 / 64;         // float4 per lane per strip (6 or 3)
    float4 rr[2][E_ITERS];
    auto out_offset = [&](int m, int i, int& c) -> size_t {
        const int e = lane + 64 * i;
        const int row = e / ROW_F4, q4 = e - row * ROW_F4;
        const int tr = wm * 64 + m * 16 + row;
        c = cg * 16 + (tr >> 3);
        return grow(tr) * (size_t)N + n0 + wn * (16 * NT) + 4 * q4;
    };
#if GM_RESID_AHEAD
    // The residual of ALL four strips is requested at once (until round 4 it was fetched strip by strip, one ahead: the epilogue paid the
    // load latency about four times per workgroup, with both co-resident workgroups' matrix pipes idle meanwhile): strips 2 and 3 into
    // registers in front of the barrier, strips 0 and 1 by LDS-DMA into the freed stage buffers right behind it (hidden from the
    // compiler - ac_lds_dma16 - so that its waits for the register loads are not turned into waits on every LDS access; the explicit
    // s_waitcnt in front of strip 0 orders the DMA, and nothing younger - no store - is outstanding at that point, so it waits for
    // exactly these loads).  Lane e of a strip owns float4 e of the strip in every form.
    static_assert(GM_MT == 4, "the residual prefetch is laid out for four strips");
    unsigned char* s_res = s_raw + O_BYTES + wave * R_BYTES;
    if (RESID) {
#pragma unroll
        for (int m = 2; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < E_ITERS; ++i) { int c; rr[m - 2][i] = *reinterpret_cast<const float4*>(resid + out_offset(m, i, c)); }
    }
    __syncthreads();                     // every wave is done reading the stage buffers
    if (RESID) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < E_ITERS; ++i) { int c; ac_lds_dma16(resid + out_offset(m, i, c), s_res + (m * E_ITERS + i) * 1024); }
    }
    // tile row & 7 = time row of the tile; 16 m and 64 wm are multiples of 8, so float4 i of this lane meets the SAME time row in
    // every strip: its maximum is carried in a register across the strips and committed once
    float vm[E_ITERS];
#pragma unroll
    for (int i = 0; i < E_ITERS; ++i) vm[i] = 0.f;
#pragma unroll
    for (int m = 0; m < GM_MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) so[(g * 4 + r) * OSTRIDE + n * 16 + px] = acc[m][n][r];
        if (RESID && m == 0) __builtin_amdgcn_s_waitcnt(0);          // the DMA'd strips have landed (this wave's own pieces only: no barrier needed)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < E_ITERS; ++i) {
            const int e = lane + 64 * i;
            const int row = e / ROW_F4, q4 = e - row * ROW_F4;
            int c;
            const size_t o = out_offset(m, i, c);
            const int tr = wm * 64 + m * 16 + row;
            const float sc = scale[c] * (w_unscale * s_inv[tr & 7]), sh = shift[c];
            float4 v = *reinterpret_cast<const float4*>(&so[row * OSTRIDE + 4 * q4]);
            v.x = fmaxf(v.x * sc + sh, 0.f); v.y = fmaxf(v.y * sc + sh, 0.f);
            v.z = fmaxf(v.z * sc + sh, 0.f); v.w = fmaxf(v.w * sc + sh, 0.f);
            if (RESID) {
                const float4 q = m >= 2 ? rr[m - 2][i] : *reinterpret_cast<const float4*>(s_res + (m * E_ITERS + i) * 1024 + lane * 16);
                v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
            }
            vm[i] = fmaxf(fmaxf(vm[i], fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            *reinterpret_cast<float4*>(y + o) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
#else        // round 3's epilogue: the residual fetched strip by strip, one ahead (kept for same-box A/B runs: EXTRA=-DGM_RESID_AHEAD=0)
    if (RESID) {
#pragma unroll
        for (int i = 0; i < E_ITERS; ++i) { int c; rr[0][i] = *reinterpret_cast<const float4*>(resid + out_offset(0, i, c)); }
    }
    __syncthreads();                     // every wave is done reading the stage buffers
    // tile row & 7 = time row of the tile; 16 m and 64 wm are multiples of 8, so float4 i of this lane meets the SAME time row in
    // every strip: its maximum is carried in a register across the strips and committed once
    float vm[E_ITERS];
#pragma unroll
    for (int i = 0; i < E_ITERS; ++i) vm[i] = 0.f;
#pragma unroll
    for (int m = 0; m < GM_MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) so[(g * 4 + r) * OSTRIDE + n * 16 + px] = acc[m][n][r];
        if (RESID && m + 1 < GM_MT) {
#pragma unroll
            for (int i = 0; i < E_ITERS; ++i) { int c; rr[(m + 1) & 1][i] = *reinterpret_cast<const float4*>(resid + out_offset(m + 1, i, c)); }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < E_ITERS; ++i) {
            const int e = lane + 64 * i;
            const int row = e / ROW_F4, q4 = e - row * ROW_F4;
            int c;
            const size_t o = out_offset(m, i, c);
            const int tr = wm * 64 + m * 16 + row;
            const float sc = scale[c] * (w_unscale * s_inv[tr & 7]), sh = shift[c];
            float4 v = *reinterpret_cast<const float4*>(&so[row * OSTRIDE + 4 * q4]);
            v.x = fmaxf(v.x * sc + sh, 0.f); v.y = fmaxf(v.y * sc + sh, 0.f);
            v.z = fmaxf(v.z * sc + sh, 0.f); v.w = fmaxf(v.w * sc + sh, 0.f);
            if (RESID) { const float4 q = rr[m & 1][i]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
            vm[i] = fmaxf(fmaxf(vm[i], fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            *reinterpret_cast<float4*>(y + o) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
#endif
    if (out_amax) {                      // max |y| per time row of the tile
        // a row is ROW_F4 = 24 or 12 float4 long: aligned groups of four lanes always share a row -> two shuffles, then one LDS
        // atomic per group (16 lanes per instruction instead of 64, E_ITERS instructions instead of 4 E_ITERS)
#pragma unroll
        for (int i = 0; i < E_ITERS; ++i) {
            float a = fmaxf(vm[i], __shfl_xor(vm[i], 1, AC_WAVE));
            a = fmaxf(a, __shfl_xor(a, 2, AC_WAVE));
            if ((lane & 3) == 0) atomicMax(&s_tmax[((lane + 64 * i) / ROW_F4) & 7], __float_as_uint(a));
        }
        __syncthreads();
        if (tid < 8 && s_tmax[tid]) atomicMax(reinterpret_cast<unsigned*>(out_amax + (size_t)item * T + tb * 8 + tid), s_tmax[tid]);
    }
}

extern "C" int ac_tdf_linear_f16x3(ac_ctx* ctx, const float* x, const void* w_packed, const float* scale, const float* shift,
                                    const float* resid, float* y, long long M, int N, int K, int T, int C, float w_unscale,
                                    const float* in_amax, float* out_amax, void* stream) {
    AC_REQUIRE(ctx && x && w_packed && scale && shift && y, "null pointer");
    AC_REQUIRE(M > 0 && C > 0 && T > 0 && C % 16 == 0 && T % 8 == 0 && M % ((long long)C * T) == 0,
               "rows = items x C x T with C % 16 == 0 and T % 8 == 0 (a tile is 16 channels x 8 time rows)");
    AC_REQUIRE(K > 0 && K % GM_BK == 0, "K % 32 == 0");
    AC_REQUIRE(N > 0 && N % 96 == 0, "N % 96 == 0");
    const bool wide = (N % 192) == 0;
    const long long n_mblk = M / GM_BM;
    const int n_nblk = N / (wide ? 192 : 96);
    long long nblk = n_mblk * n_nblk;
    AC_REQUIRE(nblk < (1LL << 31) - 8 && n_mblk < (1LL << 31), "grid too large");
    // Work order: 4 column blocks x 16 row tiles per super-group where the shape has them.  With all (8 - 16) column blocks of a row
    // tile side by side an XCD's 64 resident workgroups stream every column block's weights at once (4.7 MB at level 0: more than its
    // 4 MB L2) and the layer read 2.32x its algorithmic bytes; 4 x 16 reads 1.66x (profiles/r04c: 12.6 -> 9.0 GB, 3.77 -> 3.70 ms).
    int ord_g = n_nblk, ord_r = 1;
    if (n_nblk % 4 == 0 && n_nblk > 4 && n_mblk % 16 == 0) { ord_g = 4; ord_r = 16; }
#if AC_PROBES
    if (const char* e = getenv("AC_PROBE_TDF_ORDER")) {                       // "G,R" (tools/tdf_order_probe.py)
        int g_ = 0, r_ = 0;
        if (sscanf(e, "%d,%d", &g_, &r_) == 2 && g_ > 0 && r_ > 0 && n_nblk % g_ == 0 && n_mblk % r_ == 0) { ord_g = g_; ord_r = r_; }
    }
#endif
    dim3 grid((unsigned)nblk), block(256);
    hipStream_t st = (hipStream_t)stream;
    const f16x8* wp = (const f16x8*)w_packed;
#define GM_GO(NT_, RES_) hipLaunchKernelGGL((k_tdf_linear_f16x3<NT_, RES_>), grid, block, 0, st, x, wp, scale, shift, resid, y, (int)n_mblk, N, K, T, C, w_unscale, in_amax, out_amax, ord_g, ord_r)
    if (wide) { if (resid) GM_GO(6, true); else GM_GO(6, false); }
    else      { if (resid) GM_GO(3, true); else GM_GO(3, false); }
#undef GM_GO
    AC_LAUNCH_CHECK();
    return AC_OK;
}
