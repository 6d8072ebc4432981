// 3x3 convolution (stride 1, pad 1, NCHW float32 in/out) on the 16-bit matrix cores with a 3-term split:
//   x = xh + xl, w = wh + wl (float16 pairs; xh = f16(x), xl = f16(x - xh): 22 mantissa bits together)
//   x*w ~= xh*wh + xh*wl + xl*wh, every f16 x f16 product exact in the float32 accumulator of
//   v_mfma_f32_16x16x32_f16; the dropped xl*wl term is ~2^-22 relative.
// That is float32-class accuracy (measured per conv: ~3e-7 of peak against float64, like MIOpen's float32 conv)
// at the 16-bit MFMA rate, which on gfx950 is 16x the f32 MFMA rate — 3 products still leave 5.3x headroom.
// Ranges: weights are pre-scaled on the host by a power of two (undone exactly in the epilogue) so their low
// parts stay normal; activations are saturated to +-65504 (the f16 range; U-Net activations are O(1..1e3)) and
// their representation error is bounded by max(2^-22 |x|, 3e-8).
//
// Implicit GEMM, D[co][pixel] = sum_k A[co][k] B[k][pixel]:  A = weights (M = 48 output channels per workgroup,
// pre-packed on the host in MFMA fragment order), B = activations (N = an 8 x 32 pixel tile, 4 waves x 64 pixels).
// K is walked in blocks of 16 input channels; inside a block taps 0..7 are paired into 4 k-steps of 32 (lane groups 0-1
// carry tap 2p, groups 2-3 tap 2p+1).  Tap 8 of an even block is paired with tap 8 of the next block: the even block only
// keeps its tap-8 activation fragments in registers (lane groups 0-1), the odd block reads its own into lane groups 2-3 of
// the same registers and issues the shared k-step, so two blocks cost 9 k-steps instead of 10 (a trailing unpaired block
// issues its tap 8 against zero weights in lane groups 2-3).  Per block the
// (8+2) x (32+8) x 16 input patch is converted to f16 hi/lo once and staged in LDS as [pixel][channel].
// Staging loads are aligned float4 (a 40-column span per patch row); the epilogue (+ bias, optional ReLU) goes through
// LDS so that every output row segment leaves as one 128-byte line.
#include "ac_common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CV_TH 8
#define CV_TW 32
#define CV_PH (CV_TH + 2)
#define CV_PW (CV_TW + 2)
#define CV_CB 16                 // input channels per LDS stage
#define CV_PIX_STRIDE 16         // f16 elements per staged pixel (32 bytes, no padding: see the LDS layout note below)
#define CV_LW 40                 // staged columns per patch row: the 10 aligned float4 of a row, x0 - 4 .. x0 + 35
#define CV_COB 48                // output channels per workgroup (3 MFMA row tiles)
#define CV_MT 3

__device__ inline unsigned short f16_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }

// LDS layout of the staged patch: [row 0..9][column 0..39][16 channels] f16, hi and lo separately, 32 bytes per pixel.
// ds_read_b128 serves a wave in four fixed 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...): with 16
// consecutive pixels per group and the two channel halves 16 bytes apart, a 32-byte pixel stride is conflict-free
// (the former 48-byte stride was 2-way: SQ_LDS_BANK_CONFLICT was 46 % of the LDS cycles).  ds_write_b64 serves 16
// contiguous lanes per cycle on a 128-byte bank window, so the staging items are ordered (channel quad, column quad
// low bits) fastest and odd column quads swap their pixel pairs (`cv_phys`): writes are at most 2-way.
// staging work items: (row 0..9, column quad 0..9 (+2 idle), channel quad 0..3) -> 480 slots, 400 live
#define CV_QUADS 10
#define CV_ITEMS (CV_PH * 3 * 16)                                          // 480
#define CV_ACT_ITERS ((CV_ITEMS + 255) / 256)                              // 2
__device__ inline int cv_phys(int c) { return c ^ (((c >> 2) & 1) << 1); } // staged column of logical column c (0..39)
#define CV_WFRAGS (5 * 2 * CV_MT * 64)                                     // 1920 16-byte weight fragments per stage
#define CV_WFRAGS_PAIRS (4 * 2 * CV_MT * 64)                               // 1536: the four in-block tap pairs
#define CV_OUT_STRIDE (CV_TW + 4)                                          // floats per (co, row) line of the output staging

template <bool RELU>
__global__ __launch_bounds__(256, 2) void k_conv3x3_f16x3(const float* __restrict__ x, const f16x8* __restrict__ wpk,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int C_in, int C_out, int H, int W, float w_unscale, int bw,
                                                          const float* __restrict__ in_amax, float* __restrict__ out_amax) {
    // one LDS arena: [hi patch | lo patch | weight fragments] during the K loop, re-used as the output staging tile
    // The weight fragments are double-buffered - even stages (4 k-steps) in one buffer, odd stages (5) in the other - and filled
    // with global_load_lds (no registers, issued a stage ahead): 25,600 + 24,576 + 30,720 = 80,896 B, two workgroups per CU
    // still fit the 160 KB.
    constexpr int PATCH_BYTES = 2 * CV_PH * CV_LW * CV_PIX_STRIDE * 2;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[PATCH_BYTES + (CV_WFRAGS_PAIRS + CV_WFRAGS) * 16];
    unsigned short* s_hi = reinterpret_cast<unsigned short*>(s_raw);
    unsigned short* s_lo = s_hi + CV_PH * CV_LW * CV_PIX_STRIDE;
    f16x8* s_w0 = reinterpret_cast<f16x8*>(s_raw + PATCH_BYTES);            // even stages
    f16x8* s_w1 = s_w0 + CV_WFRAGS_PAIRS;                                      // odd stages
    float* s_out = reinterpret_cast<float*>(s_raw);       // [48 co][8 rows][CV_OUT_STRIDE] = 55296 B <= arena (56320 B)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_cob = C_out / CV_COB;
    // XCD-aware work mapping: workgroups are dealt round-robin over the 8 XCDs (blocks L and L+8 share an L2), so XCD
    // `L & 7` walks its own contiguous strip of work items.  Inside a strip the C_out blocks of one pixel tile are
    // adjacent (their input patch is fetched over the fabric once, the siblings hit L2: measured FETCH_SIZE at
    // C = 96 drops from 3.1x to 1.6x of the input) and tiles run down `bw`-wide column bands so that tiles whose halos
    // overlap are close in the walk.
    const int tiles_x = W / CV_TW, tiles_y = H / CV_TH;
    int wi = blockIdx.x;
    if ((gridDim.x & 7) == 0) wi = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int cob = wi % n_cob;
    int t = wi / n_cob;
    const int b = t / (tiles_x * tiles_y);
    t -= b * (tiles_x * tiles_y);
    const int band = t / (tiles_y * bw);
    t -= band * (tiles_y * bw);
    const int y0 = (t / bw) * CV_TH, x0 = (band * bw + t % bw) * CV_TW;
    const int n_cb = C_in / CV_CB;
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * C_in * plane;
    // time-local power-of-two activation scale (ac_common.h): log2 scale of each patch row y0 - 1 .. y0 + 8
    __shared__ int s_ex[CV_PH + 2];
    if (tid < CV_PH + 2) {
        const int gy = y0 - 1 + tid;
        s_ex[tid] = (in_amax && tid < CV_PH && gy >= 0 && gy < H) ? ac_row_ex(in_amax[(size_t)b * H + gy]) : AC_EX_NONE;
    }
    __syncthreads();
    int ex_min = AC_EX_NONE, ex_max = -AC_EX_NONE;
#pragma unroll
    for (int r = 0; r < CV_PH; ++r) {
        const int e = s_ex[r];
        if (e != AC_EX_NONE) { ex_min = e < ex_min ? e : ex_min; ex_max = e > ex_max ? e : ex_max; }
    }
    ex_min = __builtin_amdgcn_readfirstlane(ex_min);
    ex_max = __builtin_amdgcn_readfirstlane(ex_max);
    // one scale for the tile where its rows are within 2^AC_ROWX_SPREAD of each other, else the row-exact path (ac_common.h)
    const bool rowx = ex_min != AC_EX_NONE && ex_max - ex_min > AC_ROWX_SPREAD;
    const float act_s = ex_min == AC_EX_NONE ? 1.f : ldexpf(1.f, ex_min);
    const float unscale = w_unscale * (ex_min == AC_EX_NONE ? 1.f : ldexpf(1.f, -ex_min));

    f32x4 acc[CV_MT][4];
#pragma unroll
    for (int m = 0; m < CV_MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 k8h[4], k8l[4];               // tap-8 activation fragments carried from an even channel block into the next one
#pragma unroll
    for (int q = 0; q < 4; ++q) { k8h[q] = (f16x8)(_Float16)0; k8l[q] = (f16x8)(_Float16)0; }

    const int g = lane >> 4, px = lane & 15;
    const int ci_off = 8 * (g & 1);
    const f16x8* wbase = wpk + (size_t)cob * n_cb * CV_WFRAGS;

    // per-thread staging coordinates (fixed across stages): one aligned float4 (4 pixels) of 4 channels each
    int a_src[CV_ACT_ITERS];                // float offset inside the plane of the float4, -1 = outside the image (zeros)
    int a_off[CV_ACT_ITERS];                // LDS element offset of the quad's first pixel (row * CV_LW + 4 * qd) * 16 + c4 * 4
    int a_c4[CV_ACT_ITERS];                 // channel quad, -1 = idle slot
    int a_flip[CV_ACT_ITERS];               // odd column quads store their pixel pairs swapped (cv_phys)
#pragma unroll
    for (int i = 0; i < CV_ACT_ITERS; ++i) {
        const int e = tid + 256 * i;
        const int c4 = (e >> 2) & 3, ql = e & 3, rest = e >> 4;
        const int py = rest / 3, qd = (rest - py * 3) * 4 + ql;
        if (e < CV_ITEMS && qd < CV_QUADS) {
            const int gy = y0 + py - 1, gx = x0 - 4 + 4 * qd;          // aligned: x0 % 32 == 0
            a_c4[i] = c4;
            a_flip[i] = (qd & 1) << 1;
            a_off[i] = (py * CV_LW + 4 * qd) * CV_PIX_STRIDE + c4 * 4;
            a_src[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
        } else {
            a_c4[i] = -1; a_flip[i] = 0; a_off[i] = 0; a_src[i] = -1;
        }
    }
    float4 pre_x[CV_ACT_ITERS][4];
    auto prefetch = [&](int cb) {
        const f16x8* wcb = wbase + (size_t)cb * CV_WFRAGS;
        // weight fragments are already in LDS order: each wave-instruction copies 64 of them (1 KB) straight into the buffer
        f16x8* dst = (cb & 1) ? s_w1 : s_w0;
        const int n_inst = (cb & 1) ? CV_WFRAGS / 64 : CV_WFRAGS_PAIRS / 64;            // 30 or 24 (an even stage keeps 4 k-steps in LDS)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int inst = wave + 4 * i;
            if (inst < n_inst) __builtin_amdgcn_global_load_lds(wcb + inst * 64 + lane, dst + inst * 64, 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < CV_ACT_ITERS; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ci = cb * CV_CB + (a_c4[i] < 0 ? 0 : a_c4[i]) * 4 + q;
                pre_x[i][q] = (a_src[i] >= 0) ? *reinterpret_cast<const float4*>(xb + (size_t)ci * plane + a_src[i])
                                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    // C_in / 16 odd: the last stage has no partner and its buffer holds four k-steps only, so every lane keeps its
    // own fragments of that stage's tap-8 step in registers from the start (they are per-lane data: LDS is only a broadcast)
    f16x8 tail_h[CV_MT], tail_l[CV_MT];
#pragma unroll
    for (int m = 0; m < CV_MT; ++m) {
        tail_h[m] = (f16x8)(_Float16)0; tail_l[m] = (f16x8)(_Float16)0;
        if (n_cb & 1) {
            const f16x8* wlast = wbase + (size_t)(n_cb - 1) * CV_WFRAGS;
            tail_h[m] = wlast[((4 * 2 + 0) * CV_MT + m) * 64 + lane];
            tail_l[m] = wlast[((4 * 2 + 1) * CV_MT + m) * 64 + lane];
        }
    }
    prefetch(0);
    for (int cb = 0; cb < n_cb; ++cb) {
        __syncthreads();                 // previous stage fully consumed
        const bool odd = cb & 1;
        const bool shared_step = odd || cb + 1 >= n_cb;      // this block issues the tap-8 k-step (its own half, or both halves)
        const f16x8* s_w = (cb & 1) ? s_w1 : s_w0;
#pragma unroll
        for (int i = 0; i < CV_ACT_ITERS; ++i) {
            if (a_c4[i] < 0) continue;
            const float* v4[4] = {&pre_x[i][0].x, &pre_x[i][1].x, &pre_x[i][2].x, &pre_x[i][3].x};
            float a_s = act_s;
            if (rowx) {                                                // row-exact path: the item's own patch row's scale
                const int e = s_ex[a_off[i] / (CV_LW * CV_PIX_STRIDE)];
                a_s = e == AC_EX_NONE ? 1.f : ldexpf(1.f, e);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {                              // 4 pixels of the float4
                unsigned short h4[4], l4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = fminf(fmaxf(v4[q][k] * a_s, -65504.f), 65504.f);
                    const _Float16 hv = (_Float16)v;                   // v_cvt_f16_f32, round to nearest even
                    h4[q] = f16_bits(hv);
                    l4[q] = f16_bits((_Float16)(v - (float)hv));
                }
                const int off = a_off[i] + (k ^ a_flip[i]) * CV_PIX_STRIDE;
                *reinterpret_cast<uint2*>(&s_hi[off]) = make_uint2((unsigned)h4[0] | ((unsigned)h4[1] << 16), (unsigned)h4[2] | ((unsigned)h4[3] << 16));
                *reinterpret_cast<uint2*>(&s_lo[off]) = make_uint2((unsigned)l4[0] | ((unsigned)l4[1] << 16), (unsigned)l4[2] | ((unsigned)l4[3] << 16));
            }
        }
        __builtin_amdgcn_s_waitcnt(0);            // this stage's weight fragments have landed (issued before the x loads just consumed)
        __syncthreads();
        if (cb + 1 < n_cb) prefetch(cb + 1);   // global loads of the next stage fly under this stage's MFMAs
#pragma unroll
        for (int pair = 0; pair < 4; ++pair) {
            const int tap = pair * 2 + (g >> 1);
            const int dy = tap / 3, dx = tap - dy * 3;
            f16x8 ah[CV_MT], al[CV_MT];
#pragma unroll
            for (int m = 0; m < CV_MT; ++m) {
                ah[m] = s_w[((pair * 2 + 0) * CV_MT + m) * 64 + lane];
                al[m] = s_w[((pair * 2 + 1) * CV_MT + m) * 64 + lane];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ty = 2 * wave + (q >> 1), tx = (q & 1) * 16 + px;
                const int off = ((ty + dy) * CV_LW + cv_phys(tx + dx + 3)) * CV_PIX_STRIDE + ci_off;   // patch column c is staged column c + 3
                f16x8 bh = *reinterpret_cast<const f16x8*>(&s_hi[off]);
                f16x8 bl = *reinterpret_cast<const f16x8*>(&s_lo[off]);
                if (rowx) { const _Float16 f = ac_rowx_frag_factor(s_ex, ty, dy); bh *= (f16x8)f; bl *= (f16x8)f; }
#pragma unroll
                for (int m = 0; m < CV_MT; ++m) {
                    acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl, acc[m][q], 0, 0, 0);
                    acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh, acc[m][q], 0, 0, 0);
                    acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh, acc[m][q], 0, 0, 0);
                }
            }
        }
        // tap 8 (dy = dx = 2): an even block loads its fragments into every lane group (groups 0-1 are the ones that count);
        // an odd block overwrites groups 2-3 only, so k 0..15 of the shared step is the even block, k 16..31 the odd one
        if (!odd || g >= 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ty = 2 * wave + (q >> 1), tx = (q & 1) * 16 + px;
                const int off = ((ty + 2) * CV_LW + cv_phys(tx + 2 + 3)) * CV_PIX_STRIDE + ci_off;
                k8h[q] = *reinterpret_cast<const f16x8*>(&s_hi[off]);
                k8l[q] = *reinterpret_cast<const f16x8*>(&s_lo[off]);
                if (rowx) { const _Float16 f = ac_rowx_frag_factor(s_ex, ty, 2); k8h[q] *= (f16x8)f; k8l[q] *= (f16x8)f; }
            }
        }
        if (shared_step) {
            f16x8 ah[CV_MT], al[CV_MT];
#pragma unroll
            for (int m = 0; m < CV_MT; ++m) {
                if (!odd) {       // trailing unpaired stage: its fifth k-step is not in LDS (fetched per lane before the loop)
                    ah[m] = tail_h[m];
                    al[m] = tail_l[m];
                } else {
                    ah[m] = s_w[((4 * 2 + 0) * CV_MT + m) * 64 + lane];
                    al[m] = s_w[((4 * 2 + 1) * CV_MT + m) * 64 + lane];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int m = 0; m < CV_MT; ++m) {
                    acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], k8l[q], acc[m][q], 0, 0, 0);
                    acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], k8h[q], acc[m][q], 0, 0, 0);
                    acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], k8h[q], acc[m][q], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue: accumulators (D[row = (lane>>4)*4 + r][col = lane&15]) -> LDS tile [co][row][x] -> 128-byte row stores
    __syncthreads();                     // all waves done with the stage buffers
    float vmax[2] = {0.f, 0.f};          // this wave's two output rows
#pragma unroll
    for (int m = 0; m < CV_MT; ++m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ty = 2 * wave + (q >> 1), tx = (q & 1) * 16 + px;
            const float us = rowx ? w_unscale * ac_rowx_unscale(s_ex, ty) : unscale;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m * 16 + g * 4 + r;
                float v = acc[m][q][r] * us + bias[cob * CV_COB + co];
                if (RELU) v = fmaxf(v, 0.f);
                vmax[q >> 1] = fmaxf(vmax[q >> 1], fabsf(v));
                s_out[(co * CV_TH + ty) * CV_OUT_STRIDE + tx] = v;
            }
        }
    }
    __syncthreads();
    float* ob = out + ((size_t)b * C_out + (size_t)cob * CV_COB) * plane;
    // 48 co x 8 rows x 8 float4 = 3072 float4 over 256 threads: 8 consecutive threads write one 128-byte row segment
    for (int e = tid; e < CV_COB * CV_TH * (CV_TW / 4); e += 256) {
        const int line = e >> 3, q4 = e & 7;
        const int co = line >> 3, ty = line & 7;
        const float4 v = *reinterpret_cast<const float4*>(&s_out[line * CV_OUT_STRIDE + 4 * q4]);
        *reinterpret_cast<float4*>(ob + (size_t)co * plane + (size_t)(y0 + ty) * W + x0 + 4 * q4) = v;
    }
    if (out_amax) {
        ac_amax_commit(vmax[0], out_amax + (size_t)b * H + y0 + 2 * wave);
        ac_amax_commit(vmax[1], out_amax + (size_t)b * H + y0 + 2 * wave + 1);
    }
}

static int cv_launch(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* out, int B, int C_in, int C_out,
                     int H, int W, float w_unscale, int relu, void* stream, const float* in_amax, float* out_amax) {
    AC_REQUIRE(ctx && x && w_packed && bias && out, "null pointer");
    AC_REQUIRE(B > 0 && C_in > 0 && C_in % CV_CB == 0 && C_out > 0 && C_out % CV_COB == 0, "C_in % 16 == 0 and C_out % 48 == 0");
    AC_REQUIRE(H > 0 && H % CV_TH == 0 && W > 0 && W % CV_TW == 0, "H % 8 == 0 and W % 32 == 0");
    AC_REQUIRE((long long)H * W < (1LL << 31), "plane too large");
    const long long nblk = (long long)B * (C_out / CV_COB) * (H / CV_TH) * (W / CV_TW);
    AC_REQUIRE(nblk < (1LL << 31), "grid too large");
    const int tiles_x = W / CV_TW;
    const int bw = tiles_x % 4 == 0 ? 4 : (tiles_x % 3 == 0 ? 3 : (tiles_x % 2 == 0 ? 2 : 1));   // column-band width (tiles)
    dim3 grid((unsigned)nblk), block(256);
    hipStream_t st = (hipStream_t)stream;
    const f16x8* wp = (const f16x8*)w_packed;
    if (relu) hipLaunchKernelGGL((k_conv3x3_f16x3<true>), grid, block, 0, st, x, wp, bias, out, C_in, C_out, H, W, w_unscale, bw, in_amax, out_amax);
    else      hipLaunchKernelGGL((k_conv3x3_f16x3<false>), grid, block, 0, st, x, wp, bias, out, C_in, C_out, H, W, w_unscale, bw, in_amax, out_amax);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_conv3x3_f16x3(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* out, int B, int C_in,
                                 int C_out, int H, int W, float w_unscale, int relu, const float* in_amax, float* out_amax,
                                 void* stream) {
    return cv_launch(ctx, x, w_packed, bias, out, B, C_in, C_out, H, W, w_unscale, relu, stream, in_amax, out_amax);
}
