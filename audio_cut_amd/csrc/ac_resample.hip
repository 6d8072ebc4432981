// The U-Net's 2x2 / stride-2 resampling layers as ONE fused kernel each, on the 16-bit matrix cores with the 3-term
// float16 split of ac_conv.hip / ac_gemm.hip (float32-class products, float32 accumulation):
//   down: out[b][co][y][x]       = relu(bias[co] + sum_{ci,dy,dx} x[b][ci][2y+dy][2x+dx] * w[co][ci][dy][dx])
//   up:   out[b][co][2y+dy][2x+dx] = relu(bias[co] + sum_ci x[b][ci][y][x] * w[ci][co][dy][dx]) * skip[b][co][2y+dy][2x+dx]
// Both are GEMMs over pixels: A[m][k] with m = pixel (b, y, x) and k = (ci, dy, dx) (down) or ci (up), B = packed
// weights [N][K] (N = co for down, (co, dy, dx) for up; conv_pack.pack_linear with bn = 96, zero padded).  The gather
// (space-to-depth) lives in the A loader, the scatter (depth-to-space), bias, ReLU and skip product in the epilogue,
// so each layer reads its input once and writes its output once.
//
// Workgroup: 256 threads = 2 x 2 waves, tile 128 pixels x 96 columns, wave tile 64 x 48; K in stages of 32; the next
// stage's global loads are issued before the current stage's MFMAs (same structure as k_tdf_linear_f16x3).
#include "ac_common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define RS_BM 128
#define RS_BN 96
#define RS_BK 32
// Staged A tile [pixel][k] (f16 hi and lo).  The layout and the thread -> (pixel, channel) mapping of the loaders are chosen so that
// BOTH the staging stores (ds_write_b64: four groups of 16 contiguous lanes, 32 banks) and the fragment reads (ds_read_b128) are
// conflict-free; round 2's mapping put the 16 lanes of a store group on pixels 4 (up) or 2 (down) rows apart = 384 / 192 bytes =
// the same bank: 16-way / 8-way conflicts on every staging store, 67-74 % of all LDS cycles (SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE, profiles/r03f_pmc_probe.txt).
//   down: row stride 96 bytes; a store group = 8 consecutive channel quads of 2 pixel pairs
//   up:   row stride 112 bytes with odd pixels shifted by 16 bytes; a store group = 8 consecutive channel quads of 2 pixel quads
#define RS_ASTRIDE_DN 48         // f16 per staged pixel row: 32 used + 16 pad -> 96 bytes
#define RS_ASTRIDE_UP 56         // 112 bytes (+ 8 f16 for odd pixels: 80 of the 112 used)
#define RS_MT 4
#define RS_NT 3
#define RS_BFRAGS (2 * (RS_BN / 16) * 64)          // 768 16-byte weight fragments per stage
#define RS_B_ITERS (RS_BFRAGS / 256)               // 3
#ifndef RS_SKIP_AHEAD
#define RS_SKIP_AHEAD 1            // up: strips of the skip tensor requested ahead of the strip being written (1 .. RS_MT - 1; RS_MT = all at once)
#endif

__device__ inline unsigned rs_pack_hi(float a, float b, unsigned& lo_out, float s) {
    const float ca = fminf(fmaxf(a * s, -65504.f), 65504.f), cb = fminf(fmaxf(b * s, -65504.f), 65504.f);
    const _Float16 ha = (_Float16)ca, hb = (_Float16)cb;
    const _Float16 la = (_Float16)(ca - (float)ha), lb = (_Float16)(cb - (float)hb);
    lo_out = (unsigned)__builtin_bit_cast(unsigned short, la) | ((unsigned)__builtin_bit_cast(unsigned short, lb) << 16);
    return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}

__device__ inline void rs_put4(unsigned short* s_hi, unsigned short* s_lo, int off, float a, float b, float c, float d, float s) {
    unsigned l0, l1;
    const unsigned h0 = rs_pack_hi(a, b, l0, s), h1 = rs_pack_hi(c, d, l1, s);
    *reinterpret_cast<uint2*>(&s_hi[off]) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(&s_lo[off]) = make_uint2(l0, l1);
}

// MODE 0 = down (x [B][Ci][H][W] -> out [B][Co][H/2][W/2]);  MODE 1 = up (x [B][Ci][H][W] -> out [B][Co][2H][2W]).
// P = pixels per image on the GEMM's M axis (down: (H/2)*(W/2); up: H*W), P % 128 == 0.
template <int MODE>
__global__ __launch_bounds__(256, 3) void k_resample2x_f16x3(const float* __restrict__ x, const f16x8* __restrict__ wpk,
                                                             const float* __restrict__ bias, const float* __restrict__ skip,
                                                             float* __restrict__ out, int Ci, int Co, int H, int W,
                                                             int n_stage, int n_nblk, int n_mblk, float w_unscale,
                                                             const float* __restrict__ in_amax, float* __restrict__ out_amax) {
    constexpr int RS_ASTRIDE = MODE ? RS_ASTRIDE_UP : RS_ASTRIDE_DN;
    constexpr int A_BYTES = 2 * RS_BM * RS_ASTRIDE * 2;                      // 24576 (down) / 28672 (up)
    constexpr int OSTRIDE_UP = 48 + 4;                                       // up: strip [16 m][48 n]
    constexpr int OSTRIDE_DN = 64 + 4;                                       // down: strip [16 n][64 m]
    constexpr int O_BYTES = 4 * 16 * (MODE ? OSTRIDE_UP : OSTRIDE_DN) * 4;
    constexpr int ARENA = (A_BYTES + 2 * RS_BFRAGS * 16) > O_BYTES ? (A_BYTES + 2 * RS_BFRAGS * 16) : O_BYTES;   // weights double-buffered (LDS-DMA)
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[ARENA];
    unsigned short* s_hi = reinterpret_cast<unsigned short*>(s_raw);
    unsigned short* s_lo = s_hi + RS_BM * RS_ASTRIDE;
    f16x8* s_b0 = reinterpret_cast<f16x8*>(s_raw + A_BYTES);
    f16x8* s_b1 = s_b0 + RS_BFRAGS;
    float* s_out = reinterpret_cast<float*>(s_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int wi = blockIdx.x;                                   // XCD-aware order: column blocks of one pixel tile are neighbours
    if ((gridDim.x & 7) == 0) wi = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int nb = wi % n_nblk, mb = wi / n_nblk;
    if (mb >= n_mblk) return;
    const int Ho = MODE ? H : (H >> 1), Wo = MODE ? W : (W >> 1);           // the M axis walks an Ho x Wo pixel grid
    const int P = Ho * Wo;
    const int tiles_per_img = P / RS_BM;
    const int b = mb / tiles_per_img;
    const int p0 = (mb - b * tiles_per_img) * RS_BM;                         // first pixel of the tile inside its image
    const size_t plane_in = (size_t)H * W;
    const float* xb = x + (size_t)b * Ci * plane_in;
    // time-local power-of-two activation scale (ac_common.h): GEMM rows (pixels) are independent, so each is scaled by the
    // maximum of the input rows IT reads: its own row (up), rows 2 yo and 2 yo + 1 (down).  pix_scale(pixel of the M axis).
    const float* amax_in = in_amax ? in_amax + (size_t)b * H : nullptr;
    auto pix_scale = [&](int pix, float* inv) -> float {
        const int r = MODE ? pix / W : 2 * (pix / Wo);
        return ac_act_scale_lane(amax_in, r, MODE ? r : r + 1, inv);
    };

    f32x4 acc[RS_MT][RS_NT];
#pragma unroll
    for (int m = 0; m < RS_MT; ++m)
#pragma unroll
        for (int n = 0; n < RS_NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- A loader state --------------------------------------------------------------------------------------
    // down: item e = tid + 256 i: (channel e & 7 of the stage's 8, pixel pair p = e >> 3), two float4 (dy = 0, 1) = 2 pixels x 4 taps
    // up:   item (channel quad tid & 7 of the stage's 8, pixel quad q = tid >> 3), four float4 = 4 pixels x 4 channels
    constexpr int A_LD = MODE ? 4 : 4;                       // float4 loads per thread per stage (both modes: 4)
    float4 pre_a[A_LD];
    size_t dn_src[2];
    int dn_ci[2], dn_off[2];
    float dn_s[2] = {1.f, 1.f};
    float ld_inv;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i;
            const int cl = e & 7, p = e >> 3;
            const int pix = p0 + 2 * p;
            const int yo = pix / Wo, xo = pix - yo * Wo;
            dn_ci[i] = cl;
            dn_src[i] = (size_t)(2 * yo) * W + 2 * xo;
            dn_off[i] = (2 * p) * RS_ASTRIDE + cl * 4;
            dn_s[i] = pix_scale(pix, &ld_inv);                   // the two pixels of a pair share their rows (Wo even)
        }
    }
    const int up_c4 = tid & 7, up_q = tid >> 3;
    const float act_s = MODE ? pix_scale(p0 + 4 * up_q, &ld_inv) : 1.f;      // up: the four pixels this thread stages share one row (W % 4 == 0)
    const f16x8* wbase = wpk + (size_t)nb * n_stage * RS_BFRAGS;

    auto prefetch = [&](int s) {
        f16x8* dst = (s & 1) ? s_b1 : s_b0;         // weight fragments: global -> LDS by DMA, 1 KB per wave-instruction, one stage ahead
#pragma unroll
        for (int i = 0; i < RS_B_ITERS; ++i) {
            const int inst = wave + 4 * i;
            __builtin_amdgcn_global_load_lds(wbase + (size_t)s * RS_BFRAGS + inst * 64 + lane, dst + inst * 64, 16, 0, 0);
        }
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* src = xb + (size_t)(s * 8 + dn_ci[i]) * plane_in + dn_src[i];
                pre_a[2 * i + 0] = *reinterpret_cast<const float4*>(src);
                pre_a[2 * i + 1] = *reinterpret_cast<const float4*>(src + W);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ci = s * RS_BK + up_c4 * 4 + j;
                pre_a[j] = (ci < Ci) ? *reinterpret_cast<const float4*>(xb + (size_t)ci * plane_in + p0 + 4 * up_q)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto commit = [&]() {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 r0 = pre_a[2 * i], r1 = pre_a[2 * i + 1];          // (px0 dx0, px0 dx1, px1 dx0, px1 dx1) for dy = 0 / 1
                rs_put4(s_hi, s_lo, dn_off[i], r0.x, r0.y, r1.x, r1.y, dn_s[i]);
                rs_put4(s_hi, s_lo, dn_off[i] + RS_ASTRIDE, r0.z, r0.w, r1.z, r1.w, dn_s[i]);
            }
        } else {
            const int off = (4 * up_q) * RS_ASTRIDE + up_c4 * 4;            // pixel 4 up_q is even: pixels + 1 and + 3 carry the odd-pixel shift of 8 f16
            rs_put4(s_hi, s_lo, off + 0 * RS_ASTRIDE, pre_a[0].x, pre_a[1].x, pre_a[2].x, pre_a[3].x, act_s);
            rs_put4(s_hi, s_lo, off + 1 * RS_ASTRIDE + 8, pre_a[0].y, pre_a[1].y, pre_a[2].y, pre_a[3].y, act_s);
            rs_put4(s_hi, s_lo, off + 2 * RS_ASTRIDE, pre_a[0].z, pre_a[1].z, pre_a[2].z, pre_a[3].z, act_s);
            rs_put4(s_hi, s_lo, off + 3 * RS_ASTRIDE + 8, pre_a[0].w, pre_a[1].w, pre_a[2].w, pre_a[3].w, act_s);
        }
    };

    const int frag_row = lane & 15, frag_k = 8 * (lane >> 4);
    prefetch(0);
    for (int s = 0; s < n_stage; ++s) {
        __syncthreads();
        commit();
        __builtin_amdgcn_s_waitcnt(0);       // this stage's weight fragments have landed
        __syncthreads();
        if (s + 1 < n_stage) prefetch(s + 1);
        const f16x8* s_b = (s & 1) ? s_b1 : s_b0;
        f16x8 ah[RS_MT], al[RS_MT];
#pragma unroll
        for (int m = 0; m < RS_MT; ++m) {
            const int off = (wm * 64 + m * 16 + frag_row) * RS_ASTRIDE + frag_k + (MODE ? (frag_row & 1) * 8 : 0);
            ah[m] = *reinterpret_cast<const f16x8*>(&s_hi[off]);
            al[m] = *reinterpret_cast<const f16x8*>(&s_lo[off]);
        }
#pragma unroll
        for (int n = 0; n < RS_NT; ++n) {
            const f16x8 bh = s_b[(0 * (RS_BN / 16) + wn * RS_NT + n) * 64 + lane];
            const f16x8 bl = s_b[(1 * (RS_BN / 16) + wn * RS_NT + n) * 64 + lane];
#pragma unroll
            for (int m = 0; m < RS_MT; ++m) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh, acc[m][n], 0, 0, 0);
            }
        }
    }

    // ---- epilogues: D[row m = (lane >> 4) * 4 + r][col n = lane & 15]; strips are wave-private (LDS is in-order per wave)
    const int g = lane >> 4, px = lane & 15;
    __syncthreads();                     // every wave is done reading the stage buffers
    float* amax_slots = out_amax ? out_amax + (size_t)b * (MODE ? 2 * H : (H >> 1)) : nullptr;
    if (MODE == 0) {
        // down: NCHW output, pixels contiguous per channel.  Per n-tile: strip [16 n][64 m] -> 256-byte runs per channel.
        float* so = s_out + wave * 16 * OSTRIDE_DN;
        const size_t plane_out = (size_t)P;
        float vmax = 0.f;
        float inv4;                      // e & 15 == lane & 15: every float4 of this lane belongs to the same four output pixels
        (void)pix_scale(p0 + wm * 64 + 4 * (lane & 15), &inv4);
        const float unscale = w_unscale * inv4;
#pragma unroll
        for (int n = 0; n < RS_NT; ++n) {
#pragma unroll
            for (int m = 0; m < RS_MT; ++m)
                *reinterpret_cast<float4*>(&so[px * OSTRIDE_DN + m * 16 + 4 * g]) =
                    make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = lane + 64 * i;
                const int nn = e >> 4, q4 = e & 15;
                const int co = nb * RS_BN + wn * 48 + n * 16 + nn;
                if (co < Co) {
                    const float bv = bias[co];
                    float4 v = *reinterpret_cast<const float4*>(&so[nn * OSTRIDE_DN + 4 * q4]);
                    v.x = fmaxf(v.x * unscale + bv, 0.f); v.y = fmaxf(v.y * unscale + bv, 0.f);
                    v.z = fmaxf(v.z * unscale + bv, 0.f); v.w = fmaxf(v.w * unscale + bv, 0.f);
                    vmax = fmaxf(fmaxf(vmax, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                    *reinterpret_cast<float4*>(out + ((size_t)b * Co + co) * plane_out + p0 + wm * 64 + 4 * q4) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // every float4 of this lane = output pixels p0 + wm * 64 + 4 (lane & 15) .. + 3 of one output row (Wo % 4 == 0)
        if (amax_slots) ac_amax_commit_blocks(vmax, (p0 + wm * 64 + 4 * (lane & 15)) / Wo, amax_slots);
    } else {
        // up: column n = co * 4 + dy * 2 + dx.  Per m-tile: strip [16 m][48 n]; one float4 = 2 input pixels x (dx 0, 1) of one
        // (co, dy): 8 consecutive lanes write a 128-byte run of an output row.  The skip rows are fetched one strip ahead.
        float* so = s_out + wave * 16 * OSTRIDE_UP;
        const int W2 = 2 * W;
        const size_t plane_out = (size_t)4 * P;
        constexpr int SKN = RS_SKIP_AHEAD >= RS_MT ? RS_MT : RS_SKIP_AHEAD + 1;      // ring of skip strips in registers
        float4 sk[SKN][3];
        auto out_offset = [&](int m, int i, int& nn) -> size_t {
            const int e = lane + 64 * i;                   // 192 float4 per strip: (co_dy = e >> 3) in 0..23, m pair = e & 7
            const int cd = e >> 3, mp = e & 7;
            nn = cd * 2;                                   // strip column of dx = 0
            const int n_glob = nb * RS_BN + wn * 48 + nn;
            const int co = n_glob >> 2, dy = (n_glob >> 1) & 1;
            const int pix = p0 + wm * 64 + m * 16 + 2 * mp;
            const int yy = pix / W, xx = pix - yy * W;
            return ((size_t)b * Co + co) * plane_out + (size_t)(2 * yy + dy) * W2 + 2 * xx;
        };
        if (skip) {
#pragma unroll
            for (int m = 0; m < (RS_SKIP_AHEAD < RS_MT ? RS_SKIP_AHEAD : RS_MT); ++m)
#pragma unroll
                for (int i = 0; i < 3; ++i) { int nn; sk[m % SKN][i] = *reinterpret_cast<const float4*>(skip + out_offset(m, i, nn)); }
        }
        // the 64 pixels of this wave lie in at most two input rows (W >= 64): row_a and row_a + 1, the second from pixel `bnd` on.
        // Their two scales are wave-uniform; the four output maxima (2 input rows x dy) are carried across the strips.
        const int pw0 = p0 + wm * 64;
        const int row_a = __builtin_amdgcn_readfirstlane(pw0 / W);
        const int row_b = row_a + 1 < H ? row_a + 1 : row_a;
        const int bnd = (row_a + 1) * W;
        float inv_a, inv_b;
        (void)ac_act_scale(amax_in, row_a, row_a, 1.f, 0.f, &inv_a);
        (void)ac_act_scale(amax_in, row_b, row_b, 1.f, 0.f, &inv_b);
        float vrow[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int m = 0; m < RS_MT; ++m) {
#pragma unroll
            for (int n = 0; n < RS_NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) so[(g * 4 + r) * OSTRIDE_UP + n * 16 + px] = acc[m][n][r];
            if (skip && m + RS_SKIP_AHEAD < RS_MT) {
#pragma unroll
                for (int i = 0; i < 3; ++i) { int nn; sk[(m + RS_SKIP_AHEAD) % SKN][i] = *reinterpret_cast<const float4*>(skip + out_offset(m + RS_SKIP_AHEAD, i, nn)); }
            }
            __builtin_amdgcn_wave_barrier();
            float vmax[2] = {0.f, 0.f};          // output rows 2 yy (dy = 0) and 2 yy + 1 of this lane's input pixel pair
            const bool second = pw0 + m * 16 + 2 * (lane & 7) >= bnd;     // W is even: a pixel pair shares its row
            const float unscale = w_unscale * (second ? inv_b : inv_a);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                int nn;
                const size_t o = out_offset(m, i, nn);
                const int mp = (lane + 64 * i) & 7;
                const int co = (nb * RS_BN + wn * 48 + nn) >> 2;
                const int dy = ((nb * RS_BN + wn * 48 + nn) >> 1) & 1;
                const float bv = bias[co];
                const float2 a0 = *reinterpret_cast<const float2*>(&so[(2 * mp) * OSTRIDE_UP + nn]);       // pixel 2mp:   dx 0, 1
                const float2 a1 = *reinterpret_cast<const float2*>(&so[(2 * mp + 1) * OSTRIDE_UP + nn]);   // pixel 2mp+1: dx 0, 1
                float4 v = make_float4(fmaxf(a0.x * unscale + bv, 0.f), fmaxf(a0.y * unscale + bv, 0.f),
                                       fmaxf(a1.x * unscale + bv, 0.f), fmaxf(a1.y * unscale + bv, 0.f));
                if (skip) { const float4 q = sk[m % SKN][i]; v.x *= q.x; v.y *= q.y; v.z *= q.z; v.w *= q.w; }
                *reinterpret_cast<float4*>(out + o) = v;
                vmax[dy] = fmaxf(fmaxf(vmax[dy], fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
            vrow[0][0] = fmaxf(vrow[0][0], second ? 0.f : vmax[0]); vrow[0][1] = fmaxf(vrow[0][1], second ? 0.f : vmax[1]);
            vrow[1][0] = fmaxf(vrow[1][0], second ? vmax[0] : 0.f); vrow[1][1] = fmaxf(vrow[1][1], second ? vmax[1] : 0.f);
            __builtin_amdgcn_wave_barrier();
        }
        if (amax_slots) {
            ac_amax_commit(vrow[0][0], amax_slots + 2 * row_a);
            ac_amax_commit(vrow[0][1], amax_slots + 2 * row_a + 1);
            if (row_b != row_a) {
                ac_amax_commit(vrow[1][0], amax_slots + 2 * row_b);
                ac_amax_commit(vrow[1][1], amax_slots + 2 * row_b + 1);
            }
        }
    }
}

static int rs_launch(int mode, ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, const float* skip, float* out,
                     int B, int Ci, int Co, int H, int W, float w_unscale, const float* in_amax, float* out_amax, void* stream) {
    AC_REQUIRE(ctx && x && w_packed && bias && out, "null pointer");
    AC_REQUIRE(B > 0 && Ci > 0 && Co > 0 && H > 0 && W > 0, "positive sizes");
    long long P;
    int K, N;
    if (mode == 0) {
        AC_REQUIRE(H % 2 == 0 && W % 4 == 0 && Ci % 8 == 0, "down: H even, W % 4 == 0, C_in % 8 == 0");
        P = (long long)(H / 2) * (W / 2); K = 4 * Ci; N = Co;
    } else {
        AC_REQUIRE(W % 4 == 0 && (4 * Co) % RS_BN == 0, "up: W % 4 == 0, 4 * C_out % 96 == 0");
        P = (long long)H * W; K = Ci; N = 4 * Co;
    }
    AC_REQUIRE(P % RS_BM == 0, "pixels per image % 128 == 0");
    AC_REQUIRE(mode != 0 || !(in_amax || out_amax) || (W / 2) % 4 == 0, "down with amax: (W / 2) % 4 == 0 (a float4 of outputs stays in one row)");
    // the staging thread scales its four consecutive pixels with ONE row's maximum: W % 4 == 0 (required for `up` above) keeps a quad in one row
    AC_REQUIRE(mode != 1 || !(in_amax || out_amax) || (W >= 64 && W % 4 == 0), "up with amax: W >= 64 and W % 4 == 0 (a staged pixel quad lies in one row, a wave's 64 pixels in at most two)");
    AC_REQUIRE((long long)H * W * 4 < (1LL << 31), "plane too large");
    const int n_stage = (K + RS_BK - 1) / RS_BK, n_nblk = (N + RS_BN - 1) / RS_BN;
    const long long n_mblk = (long long)B * (P / RS_BM);
    const long long nblk = n_mblk * n_nblk;
    AC_REQUIRE(nblk < (1LL << 31) - 8, "grid too large");
    dim3 grid((unsigned)nblk), block(256);
    if (mode == 0)
        hipLaunchKernelGGL(k_resample2x_f16x3<0>, grid, block, 0, (hipStream_t)stream, x, (const f16x8*)w_packed, bias, skip, out,
                           Ci, Co, H, W, n_stage, n_nblk, (int)n_mblk, w_unscale, in_amax, out_amax);
    else
        hipLaunchKernelGGL(k_resample2x_f16x3<1>, grid, block, 0, (hipStream_t)stream, x, (const f16x8*)w_packed, bias, skip, out,
                           Ci, Co, H, W, n_stage, n_nblk, (int)n_mblk, w_unscale, in_amax, out_amax);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_down2x_f16x3(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* out, int B, int C_in,
                                int C_out, int H, int W, float w_unscale, const float* in_amax, float* out_amax, void* stream) {
    return rs_launch(0, ctx, x, w_packed, bias, nullptr, out, B, C_in, C_out, H, W, w_unscale, in_amax, out_amax, stream);
}

extern "C" int ac_up2x_f16x3(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, const float* skip, float* out,
                              int B, int C_in, int C_out, int H, int W, float w_unscale, const float* in_amax, float* out_amax,
                              void* stream) {
    return rs_launch(1, ctx, x, w_packed, bias, skip, out, B, C_in, C_out, H, W, w_unscale, in_amax, out_amax, stream);
}
