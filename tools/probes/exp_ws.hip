// EXPERIMENT: warp-specialised, persistent variant of k_conv3x3_f16x3.
//   512 threads: waves 0-3 consume (LDS fragments -> MFMA -> own output strip -> global), waves 4-7 produce (global f32 ->
//   f16 hi/lo -> LDS, weights -> LDS) one stage ahead into the other LDS buffer; ONE workgroup barrier per 16-channel stage.
//   One workgroup per CU walks its XCD's strip of (tile, C_out block) work items as one flat stage stream.
#include "ac_common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CV_TH 8
#define CV_TW 32
#define CV_PH (CV_TH + 2)
#define CV_CB 16
#define CV_PIX_STRIDE 16
#define CV_LW 40
#define CV_COB 48
#define CV_MT 3
#define CV_QUADS 10
#define CV_ITEMS (CV_PH * 3 * 16)
#define CV_ACT_ITERS ((CV_ITEMS + 255) / 256)
#define CV_WFRAGS (5 * 2 * CV_MT * 64)
#define CV_W_ITERS ((CV_WFRAGS + 255) / 256)
#define WS_ACT_BYTES (CV_PH * CV_LW * CV_PIX_STRIDE * 2)            // 12800 per plane (hi or lo)
#define WS_BUF_BYTES (2 * WS_ACT_BYTES + CV_WFRAGS * 16)            // 56320
#define WS_STRIP_FLOATS (CV_COB * 2 * CV_TW)                        // per consumer wave: [48 co][2 rows][32 px] = 3072 floats
#define WS_LDS_BYTES (2 * WS_BUF_BYTES + 4 * WS_STRIP_FLOATS * 4)   // 161792

__device__ inline unsigned short f16_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }
__device__ inline int cv_phys(int c) { return c ^ (((c >> 2) & 1) << 1); }

struct ws_tile { int b, cob, y0, x0; };
__device__ inline ws_tile ws_decode(int wi, int n_cob, int tiles_x, int tiles_y, int bw) {
    ws_tile t;
    t.cob = wi % n_cob;
    int r = wi / n_cob;
    t.b = r / (tiles_x * tiles_y);
    r -= t.b * (tiles_x * tiles_y);
    const int band = r / (tiles_y * bw);
    r -= band * (tiles_y * bw);
    t.y0 = (r / bw) * CV_TH;
    t.x0 = (band * bw + r % bw) * CV_TW;
    return t;
}

template <bool RELU>
__global__ __launch_bounds__(512, 1) void k_conv_ws(const float* __restrict__ x, const f16x8* __restrict__ wpk,
                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                    int C_in, int C_out, int H, int W, float w_unscale, int bw, int n_work) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int n_cob = C_out / CV_COB, n_cb = C_in / CV_CB;
    const int tiles_x = W / CV_TW, tiles_y = H / CV_TH;
    const size_t plane = (size_t)H * W;
    const int G = gridDim.x >> 3;
    const int strip = (n_work + 7) >> 3;
    const int strip_lo = (blockIdx.x & 7) * strip;
    const int strip_hi = min(n_work, strip_lo + strip);
    const int first = strip_lo + (blockIdx.x >> 3);
    if (first >= strip_hi) return;
    const int n_mine = (strip_hi - first + G - 1) / G;
    const int n_stage = n_mine * n_cb;

    if (producer) {
        const int ptid = tid - 256;
        int a_off[CV_ACT_ITERS], a_c4[CV_ACT_ITERS], a_flip[CV_ACT_ITERS], a_py[CV_ACT_ITERS], a_qd[CV_ACT_ITERS];
#pragma unroll
        for (int i = 0; i < CV_ACT_ITERS; ++i) {
            const int e = ptid + 256 * i;
            const int c4 = (e >> 2) & 3, ql = e & 3, rest = e >> 4;
            const int py = rest / 3, qd = (rest - py * 3) * 4 + ql;
            const bool live = e < CV_ITEMS && qd < CV_QUADS;
            a_c4[i] = live ? c4 : -1;
            a_flip[i] = (qd & 1) << 1;
            a_off[i] = (py * CV_LW + 4 * qd) * CV_PIX_STRIDE + c4 * 4;
            a_py[i] = py; a_qd[i] = qd;
        }
        float4 pre_x[CV_ACT_ITERS][4];
        f16x8 pre_w[CV_W_ITERS];
        int l_item = 0, l_cb = 0;
        int l_src[CV_ACT_ITERS];
        const float* l_xb = x;
        const f16x8* l_wbase = wpk;
        auto loader_tile = [&](int item) {
            const ws_tile t = ws_decode(first + item * G, n_cob, tiles_x, tiles_y, bw);
            l_xb = x + (size_t)t.b * C_in * plane;
            l_wbase = wpk + (size_t)t.cob * n_cb * CV_WFRAGS;
#pragma unroll
            for (int i = 0; i < CV_ACT_ITERS; ++i) {
                const int gy = t.y0 + a_py[i] - 1, gx = t.x0 - 4 + 4 * a_qd[i];
                l_src[i] = (a_c4[i] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
            }
        };
        auto prefetch = [&]() {
            const f16x8* wcb = l_wbase + (size_t)l_cb * CV_WFRAGS;
#pragma unroll
            for (int i = 0; i < CV_W_ITERS; ++i) {
                const int e = ptid + 256 * i;
                if (e < CV_WFRAGS) pre_w[i] = wcb[e];
            }
#pragma unroll
            for (int i = 0; i < CV_ACT_ITERS; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ci = l_cb * CV_CB + (a_c4[i] < 0 ? 0 : a_c4[i]) * 4 + q;
                    pre_x[i][q] = (l_src[i] >= 0) ? *reinterpret_cast<const float4*>(l_xb + (size_t)ci * plane + l_src[i])
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (++l_cb == n_cb) { l_cb = 0; if (++l_item < n_mine) loader_tile(l_item); }
        };
        auto commit = [&](int buf) {
            unsigned char* base = s_raw + buf * WS_BUF_BYTES;
            unsigned short* s_hi = reinterpret_cast<unsigned short*>(base);
            unsigned short* s_lo = reinterpret_cast<unsigned short*>(base + WS_ACT_BYTES);
            f16x8* s_w = reinterpret_cast<f16x8*>(base + 2 * WS_ACT_BYTES);
#pragma unroll
            for (int i = 0; i < CV_W_ITERS; ++i) {
                const int e = ptid + 256 * i;
                if (e < CV_WFRAGS) s_w[e] = pre_w[i];
            }
#pragma unroll
            for (int i = 0; i < CV_ACT_ITERS; ++i) {
                if (a_c4[i] < 0) continue;
                const float* v4[4] = {&pre_x[i][0].x, &pre_x[i][1].x, &pre_x[i][2].x, &pre_x[i][3].x};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned short h4[4], l4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = fminf(fmaxf(v4[q][k], -65504.f), 65504.f);
                        const _Float16 hv = (_Float16)v;
                        h4[q] = f16_bits(hv);
                        l4[q] = f16_bits((_Float16)(v - (float)hv));
                    }
                    const int off = a_off[i] + (k ^ a_flip[i]) * CV_PIX_STRIDE;
                    *reinterpret_cast<uint2*>(&s_hi[off]) = make_uint2((unsigned)h4[0] | ((unsigned)h4[1] << 16), (unsigned)h4[2] | ((unsigned)h4[3] << 16));
                    *reinterpret_cast<uint2*>(&s_lo[off]) = make_uint2((unsigned)l4[0] | ((unsigned)l4[1] << 16), (unsigned)l4[2] | ((unsigned)l4[3] << 16));
                }
            }
        };
        loader_tile(0);
        prefetch();                 // stage 0 -> registers
        commit(0);                  // stage 0 -> LDS buffer 0
        if (n_stage > 1) prefetch();   // stage 1 -> registers
        __syncthreads();            // barrier #0: stage 0 visible
        for (int s = 0; s < n_stage; ++s) {
            if (s + 1 < n_stage) {
                commit((s + 1) & 1);               // stage s+1 into the other buffer while the consumers work on stage s
                if (s + 2 < n_stage) prefetch();   // stage s+2 -> registers
            }
            __syncthreads();
        }
    } else {
        f32x4 acc[CV_MT][4];
#pragma unroll
        for (int m = 0; m < CV_MT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[m][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int g = lane >> 4, px = lane & 15;
        const int ci_off = 8 * (g & 1);
        float* so = reinterpret_cast<float*>(s_raw + 2 * WS_BUF_BYTES) + wave * WS_STRIP_FLOATS;
        int c_item = 0, c_cb = 0;
        __syncthreads();            // barrier #0
        for (int s = 0; s < n_stage; ++s) {
            const unsigned char* base = s_raw + (s & 1) * WS_BUF_BYTES;
            const unsigned short* s_hi = reinterpret_cast<const unsigned short*>(base);
            const unsigned short* s_lo = reinterpret_cast<const unsigned short*>(base + WS_ACT_BYTES);
            const f16x8* s_w = reinterpret_cast<const f16x8*>(base + 2 * WS_ACT_BYTES);
            // fragments of pair p+1 are fetched from LDS before the 36 MFMAs of pair p are issued (register double buffer)
            f16x8 ahA[CV_MT], alA[CV_MT], bhA[4], blA[4], ahB[CV_MT], alB[CV_MT], bhB[4], blB[4];
            auto load_pair = [&](int pair, f16x8* ah, f16x8* al, f16x8* bh, f16x8* bl) {
                int tap = pair * 2 + (g >> 1);
                if (tap > 8) tap = 8;
                const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
                for (int m = 0; m < CV_MT; ++m) {
                    ah[m] = s_w[((pair * 2 + 0) * CV_MT + m) * 64 + lane];
                    al[m] = s_w[((pair * 2 + 1) * CV_MT + m) * 64 + lane];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ty = 2 * wave + (q >> 1), tx = (q & 1) * 16 + px;
                    const int off = ((ty + dy) * CV_LW + cv_phys(tx + dx + 3)) * CV_PIX_STRIDE + ci_off;
                    bh[q] = *reinterpret_cast<const f16x8*>(&s_hi[off]);
                    bl[q] = *reinterpret_cast<const f16x8*>(&s_lo[off]);
                }
            };
            auto mfma_pair = [&](const f16x8* ah, const f16x8* al, const f16x8* bh, const f16x8* bl) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int m = 0; m < CV_MT; ++m) {
                        acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[q], acc[m][q], 0, 0, 0);
                        acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[q], acc[m][q], 0, 0, 0);
                        acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[q], acc[m][q], 0, 0, 0);
                    }
            };
            load_pair(0, ahA, alA, bhA, blA);
            load_pair(1, ahB, alB, bhB, blB);
            mfma_pair(ahA, alA, bhA, blA);
            load_pair(2, ahA, alA, bhA, blA);
            mfma_pair(ahB, alB, bhB, blB);
            load_pair(3, ahB, alB, bhB, blB);
            mfma_pair(ahA, alA, bhA, blA);
            load_pair(4, ahA, alA, bhA, blA);
            mfma_pair(ahB, alB, bhB, blB);
            mfma_pair(ahA, alA, bhA, blA);
            if (++c_cb == n_cb) {
                // tile done: own strip [48 co][2 rows][32 px] (wave-private, LDS is in-order per wave) -> 128-byte row stores
                c_cb = 0;
                const ws_tile t = ws_decode(first + c_item * G, n_cob, tiles_x, tiles_y, bw);
                ++c_item;
#pragma unroll
                for (int m = 0; m < CV_MT; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = q >> 1, tx = (q & 1) * 16 + px;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int co = m * 16 + g * 4 + r;
                            float v = acc[m][q][r] * w_unscale + bias[t.cob * CV_COB + co];
                            if (RELU) v = fmaxf(v, 0.f);
                            so[(co * 2 + row) * CV_TW + tx] = v;
                        }
                        acc[m][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                __builtin_amdgcn_wave_barrier();
                float* ob = out + ((size_t)t.b * C_out + (size_t)t.cob * CV_COB) * plane + (size_t)(t.y0 + 2 * wave) * W + t.x0;
#pragma unroll
                for (int i = 0; i < (CV_COB * 2 * (CV_TW / 4)) / 64; ++i) {     // 768 float4 per wave / 64 lanes = 12
                    const int e = lane + 64 * i;
                    const int line = e >> 3, q4 = e & 7;          // line = co * 2 + row
                    const int co = line >> 1, row = line & 1;
                    const float4 v = *reinterpret_cast<const float4*>(&so[line * CV_TW + 4 * q4]);
                    *reinterpret_cast<float4*>(ob + (size_t)co * plane + (size_t)row * W + 4 * q4) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
        }
    }
}

extern "C" int exp_conv_ws(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* out, int B, int C_in,
                            int C_out, int H, int W, float w_unscale, int relu, void* stream) {
    AC_REQUIRE(ctx && x && w_packed && bias && out, "null pointer");
    AC_REQUIRE(C_in % CV_CB == 0 && C_out % CV_COB == 0 && H % CV_TH == 0 && W % CV_TW == 0, "shape");
    const long long n_work = (long long)B * (C_out / CV_COB) * (H / CV_TH) * (W / CV_TW);
    const int tiles_x = W / CV_TW;
    const int bw = tiles_x % 4 == 0 ? 4 : (tiles_x % 3 == 0 ? 3 : (tiles_x % 2 == 0 ? 2 : 1));
    long long nblk = ctx->n_cu;
    if (nblk > n_work) nblk = n_work;
    nblk = ((nblk + 7) / 8) * 8;
    static bool attr_set = false;
    if (!attr_set) {
        AC_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv_ws<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES));
        AC_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv_ws<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES));
        attr_set = true;
    }
    if (relu)
        hipLaunchKernelGGL(k_conv_ws<true>, dim3((unsigned)nblk), dim3(512), WS_LDS_BYTES, (hipStream_t)stream, x, (const f16x8*)w_packed, bias, out, C_in, C_out, H, W, w_unscale, bw, (int)n_work);
    else
        hipLaunchKernelGGL(k_conv_ws<false>, dim3((unsigned)nblk), dim3(512), WS_LDS_BYTES, (hipStream_t)stream, x, (const f16x8*)w_packed, bias, out, C_in, C_out, H, W, w_unscale, bw, (int)n_work);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
