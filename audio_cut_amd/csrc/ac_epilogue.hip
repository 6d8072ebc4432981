// The streaming ends of the TFC-TDF U-Net (the first / last 1x1 Conv nodes of the ONNX graph the reference runs at
// separation/backends.py:358).  Everything else that touches an activation tensor is fused into the MFMA kernels' loaders and
// epilogues (ac_conv*.hip, ac_gemm.hip, ac_tdf_small.hip, ac_resample.hip); there is no library (MIOpen / rocBLAS) path.
#include <algorithm>

#include "ac_common.h"

// -------------------------------------------------------------------------------------------------
// The graph's first (4 -> g) and last (g -> 4) 1x1 convolutions: pure streaming ops (192 FMAs per pixel against
// 208 bytes), so plain float32 FMAs on the vector ALUs at the store / load rate.  One thread = 4 consecutive pixels.
//   SMALL_IN  (C_in  <= 8): inputs in registers, loop over output channels (float4 store each)
//   !SMALL_IN (C_out <= 8): accumulators in registers, loop over input channels (float4 load each)
// out[b][co][p] = act(bias[co] + sum_ci w[co][ci] * x[b][ci][p])
// -------------------------------------------------------------------------------------------------
template <bool SMALL_IN, bool RELU>
__global__ __launch_bounds__(256) void k_conv1x1_small(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int Ci, int Co, int64_t P) {
    const int64_t quads = P / 4;
    const int b = blockIdx.y;
    const float4* xb = reinterpret_cast<const float4*>(x + (int64_t)b * Ci * P);
    float4* ob = reinterpret_cast<float4*>(out + (int64_t)b * Co * P);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (int64_t)gridDim.x * 256) {
        if (SMALL_IN) {
            float4 v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (c < Ci) ? xb[(int64_t)c * quads + i] : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int co = 0; co < Co; ++co) {
                const float bv = bias[co];
                float4 a = make_float4(bv, bv, bv, bv);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (c < Ci) {
                        const float wv = w[co * Ci + c];
                        a.x = fmaf(wv, v[c].x, a.x); a.y = fmaf(wv, v[c].y, a.y); a.z = fmaf(wv, v[c].z, a.z); a.w = fmaf(wv, v[c].w, a.w);
                    }
                }
                if (RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
                ob[(int64_t)co * quads + i] = a;
            }
        } else {
            float4 a[8];
#pragma unroll
            for (int co = 0; co < 8; ++co) { const float bv = (co < Co) ? bias[co] : 0.f; a[co] = make_float4(bv, bv, bv, bv); }
#pragma unroll 4
            for (int c = 0; c < Ci; ++c) {
                const float4 v = xb[(int64_t)c * quads + i];
#pragma unroll
                for (int co = 0; co < 8; ++co) {
                    if (co < Co) {
                        const float wv = w[co * Ci + c];
                        a[co].x = fmaf(wv, v.x, a[co].x); a[co].y = fmaf(wv, v.y, a[co].y);
                        a[co].z = fmaf(wv, v.z, a[co].z); a[co].w = fmaf(wv, v.w, a[co].w);
                    }
                }
            }
#pragma unroll
            for (int co = 0; co < 8; ++co) {
                if (co < Co) {
                    float4 r = a[co];
                    if (RELU) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
                    ob[(int64_t)co * quads + i] = r;
                }
            }
        }
    }
}

extern "C" int ac_conv1x1_small(ac_ctx* ctx, const float* x, const float* w, const float* bias, float* out, int B, int C_in,
                                 int C_out, long long P, int relu, void* stream) {
    AC_REQUIRE(ctx && x && w && bias && out, "null pointer");
    AC_REQUIRE(B > 0 && B <= 65535 && C_in > 0 && C_out > 0 && P > 0 && P % 4 == 0, "B <= 65535, P % 4 == 0");
    AC_REQUIRE(C_in <= 8 || C_out <= 8, "one of C_in, C_out must be <= 8");
    const int64_t quads = P / 4;
    unsigned gx = (unsigned)std::min<int64_t>((quads + 255) / 256, 65535);
    dim3 grid(gx < 1 ? 1 : gx, (unsigned)B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (C_in <= 8) {
        if (relu) hipLaunchKernelGGL((k_conv1x1_small<true, true>), grid, block, 0, st, x, w, bias, out, C_in, C_out, (int64_t)P);
        else      hipLaunchKernelGGL((k_conv1x1_small<true, false>), grid, block, 0, st, x, w, bias, out, C_in, C_out, (int64_t)P);
    } else {
        if (relu) hipLaunchKernelGGL((k_conv1x1_small<false, true>), grid, block, 0, st, x, w, bias, out, C_in, C_out, (int64_t)P);
        else      hipLaunchKernelGGL((k_conv1x1_small<false, false>), grid, block, 0, st, x, w, bias, out, C_in, C_out, (int64_t)P);
    }
    AC_LAUNCH_CHECK();
    return AC_OK;
}
