// Fused elementwise epilogues of the TFC-TDF U-Net (the BatchNorm/ReLU/Mul/Add nodes that sit between the
// Conv / ConvTranspose / MatMul nodes of the ONNX graph the reference runs at separation/backends.py:358).
// The dense contractions stay in MIOpen / rocBLAS (MFMA); everything that touches an activation tensor
// between two contractions is ONE streaming pass here instead of 2-3 separate PyTorch elementwise kernels.
// Layout: NCHW float32, a "row" is one (batch, channel) plane of `inner` = H*W contiguous elements,
// channel = row % C.  inner % 4 == 0 (16-byte accesses).  HBM-bound: one read (+1 per extra operand) + one write.
#include "ac_common.h"

#define EP_THREADS 256

template <int MODE>
__global__ __launch_bounds__(EP_THREADS) void k_epilogue(float* __restrict__ x, const float* __restrict__ a,
                                                         const float* __restrict__ b, const float* __restrict__ other,
                                                         float* __restrict__ out, int C, int64_t inner4) {
    // MODE 0: x = relu(x + a[c])                         (conv bias + ReLU, BN folded into the conv)
    // MODE 1: x = relu(x + a[c]) * other                 (up-sampling path: bias + ReLU, then the multiplicative skip)
    // MODE 2: x = relu(x * a[c] + b[c])                  (TDF linear -> per-channel BN affine -> ReLU)
    // MODE 3: out = other + relu(x * a[c] + b[c])        (second TDF linear + affine + ReLU + residual add)
    const int64_t row = blockIdx.y;
    const int c = (int)(row % C);
    const float av = a[c];
    const float bv = (MODE >= 2) ? b[c] : 0.f;
    float4* xr = reinterpret_cast<float4*>(x) + row * inner4;
    const float4* orow = (MODE == 1 || MODE == 3) ? reinterpret_cast<const float4*>(other) + row * inner4 : nullptr;
    float4* outr = (MODE == 3) ? reinterpret_cast<float4*>(out) + row * inner4 : xr;
    for (int64_t i = (int64_t)blockIdx.x * EP_THREADS + threadIdx.x; i < inner4; i += (int64_t)gridDim.x * EP_THREADS) {
        float4 v = xr[i];
        if (MODE <= 1) {
            v.x = fmaxf(v.x + av, 0.f); v.y = fmaxf(v.y + av, 0.f); v.z = fmaxf(v.z + av, 0.f); v.w = fmaxf(v.w + av, 0.f);
        } else {
            v.x = fmaxf(v.x * av + bv, 0.f); v.y = fmaxf(v.y * av + bv, 0.f); v.z = fmaxf(v.z * av + bv, 0.f); v.w = fmaxf(v.w * av + bv, 0.f);
        }
        if (MODE == 1) { const float4 o = orow[i]; v.x *= o.x; v.y *= o.y; v.z *= o.z; v.w *= o.w; }
        if (MODE == 3) { const float4 o = orow[i]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        outr[i] = v;
    }
}

static int launch_epilogue(int mode, float* x, const float* a, const float* b, const float* other, float* out, int64_t rows, int C,
                           int64_t inner, void* stream) {
    AC_REQUIRE(x && a, "null pointer");
    AC_REQUIRE(rows > 0 && rows <= 65535 * 16LL && C > 0 && inner > 0 && inner % 4 == 0, "rows/C/inner (inner % 4 == 0)");
    AC_REQUIRE((((uintptr_t)x) & 15) == 0, "x must be 16-byte aligned");
    const int64_t inner4 = inner / 4;
    unsigned gx = (unsigned)((inner4 + EP_THREADS * 4 - 1) / (EP_THREADS * 4));   // ~4 float4 per thread
    if (gx < 1) gx = 1;
    // gridDim.y is limited to 65535: fold larger row counts into several launches of a whole number of
    // channel groups, so that channel = local_row % C stays right
    AC_REQUIRE(C <= 65535, "too many channels");
    const int64_t fold = (65535 / C) * (int64_t)C;
    for (int64_t r0 = 0; r0 < rows; r0 += fold) {
        const unsigned gy = (unsigned)((rows - r0) < fold ? (rows - r0) : fold);
        float* xo = x + r0 * inner;
        const float* oo = other ? other + r0 * inner : nullptr;
        float* outo = out ? out + r0 * inner : nullptr;
        dim3 grid(gx, gy), block(EP_THREADS);
        switch (mode) {
            case 0: hipLaunchKernelGGL(k_epilogue<0>, grid, block, 0, (hipStream_t)stream, xo, a, b, oo, outo, C, inner4); break;
            case 1: hipLaunchKernelGGL(k_epilogue<1>, grid, block, 0, (hipStream_t)stream, xo, a, b, oo, outo, C, inner4); break;
            case 2: hipLaunchKernelGGL(k_epilogue<2>, grid, block, 0, (hipStream_t)stream, xo, a, b, oo, outo, C, inner4); break;
            default: hipLaunchKernelGGL(k_epilogue<3>, grid, block, 0, (hipStream_t)stream, xo, a, b, oo, outo, C, inner4); break;
        }
        AC_LAUNCH_CHECK();
    }
    return AC_OK;
}

extern "C" int ac_bias_relu_inplace(ac_ctx* ctx, float* x, const float* bias, int64_t rows, int C, int64_t inner, void* stream) {
    AC_REQUIRE(ctx != nullptr, "ctx");
    return launch_epilogue(0, x, bias, nullptr, nullptr, nullptr, rows, C, inner, stream);
}
extern "C" int ac_bias_relu_mul_inplace(ac_ctx* ctx, float* x, const float* bias, const float* skip, int64_t rows, int C,
                                        int64_t inner, void* stream) {
    AC_REQUIRE(ctx != nullptr && skip != nullptr, "ctx/skip");
    return launch_epilogue(1, x, bias, nullptr, skip, nullptr, rows, C, inner, stream);
}
extern "C" int ac_affine_relu_inplace(ac_ctx* ctx, float* x, const float* scale, const float* shift, int64_t rows, int C,
                                      int64_t inner, void* stream) {
    AC_REQUIRE(ctx != nullptr && shift != nullptr, "ctx/shift");
    return launch_epilogue(2, x, scale, shift, nullptr, nullptr, rows, C, inner, stream);
}
extern "C" int ac_affine_relu_add(ac_ctx* ctx, const float* y, const float* scale, const float* shift, const float* residual,
                                  float* out, int64_t rows, int C, int64_t inner, void* stream) {
    AC_REQUIRE(ctx != nullptr && shift != nullptr && residual != nullptr && out != nullptr, "ctx/shift/residual/out");
    return launch_epilogue(3, const_cast<float*>(y), scale, shift, residual, out, rows, C, inner, stream);
}

// -------------------------------------------------------------------------------------------------
// 2x2 / stride-2 resampling layers of the U-Net as plain GEMMs (rocBLAS) + ONE streaming pass each:
//   down:  out = relu(W[co, (tap, ci)] @ space_to_depth(x) + b)   -> ac_space_to_depth2x, then GEMM + ac_bias_relu_inplace
//   up:    y4 = W[(tap, co), ci] @ x ; out[2y+dy, 2x+dx] = relu(y4[tap] + b) * skip  -> GEMM, then ac_depth_to_space2x_bias_relu_mul
// (MIOpen runs these as strided / transposed convolutions through im2col + GEMM + col2im or slow Winograd variants.)
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_space_to_depth2x(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W) {
    // x [B][C][H][W] -> out [B][4][C][H/2][W/2], tap = dy*2 + dx.  One thread: 4 consecutive input pixels of one row.
    const int W2 = W >> 1, H2 = H >> 1;
    const int64_t plane_in = (int64_t)H * W, plane_out = (int64_t)H2 * W2;
    const int64_t bc = blockIdx.y;                       // b*C + c
    const int64_t b = bc / C, c = bc % C;
    const float* src = x + bc * plane_in;
    float* dst = out + b * 4 * (int64_t)C * plane_out + c * plane_out;
    const int64_t quads = plane_in / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        const int64_t p = i * 4;
        const int yy = (int)(p / W), xx = (int)(p - (int64_t)yy * W);          // xx % 4 == 0
        const int dy = yy & 1;
        const int64_t o = (int64_t)(yy >> 1) * W2 + (xx >> 1);
        float* d0 = dst + (int64_t)(dy * 2 + 0) * C * plane_out + o;          // dx = 0 plane
        float* d1 = dst + (int64_t)(dy * 2 + 1) * C * plane_out + o;          // dx = 1 plane
        *reinterpret_cast<float2*>(d0) = make_float2(v.x, v.z);
        *reinterpret_cast<float2*>(d1) = make_float2(v.y, v.w);
    }
}

extern "C" int ac_space_to_depth2x(ac_ctx* ctx, const float* x, float* out, int B, int C, int H, int W, void* stream) {
    AC_REQUIRE(ctx && x && out, "null pointer");
    AC_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 4 == 0, "H even, W % 4 == 0");
    AC_REQUIRE((int64_t)B * C <= 65535, "B*C too large");
    const int64_t quads = (int64_t)H * W / 4;
    unsigned gx = (unsigned)((quads + 256 * 4 - 1) / (256 * 4));
    hipLaunchKernelGGL(k_space_to_depth2x, dim3(gx < 1 ? 1 : gx, (unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, x, out, C, H, W);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

__global__ __launch_bounds__(256) void k_depth_to_space2x(const float* __restrict__ y4, const float* __restrict__ bias,
                                                          const float* __restrict__ skip, float* __restrict__ out, int C, int H, int W) {
    // y4 [B][4][C][H][W] (tap = dy*2 + dx) -> out [B][C][2H][2W] = relu(y4 + bias[c]) * skip.  One thread: 4 output pixels of a row.
    const int W2 = 2 * W, H2 = 2 * H;
    const int64_t plane_in = (int64_t)H * W, plane_out = (int64_t)H2 * W2;
    const int64_t bc = blockIdx.y;
    const int64_t b = bc / C, c = bc % C;
    const float bv = bias[c];
    const float* src = y4 + b * 4 * (int64_t)C * plane_in + c * plane_in;
    const float* sk = skip ? skip + bc * plane_out : nullptr;
    float* dst = out + bc * plane_out;
    const int64_t quads = plane_out / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i * 4;
        const int Y = (int)(p / W2), X = (int)(p - (int64_t)Y * W2);          // X % 4 == 0
        const int dy = Y & 1;
        const int64_t o = (int64_t)(Y >> 1) * W + (X >> 1);
        const float2 a = *reinterpret_cast<const float2*>(src + (int64_t)(dy * 2 + 0) * C * plane_in + o);   // dx = 0: outputs X, X+2
        const float2 d = *reinterpret_cast<const float2*>(src + (int64_t)(dy * 2 + 1) * C * plane_in + o);   // dx = 1: outputs X+1, X+3
        float4 v = make_float4(fmaxf(a.x + bv, 0.f), fmaxf(d.x + bv, 0.f), fmaxf(a.y + bv, 0.f), fmaxf(d.y + bv, 0.f));
        if (sk) { const float4 s = reinterpret_cast<const float4*>(sk)[i]; v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w; }
        reinterpret_cast<float4*>(dst)[i] = v;
    }
}

extern "C" int ac_depth_to_space2x_bias_relu_mul(ac_ctx* ctx, const float* y4, const float* bias, const float* skip, float* out,
                                                 int B, int C, int H, int W, void* stream) {
    AC_REQUIRE(ctx && y4 && bias && out, "null pointer");
    AC_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && W % 2 == 0, "W even");
    AC_REQUIRE((int64_t)B * C <= 65535, "B*C too large");
    const int64_t quads = (int64_t)H * W;                 // (2H * 2W) / 4
    unsigned gx = (unsigned)((quads + 256 * 4 - 1) / (256 * 4));
    hipLaunchKernelGGL(k_depth_to_space2x, dim3(gx < 1 ? 1 : gx, (unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, y4, bias, skip, out, C, H, W);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

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
