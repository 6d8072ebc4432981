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
