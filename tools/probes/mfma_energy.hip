// Where do the joules go under the package power cap?  Sustained issued-MFMA rate of v_mfma_f32_16x16x32_f16 streams that add,
// one at a time, the other ingredients of the conv / GEMM kernels: LDS fragment reads at the conv's ratio (14 ds_read_b128 per
// 36 MFMAs), the split's VALU work (~1.5 VALU per MFMA), and global loads (24 dwordx4 per 180 MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int LDS, int VALU, int GLD>
__global__ __launch_bounds__(256, 2) void k_mix(const h8* __restrict__ src, const float4* __restrict__ big, float* __restrict__ sink, int iters) {
    __shared__ h8 s[3072];                                   // 48 KB
    __shared__ float4 dma[4 * 16 * 64];                      // 64 KB landing zone of the LDS-DMA variants
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3072; i += 256) s[i] = src[i & 4095];
    __syncthreads();
    h8 a[3], b[4];
    for (int i = 0; i < 3; ++i) a[i] = src[(lane * 7 + i) & 4095];
    for (int i = 0; i < 4; ++i) b[i] = src[(lane * 5 + 16 + i) & 4095];
    f4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = (f4){0, 0, 0, 0};
    float v0 = (float)a[0][0], v1 = (float)b[0][1], vs = 0.f;
    float4 g = make_float4(0, 0, 0, 0);
    float4 pre[16];
    for (int j = 0; j < 16; ++j) pre[j] = make_float4(0, 0, 0, 0);
    const size_t gbase = (size_t)blockIdx.x * 256 + tid;
    for (int it = 0; it < iters; ++it) {
        if (LDS) {                                           // 14 fragment reads per 36 MFMAs: 6 "weight" + 8 "activation" fragments
            const int o = (it * 64 + lane) & 2047;
#pragma unroll
            for (int i = 0; i < 3; ++i) a[i] = s[o + 64 * i];
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = s[(o + 256 + 64 * i) & 3071];
            h8 a2[3], b2[4];
#pragma unroll
            for (int i = 0; i < 3; ++i) a2[i] = s[(o + 512 + 64 * i) & 3071];
#pragma unroll
            for (int i = 0; i < 4; ++i) b2[i] = s[(o + 768 + 64 * i) & 3071];
#pragma unroll
            for (int i = 0; i < 3; ++i) a[i] = a[i] + a2[i] * (_Float16)0.0009765625f;
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = b[i] + b2[i] * (_Float16)0.0009765625f;
        }
        if (GLD && (it % 5) == 4) {                          // consume what was requested four iterations (144 MFMAs) ago
#pragma unroll
            for (int j = 0; j < 16; ++j) { g.x += pre[j].x; g.y += pre[j].w; }
        }
        if (GLD == 1 && (it % 5) == 0) {                     // 16 coalesced 16-B loads per 180 MFMAs out of a 2 MB (L2-resident) region
#pragma unroll
            for (int j = 0; j < 16; ++j) pre[j] = big[(gbase + (size_t)it * 4096 + j * 256) & ((1u << 17) - 1)];
        }
        if (GLD == 3 && (it % 5) == 0) {                     // 8 register loads + 8 LDS-DMA loads (1 KB per wave-instruction), L2-resident
#pragma unroll
            for (int j = 0; j < 8; ++j) pre[j] = big[(gbase + (size_t)it * 4096 + j * 256) & ((1u << 17) - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_global_load_lds(big + ((gbase + (size_t)it * 4096 + (8 + j) * 256) & ((1u << 17) - 1)),
                                                 dma + ((tid >> 6) * 8 + j) * 64, 16, 0, 0);
        }
        if (GLD == 4 && (it % 5) == 0) {                     // 16 LDS-DMA loads, no register loads
#pragma unroll
            for (int j = 0; j < 16; ++j)
                __builtin_amdgcn_global_load_lds(big + ((gbase + (size_t)it * 4096 + j * 256) & ((1u << 17) - 1)),
                                                 dma + ((tid >> 6) * 16 + j) * 64, 16, 0, 0);
        }
        if (GLD == 2 && (it % 5) == 0) {                     // 16 coalesced loads, 4 of them streaming through 2 GiB (HBM), 12 L2-resident
#pragma unroll
            for (int j = 0; j < 16; ++j)
                pre[j] = (j & 3) ? big[(gbase + (size_t)it * 4096 + j * 256) & ((1u << 17) - 1)]
                                 : big[(gbase * 4 + (size_t)it * 8388608 + j * 64) & ((1u << 27) - 1)];
        }
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[m * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[q], acc[m * 4 + q], 0, 0, 0);
                acc[m * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(m + 1) % 3], b[q], acc[m * 4 + q], 0, 0, 0);
                acc[m * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[(q + 1) & 3], acc[m * 4 + q], 0, 0, 0);
                if (VALU && (q & 1)) {                       // ~1.5 VALU per MFMA: the hi/lo split of one value per 6 MFMAs
                    const float c = fminf(fmaxf(v0 + v1, -65504.f), 65504.f);
                    const _Float16 hv = (_Float16)c;
                    const _Float16 lv = (_Float16)(c - (float)hv);
                    vs += (float)lv; v0 = v1 * 1.0001f; v1 = c * 0.5f + (float)hv;
                }
            }
    }
    if (GLD >= 3) { __syncthreads(); g.x += dma[tid].x + dma[tid + 2048].y; }
    float total = vs + g.x + g.y;
    for (int i = 0; i < 12; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (total == 123.456f) sink[0] = total;
}

template <int LDS, int VALU, int GLD>
static void run(const char* name, const h8* d, const float4* big, float* sink, double secs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 2 * 8, iters = 800;
    const double flops = (double)grid * 4 * iters * 36 * 16384.0;
    double elapsed = 0; int launches = 0; float ms;
    while (elapsed < secs * 1e3) {
        hipEventRecord(e0);
        for (int j = 0; j < 4; ++j) hipLaunchKernelGGL((k_mix<LDS, VALU, GLD>), dim3(grid), dim3(256), 0, 0, d, big, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        elapsed += ms; launches += 4;
    }
    printf("%-34s %7.1f TFLOP/s issued (%.2f ms per launch)\n", name, launches * flops / (elapsed * 1e-3) / 1e12, elapsed / launches);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    const float amp = argc > 2 ? atof(argv[2]) : 2.0f;
    std::vector<_Float16> h(4096 * 8);
    srand(7);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * amp);
    h8* d; float* sink; float4* big;
    hipMalloc(&d, h.size() * 2); hipMalloc(&sink, 4); hipMalloc(&big, (size_t)(1u << 27) * 16);     // 2 GiB
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemset(big, 0, (size_t)(1u << 27) * 16);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0, 0>("MFMA only", d, big, sink, secs);
        run<1, 0, 0>("MFMA + LDS fragment reads", d, big, sink, secs);
        run<0, 1, 0>("MFMA + split VALU", d, big, sink, secs);
        run<0, 0, 1>("MFMA + L2-resident loads", d, big, sink, secs);
        run<0, 0, 3>("MFMA + 8 reg + 8 LDS-DMA loads (L2)", d, big, sink, secs);
        run<0, 0, 4>("MFMA + 16 LDS-DMA loads (L2)", d, big, sink, secs);
        run<0, 0, 2>("MFMA + HBM streaming loads", d, big, sink, secs);
        run<1, 1, 1>("MFMA + LDS + VALU + L2 loads", d, big, sink, secs);
        run<1, 1, 2>("MFMA + LDS + VALU + HBM loads", d, big, sink, secs);
    }
    return 0;
}
