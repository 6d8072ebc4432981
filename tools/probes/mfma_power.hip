// Pure-MFMA streams under the package power cap: v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16, random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256, 2) void k_stream(const h8* __restrict__ src, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x;
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(lane * 8 + i) & 4095]; b[i] = src[(lane * 8 + 4 + i) & 4095]; }
    float total = 0.f;
    if (KIND == 0) {
        f4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (KIND == 2) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 a4[4], b4[4];
        for (int i = 0; i < 4; ++i) { a4[i] = (h4){a[i][0], a[i][1], a[i][2], a[i][3]}; b4[i] = (h4){b[i][0], b[i][1], b[i][2], b[i][3]}; }
        f4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4[i & 3], b4[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f16v acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + 1) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) total += acc[i][j];
    }
    if (total == 123.456f) sink[0] = total;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    std::vector<_Float16> h(4096 * 8);
    srand(7);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.02f);      // small: accumulators stay finite
    h8* d; float* sink;
    hipMalloc(&d, h.size() * 2); hipMalloc(&sink, 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 2 * 8, iters = 4096;                                        // 2 blocks/CU resident, 8 rounds
    for (int rep = 0; rep < 2; ++rep)
        for (int kind = 0; kind < 3; ++kind) {
            double flops_per_launch = (double)grid * 4 /*waves*/ * iters * (kind == 0 ? 8 * 16384.0 : kind == 2 ? 8 * 8192.0 : 4 * 32768.0);
            double elapsed = 0; int launches = 0; float ms;
            while (elapsed < secs * 1e3) {
                hipEventRecord(e0);
                for (int j = 0; j < 4; ++j) {
                    if (kind == 0) hipLaunchKernelGGL(k_stream<0>, dim3(grid), dim3(256), 0, 0, d, sink, iters);
                    else if (kind == 2) hipLaunchKernelGGL(k_stream<2>, dim3(grid), dim3(256), 0, 0, d, sink, iters);
                    else hipLaunchKernelGGL(k_stream<1>, dim3(grid), dim3(256), 0, 0, d, sink, iters);
                }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                elapsed += ms; launches += 4;
                if (launches == 4) printf("  kind %d first 4 launches: %.1f TF/s\n", kind, 4 * flops_per_launch / (ms * 1e-3) / 1e12);
            }
            printf("%s: sustained %.1f TF/s over %.1f s (%d launches)\n", kind == 0 ? "16x16x32_f16" : kind == 2 ? "16x16x16_f16" : "32x32x16_f16",
                   launches * flops_per_launch / (elapsed * 1e-3) / 1e12, elapsed * 1e-3, launches);
            fflush(stdout);
        }
    return 0;
}
