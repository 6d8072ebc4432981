// Minimal reproducer: do packed-float32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) of ONE wave return wrong
// results while ANOTHER wave on the same SIMD streams matrix instructions?
// tools/coresidency_bisect.py reduced the "two processes on one GPU" corruption of round 2 to exactly this pair: the iSTFT kernel is
// wrong beside the 3x3 conv only while the conv issues its MFMAs, and only when the iSTFT is built WITH hipcc's SLP-packed float math.
//   hipcc -O3 --offload-arch=gfx950 pk_mfma_hazard.hip -o build/pk_mfma_hazard && build/pk_mfma_hazard [seconds per cell]
// Victim: every lane computes r = a * b, s = a + b, t = a * b + c on float2 operands with the packed instructions (inline asm) and,
// from the same registers, with the scalar instructions; the two must agree bit for bit (same IEEE operation, same rounding mode).
// Aggressor (second stream): a register-only MFMA loop, one of four matrix instructions.  Nothing touches LDS; the victim reads no
// memory inside its loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256, 2) void k_mfma(const h8* __restrict__ src, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x;
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(lane * 8 + i) & 4095]; b[i] = src[(lane * 8 + 4 + i) & 4095]; }
    float total = 0.f;
    if (KIND == 0) {                                     // v_mfma_f32_16x16x32_f16 (what the U-Net kernels issue)
        f4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (KIND == 1) {                              // v_mfma_f32_32x32x16_f16
        f16v acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + 1) & 3], acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) total += acc[i][j];
    } else if (KIND == 2) {                              // the legacy v_mfma_f32_16x16x16_f16
        h4 a4[4], b4[4];
        for (int i = 0; i < 4; ++i) { a4[i] = (h4){a[i][0], a[i][1], a[i][2], a[i][3]}; b4[i] = (h4){b[i][0], b[i][1], b[i][2], b[i][3]}; }
        f4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4[i & 3], b4[(i >> 1) & 3], acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {                                             // v_mfma_f32_16x16x4_f32 (the exact-float32 matrix instruction of k_tdf_small)
        f4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f4){0, 0, 0, 0};
        const float fa = (float)a[0][0], fb = (float)b[0][0];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa + i, fb - i, acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    if (total == 123.456f) sink[0] = total;
}

__device__ inline float hash_f(unsigned x) {             // a finite float of moderate magnitude from an integer
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return __uint_as_float(0x3f000000u | (x & 0x007fffffu)) * ((x >> 31) ? -1.5f : 1.25f);
}

// errors[0..2]: mismatches of pk_mul / pk_add / pk_fma; errors[3]: lanes-iterations checked (low 32 bits)
__global__ __launch_bounds__(256) void k_pk_victim(unsigned* __restrict__ errors, int iters, int use_packed) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    unsigned bad_mul = 0, bad_add = 0, bad_fma = 0;
    for (int it = 0; it < iters; ++it) {
        f2 a = (f2){hash_f(gid * 3u + it), hash_f(gid * 5u + it * 7u)};
        f2 b = (f2){hash_f(gid * 11u + it * 13u), hash_f(gid * 17u + it * 19u)};
        f2 c = (f2){hash_f(gid * 23u + it * 29u), hash_f(gid * 31u + it * 37u)};
        float m0, m1, s0, s1, t0, t1;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(a.x), "v"(b.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(a.y), "v"(b.y));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(a.x), "v"(b.x));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(a.y), "v"(b.y));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(a.x), "v"(b.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(a.y), "v"(b.y), "v"(c.y));
        f2 pm, ps, pt;
        if (use_packed) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pm) : "v"(a), "v"(b));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(ps) : "v"(a), "v"(b));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pt) : "v"(a), "v"(b), "v"(c));
        } else {                                          // control: the same comparison with nothing packed in the wave
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(pm.x) : "v"(a.x), "v"(b.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(pm.y) : "v"(a.y), "v"(b.y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ps.x) : "v"(a.x), "v"(b.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ps.y) : "v"(a.y), "v"(b.y));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(pt.x) : "v"(a.x), "v"(b.x), "v"(c.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(pt.y) : "v"(a.y), "v"(b.y), "v"(c.y));
        }
        bad_mul += (__float_as_uint(pm.x) != __float_as_uint(m0)) + (__float_as_uint(pm.y) != __float_as_uint(m1));
        bad_add += (__float_as_uint(ps.x) != __float_as_uint(s0)) + (__float_as_uint(ps.y) != __float_as_uint(s1));
        bad_fma += (__float_as_uint(pt.x) != __float_as_uint(t0)) + (__float_as_uint(pt.y) != __float_as_uint(t1));
    }
    if (bad_mul) atomicAdd(errors + 0, bad_mul);
    if (bad_add) atomicAdd(errors + 1, bad_add);
    if (bad_fma) atomicAdd(errors + 2, bad_fma);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    std::vector<_Float16> h(4096 * 8);
    srand(7);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.02f);
    h8* d; float* sink; unsigned* err;
    hipMalloc(&d, h.size() * 2); hipMalloc(&sink, 4); hipMalloc(&err, 16);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    const char* names[5] = {"no aggressor", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_16x16x4_f32"};
    printf("victim: 4096 workgroups x 256 lanes x 64 iterations x (mul, add, fma) on float2, packed vs scalar instruction, bit for bit\n");
    for (int packed = 1; packed >= 0; --packed)
        for (int kind = -1; kind < 4; ++kind) {
            hipMemset(err, 0, 16);
            long long launches = 0;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, sb);
            double elapsed = 0;
            while (elapsed < secs * 1e3) {
                for (int j = 0; j < 2 && kind >= 0; ++j) {
                    if (kind == 0) hipLaunchKernelGGL(k_mfma<0>, dim3(2048), dim3(256), 0, sa, d, sink, 512);
                    if (kind == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(2048), dim3(256), 0, sa, d, sink, 512);
                    if (kind == 2) hipLaunchKernelGGL(k_mfma<2>, dim3(2048), dim3(256), 0, sa, d, sink, 512);
                    if (kind == 3) hipLaunchKernelGGL(k_mfma<3>, dim3(2048), dim3(256), 0, sa, d, sink, 512);
                }
                for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(k_pk_victim, dim3(4096), dim3(256), 0, sb, err, 64, packed);
                launches += 4;
                hipStreamSynchronize(sa); hipStreamSynchronize(sb);
                hipEventRecord(e1, sb); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); elapsed = ms;
            }
            unsigned he[4]; hipMemcpy(he, err, 16, hipMemcpyDeviceToHost);
            const double checked = (double)launches * 4096 * 256 * 64 * 2;
            printf("%-7s victim beside %-26s: mismatches mul %u  add %u  fma %u   of %.3g results each (%lld launches)\n",
                   packed ? "PACKED" : "scalar", names[kind + 1], he[0], he[1], he[2], checked, launches);
            fflush(stdout);
        }
    return 0;
}
