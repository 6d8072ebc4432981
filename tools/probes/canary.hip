// Canary kernels for tools/sharing_probe_victims.py: WHICH state of a workgroup does a co-resident 3x3-conv workgroup of
// another process disturb?  Each kernel keeps one kind of state alive for a few tens of microseconds (the life of an iSTFT
// workgroup), re-checks it against a closed form and counts mismatches.  Geometry mirrors k_mdx_istft_frames: 256 threads,
// 48 KiB of LDS where LDS is the subject, ~40 VGPRs.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC canary.hip -o build/libcanary.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ inline unsigned mixu(unsigned a, unsigned b) { return (a * 2654435761u) ^ (b * 40503u + 0x9e3779b9u); }

// VGPR canary: 32 registers per lane hold mixu(gid, i); the wave sleeps, re-checks, sleeps ...
extern "C" __global__ __launch_bounds__(256) void k_canary_vgpr(unsigned* __restrict__ errors, int rounds) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    unsigned r[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { r[i] = mixu(gid, i); asm volatile("" : "+v"(r[i])); }
    unsigned bad = 0;
    for (int k = 0; k < rounds; ++k) {
        __builtin_amdgcn_s_sleep(32);
#pragma unroll
        for (int i = 0; i < 32; ++i) { asm volatile("" : "+v"(r[i])); bad += (r[i] != mixu(gid, i)); }
    }
    if (bad) atomicAdd(errors, bad);
}

// LDS canary: 48 KiB filled with mixu(block, index); barrier; sleep; every thread re-checks its 48 words; repeat.
extern "C" __global__ __launch_bounds__(256) void k_canary_lds(unsigned* __restrict__ errors, int rounds) {
    __shared__ unsigned s[12288];
    for (int i = threadIdx.x; i < 12288; i += 256) s[i] = mixu(blockIdx.x, i);
    __syncthreads();
    unsigned bad = 0;
    for (int k = 0; k < rounds; ++k) {
        __builtin_amdgcn_s_sleep(32);
        for (int i = threadIdx.x; i < 12288; i += 256) bad += (s[i] != mixu(blockIdx.x, i));
        __syncthreads();
    }
    if (bad) atomicAdd(errors + 1, bad);
}

// LDS read-modify-write canary (what the FFT does): every round each thread adds 1 to its 48 words (ds_read + ds_write through
// a barrier), at the end word i must be mixu(block, i) + rounds.
extern "C" __global__ __launch_bounds__(256) void k_canary_lds_rmw(unsigned* __restrict__ errors, int rounds) {
    __shared__ unsigned a[6144];
    __shared__ unsigned b[6144];
    for (int i = threadIdx.x; i < 6144; i += 256) a[i] = mixu(blockIdx.x, i);
    __syncthreads();
    unsigned* src = a; unsigned* dst = b;
    for (int k = 0; k < rounds; ++k) {
        for (int i = threadIdx.x; i < 6144; i += 256) dst[(i * 5 + 1) % 6144] = src[i] + 1u;      // a permutation of the indices (5 is coprime to 6144)
        __syncthreads();
        unsigned* t = src; src = dst; dst = t;
    }
    // undo the index walk: after `rounds` applications of p(i) = 5 i + 1 the word that started at i sits at p^rounds(i)
    unsigned bad = 0;
    for (int i = threadIdx.x; i < 6144; i += 256) {
        unsigned j = i;
        for (int k = 0; k < rounds; ++k) j = (j * 5 + 1) % 6144;
        bad += (src[j] != mixu(blockIdx.x, i) + (unsigned)rounds);
    }
    if (bad) atomicAdd(errors + 2, bad);
}

// L1 canary: a 24 KiB global table t[i] = mixu(7, i) re-read `rounds` times with the FFT's strided pattern (float2-sized loads).
extern "C" __global__ __launch_bounds__(256) void k_canary_l1(const uint2* __restrict__ table, unsigned* __restrict__ errors, int rounds) {
    unsigned bad = 0;
    for (int k = 0; k < rounds; ++k) {
        for (int j = threadIdx.x; j < 1024; j += 256) {
            const int i1 = (2 * j + 6 * k) % 3072, i2 = (4 * j + k) % 3072;
            const uint2 v1 = table[i1], v2 = table[i2];
            bad += (v1.x != mixu(7, 2 * i1)) + (v1.y != mixu(7, 2 * i1 + 1)) + (v2.x != mixu(7, 2 * i2)) + (v2.y != mixu(7, 2 * i2 + 1));
        }
        __builtin_amdgcn_s_sleep(8);
    }
    if (bad) atomicAdd(errors + 3, bad);
}

// The iSTFT's LDS traffic without its arithmetic: the six Stockham passes of fft3072_f32 (ac_mdx.hip) as pure data movement of
// 8-byte words (same index arithmetic, same barriers, same two 24 KiB arrays); afterwards word y must be the one that started at
// src_of[y] (the permutation is computed on the host).  `twiddle_loads`: also issue the passes' table loads and fold them into a
// checksum that must equal the closed form (so the loads are live but the data words stay pure).
extern "C" __global__ __launch_bounds__(256) void k_canary_fft_shuffle(const int* __restrict__ src_of, const uint2* __restrict__ table,
                                                                       unsigned* __restrict__ errors, int rounds, int twiddle_loads) {
    __shared__ uint2 s_a[3072];
    __shared__ uint2 s_b[3072];
    unsigned bad = 0;
    for (int rep = 0; rep < rounds; ++rep) {
        for (int m = threadIdx.x; m < 3072; m += 256) s_a[m] = make_uint2(mixu(blockIdx.x, m), mixu(blockIdx.x + 77u, m));
        __syncthreads();
        uint2* a = s_a; uint2* b = s_b;
        int Ns = 1;
        for (int pass = 0; pass < 5; ++pass) {
            for (int j = threadIdx.x; j < 768; j += 256) {
                const int k = j & (Ns - 1);
                const uint2 v0 = a[j], v1 = a[j + 768], v2 = a[j + 1536], v3 = a[j + 2304];
                if (twiddle_loads && Ns > 1) {
                    const int q = k * (3072 / (4 * Ns));
                    const int i1 = (2 * q) % 3072, i2 = (4 * q) % 3072, i3 = (6 * q) % 3072;
                    const uint2 t1 = table[i1], t2 = table[i2], t3 = table[i3];
                    bad += (t1.x != mixu(7, 2 * i1)) + (t2.y != mixu(7, 2 * i2 + 1)) + (t3.x != mixu(7, 2 * i3));
                }
                const int base = ((j - k) << 2) + k;
                b[base] = v0; b[base + Ns] = v1; b[base + 2 * Ns] = v2; b[base + 3 * Ns] = v3;
            }
            __syncthreads();
            uint2* t = a; a = b; b = t;
            Ns <<= 2;
        }
        for (int j = threadIdx.x; j < 1024; j += 256) {
            const uint2 v0 = a[j], v1 = a[j + 1024], v2 = a[j + 2048];
            b[j] = v0; b[j + 1024] = v1; b[j + 2048] = v2;
        }
        __syncthreads();
        for (int y = threadIdx.x; y < 3072; y += 256) {
            const int x = src_of[y];
            const uint2 v = b[y];
            bad += (v.x != mixu(blockIdx.x, x)) + (v.y != mixu(blockIdx.x + 77u, x));
        }
        __syncthreads();
    }
    if (bad) atomicAdd(errors + 2, bad);
}

extern "C" __global__ void k_canary_fill_table(uint2* table) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 3072) table[i] = make_uint2(mixu(7, 2 * i), mixu(7, 2 * i + 1));
}

// ---- the register-only MFMA aggressor and the packed-float32 victim of pk_mfma_hazard.hip, callable from the Python probes
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


// The operand forms hipcc's SLP pass actually emits in k_mdx_istft_frames (op_sel / op_sel_hi / neg_lo / neg_hi modifiers, an SGPR pair
// and inline constants as packed sources), each against the scalar instruction sequence of the same IEEE operations.
// errors8[0..7]: mismatches per form.
__global__ __launch_bounds__(256) void k_pk_forms(unsigned* __restrict__ errors8, int iters, float sx, float sy) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    unsigned bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f2 sv = (f2){sx, sy};                                  // wave-uniform kernel arguments: an SGPR pair
    for (int it = 0; it < iters; ++it) {
        f2 a = (f2){hash_f(gid * 3u + it), hash_f(gid * 5u + it * 7u)};
        f2 b = (f2){hash_f(gid * 11u + it * 13u), hash_f(gid * 17u + it * 19u)};
        f2 d; float r0, r1;
#define CHECK(slot) bad[slot] += (__float_as_uint(d.x) != __float_as_uint(r0)) + (__float_as_uint(d.y) != __float_as_uint(r1))
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(a.x), "v"(b.x)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r1) : "v"(a.y), "v"(b.x)); CHECK(0);
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(a.y), "v"(b.y)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r1) : "v"(a.x), "v"(b.y)); CHECK(1);
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(a.x), "v"(b.x)); asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(a.y), "v"(b.y)); CHECK(2);
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r0) : "v"(a.x), "v"(b.y)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(r1) : "v"(a.y), "v"(b.x)); CHECK(3);
        asm volatile("v_pk_add_f32 %0, %1, 0 neg_lo:[1,1] neg_hi:[1,1]" : "=v"(d) : "v"(a));
        asm volatile("v_sub_f32 %0, 0, %1" : "=v"(r0) : "v"(a.x)); asm volatile("v_sub_f32 %0, 0, %1" : "=v"(r1) : "v"(a.y)); CHECK(4);
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "s"(sv));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(a.x), "v"(sx)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r1) : "v"(a.y), "v"(sy)); CHECK(5);
        asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(d) : "v"(a));
        r0 = a.y; r1 = a.x; CHECK(6);
        asm volatile("v_pk_mul_f32 %0, %1, -0.5 op_sel_hi:[1,0]" : "=v"(d) : "v"(a));
        asm volatile("v_mul_f32 %0, %1, -0.5" : "=v"(r0) : "v"(a.x)); asm volatile("v_mul_f32 %0, %1, -0.5" : "=v"(r1) : "v"(a.y)); CHECK(7);
#undef CHECK
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) if (bad[k]) atomicAdd(errors8 + k, bad[k]);
}

extern "C" int canary_launch(int which, unsigned* errors, const void* table, int blocks, int rounds, void* stream, const void* aux) {
    hipStream_t st = (hipStream_t)stream;
    switch (which) {
        case 0: hipLaunchKernelGGL(k_canary_vgpr, dim3(blocks), dim3(256), 0, st, errors, rounds); break;
        case 1: hipLaunchKernelGGL(k_canary_lds, dim3(blocks), dim3(256), 0, st, errors, rounds); break;
        case 2: hipLaunchKernelGGL(k_canary_lds_rmw, dim3(blocks), dim3(256), 0, st, errors, rounds); break;
        case 3: hipLaunchKernelGGL(k_canary_l1, dim3(blocks), dim3(256), 0, st, (const uint2*)table, errors, rounds); break;
        case 4: hipLaunchKernelGGL(k_canary_fill_table, dim3(12), dim3(256), 0, st, (uint2*)table); break;
        case 5: hipLaunchKernelGGL(k_canary_fft_shuffle, dim3(blocks), dim3(256), 0, st, (const int*)aux, (const uint2*)table, errors, rounds, 0); break;
        case 6: hipLaunchKernelGGL(k_canary_fft_shuffle, dim3(blocks), dim3(256), 0, st, (const int*)aux, (const uint2*)table, errors, rounds, 1); break;
        case 7: hipLaunchKernelGGL(k_pk_victim, dim3(blocks), dim3(256), 0, st, errors, rounds, 1); break;
        case 9: hipLaunchKernelGGL(k_pk_forms, dim3(blocks), dim3(256), 0, st, errors, rounds, 0.8660254f, -1.3125f); break;      // errors: 8 counters
        case 8: hipLaunchKernelGGL(k_mfma<0>, dim3(blocks), dim3(256), 0, st, (const h8*)table, (float*)errors, rounds); break;      // table: >= 64 KiB of small f16 values
        default: return -1;
    }
    return (int)hipGetLastError();
}
