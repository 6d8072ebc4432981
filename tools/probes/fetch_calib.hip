// Known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE (and the raw TCC_EA0_* request counters behind them) on gfx950,
// one launch per access shape (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your
// own access pattern").  Every kernel touches a 1 GiB buffer (4x the Infinity Cache) exactly once; the bytes each one must move are
// printed, the counters come from `rocprofv3 --pmc ... -- tools/probes/build/fetch_calib` (tools/run_r04.sh calib).
//   rd4 / rd8 / rd16      contiguous streaming loads, 4 / 8 / 16 B per lane (global_load_dword / dwordx2 / dwordx4)
//   rd16_lds              the same 16 B per lane by LDS-DMA (global_load_lds_dwordx4)
//   tail16 / tail4        ONE 16-B (4-B) load at the END of every 128-B line: what a partial-line miss fetches (128, 64 or 32 B?)
//   seg160                the 3x3 conv's patch rows: 160-B segments starting 16 B in front of every SECOND 128-B line
//                         (neighbouring tiles left out, so no other reader shares the two partial lines)
//   rd8_f64x2win          two overlapping 8-B streams (x[i] and x[i + 441]) - k_moving_meansq_db's shape
//   wr4 / wr8 / wr16      contiguous streaming stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T> __global__ void k_rd(const T* __restrict__ p, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) { T v = p[i]; acc += *reinterpret_cast<const float*>(&v); }
    if (acc == 123.456f) *sink = acc;
}
__global__ void k_rd16_lds(const float4* __restrict__ p, size_t n, float* sink) {
    __shared__ float4 s[4][64 * 4];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int wave = threadIdx.x >> 6;
    int slot = 0;
    for (; i < n; i += stride) {
        __builtin_amdgcn_global_load_lds(p + i, &s[wave][64 * slot], 16, 0, 0);
        slot = (slot + 1) & 3;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (s[wave][threadIdx.x & 63].x == 123.456f) *sink = 1.f;
}
// one load of W bytes at byte (line * 128 + 128 - W) of every 128-byte line
template <typename T> __global__ void k_tail(const unsigned char* __restrict__ p, size_t n_lines, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n_lines; i += stride) { T v = *reinterpret_cast<const T*>(p + i * 128 + 128 - sizeof(T)); acc += *reinterpret_cast<const float*>(&v); }
    if (acc == 123.456f) *sink = acc;
}
// lane -> (segment, quad 0..9): 16 B at byte 256 * seg - 16 + 16 * quad (+ 256 so that segment 0 stays inside the buffer); 60 of 64 lanes live
__global__ void k_seg160(const unsigned char* __restrict__ p, size_t n_seg, float* sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    for (size_t s0 = wave * 6; s0 < n_seg; s0 += n_waves * 6) {
        const size_t seg = s0 + lane / 10;
        if (lane < 60 && seg < n_seg) {
            float4 v = *reinterpret_cast<const float4*>(p + 256 + seg * 256 - 16 + 16 * (lane % 10));
            acc += v.x;
        }
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ void k_rd8_win(const double* __restrict__ p, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i + 441 < n; i += stride) acc += p[i] - p[i + 441];
    if (acc == 123.456) *sink = (float)acc;
}
template <typename T> __global__ void k_wr(T* __restrict__ p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    T v; for (unsigned b = 0; b < sizeof(T) / 4; ++b) reinterpret_cast<float*>(&v)[b] = 1.0f + b;
    for (; i < n; i += stride) p[i] = v;
}

int main() {
    const size_t BYTES = 1ull << 30;
    unsigned char* buf; float* sink;
    CHECK(hipMalloc(&buf, BYTES + 4096)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 0, BYTES + 4096));
    const int G = 256 * 8, T = 256;
    auto flush = [&]() { CHECK(hipDeviceSynchronize()); };
    printf("kernel,expected_bytes\n");
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_rd<float>, dim3(G), dim3(T), 0, 0, (const float*)buf, BYTES / 4, sink); flush();
        hipLaunchKernelGGL(k_rd<float2>, dim3(G), dim3(T), 0, 0, (const float2*)buf, BYTES / 8, sink); flush();
        hipLaunchKernelGGL(k_rd<float4>, dim3(G), dim3(T), 0, 0, (const float4*)buf, BYTES / 16, sink); flush();
        hipLaunchKernelGGL(k_rd16_lds, dim3(G), dim3(T), 0, 0, (const float4*)buf, BYTES / 16, sink); flush();
        hipLaunchKernelGGL(k_tail<float4>, dim3(G), dim3(T), 0, 0, buf, BYTES / 128, sink); flush();
        hipLaunchKernelGGL(k_tail<float>, dim3(G), dim3(T), 0, 0, buf, BYTES / 128, sink); flush();
        hipLaunchKernelGGL(k_seg160, dim3(G), dim3(T), 0, 0, buf, BYTES / 256 - 2, sink); flush();
        hipLaunchKernelGGL(k_rd8_win, dim3(G), dim3(T), 0, 0, (const double*)buf, BYTES / 8, sink); flush();
        hipLaunchKernelGGL(k_wr<float>, dim3(G), dim3(T), 0, 0, (float*)buf, BYTES / 4); flush();
        hipLaunchKernelGGL(k_wr<float2>, dim3(G), dim3(T), 0, 0, (float2*)buf, BYTES / 8); flush();
        hipLaunchKernelGGL(k_wr<float4>, dim3(G), dim3(T), 0, 0, (float4*)buf, BYTES / 16); flush();
    }
    printf("k_rd<float>,%zu\nk_rd<float2>,%zu\nk_rd<float4>,%zu\nk_rd16_lds,%zu\n", BYTES, BYTES, BYTES, BYTES);
    printf("k_tail<float4>,%zu (useful; %zu if whole lines are fetched)\nk_tail<float>,%zu (useful; %zu if whole lines)\n", BYTES / 8, BYTES, BYTES / 32, BYTES);
    printf("k_seg160,%zu (useful 160 B per 256; %zu if 32-B sectors, %zu if 64-B, %zu if whole lines)\n", (BYTES / 256 - 2) * 160, (BYTES / 256 - 2) * 192, (BYTES / 256 - 2) * 256, (BYTES / 256 - 2) * 384);
    printf("k_rd8_win,%zu\nk_wr<float>,%zu\nk_wr<float2>,%zu\nk_wr<float4>,%zu\n", BYTES, BYTES, BYTES, BYTES);
    CHECK(hipFree(buf)); CHECK(hipFree(sink));
    return 0;
}
