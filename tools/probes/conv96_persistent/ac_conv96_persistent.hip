// 3x3 convolution, 96 output channels per workgroup: the wide-tile sibling of ac_conv.hip for layers with C_in % 32 == 0 and
// C_out % 96 == 0 (levels 1, 3, 5 of the U-Net: C = 96, 192, 288).  Same arithmetic (3-term float16 split, float32 accumulate in
// v_mfma_f32_16x16x32_f16), same 8 x 32 pixel tile and 4 waves, but every staged activation byte and every activation fragment
// read from LDS now feeds twice as many MFMAs - under the package power cap the loads / LDS reads / VALU split around the
// MFMAs are what the time goes to.
//
// LDS is what bounds a workgroup (two per CU), so K is walked in stages of 8 input channels:
//   patch   [10 rows][50-pixel row stride][8 ch] f16, hi and lo: 2 x 8,000 B      (40 columns staged, stride 50: see below)
//   weights two DMA-filled buffers: even stages 24 KB (two k-steps x (hi, lo) x 6 row tiles x 1 KB), odd stages 36 KB
//   total   77,440 B  -> 154,880 B for two workgroups
// A k-step of 32 is 4 taps x 8 channels (lane group g carries tap 4 ks + g).  Taps 0..7 are two k-steps per stage; tap 8
// of FOUR consecutive stages shares one k-step: stage cb loads its tap-8 fragments into lane group cb & 3 of registers that
// live across the stages, and the stage with cb & 3 == 3 issues it (its weights sit behind that stage's own two k-steps in
// the odd buffer).  32 channels therefore cost 9 k-steps, none padded.
// Bank layout: a pixel is 16 B, so the 16 pixels of one tap are 256 contiguous bytes = all 64 banks once.  ds_read_b128 serves
// lanes {0-3,12-15,20-27} together: two lane groups whose taps sit in one row differ by one pixel and never collide; the
// pairs that straddle rows (taps 2|3) need the row stride to be 2 pixels mod 16 -> 50 pixels.
#include "ac_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define W9_TH 8
#define W9_PH (W9_TH + 2)
#define W9_CB 8                  // input channels per stage
// Tile geometry by GX = 16-pixel groups per tile row (two tile rows per wave): 8 x 32 pixels (GX 2: the 96-channel tile and, until round 4,
// the 48-channel one), 8 x 48 (GX 3) and 8 x 64 (GX 4) for the 48-channel tile - the dual of what the 96-channel tile did for C % 96 == 0:
// a staged weight byte and an A fragment read feed 1.5x / 2x the MFMAs, the halo falls from 1.56x to 1.46x / 1.41x of the tile.
template <int GX> struct W9Geo {
    static constexpr int TW = 16 * GX;                 // tile width in pixels
    static constexpr int NQ = 2 * GX;                  // 16-pixel groups per wave
    static constexpr int LW = TW + 8;                  // staged columns per patch row: the aligned float4 of a row, x0 - 4 .. x0 + TW + 3
    static constexpr int NQD = LW / 4;                 // float4 per staged row
    static constexpr int RS = LW + 10;                 // LDS row stride in pixels: 50 / 66 / 82, all 2 (mod 16) - see the bank note above
    static constexpr int PATCH_BYTES = 2 * W9_PH * RS * W9_CB * 2;
    static constexpr int OUT_STRIDE = TW + 4;
    static constexpr int NR = (2 * W9_PH * NQD + 255) / 256;      // staging rounds: (row, float4, channel quad) slots over 256 threads
    static constexpr int QH = NQ == 8 ? 4 : NQ;        // B fragments held at once (8 groups x hi/lo would be 64 VGPRs beside the 64 of the carried tap 8)
};
#ifndef W9_PROBE
#define W9_PROBE 0                   // bit mask of ablations for the probe builds of tools/sharing_probe_*.py (never the product):
#endif                               // 1 weights first in LDS, 2 no LDS-DMA, 4 no MFMA, 8 no activation staging, 0x10 no shared tap-8 step,
                                     // 0x20 no epilogue, 0x40 no K loop, 0x80 no activation loads, 0x100 weights fetched for stage 0 only (timing probe: what the
                                     // per-stage weight DMA costs, i.e. what LDS-resident weights could win)
#ifndef W9_PIPE
#define W9_PIPE 1                    // software-pipelined fragment reads in the K loop (0: the compiler's schedule, kept for A/B runs)
#endif
#ifndef W9_ASM_DMA_ALL
#define W9_ASM_DMA_ALL 0             // 1: the hidden (inline assembly) weight DMA in the 48-channel / row-exact variants too (A/B switch)
#endif
#ifndef W9_PERSIST
#define W9_PERSIST 1                 // 0: one tile per workgroup on every layer (A/B switch)
#endif
#define W9_MAXT 256                  // tiles a workgroup may walk (its table in LDS)
#ifndef W9_PERSIST_MIN
#define W9_PERSIST_MIN 4             // tiles per resident workgroup from which a layer is walked persistently
#endif
#ifndef W9_S8_OCC
#define W9_S8_OCC 3                  // workgroups per CU of the 48-channel variant
#endif
#ifndef W9_FIRST_OCC
#define W9_FIRST_OCC 3               // workgroups per CU of the fused first conv
#endif

// w9_dma16 = ac_lds_dma16 (ac_common.h): the LDS-DMA issued from inline assembly, hidden from the compiler's wait-count pass.
#define w9_dma16 ac_lds_dma16

__device__ __forceinline__ size_t w9_opaque(size_t v) {           // a wave-uniform value the optimiser must treat as new (stays in scalar registers)
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return ((size_t)hi << 32) | lo;
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a release / acquire fence over ALL address spaces: the compiler puts
// s_waitcnt vmcnt(0) in front of it, and in the persistent walk that would make the first stage of a tile wait for the previous tile's
// output stores to drain.  Global data this kernel produces is never read by it; what it reads through vector memory (patch loads, weight
// DMA) is waited for by counted s_waitcnt vmcnt where it is consumed.
__device__ __forceinline__ void w9_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14): vector memory operations return in
// order, so this waits for everything issued before the N youngest
template <int N> __device__ __forceinline__ void w9_wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }
// row maximum of an output row: one atomic per wave, ALWAYS issued (a maximum with 0 changes nothing): the number of vector memory
// operations behind the next tile's prefetch is then a compile-time constant (w9_wait_vm)
__device__ __forceinline__ void w9_amax_commit(float m, float* __restrict__ slot, int lane) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, AC_WAVE));
    if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(m));
}
__device__ inline unsigned short w9_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }

#if W9_PROBE & 4                      // probe build: the operands stay live, the matrix instruction is not issued
__device__ __forceinline__ f32x4 w9_mfma(f16x8 a, f16x8 b, f32x4 c) { asm volatile("" :: "v"(a), "v"(b)); return c; }
#else
// a = weight fragment, b = activation fragment everywhere below; the activation fragment is the instruction's A operand, so the accumulator is
// D[pixel][channel]: lane (g, n) holds pixels 4 g .. 4 g + 3 of its pixel group for channel n (the epilogue stores them as one float4)
__device__ __forceinline__ f32x4 w9_mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c, 0, 0, 0); }
#endif

// MT = 16-row output-channel tiles per workgroup: 6 (96 channels, two workgroups per CU) or 3 (48 channels).  GX: W9Geo.
// FIRST fuses the graph's first 1x1 convolution (C0 <= 4 spectrogram channels -> C_in, + bias + ReLU) into the loader: x is the
// [B][C0][H][W] spectrogram, a thread's four spectrogram float4 are loaded ONCE per tile and every stage's 8 channels are generated from
// them per staged pixel (s_first = [C_in][w1[0..3], b1] in LDS; float32 FMAs in ac_conv1x1_small's order, so the values are
// bit-identical to running that kernel first) - the C_in-channel tensor never touches HBM and the K loop has no activation loads.
// Every accumulator receives its products in the same order whatever MT / GX (k-step by k-step: ah*bl, al*bh, ah*bh), and the
// activation scale is a function of the tile's ROWS only, so all geometries produce bit-identical outputs.
//
// PERSISTENT WALK (round 4).  A workgroup is a chain of latencies - row maxima, first patch loads (HBM), six to thirty-six stages of
// {barrier, split + stage, barrier, MFMAs}, stores, drain, the next workgroup's launch - and with two or three workgroups per CU
// nothing else hides them: ablation builds of the kernel are ADDITIVE (profiles/r04s: no MFMA -0.73 ms, no activation loads -1.0,
// no weight DMA -0.33 of 3.63 ms at C = 48; no unit is more than half busy).  So the grid is the resident set (CUs x workgroups per CU)
// and a workgroup walks its share of the work order; before it stores a tile it has the next tile's first activation loads, first
// weight DMA and row maxima in flight.  What makes that possible is the epilogue: the MFMA computes D[pixel][channel] (the
// activation fragment is its A operand), so a lane holds four consecutive pixels of one channel and stores them as one float4
// straight from its accumulators - no LDS tile aliasing the stage buffers, no barrier between the last MFMA of a tile and the first
// stage of the next.  Tile per tile the arithmetic is the non-persistent kernel's: outputs are bit-identical (tools/conv_pf_bench.py).
template <int MT, int GX, int OCC, bool RELU, bool FIRST>
__global__ __launch_bounds__(256, OCC) void k_conv3x3_f16x3_w96(const float* __restrict__ x, const f16x8* __restrict__ wpk,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int C_in, int C_out, int H, int W, float w_unscale, int bw, int n_work,
                                                              const float* __restrict__ in_amax, float* __restrict__ out_amax,
                                                              const float* __restrict__ w1, const float* __restrict__ b1, int C0,
                                                              float amax_gain, float amax_offs) {
    using G = W9Geo<GX>;
    constexpr int W9_MT = MT, NQ = G::NQ, RS = G::RS, NR = G::NR, QH = G::QH;
    constexpr int W9_COB = 16 * MT, W9_KFR = 2 * MT * 64;     // channels per workgroup; 16-byte fragments per k-step (hi, lo)
    constexpr int K_BYTES = G::PATCH_BYTES + (2 + 3) * W9_KFR * 16;
    static_assert(K_BYTES * OCC <= 160 * 1024 - OCC * 2048, "LDS per CU");
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[K_BYTES];
    __shared__ int s_ex2[2][W9_PH + 2];                  // log2 scale of each patch row, this tile's and the next one's (read by row-exact tiles only)
    __shared__ float s_bias[W9_COB];
    __shared__ int2 s_tiles[W9_MAXT];                    // this workgroup's walk
    __shared__ float s_first[FIRST ? 5 * 64 : 1];        // FIRST: [channel][w1[0..3], b1] of the fused 1x1 conv (C_in <= 64)
#if W9_PROBE & 1                      // probe build: weight buffers first, so every LDS-DMA lands 1 KiB aligned
    f16x8* s_w0 = reinterpret_cast<f16x8*>(s_raw);
    f16x8* s_w1 = s_w0 + 2 * W9_KFR;
    unsigned short* s_hi = reinterpret_cast<unsigned short*>(s_raw + 5 * W9_KFR * 16);
    unsigned short* s_lo = s_hi + W9_PH * RS * W9_CB;
#else
    unsigned short* s_hi = reinterpret_cast<unsigned short*>(s_raw);
    unsigned short* s_lo = s_hi + W9_PH * RS * W9_CB;
    f16x8* s_w0 = reinterpret_cast<f16x8*>(s_raw + G::PATCH_BYTES);           // even stages: 2 k-steps
    f16x8* s_w1 = s_w0 + 2 * W9_KFR;                                           // odd stages: 2 k-steps (+ the shared tap-8 step)
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // the wave index in a scalar register
    const int g = lane >> 4, px = lane & 15;
    const int n_cb = C_in / W9_CB;                                             // even: the stage that issues a shared step is always odd
    const size_t plane = (size_t)H * W;
    if (FIRST) {                         // ordered by the first stage's barriers
        for (int c = tid; c < C_in; c += 256) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s_first[c * 5 + j] = (j < C0) ? w1[c * C0 + j] : 0.f;
            s_first[c * 5 + 4] = b1[c];
        }
    }
    // Work order.  Consecutive block ids go round the 8 XCDs; XCD k owns the k-th eighth of the order below and its resident workgroups
    // walk it side by side (workgroup j of the XCD takes items j, j + n, j + 2 n, ...: at any time they hold neighbours): the channel
    // blocks of a tile, then the tiles of a BAND of bw tile columns row by row, then the next band, then the next item.  Vertical and
    // in-band horizontal halos are then found in that XCD's L2; what is re-fetched is the two partial 128-byte lines per patch row at a
    // band's edges (w9_band_width picks bw).  gridDim.x == n_work is the one-tile-per-workgroup launch (small layers).
    // The walk itself is decoded ONCE, by all threads, into a table in LDS (item, channel block | tile row, tile column): inside the
    // tile loop the divisions and their loop-invariant reciprocals would live in scalar registers the K loop has none to spare of.
    const int n_cob = C_out / W9_COB, tiles_x = W / G::TW, tiles_y = H / W9_TH;
    int n_my;
    {
        int wi, wi_end, stride;
        if (((n_work | (int)gridDim.x) & 7) == 0) {
            const int per = n_work >> 3;
            wi = (blockIdx.x & 7) * per + (blockIdx.x >> 3); wi_end = ((blockIdx.x & 7) + 1) * per; stride = gridDim.x >> 3;
        } else { wi = blockIdx.x; wi_end = n_work; stride = gridDim.x; }
        if (wi >= wi_end) return;
        n_my = __builtin_amdgcn_readfirstlane((wi_end - 1 - wi) / stride + 1);       // <= W9_MAXT: the host sizes the grid for it
        for (int i = tid; i < n_my; i += 256) {
            const int w_ = wi + i * stride;
            const int cob_ = w_ % n_cob;
            int t = w_ / n_cob;
            const int b_ = t / (tiles_x * tiles_y);
            t -= b_ * (tiles_x * tiles_y);
            const int band = t / (tiles_y * bw);
            t -= band * (tiles_y * bw);
            s_tiles[i] = make_int2((b_ << 8) | cob_, ((t / bw) << 16) | (band * bw + t % bw));
        }
        __syncthreads();
    }
    auto decode = [&](int i, int& cob_, int& b_, int& y0_, int& x0_) __attribute__((always_inline)) {
        const int2 t = s_tiles[i];
        const int t0 = __builtin_amdgcn_readfirstlane(t.x), t1 = __builtin_amdgcn_readfirstlane(t.y);
        cob_ = t0 & 255; b_ = t0 >> 8; y0_ = (t1 >> 16) * W9_TH; x0_ = (t1 & 0xffff) * G::TW;
    };
    // staging: slot tid + 256 r -> (row 0..9, float4 of the row, channel quad 0..1): one aligned float4 (4 pixels) of 4 channels each
    // out-of-image / idle slots load from a clamped in-bounds address and are zeroed when staged: no divergent branch around the
    // loads (6 fewer spilled registers in the 48-channel variant).  Activation loads run one stage ahead; two stages ahead (two
    // register sets, counted vmcnt) was measured 2-12 % SLOWER on every level (profiles/r02g_conv_prefetch_depth.log).
    // Global addresses are a wave-uniform base (scalar registers, advanced per stage) + ONE 32-bit byte offset per lane (a_ld, the weight
    // DMA's lane * 16, the epilogue's o_lane): the saddr form of global_load / global_store - no 64-bit address arithmetic in vector registers.
    int a_off[NR], a_c4[NR], a_row[NR];
    unsigned a_ld[NR];                   // byte offset of the slot's float4 within the stage's first channel plane (+ its channel quad's planes)
    bool a_live[NR], a_in[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int slot = tid + 256 * r;
        a_c4[r] = slot & 1;
        const int a_rest = slot >> 1;
        a_row[r] = a_rest / G::NQD;
        const int a_qd = a_rest - a_row[r] * G::NQD;
        a_live[r] = a_rest < W9_PH * G::NQD;
        a_off[r] = ((a_row[r] * RS + 4 * a_qd) * W9_CB + a_c4[r] * 4);        // u16 elements
    }
    auto set_tile = [&](int y0_, int x0_) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int a_qd = ((tid + 256 * r) >> 1) - a_row[r] * G::NQD;
            int a_src = -1;
            if (a_live[r]) {
                const int gy = y0_ + a_row[r] - 1, gx = x0_ - 4 + 4 * a_qd;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) a_src = gy * W + gx;
            }
            a_ld[r] = 4u * (unsigned)((FIRST ? 0 : a_c4[r] * 4) * (int)plane + (a_src >= 0 ? a_src : 0));
            a_in[r] = a_src >= 0;
        }
    };
    // B fragment of (k-step ks, pixel group q): u16 offset b_tap[ks] + a compile-time constant of q (it becomes the ds_read's offset field)
    const int b_lane = ((2 * wave) * RS + px + 3) * W9_CB;                     // patch column c is staged column c + 3
    int b_tap[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { const int tap = 4 * ks + g; b_tap[ks] = b_lane + ((tap / 3) * RS + (tap % 3)) * W9_CB; }

    // fragments per stage in the packed weights: [cob][cb][3 k-steps][2][MT][64]; the third k-step exists for cb & 3 == 3 only
    // (w9_opaque: the plane size re-enters as a value the compiler cannot see through, so the per-channel / per-row-tile address terms
    // are a few scalar multiplies where they are used instead of two dozen loop-invariant scalar registers kept - and spilled - across the K loop)
    float4 pre_x[NR][4];
    auto prefetch_x = [&](int cb, const float* xb) __attribute__((always_inline)) {
        const size_t pl = w9_opaque(plane);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#if W9_PROBE & 0x80
                pre_x[r][q] = make_float4(1.f + (FIRST ? q : cb * W9_CB + a_c4[r] * 4 + q), 2.f, 3.f, 4.f);
#else
                const char* base = reinterpret_cast<const char*>(xb + (size_t)(FIRST ? q : cb * W9_CB + q) * pl);      // wave-uniform
                pre_x[r][q] = (!FIRST || q < C0) ? *reinterpret_cast<const float4*>(base + a_ld[r]) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
            }
        }
    };
    constexpr bool PIPE_ANY = W9_PIPE && MT == 6 && GX == 2;
    const unsigned w_lane = 16u * (unsigned)lane;
    auto prefetch_w = [&](int cb, const f16x8* wbase) __attribute__((always_inline)) {
        if ((W9_PROBE & 0x100) && cb > 0) return;
        const f16x8* wcb = wbase + (size_t)cb * 3 * W9_KFR;
        f16x8* dst = (cb & 1) ? s_w1 : s_w0;
        const bool third = ((cb & 3) == 3) || cb == n_cb - 1;                       // this stage carries the shared tap-8 k-step
        const int n_inst = third ? 3 * W9_KFR / 64 : 2 * W9_KFR / 64;                // wave-instructions of 1 KB
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int inst = wave + 4 * i;
#if W9_PROBE & 2                      // probe build: no LDS-DMA - the fragments go through registers (slow; an aggressor only)
            if (inst < n_inst) dst[inst * 64 + lane] = wcb[inst * 64 + lane];
            (void)w_lane;
#else
            if (inst < n_inst) {
                // the 96-channel tile issues the DMA from inline assembly (ac_lds_dma16: hidden from the compiler's wait-count pass, its
                // counted lgkmcnt waits stay exact); in the 48-channel tile that was measured 4.6 % slower (profiles/r04o)
                const char* src = reinterpret_cast<const char*>(wcb + inst * 64) + w_lane;
                if constexpr (PIPE_ANY || W9_ASM_DMA_ALL) w9_dma16(src, dst + inst * 64);
                else __builtin_amdgcn_global_load_lds(src, dst + inst * 64, 16, 0, 0);
            }
#endif
        }
    };
    // log2 scale of each patch row y0 - 1 .. y0 + 8 of a tile from the producer's row maxima (ac_common.h), by wave-uniform loads.
    // FIRST: in_amax is max |spectrogram| per row; the tensor that is split is the generated relu(w1 x + b1), bounded per row by
    // amax * max_c sum_j |w1[c][j]| + max_c |b1[c]| (amax_gain, amax_offs from the host; 1 and 0 otherwise: exact)
    // (thread 0 leaves the ten values in LDS for the row-exact path, which picks rows by run-time indices; two buffers, because the next
    // tile's scales are computed while this tile's epilogue may still read its own)
    auto row_scales = [&](int b_, int y0_, int* s_ex_, int& ex_min_, int& ex_max_) __attribute__((always_inline)) {
        ex_min_ = AC_EX_NONE; ex_max_ = -AC_EX_NONE;
        float am[W9_PH];
        const float* arow = in_amax ? in_amax + (size_t)b_ * H : x;              // (x: any readable address; the values are not used then)
#pragma unroll
        for (int r = 0; r < W9_PH; ++r) {                  // ten wave-uniform loads in flight together (rows clamped into the image)
            const int gy = y0_ - 1 + r;
            am[r] = arow[gy < 0 ? 0 : (gy >= H ? H - 1 : gy)];
        }
#pragma unroll
        for (int r = 0; r < W9_PH; ++r) {
            const int gy = y0_ - 1 + r;
            int e = __builtin_amdgcn_readfirstlane(ac_row_ex(FIRST ? am[r] * amax_gain + amax_offs : am[r]));
            if (!in_amax || gy < 0 || gy >= H) e = AC_EX_NONE;
            if (tid == 0) s_ex_[r] = e;
            if (e != AC_EX_NONE) { ex_min_ = e < ex_min_ ? e : ex_min_; ex_max_ = e > ex_max_ ? e : ex_max_; }
        }
        if (tid == 0) { s_ex_[W9_PH] = AC_EX_NONE; s_ex_[W9_PH + 1] = AC_EX_NONE; }
    };

    int cob, b, y0, x0;
    decode(0, cob, b, y0, x0);
    set_tile(y0, x0);
    // Stage 0's loads and weight DMA of the FIRST tile are issued behind its scale prologue.  Issuing them at the very top of the kernel was
    // built and measured in round 4 on the one-tile-per-workgroup kernel: bit-identical and 14 % SLOWER at C = 48 (profiles/r04i).
    int ex_min, ex_max, ex_buf = 0;
    row_scales(b, y0, s_ex2[0], ex_min, ex_max);
    prefetch_w(0, wpk + (size_t)cob * n_cb * 3 * W9_KFR);
    prefetch_x(0, x + (size_t)b * (FIRST ? C0 : C_in) * plane);
    float bias_v = bias[cob * W9_COB + (tid < W9_COB ? tid : 0)];          // this tile's channel biases (staged in LDS in its first stage)

    for (int it = 0;; ++it) {
        const bool more = it + 1 < n_my;
        int cob_n = cob, b_n = b, y0_n = y0, x0_n = x0, ex_min_n = AC_EX_NONE, ex_max_n = -AC_EX_NONE;
        const int* s_ex = s_ex2[ex_buf];

#ifndef W9_NO_ROWX
#define W9_NO_ROWX 0                 // 1: timing probe only - every tile takes the common path (wrong results on tiles that need the row-exact one)
#endif
        if (!W9_NO_ROWX && ex_min != AC_EX_NONE && ex_max - ex_min > AC_ROWX_SPREAD) {
#define W9_ROWX 1
#include "ac_conv96_tile.inc"
#undef W9_ROWX
        } else {
#define W9_ROWX 0
#include "ac_conv96_tile.inc"
#undef W9_ROWX
        }
        if (!more) break;
        cob = cob_n; b = b_n; y0 = y0_n; x0 = x0_n; ex_min = ex_min_n; ex_max = ex_max_n; ex_buf ^= 1;
    }
}

#ifndef AC_PROBES
#define AC_PROBES 0                   // 1: the probe build of tools/conv_order_probe.py - tile width and band width taken from the environment,
#endif                                // and the 8 x 48 / 8 x 64 instantiations of the 48-channel tile compiled in (measured slower: profiles/r04b)
#if AC_PROBES
static int w9_env_int(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
#endif

// Band width of the work order in tiles: the widest divisor of the tile row that keeps a band within 768 pixels.  A band edge costs two
// partial 128-byte lines per patch row that no resident workgroup shares, and the L2 fetches whole lines (profiles/r04a: one 16-byte
// load per line moves the line): at C = 48 a 4-tile band read 1.53x the tensor, 24 tiles 1.14x (3.78 -> 3.66 ms); beyond ~1000 pixels the
// resident workgroups of an XCD no longer cover two tile rows and the vertical halo starts to miss instead (96 tiles: 1.29x).
static int w9_band_width(int tiles_x, int tw) {
    int bw = 1;
    for (int d = 1; d <= tiles_x; ++d)
        if (tiles_x % d == 0 && d * tw <= 768) bw = d;
    return bw;
}

static int w9_launch(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* out, int B, int C_in, int C_out,
                     int H, int W, float w_unscale, int relu, const float* in_amax, float* out_amax, void* stream, int cob_width,
                     const float* w1 = nullptr, const float* b1 = nullptr, int C0 = 0, float amax_gain = 1.f, float amax_offs = 0.f) {
    AC_REQUIRE(ctx && x && w_packed && bias && out, "null pointer");
    AC_REQUIRE(B > 0 && C_in > 0 && C_in % 16 == 0 && C_out > 0 && C_out % cob_width == 0, "C_in % 16 == 0 and C_out % (96 or 48) == 0");
    AC_REQUIRE(H > 0 && H % W9_TH == 0 && W > 0 && W % 32 == 0, "H % 8 == 0 and W % 32 == 0");
    AC_REQUIRE(B < (1 << 23) && C_out / cob_width < 256 && W / 32 < 65536 && H / W9_TH < 32768, "walk table fields: B < 2^23, < 256 channel blocks");
    AC_REQUIRE((long long)H * W <= (1LL << 26), "plane too large (per-lane byte offsets are 32-bit: 16 planes must stay below 4 GiB)");
    int gx = 2;                          // 8 x 32 pixel tiles (W9Geo)
#if AC_PROBES
    if (cob_width == 48) {
        const int want = w9_env_int("AC_PROBE_CONV_GX");
        if (want == 3 && W % 48 == 0) gx = 3;
        if (want == 4 && W % 64 == 0) gx = 4;
    }
#endif
    const int tw = 16 * gx;
    const long long nblk = (long long)B * (C_out / cob_width) * (H / W9_TH) * (W / tw);
    AC_REQUIRE(nblk < (1LL << 31), "grid too large");
    const int tiles_x = W / tw;
    int bw = w9_band_width(tiles_x, tw);
#if AC_PROBES
    { const int v = w9_env_int("AC_PROBE_CONV_BW"); if (v > 0 && tiles_x % v == 0) bw = v; }
#endif
    dim3 block(256);
    hipStream_t st = (hipStream_t)stream;
    const f16x8* wp = (const f16x8*)w_packed;
    // persistent walk: the grid is the resident set (CUs x workgroups per CU) once every workgroup has at least W9_PERSIST_MIN tiles to walk;
    // smaller layers (levels 4 / 5) keep one tile per workgroup - the dispatcher balances those better than a static walk of 2.5 tiles
#define W9_GO(MT_, GX_, OCC_, RELU_, FIRST_) do { \
        const long long slots = (long long)(ctx->n_cu > 0 ? ctx->n_cu : 256) * (OCC_); \
        const bool persist = W9_PERSIST && nblk >= W9_PERSIST_MIN * slots && ((nblk | slots) & 7) == 0; \
        long long ngrid = persist ? slots : nblk; \
        while (persist && ngrid * W9_MAXT < nblk) ngrid += slots;       /* a workgroup's table holds W9_MAXT tiles */ \
        hipLaunchKernelGGL((k_conv3x3_f16x3_w96<MT_, GX_, OCC_, RELU_, FIRST_>), dim3((unsigned)ngrid), block, 0, st, x, wp, bias, out, \
        C_in, C_out, H, W, w_unscale, bw, (int)nblk, in_amax, out_amax, w1, b1, C0, amax_gain, amax_offs); } while (0)
#define W9_GO_R(MT_, GX_, OCC_, FIRST_) do { if (relu) W9_GO(MT_, GX_, OCC_, true, FIRST_); else W9_GO(MT_, GX_, OCC_, false, FIRST_); } while (0)
    if (w1) {
        AC_REQUIRE(b1 && C0 >= 1 && C0 <= 4 && C_in <= 64 && cob_width == 48, "fused first conv: 1 <= C0 <= 4, C_in <= 64, 48-channel workgroups");
#if AC_PROBES
        if (gx == 4) W9_GO_R(3, 4, 2, true); else if (gx == 3) W9_GO_R(3, 3, 2, true); else
#endif
        W9_GO_R(3, 2, W9_FIRST_OCC, true);
    } else if (cob_width == 96) {
        W9_GO_R(6, 2, 2, false);
    } else {
#if AC_PROBES
        if (gx == 4) W9_GO_R(3, 4, 2, false); else if (gx == 3) W9_GO_R(3, 3, 2, false); else
#endif
        W9_GO_R(3, 2, W9_S8_OCC, false);
    }
#undef W9_GO_R
#undef W9_GO
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_conv3x3_f16x3_w96(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* out, int B, int C_in,
                                     int C_out, int H, int W, float w_unscale, int relu, const float* in_amax, float* out_amax,
                                     void* stream) {
    return w9_launch(ctx, x, w_packed, bias, out, B, C_in, C_out, H, W, w_unscale, relu, in_amax, out_amax, stream, 96);
}

extern "C" int ac_conv3x3_f16x3_s8(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* out, int B, int C_in,
                                    int C_out, int H, int W, float w_unscale, int relu, const float* in_amax, float* out_amax,
                                    void* stream) {
    return w9_launch(ctx, x, w_packed, bias, out, B, C_in, C_out, H, W, w_unscale, relu, in_amax, out_amax, stream, 48);
}

// relu(conv3x3(relu(conv1x1(spec, w1) + b1))): the graph's first two convolutions in one launch (see FIRST above).  w_packed is the
// 3x3 conv's weights in the 48-channel layout of ac_conv3x3_f16x3_s8 (conv_pack.pack_conv3x3_w96(w, 48)).
extern "C" int ac_conv3x3_f16x3_first(ac_ctx* ctx, const float* spec, const float* w1, const float* b1, const void* w_packed,
                                       const float* bias, float* out, int B, int C0, int C_in, int C_out, int H, int W,
                                       float w_unscale, int relu, const float* spec_amax, float amax_gain, float amax_offs,
                                       float* out_amax, void* stream) {
    AC_REQUIRE(w1 && b1, "null pointer");
    AC_REQUIRE(amax_gain >= 0.f && amax_offs >= 0.f, "amax bound terms must be non-negative");
    return w9_launch(ctx, spec, w_packed, bias, out, B, C_in, C_out, H, W, w_unscale, relu, spec_amax, out_amax, stream, 48, w1, b1, C0,
                     spec_amax ? amax_gain : 1.f, spec_amax ? amax_offs : 0.f);
}
