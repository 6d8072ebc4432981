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
#ifndef W9_PF
#define W9_PF 1                      // activation-patch prefetch depth of the 48-channel tile in stages.  1: the product.  2 (A/B switch): two register sets and
#endif                               // LDS-only barriers; 226 VGPRs = two workgroups per CU (-DW9_S8_OCC=2); measured 10 % SLOWER than depth 1 there (profiles/r04x)
#ifndef W9_S8_OCC
#define W9_S8_OCC 3                  // workgroups per CU of the 48-channel variant
#endif
#ifndef W9_FIRST_OCC
#define W9_FIRST_OCC 3               // workgroups per CU of the fused first conv
#endif

// w9_dma16 = ac_lds_dma16 (ac_common.h): the LDS-DMA issued from inline assembly, hidden from the compiler's wait-count pass.
#define w9_dma16 ac_lds_dma16

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a release / acquire fence over ALL address spaces: the compiler puts
// s_waitcnt vmcnt(0) in front of every s_barrier, so no global load can stay in flight across a stage boundary - a patch prefetch TWO
// stages ahead (W9_PF 2) would be forced to land one stage early (round 2's depth-2 experiment, profiles/r02g, ran into exactly that).  With
// this barrier the loads do stay in flight - and the kernel is slower still (profiles/r04x): it is not waiting for its loads.  Global data the
// kernel reads (patch loads, weight DMA) is waited for by counted s_waitcnt vmcnt where it is consumed; what it writes it never reads.
__device__ __forceinline__ void w9_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14): vector memory operations return in
// order, so this waits for everything issued before the N youngest
template <int N> __device__ __forceinline__ void w9_wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }
__device__ inline unsigned short w9_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }

#if W9_PROBE & 4                      // probe build: the operands stay live, the matrix instruction is not issued
__device__ __forceinline__ f32x4 w9_mfma(f16x8 a, f16x8 b, f32x4 c) { asm volatile("" :: "v"(a), "v"(b)); return c; }
#else
__device__ __forceinline__ f32x4 w9_mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
#endif

// MT = 16-row output-channel tiles per workgroup: 6 (96 channels, two workgroups per CU) or 3 (48 channels).  GX: W9Geo.
// ROWX: the row-exact path of ac_common.h (per-row staging scales, power-of-two fragment factors); s_ex = log2 scale per patch row.
// FIRST fuses the graph's first 1x1 convolution (C0 <= 4 spectrogram channels -> C_in, + bias + ReLU) into the loader: x is the
// [B][C0][H][W] spectrogram, a thread's four spectrogram float4 are loaded ONCE and every stage's 8 channels are generated from
// them per staged pixel (s_first = [C_in][w1[0..3], b1] in LDS; float32 FMAs in ac_conv1x1_small's order, so the values are
// bit-identical to running that kernel first) - the C_in-channel tensor never touches HBM and the K loop has no activation loads.
// Every accumulator receives its products in the same order whatever MT / GX (k-step by k-step: ah*bl, al*bh, ah*bh), and the
// activation scale is a function of the tile's ROWS only, so all geometries produce bit-identical outputs.
template <int MT, int GX, int EP_M, int PF, bool RELU, bool ROWX, bool FIRST>
__device__ __forceinline__ void w9_tile(const float* __restrict__ x, const f16x8* __restrict__ wbase /* this channel block's fragments */, const float* __restrict__ bias,
                                        float* __restrict__ out, int C_in, int C_out, int H, int W, float w_unscale,
                                        float* __restrict__ out_amax, unsigned char* s_raw, const int* s_ex, int ex_min,
                                        int co_base, int b, int y0, int x0, const float* s_first, int C0) {
    using G = W9Geo<GX>;
    constexpr int W9_MT = MT, NQ = G::NQ, RS = G::RS, NR = G::NR;
    constexpr int QH = G::QH;
    constexpr bool PIPE = W9_PIPE && MT == 6 && GX == 2 && !ROWX;  // the 48-channel variants have no registers for it (measured on 8 x 32: spills, 40 % slower)
    constexpr int W9_KFR = 2 * MT * 64;                       // 16-byte fragments per k-step (hi, lo)
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
    float* s_out = reinterpret_cast<float*>(s_raw);                           // [16 EP_M co][8 rows][TW + 4], MT / EP_M passes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, px = lane & 15;
    const int n_cb = C_in / W9_CB;                                             // even: the stage that issues a shared step is always odd
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * (FIRST ? C0 : C_in) * plane;
    // staging: slot tid + 256 r -> (row 0..9, float4 of the row, channel quad 0..1): one aligned float4 (4 pixels) of 4 channels each
    // out-of-image / idle slots load from a clamped in-bounds address and are zeroed when staged: no divergent branch around the
    // loads (6 fewer spilled registers in the 48-channel variant).  Activation loads run one stage ahead; two stages ahead (two
    // register sets, counted vmcnt) was measured 2-12 % SLOWER on every level (profiles/r02g_conv_prefetch_depth.log).
    int a_off[NR], a_ld[NR], a_c4[NR];
    float a_zs[NR];                      // the staging scale of the slot; 0 for zero padding and the idle slots of a row
    bool a_live[NR];
    const float unscale = w_unscale * (ex_min == AC_EX_NONE ? 1.f : ldexpf(1.f, -ex_min));
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int slot = tid + 256 * r;
        a_c4[r] = slot & 1;
        const int a_rest = slot >> 1;
        const int a_row = a_rest / G::NQD, a_qd = a_rest - a_row * G::NQD;
        a_live[r] = a_rest < W9_PH * G::NQD;
        // common path: one scale for the tile = the maximum over exactly the patch rows y0 - 1 .. y0 + 8 (ex_min is its log2)
        float act_s = ex_min == AC_EX_NONE ? 1.f : ldexpf(1.f, ex_min);
        if (ROWX) {                      // row-exact path: this slot stages one patch row, at that row's own scale
            const int e_row = a_live[r] ? s_ex[a_row] : AC_EX_NONE;
            act_s = e_row == AC_EX_NONE ? 1.f : ldexpf(1.f, e_row);
        }
        a_off[r] = ((a_row * RS + 4 * a_qd) * W9_CB + a_c4[r] * 4);            // u16 elements
        int a_src = -1;
        if (a_live[r]) {
            const int gy = y0 + a_row - 1, gx = x0 - 4 + 4 * a_qd;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) a_src = gy * W + gx;
        }
        a_ld[r] = a_src >= 0 ? a_src : 0;
        a_zs[r] = a_src >= 0 ? act_s : 0.f;
    }

    // B fragment of (k-step ks, pixel group q): u16 offset b_tap[ks] + a compile-time constant of q (it becomes the ds_read's offset field)
    const int b_lane = ((2 * wave) * RS + px + 3) * W9_CB;                     // patch column c is staged column c + 3
    int b_tap[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { const int tap = 4 * ks + g; b_tap[ks] = b_lane + ((tap / 3) * RS + (tap % 3)) * W9_CB; }

    f32x4 acc[W9_MT][NQ];
#pragma unroll
    for (int m = 0; m < W9_MT; ++m)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[m][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 k8h[NQ], k8l[NQ];             // tap-8 activation fragments of four consecutive stages, one lane group each
#pragma unroll
    for (int q = 0; q < NQ; ++q) { k8h[q] = (f16x8)(_Float16)0; k8l[q] = (f16x8)(_Float16)0; }

    // fragments per stage in the packed weights: [cob][cb][3 k-steps][2][MT][64]; the third k-step exists for cb & 3 == 3 only
    float4 pre_x[PF][NR][4];            // PF register sets: stage cb's patch sits in set cb % PF (static after the unroll by PF below)

    auto prefetch_x = [&](int cb, int set) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ci = FIRST ? q : cb * W9_CB + a_c4[r] * 4 + q;            // FIRST: the spectrogram's channels, fetched once
#if W9_PROBE & 0x80
                pre_x[set][r][q] = make_float4(1.f + ci, 2.f, 3.f, 4.f);
#else
                pre_x[set][r][q] = (!FIRST || q < C0) ? *reinterpret_cast<const float4*>(xb + (size_t)ci * plane + a_ld[r]) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
            }
        }
    };
    auto prefetch_w = [&](int cb) {
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
#else
            if (inst < n_inst) {
#if W9_ASM_DMA_ALL
                w9_dma16(wcb + inst * 64 + lane, dst + inst * 64);
#else
                if constexpr (PIPE) w9_dma16(wcb + inst * 64 + lane, dst + inst * 64);
                else __builtin_amdgcn_global_load_lds(wcb + inst * 64 + lane, dst + inst * 64, 16, 0, 0);
#endif
            }
#endif
        }
    };
    // split one slot's 4 pixels x 4 channels to f16 hi / lo and store them: two 8-byte stores per pixel
    auto stage_slot = [&](int cb, int r, int set) {
        const float zs = a_zs[r];
        const float* v4[4] = {&pre_x[set][r][0].x, &pre_x[set][r][1].x, &pre_x[set][r][2].x, &pre_x[set][r][3].x};
        if (FIRST) {
            // two pixels at a time: all sixteen generated values at once cost 2 GB of scratch per launch, one pixel at a time
            // re-reads the 1x1 weights from LDS four times (measured 30 % slower)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                float gv[2][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float* wf = s_first + (cb * W9_CB + a_c4[r] * 4 + q) * 5;
                    const float wv[4] = {wf[0], wf[1], wf[2], wf[3]};
                    const float bv = wf[4];
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        float a = bv;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j < C0) a = fmaf(wv[j], v4[j][2 * kp + p2], a);
                        gv[p2][q] = fmaxf(a, 0.f);         // zero padding applies to the conv input (zs = 0), not to relu(b1)
                    }
                }
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    unsigned short h4[4], l4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = fminf(fmaxf(gv[p2][q] * zs, -65504.f), 65504.f);
                        const _Float16 hv = (_Float16)v;
                        h4[q] = w9_bits(hv);
                        l4[q] = w9_bits((_Float16)(v - (float)hv));
                    }
                    const int off = a_off[r] + (2 * kp + p2) * W9_CB;
                    *reinterpret_cast<uint2*>(&s_hi[off]) = make_uint2((unsigned)h4[0] | ((unsigned)h4[1] << 16), (unsigned)h4[2] | ((unsigned)h4[3] << 16));
                    *reinterpret_cast<uint2*>(&s_lo[off]) = make_uint2((unsigned)l4[0] | ((unsigned)l4[1] << 16), (unsigned)l4[2] | ((unsigned)l4[3] << 16));
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned short h4[4], l4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = fminf(fmaxf(v4[q][k] * zs, -65504.f), 65504.f);
                    const _Float16 hv = (_Float16)v;
                    h4[q] = w9_bits(hv);
                    l4[q] = w9_bits((_Float16)(v - (float)hv));
                }
                const int off = a_off[r] + k * W9_CB;
                *reinterpret_cast<uint2*>(&s_hi[off]) = make_uint2((unsigned)h4[0] | ((unsigned)h4[1] << 16), (unsigned)h4[2] | ((unsigned)h4[3] << 16));
                *reinterpret_cast<uint2*>(&s_lo[off]) = make_uint2((unsigned)l4[0] | ((unsigned)l4[1] << 16), (unsigned)l4[2] | ((unsigned)l4[3] << 16));
            }
        }
    };

    // Stage 0's loads and weight DMA are issued HERE, behind the scale prologue.  Issuing them at the very top of the kernel (in flight while
    // the ten row maxima arrive) was built and measured in round 4: bit-identical and 14 % SLOWER at C = 48 (3.62 -> 4.14 ms), 2 % at C = 96
    // (profiles/r04i): vmcnt returns in order, so the wait for the (L2-resident) maxima then also waits for the HBM-bound patch loads,
    // and the three workgroups of a CU stop overlapping their prologues with one another's K loops.
    static_assert(PF == 1 || !FIRST, "the fused first conv fetches its spectrogram patch once per tile");
    prefetch_w(0); prefetch_x(0, 0);
    if (PF == 2) prefetch_x(1, 1);       // n_cb >= 2 (C_in % 16 == 0)
    // In flight at the top of stage cb, oldest first: [PF 2: X(cb), issued two stages ago] W(cb) X(cb + 1).  The compiler waits for X(cb) where
    // stage_slot reads it (counted: W and X(cb + 1) stay in flight); W(cb) is waited for in front of the stage's second barrier with the
    // X(cb + 1) loads (4 NR instructions per lane) still out.  n_cb is even, so the loop below is unrolled by PF and the set index is static.
    for (int cb0 = 0; cb0 < ((W9_PROBE & 0x40) ? 0 : n_cb); cb0 += PF) {
#pragma unroll
    for (int h = 0; h < PF; ++h) {
        const int cb = cb0 + h;
        if (PF == 2) w9_barrier(); else __syncthreads();      // previous stage fully consumed
        if (!(W9_PROBE & 8)) {
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (a_live[r]) stage_slot(cb, r, h);
        }
        // this stage's weight fragments have landed
        if (PF == 2) { if (cb + 1 < n_cb) w9_wait_vm<4 * NR>(); else w9_wait_vm<0>(); w9_barrier(); }
        else { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); }
        if (cb + 1 < n_cb) prefetch_w(cb + 1);
        if (!FIRST && cb + PF < n_cb) prefetch_x(cb + PF, h);
        const f16x8* s_w = (cb & 1) ? s_w1 : s_w0;
        if constexpr (PIPE) {
        // Fragment reads run AHEAD of the MFMAs that consume them (the compiler's own schedule drained lgkmcnt to 0 in front of every
        // group of 12 MFMAs: six exposed LDS round trips per stage and wave).  A unit = 12 MFMAs of one row tile m of one k-step; the
        // A fragments (2 per unit) are double-buffered and requested one unit ahead; the B fragments of the next k-step replace the
        // current ones pair by pair inside the last unit of a k-step, behind the MFMAs that read them last.  sched_barriers pin the
        // order; the waits are the compiler's counted lgkmcnt(N).
        const bool third = (((cb & 3) == 3) || cb == n_cb - 1) && !(W9_PROBE & 0x10);
        f16x8 bh[4], bl[4], ah[2], al[2];
        auto ld_b = [&](int ks, int q) {
            const int off = b_tap[ks] + ((q >> 1) * RS + (q & 1) * 16) * W9_CB;
            bh[q] = *reinterpret_cast<const f16x8*>(&s_hi[off]);
            bl[q] = *reinterpret_cast<const f16x8*>(&s_lo[off]);
        };
        auto ld_a = [&](int ks, int m, int slot) {
            ah[slot] = s_w[((ks * 2 + 0) * W9_MT + m) * 64 + lane];
            al[slot] = s_w[((ks * 2 + 1) * W9_MT + m) * 64 + lane];
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) ld_b(0, q);
        ld_a(0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int m = 0; m < W9_MT; ++m) {
                const int slot = (ks * W9_MT + m) & 1;
                const bool last_m = m == W9_MT - 1;
                if (!last_m) ld_a(ks, m + 1, slot ^ 1);
                else if (ks == 0) ld_a(1, 0, slot ^ 1);
                else if (third) ld_a(2, 0, slot ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                if (last_m && ks == 0) {
#pragma unroll
                    for (int qp = 0; qp < 4; qp += 2) {
#pragma unroll
                        for (int q = qp; q < qp + 2; ++q) acc[m][q] = w9_mfma(ah[slot], bl[q], acc[m][q]);
#pragma unroll
                        for (int q = qp; q < qp + 2; ++q) acc[m][q] = w9_mfma(al[slot], bh[q], acc[m][q]);
#pragma unroll
                        for (int q = qp; q < qp + 2; ++q) acc[m][q] = w9_mfma(ah[slot], bh[q], acc[m][q]);
                        __builtin_amdgcn_sched_barrier(0);
                        ld_b(1, qp); ld_b(1, qp + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    if (last_m && ks == 1 && !(W9_PROBE & 0x10)) {
                        // tap 8 (dy = dx = 2) of this stage goes into lane group cb & 3 of the carried fragments; requested in front
                        // of the stage's last unit so that a shared step that follows finds them landed
                        if (g == (cb & 3)) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int off = b_lane + ((2 + (q >> 1)) * RS + 2 + (q & 1) * 16) * W9_CB;
                                k8h[q] = *reinterpret_cast<const f16x8*>(&s_hi[off]);
                                k8l[q] = *reinterpret_cast<const f16x8*>(&s_lo[off]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[m][q] = w9_mfma(ah[slot], bl[q], acc[m][q]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[m][q] = w9_mfma(al[slot], bh[q], acc[m][q]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[m][q] = w9_mfma(ah[slot], bh[q], acc[m][q]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (third) {                     // a trailing group of two stages: lane groups 2-3 meet zero weights
#pragma unroll
            for (int m = 0; m < W9_MT; ++m) {
                const int slot = (2 * W9_MT + m) & 1;
                if (m + 1 < W9_MT) ld_a(2, m + 1, slot ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[m][q] = w9_mfma(ah[slot], k8l[q], acc[m][q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[m][q] = w9_mfma(al[slot], k8h[q], acc[m][q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[m][q] = w9_mfma(ah[slot], k8h[q], acc[m][q]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int dy = (4 * ks + g) / 3;              // this lane group's tap row (ROWX only)
#pragma unroll
            for (int q0 = 0; q0 < NQ; q0 += QH) {         // QH pixel groups at a time (8 x 64 tile: two halves, A fragments read per half)
                f16x8 bh[QH], bl[QH];
#pragma unroll
                for (int qq = 0; qq < QH; ++qq) {
                    const int q = q0 + qq, ty = 2 * wave + q / GX;
                    const int off = b_tap[ks] + ((q / GX) * RS + (q % GX) * 16) * W9_CB;    // one address register per k-step, q in the offset field
                    bh[qq] = *reinterpret_cast<const f16x8*>(&s_hi[off]);
                    bl[qq] = *reinterpret_cast<const f16x8*>(&s_lo[off]);
                    if (ROWX) { const _Float16 f = ac_rowx_frag_factor(s_ex, ty, dy); bh[qq] *= (f16x8)f; bl[qq] *= (f16x8)f; }
                }
#pragma unroll
                for (int m = 0; m < W9_MT; ++m) {
                    const f16x8 ah = s_w[((ks * 2 + 0) * W9_MT + m) * 64 + lane];
                    const f16x8 al = s_w[((ks * 2 + 1) * W9_MT + m) * 64 + lane];
#pragma unroll
                    for (int qq = 0; qq < QH; ++qq) {
                        acc[m][q0 + qq] = w9_mfma(ah, bl[qq], acc[m][q0 + qq]);
                        acc[m][q0 + qq] = w9_mfma(al, bh[qq], acc[m][q0 + qq]);
                        acc[m][q0 + qq] = w9_mfma(ah, bh[qq], acc[m][q0 + qq]);
                    }
                }
            }
        }
        // tap 8 (dy = dx = 2) of this stage goes into lane group cb & 3 of the carried fragments
        if (g == (cb & 3) && !(W9_PROBE & 0x10)) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int ty = 2 * wave + q / GX;
                const int off = b_lane + ((2 + q / GX) * RS + 2 + (q % GX) * 16) * W9_CB;
                k8h[q] = *reinterpret_cast<const f16x8*>(&s_hi[off]);
                k8l[q] = *reinterpret_cast<const f16x8*>(&s_lo[off]);
                if (ROWX) { const _Float16 f = ac_rowx_frag_factor(s_ex, ty, 2); k8h[q] *= (f16x8)f; k8l[q] *= (f16x8)f; }
            }
        }
        if (((cb & 3) == 3 || cb == n_cb - 1) && !(W9_PROBE & 0x10)) {           // a trailing group of two stages: lane groups 2-3 meet zero weights
#pragma unroll
            for (int m = 0; m < W9_MT; ++m) {
                const f16x8 ah = s_w[((2 * 2 + 0) * W9_MT + m) * 64 + lane];
                const f16x8 al = s_w[((2 * 2 + 1) * W9_MT + m) * 64 + lane];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    acc[m][q] = w9_mfma(ah, k8l[q], acc[m][q]);
                    acc[m][q] = w9_mfma(al, k8h[q], acc[m][q]);
                    acc[m][q] = w9_mfma(ah, k8h[q], acc[m][q]);
                }
            }
        }
            }
    }
    }
    // ---- epilogue in passes of EP_M row tiles through the LDS tile [co][row][x] -> whole row segments of the tile per store
    float vmax[2] = {0.f, 0.f};          // this wave's two output rows (ty = 2 wave + q / GX)
#if W9_PROBE & 0x20
    if (acc[0][0][0] == 12345.678f) out[0] = acc[0][0][0] + acc[W9_MT - 1][NQ - 1][3];      // probe build: no epilogue (the accumulators stay live)
    return;
#endif
#pragma unroll
    for (int m0 = 0; m0 < W9_MT; m0 += EP_M) {
        const int n_m = (W9_MT - m0) < EP_M ? (W9_MT - m0) : EP_M;
        __syncthreads();                 // stage buffers (first pass) / the previous pass's tile are done with
#pragma unroll
        for (int mm = 0; mm < EP_M; ++mm) {
            if (mm < n_m) {
                const int m = m0 + mm;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int ty = 2 * wave + q / GX, tx = (q % GX) * 16 + px;
                    const float us = ROWX ? w_unscale * ac_rowx_unscale(s_ex, ty) : unscale;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = mm * 16 + g * 4 + r;
                        float v = acc[m][q][r] * us + bias[co_base + m0 * 16 + co];
                        if (RELU) v = fmaxf(v, 0.f);
                        vmax[q / GX] = fmaxf(vmax[q / GX], fabsf(v));
                        s_out[(co * W9_TH + ty) * G::OUT_STRIDE + tx] = v;
                    }
                }
            }
        }
        __syncthreads();
        float* ob = out + ((size_t)b * C_out + (size_t)co_base + m0 * 16) * plane;
        constexpr int F4 = G::TW / 4;    // float4 per row segment
        for (int e = tid; e < n_m * 16 * W9_TH * F4; e += 256) {
            const int line = e / F4, q4 = e - line * F4;
            const int co = line >> 3, ty = line & 7;
            const float4 v = *reinterpret_cast<const float4*>(&s_out[line * G::OUT_STRIDE + 4 * q4]);
            *reinterpret_cast<float4*>(ob + (size_t)co * plane + (size_t)(y0 + ty) * W + x0 + 4 * q4) = v;
        }
    }
    if (out_amax) {
        ac_amax_commit(vmax[0], out_amax + (size_t)b * H + y0 + 2 * wave);
        ac_amax_commit(vmax[1], out_amax + (size_t)b * H + y0 + 2 * wave + 1);
    }
}

template <int MT, int GX, int OCC, bool RELU, bool FIRST>
__global__ __launch_bounds__(256, OCC) void k_conv3x3_f16x3_w96(const float* __restrict__ x, const f16x8* __restrict__ wpk,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int C_in, int C_out, int H, int W, float w_unscale, int bw,
                                                              const float* __restrict__ in_amax, float* __restrict__ out_amax,
                                                              const float* __restrict__ w1, const float* __restrict__ b1, int C0,
                                                              float amax_gain, float amax_offs) {
    using G = W9Geo<GX>;
    constexpr int W9_COB = 16 * MT, W9_KFR = 2 * MT * 64;
    // row tiles per epilogue pass: the output tile [16 EP_M][8][TW + 4] must fit the workgroup's share of the LDS (160 KB / OCC)
    constexpr int EP_M = (MT == 6) ? 3 : (GX == 3 ? 3 : 2);
    constexpr int PF = (MT == 3 && !FIRST) ? W9_PF : 1;        // the 96-channel tile has no registers for a second set (252 of 256), the first conv no per-stage loads
    constexpr int EP_BYTES = EP_M * 16 * W9_TH * G::OUT_STRIDE * 4;
    constexpr int K_BYTES = G::PATCH_BYTES + (2 + 3) * W9_KFR * 16;
    static_assert((K_BYTES > EP_BYTES ? K_BYTES : EP_BYTES) * OCC <= 160 * 1024 - OCC * 2048, "LDS per CU");
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[K_BYTES > EP_BYTES ? K_BYTES : EP_BYTES];
    __shared__ int s_ex[W9_PH + 2];
    __shared__ float s_first[FIRST ? 5 * 64 : 1];        // FIRST: [channel][w1[0..3], b1] of the fused 1x1 conv (C_in <= 64)
    const int tid = threadIdx.x;
    if (FIRST) {
        for (int c = tid; c < C_in; c += 256) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s_first[c * 5 + j] = (j < C0) ? w1[c * C0 + j] : 0.f;
            s_first[c * 5 + 4] = b1[c];
        }
    }
    const int n_cob = C_out / W9_COB;
    const int tiles_x = W / G::TW, tiles_y = H / W9_TH;
    // Work order.  Consecutive block ids go round the 8 XCDs, so XCD k walks the k-th eighth of the order below and its ~64-96 resident
    // workgroups are consecutive in it: the channel blocks of a tile, then the tiles of a BAND of bw tile columns row by row, then
    // the next band, then the next item.  Vertical and in-band horizontal halos are then found in that XCD's L2; what is re-fetched
    // is the two partial 128-byte lines per patch row at a band's edges (w9_band_width picks bw).
    int wi = blockIdx.x;
    if ((gridDim.x & 7) == 0) wi = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int cob = wi % n_cob;
    int t = wi / n_cob;
    const int b = t / (tiles_x * tiles_y);
    t -= b * (tiles_x * tiles_y);
    const int band = t / (tiles_y * bw);
    t -= band * (tiles_y * bw);
    const int y0 = (t / bw) * W9_TH, x0 = (band * bw + t % bw) * G::TW;
    // time-local power-of-two activation scale (ac_common.h): log2 scale of each patch row y0 - 1 .. y0 + 8.  FIRST: in_amax is
    // max |spectrogram| per row; the tensor that is split is the generated relu(w1 x + b1), bounded per row by
    // amax * max_c sum_j |w1[c][j]| + max_c |b1[c]| (amax_gain, amax_offs from the host; 1 and 0 otherwise: exact)
    if (tid < W9_PH + 2) {
        const int gy = y0 - 1 + tid;
        s_ex[tid] = (in_amax && tid < W9_PH && gy >= 0 && gy < H) ? ac_row_ex(in_amax[(size_t)b * H + gy] * amax_gain + amax_offs) : AC_EX_NONE;
    }
    __syncthreads();
    int ex_min = AC_EX_NONE, ex_max = -AC_EX_NONE;
#pragma unroll
    for (int r = 0; r < W9_PH; ++r) {
        const int e = s_ex[r];
        if (e != AC_EX_NONE) { ex_min = e < ex_min ? e : ex_min; ex_max = e > ex_max ? e : ex_max; }
    }
    ex_min = __builtin_amdgcn_readfirstlane(ex_min);
    ex_max = __builtin_amdgcn_readfirstlane(ex_max);
    if (ex_min != AC_EX_NONE && ex_max - ex_min > AC_ROWX_SPREAD)
        w9_tile<MT, GX, EP_M, PF, RELU, true, FIRST>(x, wpk + (size_t)cob * (C_in / W9_CB) * 3 * W9_KFR, bias, out, C_in, C_out, H, W, w_unscale, out_amax, s_raw, s_ex, ex_min, cob * W9_COB, b, y0, x0, s_first, C0);
    else
        w9_tile<MT, GX, EP_M, PF, RELU, false, FIRST>(x, wpk + (size_t)cob * (C_in / W9_CB) * 3 * W9_KFR, bias, out, C_in, C_out, H, W, w_unscale, out_amax, s_raw, s_ex, ex_min, cob * W9_COB, b, y0, x0, s_first, C0);
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
    AC_REQUIRE((long long)H * W < (1LL << 31), "plane too large");
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
    dim3 grid((unsigned)nblk), block(256);
    hipStream_t st = (hipStream_t)stream;
    const f16x8* wp = (const f16x8*)w_packed;
#define W9_GO(MT_, GX_, OCC_, RELU_, FIRST_) hipLaunchKernelGGL((k_conv3x3_f16x3_w96<MT_, GX_, OCC_, RELU_, FIRST_>), grid, block, 0, st, x, wp, bias, out, \
        C_in, C_out, H, W, w_unscale, bw, in_amax, out_amax, w1, b1, C0, amax_gain, amax_offs)
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
