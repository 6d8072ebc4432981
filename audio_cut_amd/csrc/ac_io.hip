// Loader / exporter kernels.
//  * ac_resample_poly: rational-rate polyphase FIR resampling of a resident track (the loader's 48 kHz -> 44.1 kHz and
//    the VAD's 44.1 kHz -> 16 kHz, `audio_processor.py:45-49`, `vocal_pause_detector.py:189`).  The reference's
//    resampler is soxr_hq through librosa; libsoxr's coefficients are not available offline, so the host designs a low-pass to
//    soxr's published HQ specification (pass band to 0.9136 of the lower Nyquist, stop band from it, 126 dB: _native.py
//    `_resample_filter`) and hands it over in scipy.signal.resample_poly's framing (scaled by `up`, front-padded for alignment,
//    zero extension) as polyphase rows, one wave per output (ac_common.h `ac_polyphase_dot_wave`).
//  * ac_pack_pcm24: float32 [-1, 1] -> little-endian 24-bit PCM exactly as soundfile.write(subtype="PCM_24") produces it
//    (`audio_export.py:109-111`).  python-soundfile switches libsndfile's clipping on (SFC_SET_CLIPPING), so the conversion
//    is libsndfile pcm.c f2let_clip_array: s = x * 2^31 in float32; s >= 2^31 - 1 -> 0x7FFFFF, s <= -2^31 -> 0x800000,
//    else the top three bytes of lrintf(s), i.e. lrintf(x * 2^31) >> 8 (a floor, not a rounding, of x * 2^23).
//    Four samples (12 bytes) per thread.
#include "ac_common.h"

__global__ __launch_bounds__(256) void k_resample_poly(const float* __restrict__ x, int64_t n, int up, int down,
                                                       const float* __restrict__ hp, int tpp, int64_t n_pre_remove,
                                                       float* __restrict__ out, int64_t n_out) {
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * AC_RS_PER_WAVE;      // a wave walks AC_RS_PER_WAVE consecutive outputs
    for (int64_t m = m0; m < m0 + AC_RS_PER_WAVE && m < n_out; ++m) {
        const float v = ac_polyphase_dot_wave(x, n, hp, up, tpp, (m + n_pre_remove) * (int64_t)down);
        if ((threadIdx.x & 63) == 0) out[m] = v;
    }
}

extern "C" int ac_resample_poly(ac_ctx* ctx, const float* x, int64_t n, int up, int down, const float* hp, int64_t hlen,
                                 int64_t n_pre_remove, float* out, int64_t n_out, void* stream) {
    AC_REQUIRE(ctx && x && hp && out, "null pointer");
    AC_REQUIRE(n > 0 && up > 0 && down > 0 && hlen > 0 && n_pre_remove >= 0 && n_out > 0, "sizes must be positive");
    AC_REQUIRE(hlen % up == 0 && hlen / up < (1LL << 31), "hp is [up][hlen / up] polyphase rows");
    const int64_t blocks = (n_out + 4 * AC_RS_PER_WAVE - 1) / (4 * AC_RS_PER_WAVE);
    AC_REQUIRE(blocks < (1LL << 31), "output too long");
    hipLaunchKernelGGL(k_resample_poly, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, up, down, hp,
                       (int)(hlen / up), n_pre_remove, out, n_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

__device__ inline int pcm24(float v) {
    const float s = v * 2147483648.0f;                   // libsndfile: normfact = 8.0 * 0x10000000, product in float32
    if (s >= 2147483647.0f) return 8388607;              // (float)0x7FFFFFFF == 2^31: every s >= 1.0 * 0x7FFFFFFF
    if (s <= -2147483648.0f) return -8388608;
    if (!(s == s)) return 0;                             // NaN: lrintf is undefined there; silence
    return ((int)rintf(s)) >> 8;                         // lrintf (half to even), then the three high bytes
}

__global__ __launch_bounds__(256) void k_pack_pcm24(const float* __restrict__ x, int64_t n, unsigned char* __restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;       // quad of samples
    const int64_t s0 = q * 4;
    if (s0 >= n) return;
    if (s0 + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(x + s0);
        const unsigned a = (unsigned)pcm24(v.x) & 0xFFFFFF, b = (unsigned)pcm24(v.y) & 0xFFFFFF;
        const unsigned c = (unsigned)pcm24(v.z) & 0xFFFFFF, d = (unsigned)pcm24(v.w) & 0xFFFFFF;
        unsigned* o = reinterpret_cast<unsigned*>(out + s0 * 3);     // 12-byte aligned: s0 % 4 == 0
        o[0] = a | (b << 24);
        o[1] = (b >> 8) | (c << 16);
        o[2] = (c >> 16) | (d << 8);
    } else {
        for (int64_t s = s0; s < n; ++s) {
            const unsigned a = (unsigned)pcm24(x[s]) & 0xFFFFFF;
            out[s * 3] = (unsigned char)a; out[s * 3 + 1] = (unsigned char)(a >> 8); out[s * 3 + 2] = (unsigned char)(a >> 16);
        }
    }
}

extern "C" int ac_pack_pcm24(ac_ctx* ctx, const float* x, int64_t n, unsigned char* out, void* stream) {
    AC_REQUIRE(ctx && x && out, "null pointer");
    AC_REQUIRE(n > 0, "n must be positive");
    const int64_t quads = (n + 3) / 4;
    AC_REQUIRE((quads + 255) / 256 < (1LL << 31), "input too long");
    AC_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)out) & 3) == 0, "x 16-byte and out 4-byte aligned");
    hipLaunchKernelGGL(k_pack_pcm24, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
