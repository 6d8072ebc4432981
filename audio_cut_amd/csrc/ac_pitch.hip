// Kernels of the detector's multi-feature branch (pure_vocal_pause_detector.py:410-459,937-1018):
// probabilistic YIN (librosa.pyin: trough probabilities -> pitch-bin observations -> Viterbi), LPC-12 formant peaks,
// zero-crossing rate.  The CMND curves (float64) come from ac_yin_f0 (ac_frames.hip).
#include <math.h>

#include "ac_common.h"

// =================================================================================================
// pyin observation probabilities, one workgroup per frame (librosa.core.pitch.__pyin_helper).
//   troughs of the CMND curve; for threshold k (1..100) the troughs below it form a Boltzmann-weighted prior by rank;
//   probs[trough] = sum_k prior * beta_probs[k-1]; the global minimum also takes the "no trough" mass;
//   each trough votes its probability into a pitch bin (later trough of the same bin overwrites);
//   voiced_prob = clip(sum bins); the unvoiced states share (1 - voiced_prob) / n_bins.
// Threshold membership is monotone in k, so the rank of trough j at threshold k is #{j' < j : kmin[j'] <= k}.
// Outputs are the log-probabilities the Viterbi pass needs.
// =================================================================================================
#define PY_MAX_LAGS 1024
#define PY_MAX_TROUGH 512
#define PY_NTHR 100

__global__ __launch_bounds__(256) void k_pyin_observe(const double* __restrict__ cmnd, int n_lags, int min_period, double sr,
                                                      double fmin, int n_bins, int bins_per_semitone,
                                                      const double* __restrict__ thresholds,   // [101]
                                                      const double* __restrict__ beta_probs,   // [100]
                                                      const double* __restrict__ beta_cum,     // [101]: sum(beta_probs[:n])
                                                      const double* __restrict__ boltz_fact,   // [PY_MAX_TROUGH + 1]: (1-e^-l)/(1-e^(-l N))
                                                      const double* __restrict__ boltz_exp,    // [PY_MAX_TROUGH]: e^(-l k)
                                                      double no_trough_prob, double tiny_val,
                                                      double* __restrict__ logv,               // [n_frames][n_bins]
                                                      double* __restrict__ logu,               // [n_frames]
                                                      double* __restrict__ voiced_prob) {      // [n_frames]
    __shared__ double s_c[PY_MAX_LAGS];
    __shared__ unsigned short s_tidx[PY_MAX_TROUGH];
    __shared__ unsigned char s_kmin[PY_MAX_TROUGH];          // smallest k in 1..100 with height < thresholds[k]; 101 = never
    __shared__ double s_prob[PY_MAX_TROUGH];
    __shared__ unsigned short s_ntr[PY_NTHR + 2];            // troughs below threshold k
    __shared__ double s_obs[1024];
    __shared__ int s_T;
    const int64_t f = blockIdx.x;
    const double* row = cmnd + f * (int64_t)n_lags;
    for (int i = threadIdx.x; i < n_lags; i += 256) s_c[i] = row[i];
    for (int i = threadIdx.x; i < n_bins + 1; i += 256) s_obs[i] = 0.0;
    __syncthreads();
    // trough list in lag order (thread 0: at most a few hundred lags)
    if (threadIdx.x == 0) {
        int T = 0;
        for (int i = 0; i < n_lags; ++i) {
            const double v = s_c[i];
            const double l = s_c[i > 0 ? i - 1 : 0], r = s_c[i < n_lags - 1 ? i + 1 : n_lags - 1];
            const bool tr = (i == 0) ? (n_lags > 1 && v < s_c[1]) : (v < l && v <= r);
            if (tr && T < PY_MAX_TROUGH) s_tidx[T++] = (unsigned short)i;
        }
        s_T = T;
    }
    __syncthreads();
    const int T = s_T;
    if (T > 0) {
        for (int j = threadIdx.x; j < T; j += 256) {
            const double h = s_c[s_tidx[j]];
            int k = 1;
            while (k <= PY_NTHR && !(h < thresholds[k])) ++k;
            s_kmin[j] = (unsigned char)k;
            s_prob[j] = 0.0;
        }
        for (int k = threadIdx.x; k <= PY_NTHR + 1; k += 256) s_ntr[k] = 0;
        __syncthreads();
        // n_troughs[k] = #{j : kmin[j] <= k}
        if (threadIdx.x >= 1 && threadIdx.x <= PY_NTHR) {
            const int k = threadIdx.x;
            int c = 0;
            for (int j = 0; j < T; ++j) c += (s_kmin[j] <= k);
            s_ntr[k] = (unsigned short)c;
        }
        __syncthreads();
        // probs[j] = sum_{k >= kmin[j]} fact[N_k] * exp(-l * rank_jk) * beta[k-1], rank_jk = #{j' < j : kmin[j'] <= k}
        for (int j = threadIdx.x; j < T; j += 256) {
            const int km = s_kmin[j];
            double p = 0.0;
            if (km <= PY_NTHR) {
                // ranks for all k at once: count earlier troughs by their kmin (histogram prefix)
                unsigned short cnt[PY_NTHR + 2];
                for (int k = 0; k <= PY_NTHR + 1; ++k) cnt[k] = 0;
                for (int q = 0; q < j; ++q) cnt[s_kmin[q]]++;
                int below = 0;
                for (int k = 1; k <= PY_NTHR; ++k) {
                    below += cnt[k];                          // earlier troughs with kmin <= k
                    if (k >= km) p += boltz_fact[s_ntr[k]] * boltz_exp[below] * beta_probs[k - 1];
                }
            }
            s_prob[j] = p;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int g = 0;
            double gh = s_c[s_tidx[0]];
            for (int j = 1; j < T; ++j) { const double h = s_c[s_tidx[j]]; if (h < gh) { gh = h; g = j; } }
            const int n_below_min = (int)s_kmin[g] - 1;       // thresholds 1..kmin-1 are not above the minimum
            s_prob[g] += no_trough_prob * beta_cum[n_below_min];
            // votes in lag order: a later trough of the same bin overwrites (observation_probs[bin, frame] = ...)
            for (int j = 0; j < T; ++j) {
                const double p = s_prob[j];
                if (p == 0.0) continue;
                const int i = s_tidx[j];
                double shift = 0.0;
                if (i > 0 && i < n_lags - 1) {
                    const double xm = s_c[i - 1], x0 = s_c[i], xp = s_c[i + 1];
                    const double a = xp + xm - 2.0 * x0;
                    const double b = (xp - xm) / 2.0;
                    if (!(fabs(b) >= fabs(a))) shift = -b / a;
                }
                const double period = (double)(min_period + i) + shift;
                const double f0 = sr / period;
                double bin = 12.0 * (double)bins_per_semitone * log2(f0 / fmin);
                bin = rint(bin);                              // np.round: half to even
                int bi = bin < 0.0 ? 0 : (bin > (double)n_bins ? n_bins : (int)bin);
                s_obs[bi] = p;                                // bi == n_bins lands in the unvoiced block and is dropped
            }
        }
        __syncthreads();
    }
    // voiced probability: sum over the voiced bins in bin order (numpy reduces axis 0 row by row)
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < n_bins; ++b) s += s_obs[b];
        s = s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s);
        voiced_prob[f] = s;
        logu[f] = log((1.0 - s) / (double)n_bins + tiny_val);
    }
    __syncthreads();
    double* out = logv + f * (int64_t)n_bins;
    for (int b = threadIdx.x; b < n_bins; b += 256) out[b] = log(s_obs[b] + tiny_val);
}

extern "C" int ac_pyin_observe(ac_ctx* ctx, const double* cmnd, int64_t n_frames, int n_lags, int min_period, double sr, double fmin,
                               int n_bins, int bins_per_semitone, const double* thresholds, const double* beta_probs,
                               const double* beta_cum, const double* boltz_fact, const double* boltz_exp, double no_trough_prob,
                               double tiny_val, double* logv, double* logu, double* voiced_prob, void* stream) {
    AC_REQUIRE(ctx && cmnd && thresholds && beta_probs && beta_cum && boltz_fact && boltz_exp && logv && logu && voiced_prob, "null pointer");
    AC_REQUIRE(n_frames > 0 && n_frames < (1LL << 31), "frame count");
    AC_REQUIRE(n_lags >= 3 && n_lags <= PY_MAX_LAGS, "3 <= n_lags <= 1024");
    AC_REQUIRE(n_bins >= 1 && n_bins < 1024 && bins_per_semitone >= 1 && min_period >= 1, "bin layout");
    hipLaunchKernelGGL(k_pyin_observe, dim3((unsigned)n_frames), dim3(256), 0, (hipStream_t)stream, cmnd, n_lags, min_period, sr, fmin,
                       n_bins, bins_per_semitone, thresholds, beta_probs, beta_cum, boltz_fact, boltz_exp, no_trough_prob, tiny_val,
                       logv, logu, voiced_prob);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// Viterbi over the 2 * n_bins pitch/voicing states (librosa.sequence.viterbi), ONE workgroup, sequential in time.
// The transition matrix kron([[1-s, s], [s, 1-s]], triangle band) is zero outside |i - j| <= half, where librosa's
// log(0 + tiny) still allows the jump: best predecessor = max(band candidates, global max + log(tiny)), first index
// on ties.  lt_same / lt_cross hold log(transition + tiny) for the band, [n_bins][2*half+1], column d <-> i = j-half+d
// stored per DESTINATION j.  value/ptr follow librosa: value[t][j] = logprob[t][j] + max_i(value[t-1][i] + lt[i][j]).
// =================================================================================================
#define VT_THREADS 1024
#define VT_MAX_STATES 2048

__global__ __launch_bounds__(VT_THREADS) void k_pyin_viterbi(const double* __restrict__ logv, const double* __restrict__ logu,
                                                             int64_t n_frames, int n_bins, int half,
                                                             const double* __restrict__ lt_same, const double* __restrict__ lt_cross,
                                                             double lt_zero, const double* __restrict__ log_p_init,
                                                             unsigned short* __restrict__ ptr, int* __restrict__ states) {
    __shared__ double s_val[2][VT_MAX_STATES];
    __shared__ double s_rmax[VT_THREADS / 64];
    __shared__ int s_ridx[VT_THREADS / 64];
    __shared__ double s_gmax;
    __shared__ int s_gidx;
    const int S = 2 * n_bins;
    const int W = 2 * half + 1;
    const int tid = threadIdx.x;
    for (int j = tid; j < S; j += VT_THREADS)
        s_val[0][j] = (j < n_bins ? logv[j] : logu[0]) + log_p_init[j];
    __syncthreads();
    int cur = 0;
    for (int64_t t = 1; t < n_frames; ++t) {
        // global first maximum of value[t-1]
        double bm = -INFINITY; int bi = 0x7fffffff;
        for (int j = tid; j < S; j += VT_THREADS) { const double v = s_val[cur][j]; if (v > bm || (v == bm && j < bi)) { bm = v; bi = j; } }
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_down(bm, off, AC_WAVE); const int oi = __shfl_down(bi, off, AC_WAVE);
            if (ov > bm || (ov == bm && oi < bi)) { bm = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { s_rmax[tid >> 6] = bm; s_ridx[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            double gm = s_rmax[0]; int gi = s_ridx[0];
            for (int w = 1; w < VT_THREADS / 64; ++w)
                if (s_rmax[w] > gm || (s_rmax[w] == gm && s_ridx[w] < gi)) { gm = s_rmax[w]; gi = s_ridx[w]; }
            s_gmax = gm; s_gidx = gi;
        }
        __syncthreads();
        const double oob = s_gmax + lt_zero; const int oob_i = s_gidx;
        const double lu = logu[t];
        const double* lv = logv + t * (int64_t)n_bins;
        unsigned short* prow = ptr + t * (int64_t)S;
        for (int j = tid; j < S; j += VT_THREADS) {
            const int blk = j >= n_bins, jb = j - blk * n_bins;
            double best = -INFINITY; int besti = 0x7fffffff;
            const double* ls = lt_same + (size_t)jb * W;
            const double* lc = lt_cross + (size_t)jb * W;
            // voiced predecessors (indices < n_bins) come first in librosa's argmax order
#pragma unroll 1
            for (int pb = 0; pb < 2; ++pb) {
                const double* lt = (pb == blk) ? ls : lc;
                for (int d = 0; d < W; ++d) {
                    const int ib = jb - half + d;
                    if (ib < 0 || ib >= n_bins) continue;
                    const int i = pb * n_bins + ib;
                    const double v = s_val[cur][i] + lt[d];
                    if (v > best || (v == best && i < besti)) { best = v; besti = i; }
                }
            }
            if (oob > best || (oob == best && oob_i < besti)) { best = oob; besti = oob_i; }
            s_val[cur ^ 1][j] = (blk ? lu : lv[jb]) + best;
            prow[j] = (unsigned short)besti;
        }
        __syncthreads();
        cur ^= 1;
    }
    // last state = first maximum of value[-1]; back-track on one thread
    {
        double bm = -INFINITY; int bi = 0x7fffffff;
        for (int j = tid; j < S; j += VT_THREADS) { const double v = s_val[cur][j]; if (v > bm || (v == bm && j < bi)) { bm = v; bi = j; } }
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_down(bm, off, AC_WAVE); const int oi = __shfl_down(bi, off, AC_WAVE);
            if (ov > bm || (ov == bm && oi < bi)) { bm = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { s_rmax[tid >> 6] = bm; s_ridx[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            double gm = s_rmax[0]; int gi = s_ridx[0];
            for (int w = 1; w < VT_THREADS / 64; ++w)
                if (s_rmax[w] > gm || (s_rmax[w] == gm && s_ridx[w] < gi)) { gm = s_rmax[w]; gi = s_ridx[w]; }
            int st = gi;
            states[n_frames - 1] = st;
            for (int64_t t = n_frames - 2; t >= 0; --t) {
                st = ptr[(t + 1) * (int64_t)S + st];
                states[t] = st;
            }
        }
    }
}

extern "C" int ac_pyin_viterbi(ac_ctx* ctx, const double* logv, const double* logu, int64_t n_frames, int n_bins, int half,
                               const double* lt_same, const double* lt_cross, double lt_zero, const double* log_p_init,
                               unsigned short* ptr_scratch, int* states, void* stream) {
    AC_REQUIRE(ctx && logv && logu && lt_same && lt_cross && log_p_init && ptr_scratch && states, "null pointer");
    AC_REQUIRE(n_frames > 0 && n_bins >= 1 && 2 * n_bins <= VT_MAX_STATES && half >= 0 && half < n_bins, "state layout");
    hipLaunchKernelGGL(k_pyin_viterbi, dim3(1), dim3(VT_THREADS), 0, (hipStream_t)stream, logv, logu, n_frames, n_bins, half, lt_same,
                       lt_cross, lt_zero, log_p_init, ptr_scratch, states);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// LPC-12 formant peaks per 25 ms frame (`_extract_formants`, pure_vocal_pause_detector.py:959-1018):
// pre-emphasis 0.95, Burg recursion (librosa.lpc, float32 series with float64 reductions), |1 / A(e^jw)| on 512
// points of [0, pi), strict local maxima >= 10 % of the maximum, the three lowest-frequency peaks.
// One workgroup per frame.  out_count[f] = number of peaks found (may exceed 3), out_mag[f][0..2] their magnitudes.
// =================================================================================================
#define LP_MAX_FRAME 2048
#define LP_ORDER_MAX 32

__global__ __launch_bounds__(256) void k_lpc_formants(const float* __restrict__ x, int64_t n, int frame_len, int hop, int order,
                                                      float preemph, int* __restrict__ out_count, double* __restrict__ out_mag) {
    __shared__ float s_f[LP_MAX_FRAME];
    __shared__ float s_b[LP_MAX_FRAME];
    __shared__ double s_red[8];
    __shared__ float s_ar[2][LP_ORDER_MAX + 1];
    __shared__ double s_mag[512];
    __shared__ float s_den, s_rc;
    const int64_t f = blockIdx.x;
    const float* src = x + f * (int64_t)hop;
    const int tid = threadIdx.x;
    // pre-emphasised frame y[0] = x[0], y[i] = x[i] - 0.95 x[i-1] (float32); fwd = y[1:], bwd = y[:-1]
    const int m0 = frame_len - 1;
    for (int i = tid; i < frame_len; i += 256) {
        const float y = (i == 0) ? src[0] : src[i] - preemph * src[i - 1];
        if (i >= 1) s_f[i - 1] = y;
        if (i < frame_len - 1) s_b[i] = y;
    }
    for (int i = tid; i <= order; i += 256) { s_ar[0][i] = (i == 0) ? 1.f : 0.f; s_ar[1][i] = (i == 0) ? 1.f : 0.f; }
    __syncthreads();
    {
        double acc = 0.0;
        for (int i = tid; i < m0; i += 256) { const float a = s_f[i], b = s_b[i]; acc += (double)(a * a + b * b); }
        const double tot = block_sum_f64_256(acc, s_red);
        if (tid == 0) s_den = (float)tot;
    }
    __syncthreads();
    int off = 0;                 // fwd window starts at s_f[off]; both windows have m elements
    int m = m0;
    int cur = 0;                 // s_ar[cur] = ar_coeffs, s_ar[cur ^ 1] = ar_coeffs_prev (swapped every iteration)
    for (int it = 0; it < order; ++it) {
        double acc = 0.0;
        for (int i = tid; i < m; i += 256) acc += (double)(s_b[i] * s_f[off + i]);
        const double dot = block_sum_f64_256(acc, s_red + 4);
        if (tid == 0) {
            float rc = (float)dot;
            rc = rc * -2.0f;
            rc = rc / (s_den + 1.17549435e-38f);
            s_rc = rc;
        }
        __syncthreads();
        const float rc = s_rc;
        cur ^= 1;                // swap: the old coefficients are now "prev"
        if (tid >= 1 && tid <= it + 1) s_ar[cur][tid] = s_ar[cur ^ 1][tid] + rc * s_ar[cur ^ 1][it - tid + 1];
        for (int i = tid; i < m; i += 256) {
            const float fo = s_f[off + i], bo = s_b[i];
            s_f[off + i] = fo + rc * bo;
            s_b[i] = bo + rc * fo;
        }
        __syncthreads();
        if (tid == 0) {
            const float q = 1.0f - rc * rc;
            const float bl = s_b[m - 1], f0 = s_f[off];
            s_den = q * s_den - bl * bl - f0 * f0;
        }
        off += 1; m -= 1;
        __syncthreads();
    }
    // ar_coeffs of the last iteration live in s_ar[cur]; entries above `order` untouched.  (librosa copies the
    // untouched higher entries from two iterations back; they are never written before their own iteration, so zero.)
    double a[LP_ORDER_MAX + 1];
    for (int k = 0; k <= order; ++k) a[k] = (double)s_ar[cur][k];
    for (int k = tid; k < 512; k += 256) {
        const double w = M_PI * (double)k / 512.0;
        double re = 0.0, im = 0.0;
        for (int q = 0; q <= order; ++q) { double sn, cs; sincos(w * (double)q, &sn, &cs); re += a[q] * cs; im -= a[q] * sn; }
        s_mag[k] = 1.0 / sqrt(re * re + im * im);
    }
    __syncthreads();
    if (tid == 0) {
        double mx = 0.0;
        for (int k = 0; k < 512; ++k) mx = fmax(mx, s_mag[k]);
        const double hmin = mx * 0.1;
        int cnt = 0;
        // scipy.signal.find_peaks: strict rise then fall, flat tops report their middle sample
        int k = 1;
        while (k < 511) {
            if (s_mag[k - 1] < s_mag[k]) {
                int ahead = k + 1;
                while (ahead < 511 && s_mag[ahead] == s_mag[k]) ++ahead;
                if (s_mag[ahead] < s_mag[k]) {
                    const int mid = (k + ahead - 1) / 2;
                    if (s_mag[mid] >= hmin) { if (cnt < 3) out_mag[f * 3 + cnt] = s_mag[mid]; ++cnt; }
                    k = ahead;
                    continue;
                }
            }
            ++k;
        }
        for (int q = cnt; q < 3; ++q) out_mag[f * 3 + q] = 0.0;
        out_count[f] = cnt;
    }
}

extern "C" int ac_lpc_formants(ac_ctx* ctx, const float* x, int64_t n, int frame_len, int hop, int order, float preemph,
                               int* out_count, double* out_mag, int64_t n_frames, void* stream) {
    AC_REQUIRE(ctx && x && out_count && out_mag, "null pointer");
    AC_REQUIRE(frame_len >= 4 && frame_len <= LP_MAX_FRAME && hop > 0 && order >= 1 && order <= LP_ORDER_MAX && order < frame_len - 1, "frame / order");
    AC_REQUIRE(n_frames > 0 && n_frames < (1LL << 31) && (n_frames - 1) * (int64_t)hop + frame_len <= n, "frames must lie inside the signal");
    hipLaunchKernelGGL(k_lpc_formants, dim3((unsigned)n_frames), dim3(256), 0, (hipStream_t)stream, x, n, frame_len, hop, order, preemph,
                       out_count, out_mag);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// =================================================================================================
// Zero-crossing rate (librosa.feature.zero_crossing_rate: edge-padded centred frames, sign-bit changes, mean).
// One wave per frame.
// =================================================================================================
__global__ __launch_bounds__(256) void k_zcr(const float* __restrict__ x, int64_t n, int frame_len, int hop, double* __restrict__ out,
                                             int64_t n_frames) {
    const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= n_frames) return;
    const int lane = threadIdx.x & 63;
    const int64_t s0 = f * (int64_t)hop - frame_len / 2;
    int c = 0;
    for (int i = 1 + lane; i < frame_len; i += 64) {
        int64_t g1 = s0 + i, g0 = g1 - 1;
        g1 = g1 < 0 ? 0 : (g1 >= n ? n - 1 : g1);
        g0 = g0 < 0 ? 0 : (g0 >= n ? n - 1 : g0);
        // librosa.zero_crossings: |y| <= 1e-10 is clipped to +0 before the sign test (zero_pos=True)
        float v1 = x[g1], v0 = x[g0];
        v1 = fabsf(v1) <= 1e-10f ? 0.f : v1;
        v0 = fabsf(v0) <= 1e-10f ? 0.f : v0;
        c += (signbit(v1) != signbit(v0));
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, AC_WAVE);
    if (lane == 0) out[f] = (double)c / (double)frame_len;
}

extern "C" int ac_zero_crossing_rate(ac_ctx* ctx, const float* x, int64_t n, int frame_len, int hop, double* out, int64_t n_frames,
                                     void* stream) {
    AC_REQUIRE(ctx && x && out, "null pointer");
    AC_REQUIRE(n > 0 && frame_len > 1 && hop > 0 && n_frames == 1 + n / hop, "n_frames != 1 + n/hop");
    hipLaunchKernelGGL(k_zcr, dim3((unsigned)((n_frames + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, n, frame_len, hop, out, n_frames);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
