/*
 * audiocut_hip.h — C ABI of libaudiocut_hip.so: the MI355X (gfx950) kernels under the audio-cut
 * separate+detect hot path.  Loaded from Python with ctypes
 * (audio_cut_amd/_native.py); see INTEGRATION.md for the stub a maintainer of the reference adds.
 *
 * The reference (BDMstudio/audio-cut) has no FFI: the path is Python calling librosa / torch /
 * onnxruntime.  Every entry point below therefore cites the reference *call site* whose per-sample
 * arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd by the caller; PyTorch owns the memory in the
 *     Python host) unless the name ends in _h; sizes are element counts;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous on
 *     that stream and never synchronise, allocate or free device memory (graph-capture safe);
 *   - return value 0 = ok, negative = error (AC_E_*); text via ac_last_error() (thread-local);
 *   - no exceptions cross the ABI, no global state besides the per-device context handle.
 */
#ifndef AUDIOCUT_HIP_H
#define AUDIOCUT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AC_OK 0
#define AC_E_INVALID (-1)   /* bad argument (shape/size the kernels do not support) */
#define AC_E_HIP (-2)       /* a HIP runtime call failed */
#define AC_E_NOMEM (-3)

#define AC_ABI_VERSION 6        /* 5: + ac_frame_rms_multi; 6: + ac_window_sum_squares */

typedef struct ac_ctx ac_ctx;

int ac_abi_version(void);
const char* ac_last_error(void);

/* Per-device context: twiddle / window / mel tables resident in HBM.  (new; no reference analogue) */
int ac_ctx_create(int device, ac_ctx** out);
int ac_ctx_destroy(ac_ctx* ctx);

/* ---- framewise features ------------------------------------------------------------------- */

/* librosa.feature.rms(y, frame_length, hop_length, center=True, pad_mode="constant")
 * call sites: analysis/features_cache.py:182 (4410/2205), core/pure_vocal_pause_detector.py:1113
 * (1102/441), :1397 (2048/441), core/seamless_splitter.py:1714,1848 (2048/441),
 * core/vocal_separator.py:483 (2205/882).  out[n_frames] f32, n_frames = 1 + (n + 2*(frame/2) - frame)/hop
 * when center, else 1 + (n - frame)/hop. */
int ac_frame_rms(ac_ctx* ctx, const float* x, int64_t n, int frame, int hop, int center,
                 float* out, int64_t n_frames, void* stream);

/* The same for up to AC_RMS_MULTI_MAX (frame, hop) configurations of ONE wave in one pass over it (center = True): the three series
 * the reference takes of the vocal stem after separation - core/vocal_separator.py:483 (2205/882), core/pure_vocal_pause_detector.py:1113
 * (1102/441), core/seamless_splitter.py:1714 (2048/441) - read the 42 MB stem once instead of three times.  Every frame is summed in
 * ac_frame_rms's order, so out[c] is bit-identical to ac_frame_rms(x, n, frame[c], hop[c], 1, ...).  frame / hop / out / n_frames are
 * HOST arrays of n_cfg entries (out[c]: device pointer to n_frames[c] floats). */
#define AC_RMS_MULTI_MAX 4
int ac_frame_rms_multi(ac_ctx* ctx, const float* x, int64_t n, int n_cfg, const int* frame, const int* hop,
                       float* const* out, const int64_t* n_frames, void* stream);

/* STFT(n_fft=2048, periodic Hann, center, zero pad) -> |X|^2 -> spectral flatness and/or mel-128
 * power, one pass.  Replaces librosa.feature.spectral_flatness (features_cache.py:183,
 * pure_vocal_pause_detector.py:1117) and the melspectrogram inside librosa.onset.onset_strength
 * (features_cache.py:184, adaptive_vad_enhancer.py:143).  Frame f is centred on sample
 * frame_center[f] (NULL: f*hop) and sees zeros outside [frame_lo[f], frame_hi[f]) (NULL: [0, n)) —
 * that is how the per-chunk calls of features_cache.py:181-187 (chunk-local zero padding) are
 * evaluated for all chunks of a track in one launch.
 * flat_out[n_frames] f32 (may be NULL), mel_out[n_frames*128] f32 frame-major (may be NULL). */
int ac_stft2048_features(ac_ctx* ctx, const float* x, int64_t n, int hop, const int64_t* frame_center,
                         const int64_t* frame_lo, const int64_t* frame_hi, float* flat_out,
                         float* mel_out, int64_t n_frames, void* stream);

/* power_to_db(top_db=80) + lag-1 positive difference + aggregate over 128 mels
 * (librosa.onset.onset_strength: aggregate 0 = mean (features_cache.py:184), 1 = median
 * (adaptive_vad_enhancer.py:143-148)).  mel is processed in `n_groups` independent frame groups
 * (group g = frames [group_start[g], group_start[g+1])) each with its own top_db reference, which is
 * how the per-chunk calls of features_cache.py:181-187 clip.  env_out[n_frames] f32 with the
 * librosa left padding (lag + 1024/hop) applied inside every group.  scratch: n_groups floats. */
int ac_onset_strength(ac_ctx* ctx, const float* mel, int64_t n_frames, const int64_t* group_start,
                      int n_groups, int hop, int aggregate, float* env_out, float* scratch, void* stream);

/* librosa.feature.tempogram(win_length, center, hann, norm=inf) reduced on the fly:
 * mean_out[win] f64 = mean over frames of the normalised autocorrelation (what tempo(aggregate=mean)
 * consumes: adaptive_vad_enhancer.py:61 via beat_track, features_cache.py:289), and
 * argmax_out[n] i32 = argmax_lag(log1p(1e6*tg) + logprior) per frame (tempo(aggregate=None):
 * adaptive_vad_enhancer.py:151, features_cache.py:283).  logprior[win] f64 is supplied by the host.
 * scratch: n_parts*win doubles, n_parts = ac_tempogram_parts(n). */
int ac_tempogram_parts(int64_t n);
int ac_tempogram_reduce(ac_ctx* ctx, const float* env, int64_t n, int win, const double* logprior,
                        double* mean_out, int32_t* argmax_out, double* scratch, void* stream);

/* librosa.yin — the deterministic autocorrelation-F0 stage of librosa.pyin(fmin=C2, fmax=C7, frame 2048, hop 441)
 * (core/pure_vocal_pause_detector.py:422-428, dormant multi-feature branch): centred frames, difference function
 * over tau in [0, max_period] with W = frame_length/2, cumulative-mean normalisation, first trough below
 * `threshold` (else the global minimum), parabolic refinement.  period_out[n_frames] f64 (f0 = sr / period);
 * cmnd_out[n_frames * (max_period - min_period + 1)] f64 frame-major may be NULL.  Precision follows librosa under the
 * reference's pinned numpy < 2: float64 autocorrelation (np.fft works in double there), float32 cumulative-sum
 * energies, float64 series from the difference function on. */
int ac_yin_f0(ac_ctx* ctx, const float* x, int64_t n, int frame_length, int hop, int min_period, int max_period,
              double threshold, double* period_out, double* cmnd_out, int64_t n_frames, void* stream);

/* ---- quiet guard / cut refinement --------------------------------------------------------- */

/* cutting/refine.py:170-174: float64 moving mean of x^2 (np.convolve(...,'same'), window `win`)
 * -> 20*log10(sqrt(ms + 1e-12) + 1e-12).  db_out[n] f64. */
int ac_moving_meansq_db_f64(ac_ctx* ctx, const float* x, int64_t n, int win, double* db_out, void* stream);

/* cutting/refine.py:175-180: next_out[i] = smallest j >= i with db[j] <= floor_db, else -1.
 * scratch: ac_next_leq_scratch(n) int64 elements. */
int64_t ac_next_leq_scratch(int64_t n);
int ac_next_leq_scan(ac_ctx* ctx, const double* db, int64_t n, double floor_db, int64_t* next_out,
                     int64_t* scratch, void* stream);

/* cutting/refine.py:203-207: first argmin of db[start[q] : start[q]+len[q]) for k windows.
 * arg_out[k] i64 (absolute index), val_out[2k] f64 = (db[start], db[argmin]). */
int ac_window_argmin_f64(ac_ctx* ctx, const double* db, int64_t n, const int64_t* start,
                         const int64_t* len, int k, int64_t* arg_out, double* val_out, void* stream);

/* cutting/refine.py:72-110: nearest (fractional) zero crossing to idx[q] within +-half[q] samples;
 * pos_out[k] f64 (NaN = none).  Scalar promotion follows numpy<2 (float64), see DESIGN.md. */
int ac_zero_cross_nearest(ac_ctx* ctx, const float* x, int64_t n, const int64_t* idx, int half,
                          int k, double* pos_out, void* stream);

/* cutting/refine.py:113-157 (slow guard): edge-padded `win`-sample RMS over x[idx : idx+span),
 * 'valid' window sums, dB, first argmin.  arg_out[k] i64 = offset of the minimum (or -1 when the
 * reference returns early), val_out[2k] f64 = (db[0], db[argmin]). */
int ac_quiet_guard_slow(ac_ctx* ctx, const float* x, int64_t n, const int64_t* idx, int span, int win,
                        int k, int64_t* arg_out, double* val_out, void* stream);

/* core/pure_vocal_pause_detector.py:1047-1078: per pause, segment-local 'same' moving RMS (float32
 * semantics, window `win`) argmin over x[a:b), then the look-ahead argmin over x[cut : cut+guard);
 * cut_out[k] i64 = sample index after the look-ahead; aux_out[2k] i64 = (zeros in |x[a:b)|, 1 if
 * x[cut] != 0 else 0) for the silence-floor test at :1080-1083. */
int ac_pause_cut_points(ac_ctx* ctx, const float* x, int64_t n, const int64_t* a, const int64_t* b,
                        int k, int win, int guard, int64_t* cut_out, int64_t* aux_out, void* stream);

/* ---- MDX23 separator front/back end -------------------------------------------------------- */

/* separation/backends.py:306-330 (windowing) + external Conv_TDF_net_trim_model.stft (:355):
 * for item q = (chunk_start[q], chunk_len[q], win_index[q]) build the 261120-sample window with
 * 3072-sample zero margins straight from the resident mono track, reflect-pad, Hann(6144) STFT,
 * hop 1024, keep bins 0..3071.  spec_out[n_items][4][256][3072] f32 (T-major: the U-Net's internal
 * layout; channels L.re, L.im, R.re, R.im of the mono-duplicated input, backends.py:269-270). */
int ac_mdx_stft(ac_ctx* ctx, const float* track, int64_t n, const int64_t* chunk_start,
                const int64_t* chunk_len, const int32_t* win_index, int n_items, float* spec_out,
                float* spec_amax /* [n_items][256], zeroed by the caller, may be NULL: max |spec| per item and frame (see "amax" below) */,
                void* stream);

/* external Conv_TDF_net_trim_model.istft (backends.py:376): spec[n_items][4][256][3072] ->
 * wave_out[n_items][2][261120] f32 (zero top bin, inverse FFT, Hann, overlap-add / window envelope).
 * scratch: n_items*2*256*6144 floats. */
int ac_mdx_istft(ac_ctx* ctx, const float* spec, int n_items, float* wave_out, float* scratch, void* stream);

/* backends.py:377,389-406 + core/enhanced_vocal_separator.py:423-437,456-458: trim the 3072 margins,
 * place the windows, crop, instrumental = mix - vocal, channel mean, then uniform overlap-add of the
 * effective regions of all chunks.  chunk tables: start/len (samples), eff_start/eff_end (absolute),
 * item_base[c] = first item of chunk c.  vocal_out[n], inst_out[n] f32. */
int ac_mdx_assemble_ola(ac_ctx* ctx, const float* track, int64_t n, const float* wave,
                        const int64_t* chunk_start, const int64_t* chunk_len, const int64_t* eff_start,
                        const int64_t* eff_end, const int32_t* item_base, int n_chunks,
                        float* vocal_out, float* inst_out, void* stream);

/* backends.py:389-406 per chunk: the mono vocal of every chunk before the overlap-add (the chunked VAD
 * input, enhanced_vocal_separator.py:412-417).  out = concatenation, chunk c at out_offset[c]. */
int ac_mdx_chunk_vocal(ac_ctx* ctx, const float* wave, const int64_t* chunk_len, const int64_t* out_offset,
                       const int32_t* item_base, int n_chunks, float* out, void* stream);

/* enhanced_vocal_separator.py:490-501: float64 partial sums of x^2 (n_partials blocks, summed by the host). */
int ac_sum_squares(ac_ctx* ctx, const float* x, int64_t n, double* partials, int n_partials, void* stream);

/* src/audio_cut/cutting/beat_candidates.py:97-109 (`_VocalRisk`: sqrt(mean(vocal[c - w : c + w]^2)) / peak per beat candidate): the float64 sums
 * of x^2 over n_windows windows [w_start[i], w_end[i]) (device arrays; every window shorter than 8192 samples) in ONE launch; out[i] is
 * bit-identical to ac_sum_squares over that slice with one partial. */
int ac_window_sum_squares(ac_ctx* ctx, const float* x, int64_t n, const int64_t* w_start, const int64_t* w_end, int n_windows,
                          double* out, void* stream);

/* ---- U-Net layers (MFMA kernels; no MIOpen / rocBLAS path) ----------------------------------- */

/* 3x3 convolution (stride 1, pad 1) of the U-Net, NCHW float32 in/out, on the 16-bit matrix cores with a 3-term
 * float16 hi/lo split (x*w ~= xh*wh + xh*wl + xl*wh; products exact in the float32 MFMA accumulator): float32-class
 * accuracy (~3e-7 of peak per conv) at the f16 MFMA rate.  Replaces the Conv nodes of the graph run at
 * separation/backends.py:358.  w_packed = weights (BatchNorm folded, scaled by a power of two 1/w_unscale)
 * pre-arranged in MFMA fragment order by audio_cut_amd.separation.conv_pack.pack_conv3x3
 * ([C_out/48][C_in/16][5][hi,lo][3][64] fragments of 8 f16).  C_in % 16 == 0, C_out % 48 == 0, H % 8 == 0,
 * W % 32 == 0.  out = conv(x) * w_unscale + bias[c], followed by ReLU when relu != 0.
 *
 * "amax" (every split-float16 kernel below takes the pair): in_amax [B][H] float32 = max |x| of each batch item and row of the
 * time axis H (H = T for the TDF kernels) of the INPUT tensor, as written by the kernel that produced it.  A kernel takes the
 * maximum over exactly the rows that enter one accumulation (the 10 patch rows of a 3x3 conv tile; the single time row of a
 * TDF GEMM row; the two input rows of a 2x2 down-sampling pixel, the one row of an up-sampling pixel) and scales the
 * activations by the power of two that puts it in [2^14, 2^15) before the float16 split (undone exactly in the epilogue): the
 * low part stays a normal float16 down to 2^-17 of the LOCAL peak and the representation error is
 * max(2^-22 |x|, 2^-40 local max) instead of max(2^-22 |x|, 3e-8) - float32-class relative accuracy for quiet items, for
 * decays into silence and for the leakage next to a loud passage (where the quiet guard decides), and no saturation at 65504.
 * A 3x3 conv tile whose ten row maxima span more than 2^12 takes the kernels' row-exact path instead: every patch row staged at its
 * own scale, every output row accumulated at the scale of the loudest of its three input rows (csrc/ac_common.h).
 * out_amax [B][H_out] (zeroed by the caller before the launch) receives max |out| per item and row by ordered-bits atomicMax.
 * Either may be NULL: no scaling (the pre-ABI-2 behaviour) / no reduction. */
int ac_conv3x3_f16x3(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* out, int B, int C_in,
                     int C_out, int H, int W, float w_unscale, int relu, const float* in_amax, float* out_amax, void* stream);

/* The graph's first (4 -> g) and last (g -> 4) 1x1 convolutions (Conv nodes at the two ends of the graph run at
 * separation/backends.py:358; oracle/separator.py:78,94): float32 FMAs at streaming rate.
 * out[b][co][p] = act(bias[co] + sum_ci w[co][ci] x[b][ci][p]); x [B][C_in][P], P % 4 == 0, min(C_in, C_out) <= 8. */
int ac_conv1x1_small(ac_ctx* ctx, const float* x, const float* w, const float* bias, float* out, int B, int C_in, int C_out,
                     long long P, int relu, void* stream);

/* TDF layer of a TFC-TDF block (bias-free Linear over the frequency axis + eval BatchNorm2d over channels + ReLU;
 * the MatMul / BatchNormalization / Relu (/ Add) nodes of the graph run at separation/backends.py:358, restated in
 * oracle/separator.py:_tfc_tdf), same 3-term float16 split as ac_conv3x3_f16x3:
 *   y[m][n] = (resid ? resid[m][n] : 0) + relu(scale[c] * w_unscale * sum_k x[m][k] w[n][k] + shift[c]),  c = (m / T) % C
 * x [M][K] float32 (the NCHW activation as rows (b, c, t)), y / resid [M][N]; w_packed from
 * audio_cut_amd.separation.conv_pack.pack_linear ([N/BN][K/32][hi,lo][BN/16][64] fragments of 8 f16, BN = 192 when
 * N % 192 == 0 else 96).  M = items * C * T with C % 16 == 0, T % 8 == 0 (a workgroup tile is 16 channels x 8 time rows x 192 or
 * 96 columns), K % 32 == 0, N % 96 == 0. */
int ac_tdf_linear_f16x3(ac_ctx* ctx, const float* x, const void* w_packed, const float* scale, const float* shift,
                        const float* resid, float* y, long long M, int N, int K, int T, int C, float w_unscale,
                        const float* in_amax, float* out_amax /* [M / (C*T)][T] */, void* stream);

/* Both TDF layers + residual of a block at the deep levels (F = 384 / 192 / 96, bottleneck Hd = F / 8 <= 48: too narrow for
 * ac_tdf_linear_f16x3), one kernel, exact float32 on v_mfma_f32_16x16x4_f32 (same graph nodes as ac_tdf_linear_f16x3):
 *   y[m][n] = x[m][n] + relu(scale2[c] * sum_j relu(scale1[c] * sum_f x[m][f] w1[j][f] + shift1[c]) w2[n][j] + shift2[c])
 * w1_packed / w2_packed from conv_pack.pack_tdf_small.  M % 32 == 0, F % 32 == 0, x != y.  out_amax [M / (C*T)][T] as above
 * (needs (C * T) % 32 == 0); no in_amax: nothing is split. */
int ac_tdf_small_fused(ac_ctx* ctx, const float* x, const void* w1_packed, const void* w2_packed, const float* scale1,
                       const float* shift1, const float* scale2, const float* shift2, float* y, long long M, int F, int Hd,
                       int T, int C, float* out_amax, void* stream);

/* The U-Net's 2x2 / stride-2 resampling layers (Conv stride 2 / ConvTranspose stride 2 + BatchNormalization + Relu
 * (+ Mul with the encoder skip) nodes of the graph run at separation/backends.py:358; oracle/separator.py:85-91),
 * one fused kernel each, 3-term float16 split on the matrix cores:
 *   down: out[b][co][y][x]         = relu(bias[co] + w_unscale * sum x[b][ci][2y+dy][2x+dx] w[co][ci][dy][dx])
 *   up:   out[b][co][2y+dy][2x+dx] = relu(bias[co] + w_unscale * sum x[b][ci][y][x] w[ci][co][dy][dx]) * skip[...]  (skip may be NULL)
 * x [B][C_in][H][W] float32 NCHW; w_packed = conv_pack.pack_linear(W, bn=96) of W[co][(ci,dy,dx)] (down) or
 * W[(co,dy,dx)][ci] (up), zero padded to N % 96 == 0, K % 32 == 0.  Pixels per image on the GEMM's M axis
 * ((H/2)*(W/2) down, H*W up) % 128 == 0; W % 4 == 0; down: H even, C_in % 8 == 0; up: 4*C_out % 96 == 0.
 * With in_amax / out_amax: down needs (W/2) % 4 == 0, up needs W >= 64 and W % 4 == 0 (a staged quad of pixels is scaled with
 * the maximum of ONE input row, so it must not straddle two rows). */
int ac_down2x_f16x3(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* out, int B, int C_in, int C_out,
                    int H, int W, float w_unscale, const float* in_amax, float* out_amax, void* stream);
int ac_up2x_f16x3(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, const float* skip, float* out, int B,
                  int C_in, int C_out, int H, int W, float w_unscale, const float* in_amax, float* out_amax, void* stream);

/* ---- multi-feature detector branch (pure_vocal_pause_detector.py:410-459,937-1018) ----------- */

/* librosa.pyin after the CMND stage (ac_yin_f0's cmnd_out [n_frames][n_lags]): trough probabilities (beta-distributed
 * thresholds, Boltzmann rank prior, no-trough mass on the global minimum) voted into pitch bins.  Tables come from the
 * host (numpy / scipy values, so the products are bit-identical to librosa's): thresholds[101] = linspace(0,1,101),
 * beta_probs[100], beta_cum[n] = sum(beta_probs[:n]), boltz_fact[N] = (1-e^-2)/(1-e^(-2N)), boltz_exp[k] = e^(-2k).
 * Out: logv [n_frames][n_bins] = log(obs + tiny), logu [n_frames] = log((1 - voiced_prob)/n_bins + tiny), voiced_prob. */
int ac_pyin_observe(ac_ctx* ctx, const double* cmnd, int64_t n_frames, int n_lags, int min_period, double sr, double fmin, int n_bins,
                    int bins_per_semitone, const double* thresholds, const double* beta_probs, const double* beta_cum,
                    const double* boltz_fact, const double* boltz_exp, double no_trough_prob, double tiny_val, double* logv,
                    double* logu, double* voiced_prob, void* stream);
/* librosa.sequence.viterbi over the 2*n_bins pitch/voicing states with the banded pyin transition matrix
 * (lt_same / lt_cross [n_bins][2*half+1] = log(transition + tiny) per destination bin, lt_zero = log(tiny) outside the
 * band, first-index ties).  ptr_scratch [n_frames][2*n_bins] u16.  states [n_frames] i32. */
int ac_pyin_viterbi(ac_ctx* ctx, const double* logv, const double* logu, int64_t n_frames, int n_bins, int half,
                    const double* lt_same, const double* lt_cross, double lt_zero, const double* log_p_init,
                    unsigned short* ptr_scratch, int* states, void* stream);
/* `_extract_formants` (pure_vocal_pause_detector.py:959-1018): per frame pre-emphasis, Burg LPC (librosa.lpc), |1/A| on
 * 512 points, scipy.signal.find_peaks(height = 10 % of max); out_count[f] peaks found, out_mag[f][3] the lowest three. */
int ac_lpc_formants(ac_ctx* ctx, const float* x, int64_t n, int frame_len, int hop, int order, float preemph, int* out_count,
                    double* out_mag, int64_t n_frames, void* stream);
/* librosa.feature.zero_crossing_rate (edge-padded centred frames); out f64 [1 + n/hop]. */
int ac_zero_crossing_rate(ac_ctx* ctx, const float* x, int64_t n, int frame_len, int hop, double* out, int64_t n_frames,
                          void* stream);
/* librosa.feature.spectral_centroid (n_fft 2048) and the low-third magnitude ratio of
 * `_calculate_harmonic_ratio_direct` (pure_vocal_pause_detector.py:936-957), one STFT pass. */
int ac_stft2048_spectral(ac_ctx* ctx, const float* x, int64_t n, int hop, double sr, double* centroid_out, float* ratio_out,
                         int64_t n_frames, void* stream);

/* ---- post-path boundary policy ------------------------------------------------------------ */

/* Framed RMS of every segment [seg_start[s], seg_end[s]) in one launch; a frame sees zeros outside its own segment.
 * center = 1: librosa.feature.rms(y=segment, frame, hop) (centred frames, constant padding; segment s has
 * 1 + len / hop frames) - `_classify_segments_vocal_presence` (core/seamless_splitter.py:2335-2342).
 * center = 0: frames start at seg_start + j * hop (the last one zero padded; the caller picks the count) - the
 * per-chunk VAD windows of detectors/silero_chunk_vad.py.  frame_off[s] = index of segment s's first frame in `out`. */
int ac_segment_frame_rms(ac_ctx* ctx, const float* x, int64_t n, const int64_t* seg_start, const int64_t* seg_end,
                         const int64_t* frame_off, int n_seg, int frame, int hop, int center, float* out, int64_t n_frames,
                         void* stream);
/* `_refine_boundaries_local_valley` (core/seamless_splitter.py:2646-2661): for boundary c the window [c - radius, c + radius)
 * clipped to the signal, float64 'valid' moving mean of x^2 over `win`, dB = 20 log10(sqrt(mean + 1e-12) + 1e-12);
 * orig_db[k] = dB at clip(c - start - win/2), min_db / min_idx[k] = first minimum (index into the 'valid' series;
 * -1 when the window is not longer than `win`).  The search of one boundary is split over ac_local_valley_tiles(radius, win)
 * workgroups (1024 outputs each); part_v / part_i [k * that many] are scratch for their partial minima. */
int ac_local_valley_tiles(int radius, int win);
int ac_local_valley(ac_ctx* ctx, const float* x, int64_t n, const int64_t* centers, int k, int radius, int win, double* orig_db,
                    double* min_db, int64_t* min_idx, double* part_v, int64_t* part_i, void* stream);

/* per-segment sum of x^2 (float64) and peak |x| over [seg_start[s], seg_end[s]) (host bounds inside [0, n]) as 16
 * partials per segment (sumsq / peak are [n_seg][16]; the caller adds / maxes them in order):
 * `_merge_short_weak_human_tails_into_following_music` stats (core/seamless_splitter.py:2179-2196) and the classifier's
 * short-segment branch (:2349-2358). */
int ac_segment_sumsq_peak(ac_ctx* ctx, const float* x, int64_t n, const int64_t* seg_start, const int64_t* seg_end, int n_seg,
                          double* sumsq, float* peak, void* stream);

/* ---- loader / exporter ----------------------------------------------------------- */

/* Rational-rate polyphase FIR resampling in scipy.signal.resample_poly's framing (zero extension, output 0 aligned with input 0,
 * ceil(n * up / down) outputs): the loader's `librosa.load(sr=44100)` at audio_processor.py:45-49 and the VAD's 16 kHz resample at
 * vocal_pause_detector.py:189, both librosa's default soxr_hq.  libsoxr's coefficients are not available offline; the host designs
 * the low-pass to soxr's published HQ specification.  out[m] = sum_q hfull[(m + n_pre_remove) * down - q * up] x[q],
 * float64 accumulation, where hfull = taps * up behind the n_pre_pad leading zeros of resample_poly; it arrives as polyphase rows
 * hp [up][hlen / up] float32, hp[p][t] = hfull[p + t * up] (zero beyond hfull's end; hlen % up == 0).  (ABI 4: rows; ABI <= 3 took hfull.) */
int ac_resample_poly(ac_ctx* ctx, const float* x, int64_t n, int up, int down, const float* hp, int64_t hlen, int64_t n_pre_remove,
                     float* out, int64_t n_out, void* stream);
/* float32 -> little-endian PCM_24 as soundfile.write(subtype="PCM_24") writes it (vocal_smart_splitter/utils/audio_export.py:109-111):
 * libsndfile's clipping conversion (python-soundfile sets SFC_SET_CLIPPING): lrintf(x * 2^31) >> 8, saturating at
 * 0x7FFFFF / 0x800000 (pcm.c f2let_clip_array).  out [3 * n] bytes. */
int ac_pack_pcm24(ac_ctx* ctx, const float* x, int64_t n, unsigned char* out, void* stream);

/* ac_conv3x3_f16x3_s8 (w_packed in its 48-channel layout: conv_pack.pack_conv3x3_w96(w, 48); C_in <= 64) with the graph's first
 * 1x1 convolution (spec [B][C0][H][W], C0 <= 4, w1 [C_in][C0], b1 [C_in], + ReLU; the first Conv + BatchNormalization + Relu
 * nodes at separation/backends.py:358) fused into its loader: a thread's spectrogram float4s are loaded once and the
 * C_in-channel tensor is generated per staged pixel with ac_conv1x1_small's arithmetic (bit-identical), never written to HBM.
 * spec_amax [B][H] = max |spec| per item and row (ac_mdx_stft); the generated tensor's maximum is bounded by
 * spec_amax * amax_gain + amax_offs with amax_gain = max_c sum_j |w1[c][j]|, amax_offs = max_c |b1[c]| (host constants). */
int ac_conv3x3_f16x3_first(ac_ctx* ctx, const float* spec, const float* w1, const float* b1, const void* w_packed,
                           const float* bias, float* out, int B, int C0, int C_in, int C_out, int H, int W, float w_unscale,
                           int relu, const float* spec_amax, float amax_gain, float amax_offs, float* out_amax, void* stream);

/* ac_conv3x3_f16x3 with 96 output channels per workgroup (C_in % 16 == 0, C_out % 96 == 0; weights packed by
 * conv_pack.pack_conv3x3_w96: 8-channel stages, tap 8 of four consecutive stages in one k-step).  Same replaced graph nodes
 * (Conv 3x3 + folded BatchNormalization + Relu of Kim_Vocal_1.onnx, reference backends.py:358), same arithmetic; every staged
 * activation byte feeds twice the MFMAs. */
int ac_conv3x3_f16x3_w96(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* out, int B, int C_in,
                         int C_out, int H, int W, float w_unscale, int relu, const float* in_amax, float* out_amax, void* stream);

/* The 8-channel-stage kernel with 48 output channels per workgroup, three workgroups per CU (C_in % 16 == 0, C_out % 48 == 0;
 * conv_pack.pack_conv3x3_w96(w, cob=48)): the layers ac_conv3x3_f16x3_w96 cannot take. */
int ac_conv3x3_f16x3_s8(ac_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* out, int B, int C_in,
                        int C_out, int H, int W, float w_unscale, int relu, const float* in_amax, float* out_amax, void* stream);


/* ---- Silero VAD (vocal_pause_detector.py:175-296 behind silero_chunk_vad.py:56-117) ----------------------- */

/* ac_resample_poly for n_seg independent segments of one packed buffer in one launch (every chunk's 44.1 kHz -> 16 kHz
 * resampling, vocal_pause_detector.py:189): segment s = x[in_off[s] .. + in_len[s]) is resampled on its own (zero extension at
 * its edges) into out[out_off[s] .. + out_len[s]); out_off[s + 1] - out_off[s] >= out_len[s], the slack (the 4096-sample bucket
 * padding of :192-196) is left untouched - the caller zeroes `out`.  in_off / in_len / out_off / out_len: device int64 [n_seg],
 * out_off increasing; n_out_total = out_off[n_seg - 1] + padded length of the last segment.  h as for ac_resample_poly. */
int ac_resample_poly_segments(ac_ctx* ctx, const float* x, const int64_t* in_off, const int64_t* in_len, const int64_t* out_off,
                              const int64_t* out_len, int n_seg, int up, int down, const float* h, int64_t hlen,
                              int64_t n_pre_remove, float* out, int64_t n_out_total, void* stream);

/* The Silero VAD v5 16 kHz network up to the LSTM's input gates, for n_windows independent windows (8 per workgroup):
 * window w = 64 samples of context + 512 samples starting at x16[win_start[w]] (a chunk's FIRST window is flagged by
 * win_start = -(index) - 1: its context is zeros), reflect-padded by 64, STFT as a stride-128 convolution with
 * forward_basis_buffer (basis_t [256][258], transposed), magnitude, Conv1d(129,128,3) / (128,64,3,s2) / (64,64,3,s2) /
 * (64,128,3) each + ReLU (weights as [c_in * 3 + tap][c_out]), then gates_x[w][512] = weight_ih feat + bias_ih + bias_hh
 * (wih_t [128][512], bias_sum [512]).  x16 must be readable 64 samples before every non-first window and 576 after its start. */
int ac_silero_frontend(ac_ctx* ctx, const float* x16, const int64_t* win_start, int n_windows, const float* basis_t,
                       const float* c1, const float* b1, const float* c2, const float* b2, const float* c3, const float* b3,
                       const float* c4, const float* b4, const float* wih_t, const float* bias_sum, float* gates_x, void* stream);

/* LSTMCell(128, 128) over the windows of each chunk (state zero at the chunk's first window, as silero_vad's
 * get_speech_timestamps resets it per call): chunk s owns windows seg_first_window[s] .. + seg_window_count[s] of gates_x;
 * h_out[w][128] = hidden state after window w.  whh_t [128][512] = weight_hh transposed.  One workgroup per chunk. */
int ac_silero_lstm(ac_ctx* ctx, const float* gates_x, const int* seg_first_window, const int* seg_window_count, int n_seg,
                   const float* whh_t, float* h_out, void* stream);

/* probs[w] = sigmoid(b_out + sum_j w_out[j] relu(h[w][j])): the decoder's ReLU + Conv1d(128, 1, 1) + Sigmoid. */
int ac_silero_out(ac_ctx* ctx, const float* h, const float* w_out, float b_out, int n_windows, float* probs, void* stream);

/* ---- host-side sequential helper (runs on the CPU; pointers are HOST pointers) ------------- */

/* librosa.beat.__beat_track_dp: the O(n * period) dynamic programme over the local score
 * (adaptive_vad_enhancer.py:61, features_cache.py:289).  backlink_h[n] i64, cumscore_h[n] f64. */
int ac_host_beat_dp(const double* localscore_h, int64_t n, double period, double tightness,
                    int64_t* backlink_h, double* cumscore_h);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOCUT_HIP_H */
