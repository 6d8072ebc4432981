// Context, error handling and host-side helpers of libaudiocut_hip.so.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <vector>

#include "ac_common.h"

static thread_local char g_err[512] = "";

void ac_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ac_last_error(void) { return g_err; }
extern "C" int ac_abi_version(void) { return AC_ABI_VERSION; }

// ---- Slaney mel scale (librosa.filters.mel(htk=False, norm="slaney"), published algorithm) ----
static double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

static void build_mel(double sr, int n_fft, int n_mels, std::vector<float>& w, std::vector<int>& lo, std::vector<int>& hi) {
    const int nb = n_fft / 2 + 1;
    std::vector<double> mel_f(n_mels + 2);
    const double m0 = hz_to_mel(0.0), m1 = hz_to_mel(sr / 2);
    for (int i = 0; i < n_mels + 2; ++i) {
        // np.linspace(m0, m1, n): start + i*step, last element forced to stop
        double step = (m1 - m0) / (n_mels + 1);
        double v = (i == n_mels + 1) ? m1 : m0 + i * step;
        mel_f[i] = mel_to_hz(v);
    }
    w.assign((size_t)n_mels * nb, 0.f);
    lo.assign(n_mels, 0);
    hi.assign(n_mels, 0);
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        int l = nb, h = 0;
        for (int k = 0; k < nb; ++k) {
            const double fk = (double)k * sr / n_fft;
            const double lower = -(mel_f[i] - fk) / fd0;
            const double upper = (mel_f[i + 2] - fk) / fd1;
            double v = fmin(lower, upper);
            if (v < 0) v = 0;
            float vf = (float)v;          // weights array is float32 before the slaney scaling
            vf = vf * (float)enorm;       // `weights *= enorm[:, None]` in float32
            w[(size_t)i * nb + k] = vf;
            if (vf != 0.f) {
                if (k < l) l = k;
                if (k + 1 > h) h = k + 1;
            }
        }
        if (h <= l) { l = 0; h = 0; }
        lo[i] = l;
        hi[i] = h;
    }
}

template <typename T>
static int upload(T** dst, const std::vector<T>& v) {
    AC_CHECK_HIP(hipMalloc((void**)dst, v.size() * sizeof(T)));
    AC_CHECK_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return AC_OK;
}

// everything after the calloc: a failure leaves a partly filled context that the caller (ac_ctx_create) destroys
static int ctx_fill(ac_ctx* c, int device) {
    AC_CHECK_HIP(hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device));
    if (c->n_cu <= 0) c->n_cu = 256;
    int rc;
    {
        std::vector<double2> tw(1024);
        for (int k = 0; k < 1024; ++k) { double a = -2.0 * M_PI * k / 2048.0; tw[k] = make_double2(cos(a), sin(a)); }
        if ((rc = upload(&c->tw2048, tw))) return rc;
        std::vector<double> hw(2048);
        for (int n = 0; n < 2048; ++n) hw[n] = 0.5 - 0.5 * cos(2.0 * M_PI * n / 2048.0);
        if ((rc = upload(&c->hann2048, hw))) return rc;
    }
    {
        std::vector<float> w; std::vector<int> lo, hi;
        build_mel(44100.0, 2048, 128, w, lo, hi);
        if ((rc = upload(&c->mel_w, w))) return rc;
        if ((rc = upload(&c->mel_lo, lo))) return rc;
        if ((rc = upload(&c->mel_hi, hi))) return rc;
    }
    {
        std::vector<float2> tw(3072);
        for (int k = 0; k < 3072; ++k) { double a = -2.0 * M_PI * k / 6144.0; tw[k] = make_float2((float)cos(a), (float)sin(a)); }
        if ((rc = upload(&c->tw6144, tw))) return rc;
        std::vector<float> hw(6144);
        for (int n = 0; n < 6144; ++n) hw[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / 6144.0));
        if ((rc = upload(&c->hann6144, hw))) return rc;
        // window envelope of torch.istft over the 256-frame lattice (float32 accumulation like torch)
        const int T = 256, H = 1024, N = 6144, P = H * (T - 1) + N;
        std::vector<float> env(P, 0.f);
        for (int t = 0; t < T; ++t)
            for (int n = 0; n < N; ++n) env[t * H + n] += hw[n] * hw[n];
        if ((rc = upload(&c->ola_env6144, env))) return rc;
    }
    return AC_OK;
}

extern "C" int ac_ctx_create(int device, ac_ctx** out) {
    AC_REQUIRE(out != nullptr, "out pointer");
    *out = nullptr;
    int count = 0;
    AC_CHECK_HIP(hipGetDeviceCount(&count));
    AC_REQUIRE(device >= 0 && device < count, "device index");
    int previous = -1;
    (void)hipGetDevice(&previous);                 // the caller's current device is restored on every path
    AC_CHECK_HIP(hipSetDevice(device));
    ac_ctx* c = (ac_ctx*)calloc(1, sizeof(ac_ctx));
    int rc = AC_OK;
    if (!c) { ac_set_error("out of host memory"); rc = AC_E_NOMEM; }
    else {
        c->device = device;
        rc = ctx_fill(c, device);
        if (rc != AC_OK) { ac_ctx_destroy(c); c = nullptr; }     // hipFree(nullptr) is a no-op: frees whatever was uploaded
    }
    if (previous >= 0 && previous != device) (void)hipSetDevice(previous);
    if (rc == AC_OK) *out = c;
    return rc;
}

extern "C" int ac_ctx_destroy(ac_ctx* c) {
    if (!c) return AC_OK;
    int previous = -1;
    (void)hipGetDevice(&previous);
    (void)hipSetDevice(c->device);
    (void)hipFree(c->tw2048); (void)hipFree(c->hann2048); (void)hipFree(c->mel_w); (void)hipFree(c->mel_lo); (void)hipFree(c->mel_hi);
    (void)hipFree(c->tw6144); (void)hipFree(c->hann6144); (void)hipFree(c->ola_env6144);
    if (previous >= 0 && previous != c->device) (void)hipSetDevice(previous);
    free(c);
    return AC_OK;
}

// ---- host-side DP of librosa's beat tracker (sequential by construction) -----------------------
extern "C" int ac_host_beat_dp(const double* localscore, int64_t n, double period, double tightness,
                               int64_t* backlink, double* cumscore) {
    AC_REQUIRE(localscore && backlink && cumscore, "null pointer");
    AC_REQUIRE(n > 0 && period > 0 && tightness > 0, "n, period, tightness must be positive");
    // window = arange(-2*period, -round(period/2) + 1) as integers
    const int64_t w0 = (int64_t)(-2 * period);
    const int64_t w1 = (int64_t)(-nearbyint(period / 2)) + 1;   // exclusive
    const int64_t wn = w1 - w0;
    AC_REQUIRE(wn > 0, "empty search window");
    std::vector<double> txwt(wn);
    for (int64_t j = 0; j < wn; ++j) {
        double l = log(-(double)(w0 + j) / period);
        txwt[j] = -tightness * (l * l);           // numpy's association (-tightness * log(.) ** 2): (-tightness * l) * l differs by an ulp and breaks ties the other way
    }
    double maxscore = localscore[0];
    for (int64_t i = 1; i < n; ++i) if (localscore[i] > maxscore) maxscore = localscore[i];
    const double thresh = 0.01 * maxscore;
    bool first_beat = true;
    for (int64_t i = 0; i < n; ++i) {
        // z_pad = max(0, min(-(w0 + i), wn)): predecessors before time 0 keep the bare transition cost
        int64_t zpad = -(w0 + i);
        if (zpad > wn) zpad = wn;
        if (zpad < 0) zpad = 0;
        double best = -INFINITY;
        int64_t best_j = 0;
        for (int64_t j = 0; j < wn; ++j) {
            double cand = txwt[j];
            if (j >= zpad) cand += cumscore[i + w0 + j];
            if (cand > best) { best = cand; best_j = j; }
        }
        cumscore[i] = localscore[i] + best;
        if (first_beat && localscore[i] < thresh) {
            backlink[i] = -1;
        } else {
            backlink[i] = i + w0 + best_j;
            first_beat = false;
        }
    }
    return AC_OK;
}
