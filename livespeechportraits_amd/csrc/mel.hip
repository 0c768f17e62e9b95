// gfx950: the mel front-end (include/lspmel.h) -- every window of an utterance in one batch, two fp32 MFMA GEMMs.
//   mel_gather     window i starts at int(i * 133.33..) (utils.py:74); the reflect pad of 189 and the position of the 266-tap
//                  window inside the 512-point frame reduce to x[n] = clip[|n - 66|], n = 0..265 (audio_funcs.py:59-68)
//   gemm_f32       [nwin][268] x [514][268]^T : real and imaginary parts of bins 0..256 (window folded into the basis)
//   mel_magnitude  sqrt(re^2 + im^2) -> [nwin][260]
//   gemm_f32       x [80][260]^T : the filterbank
//   mel_log        log(max(x, 1e-5)), (x - log 1e-5) / -log 1e-5
#include "../../include/lspmel.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "gemm_f32.h"

namespace lspmel {

constexpr int KA = 268;        // 266 taps, padded to a multiple of 4
constexpr int NB = 257;        // bins
constexpr int KM = 260;        // 257 bins, padded
constexpr int PAD = 189;       // (n_fft - hop) / 2
constexpr int OFF = 123;       // (n_fft - win_length) / 2: where torch.stft centres the window in the frame

__global__ __launch_bounds__(256) void mel_gather(const float *audio, long long nsamples, int nwin, float *A)
{
    const int i = blockIdx.x, n = threadIdx.x + blockIdx.y * 256;
    if (i >= nwin || n >= KA) return;
    float v = 0.f;
    if (n < LSPMEL_WIN) {
        const long long st = (long long)((double)i * (16000.0 * (0.5 / 60)));      // int(i * mel_frame_step), same double product
        int k = n + OFF - PAD;                                                       // index into the 266-sample clip
        k = k < 0 ? -k : k;                                                          // reflect (no edge repeat); the right edge is never reached
        const long long s = st + k;
        v = s < nsamples ? audio[s] : 0.f;                                           // zero padding of a short last clip (utils.py:76-77)
    }
    A[(size_t)i * KA + n] = v;
}

__global__ __launch_bounds__(256) void mel_magnitude(const float *C, int nwin, float *mag)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nwin * KM) return;
    const int i = idx / KM, k = idx - i * KM;
    float v = 0.f;
    if (k < NB) {
        const float re = C[(size_t)i * (2 * NB) + k], im = C[(size_t)i * (2 * NB) + NB + k];
        v = sqrtf(re * re + im * im);
    }
    mag[idx] = v;
}

__global__ __launch_bounds__(256) void mel_log(float *mel, int total)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const float min_mel = -11.512925464970229f;                // math.log(1e-5)
    mel[idx] = (logf(fmaxf(mel[idx], 1e-5f)) - min_mel) / -min_mel;
}

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// librosa 0.7.0 filters.mel(16000, 512, 80, 90, 7600), htk=False, norm=1: Slaney scale, area-normalised triangles (double, rounded once)
static void slaney_filterbank(std::vector<float> &w)
{
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    auto hz2mel = [&](double f) { return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp; };
    auto mel2hz = [&](double m) { return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m; };
    const int nm = LSPMEL_N_MELS;
    std::vector<double> mel_f(nm + 2);
    const double lo = hz2mel(90.0), hi = hz2mel(7600.0);
    for (int i = 0; i < nm + 2; ++i) {
        // np.linspace(lo, hi, nm + 2): lo + i * step, the last point set to hi exactly
        const double step = (hi - lo) / (nm + 1);
        mel_f[i] = mel2hz(i == nm + 1 ? hi : lo + i * step);
    }
    w.assign((size_t)nm * KM, 0.f);
    for (int i = 0; i < nm; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < NB; ++k) {
            const double f = k == NB - 1 ? 8000.0 : k * (8000.0 / (NB - 1));       // np.linspace(0, sr / 2, 257)
            const double lower = -(mel_f[i] - f) / fd0, upper = (mel_f[i + 2] - f) / fd1;
            const double t = std::fmax(0.0, std::fmin(lower, upper));
            // weights is float32: the triangle is stored rounded; `weights *= enorm[:, None]` then multiplies in double and rounds again
            w[(size_t)i * KM + k] = (float)((double)(float)t * enorm);
        }
    }
}

}  // namespace lspmel

using namespace lspmel;

extern "C" {

const char *lspmel_last_error(void) { return g_err.c_str(); }

int lspmel_num_windows(int64_t nsamples)
{
    if (nsamples < 0) return 0;
    return 2 * (int)((double)nsamples / 16000 * 60);
}

size_t lspmel_basis_floats(void) { return (size_t)2 * NB * KA + (size_t)LSPMEL_N_MELS * KM; }

int lspmel_make_basis(float *host_blob, size_t nfloats)
{
    if (!host_blob || nfloats < lspmel_basis_floats()) return fail(LSPMEL_ERR_INVALID_ARGUMENT, "basis buffer missing or too small");
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < NB; ++k)
        for (int n = 0; n < KA; ++n) {
            double c = 0.0, s = 0.0;
            if (n < LSPMEL_WIN) {
                // torch.hann_window(266) is periodic and float32: the window value is rounded to float32 before it multiplies
                const double w = (double)(float)(0.5 - 0.5 * std::cos(2.0 * pi * n / LSPMEL_WIN));
                // exact argument reduction: k * (n + 123) mod 512
                const int ph = (int)(((long long)k * (n + OFF)) % LSPMEL_N_FFT);
                const double a = 2.0 * pi * ph / LSPMEL_N_FFT;
                c = w * std::cos(a);
                s = -w * std::sin(a);
            }
            host_blob[(size_t)k * KA + n] = (float)c;
            host_blob[(size_t)(NB + k) * KA + n] = (float)s;
        }
    std::vector<float> fb;
    slaney_filterbank(fb);
    for (size_t i = 0; i < fb.size(); ++i) host_blob[(size_t)2 * NB * KA + i] = fb[i];
    return LSPMEL_OK;
}

size_t lspmel_workspace_bytes(int nwindows)
{
    if (nwindows < 1) return 0;
    return align256((size_t)nwindows * KA * 4) + align256((size_t)nwindows * 2 * NB * 4) + align256((size_t)nwindows * KM * 4);
}

int lspmel_compute(const float *audio_dev, int64_t nsamples, const float *basis_dev, int nwindows, float *mel_dev,
                   void *workspace_dev, size_t workspace_bytes, void *hip_stream)
{
    if (!audio_dev || !basis_dev || !mel_dev || !workspace_dev) return fail(LSPMEL_ERR_INVALID_ARGUMENT, "null argument");
    if (nwindows < 1 || nwindows != lspmel_num_windows(nsamples)) return fail(LSPMEL_ERR_SHAPE, "nwindows must equal lspmel_num_windows(nsamples) and be >= 1");
    if (workspace_bytes < lspmel_workspace_bytes(nwindows)) return fail(LSPMEL_ERR_SHAPE, "workspace smaller than lspmel_workspace_bytes()");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    char *w = static_cast<char *>(workspace_dev);
    float *A = reinterpret_cast<float *>(w);
    float *C = reinterpret_cast<float *>(w + align256((size_t)nwindows * KA * 4));
    float *mag = reinterpret_cast<float *>(w + align256((size_t)nwindows * KA * 4) + align256((size_t)nwindows * 2 * NB * 4));
    hipLaunchKernelGGL(mel_gather, dim3(nwindows, (KA + 255) / 256), dim3(256), 0, s, audio_dev, (long long)nsamples, nwindows, A);
    lspgemm::GemmParams g1{A, basis_dev, nullptr, nullptr, nullptr, C, nwindows, 2 * NB, KA, 1.0f, 0};
    hipError_t e = lspgemm::launch_gemm_f32(g1, s);
    if (e != hipSuccess) return fail(LSPMEL_ERR_HIP, std::string("DFT gemm launch: ") + hipGetErrorString(e));
    hipLaunchKernelGGL(mel_magnitude, dim3((nwindows * KM + 255) / 256), dim3(256), 0, s, C, nwindows, mag);
    lspgemm::GemmParams g2{mag, basis_dev + (size_t)2 * NB * KA, nullptr, nullptr, nullptr, mel_dev, nwindows, LSPMEL_N_MELS, KM, 1.0f, 0};
    e = lspgemm::launch_gemm_f32(g2, s);
    if (e != hipSuccess) return fail(LSPMEL_ERR_HIP, std::string("filterbank gemm launch: ") + hipGetErrorString(e));
    hipLaunchKernelGGL(mel_log, dim3((nwindows * LSPMEL_N_MELS + 255) / 256), dim3(256), 0, s, mel_dev, nwindows * LSPMEL_N_MELS);
    e = hipGetLastError();
    return e == hipSuccess ? LSPMEL_OK : fail(LSPMEL_ERR_HIP, std::string("mel kernels: ") + hipGetErrorString(e));
}

}  // extern "C"
