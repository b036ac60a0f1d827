// Host-side construction (fp64) of the constant tables the MFCC kernels read.
//
// These restate the published definitions of the third-party routines the reference calls at
// extract_features_construct_dataset.py:27,30 (librosa.load / librosa.feature.mfcc, librosa <= 0.9
// defaults; resampy 0.2.x 'kaiser_best'; scipy.fftpack.dct).  Pure C++ (no HIP) so that the CPU test
// suite can fetch them through lipasr_debug_table() and compare with the oracle's NumPy tables.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace lipasr {
namespace tables {

constexpr int kNFft = 2048, kHop = 512, kNMels = 128, kNMfcc = 20, kSr = 22050, kNBins = 1025;
constexpr int kKbZeros = 64, kKbTable = 512;  // resampy kaiser_best: 64 zero crossings, 2^9 table steps
constexpr double kKbRolloff = 0.9475937167399596, kKbBeta = 14.769656459379492;
constexpr double kPi = 3.14159265358979323846;

inline double bessel_i0(double x) {
  // power series; converges for the |x| <= 15 used here to 1 ulp-level relative accuracy
  const double q = x * x / 4.0;
  double term = 1.0, sum = 1.0;
  for (int k = 1; k < 200; ++k) {
    term *= q / ((double)k * (double)k);
    sum += term;
    if (term < 1e-18 * sum) break;
  }
  return sum;
}

inline double sinc(double x) {
  if (x == 0.0) return 1.0;
  const double px = kPi * x;
  return std::sin(px) / px;
}

// resampy.filters.sinc_window(64, 9, kaiser(beta), rolloff): half window of 512*64+1 samples
inline std::vector<double> kaiser_best_half_window() {
  const int n = kKbTable * kKbZeros;
  std::vector<double> w(n + 1);
  const double i0b = bessel_i0(kKbBeta);
  for (int t = 0; t <= n; ++t) {
    const double pos = (double)kKbZeros * (double)t / (double)n;  // np.linspace(0, 64, n+1)
    const double s = kKbRolloff * sinc(kKbRolloff * pos);
    const double r = (double)t / (double)n;                        // np.kaiser(2n+1, beta)[n:]
    const double arg = 1.0 - r * r;
    const double taper = bessel_i0(kKbBeta * std::sqrt(arg > 0.0 ? arg : 0.0)) / i0b;
    w[t] = taper * s;
  }
  return w;
}

inline int gcd_int(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

struct Polyphase {
  int up = 1, down = 1;   // sr_new/sr_orig = up/down in lowest terms; output t = up*q + p
  int taps = 0;           // taps per phase (left wing + right wing), padded to a multiple of 4
  int left = 0;           // taps [0, left) apply to x[n - (left-1) + k]
  std::vector<float> h;   // [up][taps]
  std::vector<int> n_off; // [up]: floor(p*down/up)
};

// resampy.resample_f restated per output phase: for ratio up/down every `up` outputs see the same
// fractional positions, so the interpolated taps  win[off + i*step] + eta*delta[off + i*step]
// are tabulated once (computed in fp64, stored fp32).
inline Polyphase build_polyphase(int sr_orig, int sr_new) {
  Polyphase pp;
  const int g = gcd_int(sr_orig, sr_new);
  pp.up = sr_new / g;
  pp.down = sr_orig / g;
  std::vector<double> win = kaiser_best_half_window();
  const double ratio = (double)sr_new / (double)sr_orig;
  if (ratio < 1.0)
    for (double& v : win) v *= ratio;
  const int nwin = (int)win.size();
  std::vector<double> delta(nwin, 0.0);
  for (int i = 0; i + 1 < nwin; ++i) delta[i] = win[i + 1] - win[i];
  const double scale = ratio < 1.0 ? ratio : 1.0;
  const int step = (int)(scale * kKbTable);
  const int wing = nwin / step;  // max over offsets of (nwin - offset) / step: taps per wing
  pp.left = wing;
  pp.taps = (2 * wing + 3) & ~3;
  pp.h.assign((size_t)pp.up * pp.taps, 0.0f);
  pp.n_off.assign(pp.up, 0);
  for (int p = 0; p < pp.up; ++p) {
    const long long num = (long long)p * pp.down;
    const int n = (int)(num / pp.up);
    const double fr = (double)(num % pp.up) / (double)pp.up;  // time_register - n, exact rational
    pp.n_off[p] = n;
    float* row = pp.h.data() + (size_t)p * pp.taps;
    // left wing: x[n - i]
    double frac = scale * fr;
    double index_frac = frac * kKbTable;
    int offset = (int)index_frac;
    double eta = index_frac - offset;
    int i_max = (nwin - offset) / step;
    for (int i = 0; i < i_max && i < wing; ++i) {
      const double w = win[offset + i * step] + eta * delta[offset + i * step];
      row[(wing - 1) - i] = (float)w;  // k = left-1-i  <->  x[n - i]
    }
    // right wing: x[n + 1 + k]
    frac = scale - frac;
    index_frac = frac * kKbTable;
    offset = (int)index_frac;
    eta = index_frac - offset;
    int k_max = (nwin - offset) / step;
    for (int k = 0; k < k_max && k < wing; ++k) {
      const double w = win[offset + k * step] + eta * delta[offset + k * step];
      row[wing + k] = (float)w;  // x[n + 1 + k]
    }
  }
  return pp;
}

inline std::vector<double> hann_periodic_f64(int n_fft) {
  std::vector<double> w(n_fft);
  for (int n = 0; n < n_fft; ++n) w[n] = 0.5 - 0.5 * std::cos(2.0 * kPi * (double)n / (double)n_fft);
  return w;
}
inline std::vector<float> hann_periodic(int n_fft = kNFft) {
  std::vector<double> d = hann_periodic_f64(n_fft);
  return std::vector<float>(d.begin(), d.end());
}

// exp(-2 pi i k / 2048), interleaved (cos, sin)
inline std::vector<float> twiddles() {
  std::vector<float> t(2 * kNFft);
  for (int k = 0; k < kNFft; ++k) {
    const double a = -2.0 * kPi * (double)k / (double)kNFft;
    t[2 * k] = (float)std::cos(a);
    t[2 * k + 1] = (float)std::sin(a);
  }
  return t;
}

inline double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
inline double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// librosa.filters.mel(sr=22050, n_fft=2048, n_mels=128, fmin=0, fmax=sr/2, htk=False, norm='slaney'),
// dense float32 [128][1 + n_fft/2].  The bin frequencies are librosa <= 0.9's fft_frequencies,
// np.linspace(0, sr/2, 1 + n_fft//2) -- also for an odd n_fft (Speaker recognition, n_fft = 441), where it
// differs from the true k*sr/n_fft grid; the reference's era of librosa is what is restated.
inline std::vector<float> mel_dense(int n_fft = kNFft) {
  const int n_bins = 1 + n_fft / 2;
  std::vector<float> W((size_t)kNMels * n_bins, 0.0f);
  std::vector<double> fftfreqs(n_bins), mel_f(kNMels + 2);
  const double fmax = (double)kSr / 2.0;
  for (int i = 0; i < n_bins; ++i) fftfreqs[i] = fmax * (double)i / (double)(n_bins - 1);
  const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(fmax);
  for (int i = 0; i < kNMels + 2; ++i) mel_f[i] = mel_to_hz(m_lo + (m_hi - m_lo) * (double)i / (double)(kNMels + 1));
  for (int i = 0; i < kNMels; ++i) {
    const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
    const float enorm = (float)(2.0 / (mel_f[i + 2] - mel_f[i]));
    for (int b = 0; b < n_bins; ++b) {
      const double lower = -(mel_f[i] - fftfreqs[b]) / fd0;
      const double upper = (mel_f[i + 2] - fftfreqs[b]) / fd1;
      double v = lower < upper ? lower : upper;
      if (v < 0.0) v = 0.0;
      W[(size_t)i * n_bins + b] = (float)v * enorm;
    }
  }
  return W;
}

struct MelSparse {
  std::vector<int> start, len, off;  // [128]
  std::vector<float> w;              // concatenated non-zero runs
};
inline MelSparse mel_sparse(int n_fft = kNFft) {
  const int n_bins = 1 + n_fft / 2;
  MelSparse s;
  std::vector<float> W = mel_dense(n_fft);
  s.start.assign(kNMels, 0); s.len.assign(kNMels, 0); s.off.assign(kNMels, 0);
  for (int i = 0; i < kNMels; ++i) {
    int lo = n_bins, hi = -1;
    for (int b = 0; b < n_bins; ++b)
      if (W[(size_t)i * n_bins + b] != 0.0f) { if (b < lo) lo = b; hi = b; }
    s.off[i] = (int)s.w.size();
    if (hi >= lo) {
      s.start[i] = lo;
      s.len[i] = hi - lo + 1;
      for (int b = lo; b <= hi; ++b) s.w.push_back(W[(size_t)i * n_bins + b]);
    }
  }
  return s;
}

// Two-filters-per-bin form of the same bank: bin k feeds filter lo[k] with weight wlo[k] and filter lo[k]+1
// with weight whi[k]; lo is non-decreasing, so the bins of filter m form the run [start[m], start[m]+len[m]):
//   mel[m] = sum_{k in run(m)} wlo[k] P[k] + sum_{k in run(m-1)} whi[k] P[k].
// ok = false if the bank does not have that structure (then the CSR form is used).
struct MelPairs {
  std::vector<float> wlo, whi;  // [1025]
  std::vector<int> start, len;  // [128]
  bool ok = true;
};
inline MelPairs mel_pairs() {
  MelPairs mp;
  std::vector<float> W = mel_dense();
  mp.wlo.assign(kNBins, 0.0f); mp.whi.assign(kNBins, 0.0f);
  mp.start.assign(kNMels, 0); mp.len.assign(kNMels, 0);
  std::vector<int> lo(kNBins, 0);
  int prev = 0;
  for (int b = 0; b < kNBins; ++b) {
    int first = -1, count = 0;
    for (int m = 0; m < kNMels; ++m)
      if (W[(size_t)m * kNBins + b] != 0.0f) { if (first < 0) first = m; ++count; }
    if (count == 0) { lo[b] = prev; continue; }
    if (count > 2 || first < prev) { mp.ok = false; return mp; }
    if (count == 2 && W[(size_t)(first + 1) * kNBins + b] == 0.0f) { mp.ok = false; return mp; }
    lo[b] = first;
    mp.wlo[b] = W[(size_t)first * kNBins + b];
    if (count == 2) mp.whi[b] = W[(size_t)(first + 1) * kNBins + b];
    prev = first;
  }
  for (int m = 0; m < kNMels; ++m) { mp.start[m] = kNBins; }
  for (int b = 0; b < kNBins; ++b) {
    const int m = lo[b];
    if (mp.len[m] == 0) mp.start[m] = b;
    mp.len[m]++;
  }
  for (int m = 0; m < kNMels; ++m)
    if (mp.len[m] == 0) mp.start[m] = 0;
  return mp;
}

// Windowed real-DFT matrix for an STFT evaluated as a contraction (n_fft <= 510, any parity), folded on the
// symmetry of a real input under a symmetric window (periodic Hann: w[N-n] = w[n]):
//   Re X[b] =  sum_{n=0}^{N/2} c_n w[n] cos(2 pi n b / N) * (x[n] + x[(N-n) mod N])
//   Im X[b] = -sum_{n=0}^{N/2} c_n w[n] sin(2 pi n b / N) * (x[n] - x[(N-n) mod N])
// with c_n = 1/2 where n pairs with itself (n = 0, and n = N/2 for even N) and 1 elsewhere: half the rows of the
// plain DFT matrix.  Layout: T[n][t*64 + j] = re weight, T[n][t*64 + 32 + j] = im weight, b = 32 t + j; rows
// n > N/2 and bins b > N/2 are zero; k_rows rows of n_tiles*64 floats.  Angles are reduced exactly (n*b mod N)
// before the fp64 evaluation.
inline std::vector<float> dft_table(int n_fft, int k_rows, int n_tiles) {
  const int ld = n_tiles * 64, n_bins = 1 + n_fft / 2;
  std::vector<float> T((size_t)k_rows * ld, 0.0f);
  std::vector<double> w = hann_periodic_f64(n_fft);
  for (int n = 0; n <= n_fft / 2; ++n) {
    const double c = (n == 0 || 2 * n == n_fft) ? 0.5 : 1.0;
    for (int b = 0; b < n_bins; ++b) {
      const long long r = ((long long)n * b) % n_fft;
      const double ang = 2.0 * kPi * (double)r / (double)n_fft;
      const int t = b >> 5, j = b & 31;
      T[(size_t)n * ld + t * 64 + j] = (float)(c * w[n] * std::cos(ang));
      T[(size_t)n * ld + t * 64 + 32 + j] = (float)(-c * w[n] * std::sin(ang));
    }
  }
  return T;
}

// rows 0..19 of scipy.fftpack.dct(type=2, norm='ortho') over 128 points, float32 [20][128]
inline std::vector<float> dct_matrix() {
  std::vector<float> D((size_t)kNMfcc * kNMels);
  for (int k = 0; k < kNMfcc; ++k)
    for (int n = 0; n < kNMels; ++n) {
      double d = std::cos(kPi * (double)k * (double)(2 * n + 1) / (2.0 * kNMels)) * std::sqrt(2.0 / kNMels);
      if (k == 0) d *= std::sqrt(0.5);
      D[(size_t)k * kNMels + n] = (float)d;
    }
  return D;
}

// librosa.load -> resampy: int(n * ratio) samples, then librosa.resample's fix_length to ceil(n * ratio)
inline void resampled_lengths(int n_samp, int sr_orig, int sr_new, int* n_valid, int* n_fixed) {
  if (sr_orig == sr_new) { *n_valid = n_samp; *n_fixed = n_samp; return; }
  const double ratio = (double)sr_new / (double)sr_orig;
  *n_valid = (int)((double)n_samp * ratio);
  *n_fixed = (int)std::ceil((double)n_samp * ratio);
}

}  // namespace tables
}  // namespace lipasr
