// Host-side logic of the drop-in: window tables, Stft.Config / Mel.Config
// construction and validation, the integer frame grid, boundary indexing and
// FIR design.  Everything here is float64 / integer host arithmetic that the
// reference also runs once per configuration on the CPU (SURVEY 8a rows a1-a4,
// a11); the results are uploaded as tables for the HIP kernels.  Error messages
// are the reference's, verbatim, because its tests match them as strings.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <limits>

#include <algorithm>

#include "smx_internal.hpp"

namespace smx {

std::string format(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  va_list ap2;
  va_copy(ap2, ap);
  int len = vsnprintf(nullptr, 0, fmt, ap);
  va_end(ap);
  std::string out((size_t)(len > 0 ? len : 0), '\0');
  if (len > 0) vsnprintf(&out[0], (size_t)len + 1, fmt, ap2);
  va_end(ap2);
  return out;
}

// ---- Window (window.ml) ---------------------------------------------------------

namespace {

// window.ml:147-164 cosine_fill: sum_k a_k cos(k theta_i), theta_i = (2i-(m-1)) pi/(m-1);
// harmonics by the Chebyshev recurrence from ONE cosine per sample; both mirrored
// slots are written from the same evaluation; slots >= len are dropped (:135-136).
void cosine_fill(double *buf, int64_t len, const double *coef, int order, int64_t m) {
  const double step = M_PI / (double)(m - 1);
  for (int64_t i = 0; i <= (m - 1) / 2; ++i) {
    const double c = std::cos((double)(2 * i - (m - 1)) * step);
    double acc = coef[0] + coef[1] * c;
    double previous = 1.0, current = c;
    for (int k = 2; k < order; ++k) {
      const double t = 2.0 * c * current - previous;
      acc = acc + coef[k] * t;
      previous = current;
      current = t;
    }
    if (i < len) buf[i] = acc;
    if (m - 1 - i < len) buf[m - 1 - i] = acc;
  }
}

}  // namespace

namespace {
// the first half evaluated, every value stored at i and m - 1 - i; slots at or past `len` fall off (window.ml:121-136)
template <typename F>
void mirror_fill(double *out, int64_t len, int64_t m, F value) {
  for (int64_t i = 0; i <= (m - 1) / 2; ++i) {
    const double v = value(i);
    if (i < len) out[i] = v;
    if (m - 1 - i < len) out[m - 1 - i] = v;
  }
}
}  // namespace

void window_make(int kind, bool periodic, int64_t n, double *out) { window_make_param(kind, 0.0, periodic, n, out); }

// the shape parameter: Kaiser beta, Gaussian standard deviation (samples), Tukey taper fraction (window.ml:77-97)
void window_make_param(int kind, double param, bool periodic, int64_t n, double *out) {
  if (kind == SMX_WINDOW_KAISER && !(std::isfinite(param) && param >= 0.0))
    throw InvalidArgument(format("make: cannot use a kaiser window with beta %g (beta must be finite and non-negative)", param));
  if (kind == SMX_WINDOW_GAUSSIAN && !(std::isfinite(param) && param > 0.0))
    throw InvalidArgument(format(
        "make: cannot use a gaussian window with standard deviation %g (standard deviation must be finite and positive)", param));
  if (kind == SMX_WINDOW_TUKEY && !(param >= 0.0 && param <= 1.0))
    throw InvalidArgument(format("make: cannot use a tukey window with taper %g (taper must lie in [0, 1])", param));
  if (n < 1)
    throw InvalidArgument(format(
        "make: cannot make a %lld-point window (length must be at least 1)", (long long)n));
  if (n == 1) {  // window.ml:369-371
    out[0] = 1.0;
    return;
  }
  const int64_t m = periodic ? n + 1 : n;
  static const double hann[] = {0.5, 0.5};
  static const double hamming[] = {0.54, 0.46};
  static const double blackman[] = {0.42, 0.5, 0.08};
  static const double bh[] = {0.35875, 0.48829, 0.14128, 0.01168};
  static const double nuttall[] = {0.3635819, 0.4891775, 0.1365995, 0.0106411};
  static const double flat_top[] = {0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368};
  const double last = (double)(m - 1);
  if (kind == SMX_WINDOW_TUKEY) {               // window.ml:352-355: the ends of the family are the other two windows
    if (param <= 0.0) kind = SMX_WINDOW_RECTANGULAR;
    else if (param >= 1.0) kind = SMX_WINDOW_HANN;
  }
  switch (kind) {
    case SMX_WINDOW_RECTANGULAR:
      for (int64_t i = 0; i < n; ++i) out[i] = 1.0;
      return;
    case SMX_WINDOW_BARTLETT:                   // window.ml:166-174
      mirror_fill(out, n, m, [&](int64_t i) { return 2.0 * (double)i / last; });
      return;
    case SMX_WINDOW_GAUSSIAN: {                 // window.ml:176-186
      const double half = last / 2.0, scale = -1.0 / (2.0 * param * param);
      mirror_fill(out, n, m, [&](int64_t i) {
        const double x = (double)i - half;
        return std::exp(x * x * scale);
      });
      return;
    }
    case SMX_WINDOW_TUKEY: {                    // window.ml:188-205
      const int64_t width = (int64_t)std::floor(param * last / 2.0);
      const double step = 2.0 / param / last;
      mirror_fill(out, n, m, [&](int64_t i) {
        return i <= width ? 0.5 * (1.0 + std::cos(M_PI * (-1.0 + step * (double)i))) : 1.0;
      });
      return;
    }
    case SMX_WINDOW_KAISER: {                   // window.ml:296-318; I0 by its power series throughout (bessel_i0 below:
      const double alpha = last / 2.0;          // the reference's minimax branches agree with it to a few ulp)
      const double denominator = bessel_i0(param);
      mirror_fill(out, n, m, [&](int64_t i) {
        const double r = ((double)i - alpha) / alpha;
        const double inner = 1.0 - r * r;
        return bessel_i0(param * std::sqrt(inner > 0.0 ? inner : 0.0)) / denominator;
      });
      return;
    }
    case SMX_WINDOW_HANN: cosine_fill(out, n, hann, 2, m); return;
    case SMX_WINDOW_HAMMING: cosine_fill(out, n, hamming, 2, m); return;
    case SMX_WINDOW_BLACKMAN: cosine_fill(out, n, blackman, 3, m); return;
    case SMX_WINDOW_BLACKMAN_HARRIS: cosine_fill(out, n, bh, 4, m); return;
    case SMX_WINDOW_NUTTALL: cosine_fill(out, n, nuttall, 4, m); return;
    case SMX_WINDOW_FLAT_TOP: cosine_fill(out, n, flat_top, 5, m); return;
    default:
      throw InvalidArgument(format("make: unknown window family %d", kind));
  }
}

// window.ml:407-434 cola: the periodic window's shifts by `hop` sum to a constant within 1e-10 of their mean
bool window_cola(int kind, double param, int64_t length, int64_t hop) {
  if (length < 1)
    throw InvalidArgument(format("cola: cannot check overlap-add of a %lld-point window (length must be at least 1)",
                                 (long long)length));
  if (hop < 1 || hop > length)
    throw InvalidArgument(format("cola: cannot check overlap-add at hop %lld (hop must lie in [1, %lld])", (long long)hop,
                                 (long long)length));
  std::vector<double> w((size_t)length), sums((size_t)hop, 0.0);
  window_make_param(kind, param, true, length, w.data());
  for (int64_t i = 0; i < length; ++i) sums[(size_t)(i % hop)] += w[(size_t)i];
  double mean = 0.0;
  for (double v : sums) mean += v;
  mean /= (double)hop;
  if (!(mean > 0.0)) return false;
  for (double v : sums)
    if (!(std::fabs(v - mean) <= 1e-10 * mean)) return false;
  return true;
}

// ---- Stft.Config (stft.ml:61-111) -------------------------------------------------

smx_stft_config *stft_config_create(int64_t fft_size, int64_t win_length, int64_t hop,
                                    int alignment, int pad, double pad_value, int scale,
                                    int window_kind, const double *custom_window) {
  if (fft_size < 1)
    throw InvalidArgument(format(
        "create: cannot use an FFT of size %lld (fft_size must be at least 1)", (long long)fft_size));
  if (win_length == SMX_DEFAULT) win_length = fft_size;
  if (win_length < 1 || win_length > fft_size)
    throw InvalidArgument(format(
        "create: cannot use a %lld-point window with an FFT of size %lld (win_length must lie in "
        "[1, fft_size])",
        (long long)win_length, (long long)fft_size));
  if (hop == SMX_DEFAULT) hop = fft_size / 4 > 1 ? fft_size / 4 : 1;
  if (hop < 1)
    throw InvalidArgument(format(
        "create: cannot advance frames by %lld samples (hop must be at least 1)", (long long)hop));
  if (alignment < SMX_ALIGN_CENTERED || alignment > SMX_ALIGN_RIGHT)
    throw InvalidArgument(format("create: unknown alignment %d", alignment));
  if (pad < SMX_PAD_REFLECT || pad > SMX_PAD_EDGE)
    throw InvalidArgument(format("create: unknown pad mode %d", pad));
  if (scale < SMX_SCALE_NONE || scale > SMX_SCALE_PSD)
    throw InvalidArgument(format("create: unknown scale %d", scale));

  std::vector<double> coefficients((size_t)win_length);
  if (window_kind == SMX_WINDOW_CUSTOM) {
    if (!custom_window) throw InvalidArgument("create: custom window table is missing");
    for (int64_t i = 0; i < win_length; ++i) coefficients[(size_t)i] = custom_window[i];
  } else {
    try {
      window_make(window_kind, true, win_length, coefficients.data());
    } catch (const InvalidArgument &e) {  // relabel with this entry point (stft.ml:86-96)
      std::string m = e.what();
      size_t colon = m.find(':');
      throw InvalidArgument(colon == std::string::npos ? "create: " + m : "create" + m.substr(colon));
    }
  }
  auto *c = new smx_stft_config();
  c->fft_size = fft_size;
  c->win_length = win_length;
  c->hop = hop;
  c->alignment = alignment;
  c->pad = pad;
  c->pad_value = pad_value;
  c->scale = scale;
  c->window_kind = window_kind;
  c->analysis_window.assign((size_t)fft_size, 0.0);
  const int64_t left = (fft_size - win_length) / 2;  // stft.ml:100
  for (int64_t i = 0; i < win_length; ++i) c->analysis_window[(size_t)(left + i)] = coefficients[(size_t)i];
  if (scale == SMX_SCALE_MAGNITUDE) {  // stft.ml:106-107
    double sum = 0.0;
    for (double v : c->analysis_window) sum += v;
    for (double &v : c->analysis_window) v = v / sum;
  } else if (scale == SMX_SCALE_PSD) {  // stft.ml:108-109
    double sum = 0.0;
    for (double v : c->analysis_window) sum += v * v;
    const double root = std::sqrt(sum);
    for (double &v : c->analysis_window) v = v / root;
  }
  return c;
}

}  // namespace smx

int64_t smx_stft_config::left_width() const {
  switch (alignment) {
    case SMX_ALIGN_CENTERED: return fft_size / 2;
    case SMX_ALIGN_LEFT: return 0;
    default: return fft_size - 1;
  }
}

int64_t smx_stft_config::right_width() const {
  return alignment == SMX_ALIGN_CENTERED ? fft_size / 2 : 0;
}

int64_t smx_stft_config::frames(int64_t n) const {  // stft.ml:217-223
  if (n < 0)
    throw smx::InvalidArgument(smx::format(
        "frames: cannot analyse a signal of length %lld (length must be non-negative)", (long long)n));
  if (n == 0) return 0;
  const int64_t padded = n + left_width() + right_width();
  if (padded < fft_size) return 0;
  return 1 + (padded - fft_size) / hop;
}

namespace smx {

int64_t stft_first_complete(const smx_stft_config &c) {
  return (c.left_width() + c.hop - 1) / c.hop;
}

// ---- least-squares synthesis bookkeeping (stft.ml:708-889) --------------------------------------
namespace {
std::vector<double> folded_square_window(const smx_stft_config &c) {   // stft.ml:712-720, j ascending
  std::vector<double> folded((size_t)c.hop, 0.0);
  for (int64_t j = 0; j < c.fft_size; ++j) {
    const double w = c.analysis_window[(size_t)j];
    folded[(size_t)(j % c.hop)] += w * w;
  }
  return folded;
}
inline double guard(double v) { return v == 0.0 ? 1.0 : v; }   // stft.ml:836
inline int64_t ceil_div(int64_t a, int64_t b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }
}  // namespace

bool stft_nola(const smx_stft_config &c) {
  if (c.hop > c.fft_size) return false;
  const std::vector<double> folded = folded_square_window(c);
  double lo = folded[0], hi = 0.0;
  for (double v : folded) {
    if (v < lo) lo = v;
    if (v > hi) hi = v;
  }
  return lo > 1e-10 * hi;
}

int64_t stft_output_length(const smx_stft_config &c, int64_t frames) {
  if (frames == 0) return 0;
  return (frames - 1) * c.hop + c.fft_size - c.left_width() - c.right_width();
}

void stft_envelope(const smx_stft_config &c, int64_t frames, std::vector<double> &head, std::vector<double> &period,
                   std::vector<double> &tail, int64_t &head_n, int64_t &stop) {
  const int64_t fft = c.fft_size, hop = c.hop;
  const int64_t span = (frames - 1) * hop + fft;
  head_n = span < fft - hop ? span : fft - hop;
  int64_t interior_end = span < frames * hop ? span : frames * hop;
  stop = head_n > interior_end ? head_n : interior_end;
  auto partial = [&](int64_t q) {   // stft.ml:844-853: p ascending, capped at the last frame
    const int64_t first = std::max<int64_t>(0, ceil_div(q - fft + 1, hop));
    const int64_t last = std::min<int64_t>(frames - 1, q / hop);
    double total = 0.0;
    for (int64_t p = first; p <= last; ++p) {
      const double w = c.analysis_window[(size_t)(q - p * hop)];
      total += w * w;
    }
    return guard(total);
  };
  head.resize((size_t)head_n);
  for (int64_t q = 0; q < head_n; ++q) head[(size_t)q] = partial(q);
  const std::vector<double> folded = folded_square_window(c);
  period.resize((size_t)hop);
  for (int64_t r = 0; r < hop; ++r) period[(size_t)r] = guard(folded[(size_t)r]);
  tail.resize((size_t)(span - stop));
  for (int64_t q = stop; q < span; ++q) tail[(size_t)(q - stop)] = partial(q);
}

int64_t stft_last_complete(const smx_stft_config &c, int64_t n) {
  if (n < 0)
    throw InvalidArgument(format(
        "last_complete: cannot analyse a signal of length %lld (length must be non-negative)",
        (long long)n));
  const int64_t total = c.frames(n);
  if (n == 0) return 0;
  const int64_t reach = n + c.left_width() - c.fft_size;
  if (reach < 0) return 0;
  const int64_t k = reach / c.hop + 1;
  return total < k ? total : k;
}

int64_t source_index(const smx_stft_config &c, int64_t n, int64_t q) {
  if (q >= 0 && q < n) return q;
  switch (c.pad) {
    case SMX_PAD_REFLECT: {  // stft.ml:300-305
      if (n == 1) return 0;
      const int64_t period = 2 * (n - 1);
      const int64_t m = ((q % period) + period) % period;
      return m < n ? m : period - m;
    }
    case SMX_PAD_EDGE: return q < 0 ? 0 : n - 1;
    default: return -1;
  }
}

// ---- Mel.Config (mel.ml:39-164, convert.ml:70-102) -----------------------------------

namespace {

const double kFsp = 200.0 / 3.0;            // convert.ml:72
const double kMinLogHz = 1000.0;            // convert.ml:74
const double kMinLogMel = kMinLogHz / kFsp; // convert.ml:76
const double kLogStep = std::log(6.4) / 27.0;  // convert.ml:78

}  // namespace

double hz_to_mel(double f, int scale) {  // convert.ml:80-90
  if (scale == SMX_MEL_HTK) return std::log(f / 700.0 + 1.0) * (2595.0 / std::log(10.0));
  if (f < kMinLogHz) return f / kFsp;
  return std::log(f / kMinLogHz) / kLogStep + kMinLogMel;
}

double mel_to_hz(double m, int scale) {  // convert.ml:92-102
  if (scale == SMX_MEL_HTK) return (std::exp(m * (std::log(10.0) / 2595.0)) - 1.0) * 700.0;
  if (m < kMinLogMel) return m * kFsp;
  return std::exp((m - kMinLogMel) * kLogStep) * kMinLogHz;
}

smx_mel_config *mel_config_create(int64_t n_mels, int64_t sample_rate, int64_t fft_size,
                                  double f_min, bool has_f_max, double f_max, int scale, int norm) {
  if (n_mels < 1)
    throw InvalidArgument(format("create: cannot build %lld mel bands (n_mels must be at least 1)",
                                 (long long)n_mels));
  if (sample_rate < 1)
    throw InvalidArgument(format(
        "create: cannot use a sample rate of %lld Hz (sample_rate must be at least 1)",
        (long long)sample_rate));
  if (fft_size < 1)
    throw InvalidArgument(format(
        "create: cannot use an FFT of size %lld (fft_size must be at least 1)", (long long)fft_size));
  if (!(std::isfinite(f_min) && f_min >= 0.0))
    throw InvalidArgument(format(
        "create: cannot start the filterbank at %g Hz (f_min must be finite and non-negative)", f_min));
  const double nyquist = (double)sample_rate / 2.0;
  if (!has_f_max) f_max = nyquist;
  if (!(std::isfinite(f_max) && f_max > f_min))
    throw InvalidArgument(format(
        "create: cannot span [%g, %g] Hz (f_max must be finite and greater than f_min)", f_min, f_max));
  if (f_max > nyquist)
    throw InvalidArgument(format(
        "create: cannot extend the filterbank to %.17g Hz at a sample rate of %lld Hz (f_max must "
        "not exceed the Nyquist frequency %g)",
        f_max, (long long)sample_rate, nyquist));
  if (scale != SMX_MEL_SLANEY && scale != SMX_MEL_HTK)
    throw InvalidArgument(format("create: unknown mel scale %d", scale));
  if (norm != SMX_NORM_SLANEY && norm != SMX_NORM_NONE)
    throw InvalidArgument(format("create: unknown mel norm %d", norm));

  // mel.ml:67-117 weights_of
  const int64_t bins = fft_size / 2 + 1;
  const int64_t count = n_mels + 2;
  const double step = 1.0 / ((double)fft_size * (1.0 / (double)sample_rate));  // mel.ml:39-43
  const double mel_min = hz_to_mel(f_min, scale), mel_max = hz_to_mel(f_max, scale);
  const double mstep = (mel_max - mel_min) / (double)(count - 1);               // mel.ml:50-60
  std::vector<double> points((size_t)count);
  for (int64_t i = 0; i < count; ++i)
    points[(size_t)i] = mel_to_hz(i == count - 1 ? mel_max : (double)i * mstep + mel_min, scale);
  std::vector<double> steps((size_t)count - 1);
  for (int64_t i = 0; i + 1 < count; ++i) {
    steps[(size_t)i] = points[(size_t)i + 1] - points[(size_t)i];
    if (steps[(size_t)i] <= 0.0)
      throw InvalidArgument(format(
          "create: cannot resolve %lld mel bands between %g and %g Hz (adjacent breakpoints "
          "collapse in double precision)",
          (long long)n_mels, f_min, f_max));
  }
  auto *c = new smx_mel_config();
  c->f_min = f_min;
  c->f_max = f_max;
  c->scale = scale;
  c->norm = norm;
  c->n_mels = n_mels;
  c->sample_rate = sample_rate;
  c->fft_size = fft_size;
  c->weights.assign((size_t)(n_mels * bins), 0.0);
  for (int64_t m = 0; m < n_mels; ++m) {
    double support = 0.0;
    for (int64_t b = 0; b < bins; ++b) {
      const double freq = (double)b * step;
      const double lower = (-(points[(size_t)m] - freq)) / steps[(size_t)m];
      const double upper = (points[(size_t)m + 2] - freq) / steps[(size_t)m + 1];
      double w = lower < upper ? lower : upper;
      if (!(w > 0.0)) w = 0.0;
      c->weights[(size_t)(m * bins + b)] = w;
      if (w > support) support = w;
    }
    if (support <= 0.0) {
      delete c;
      throw InvalidArgument(format(
          "create: cannot support %lld mel bands with an FFT of size %lld (at least one filter "
          "spans no FFT bin; raise fft_size or lower n_mels)",
          (long long)n_mels, (long long)fft_size));
    }
  }
  if (norm == SMX_NORM_SLANEY) {  // mel.ml:108-117
    for (int64_t m = 0; m < n_mels; ++m) {
      const double enorm = 2.0 / (points[(size_t)m + 2] - points[(size_t)m]);
      for (int64_t b = 0; b < bins; ++b) c->weights[(size_t)(m * bins + b)] *= enorm;
    }
  }
  return c;
}

// ---- FIR design (model: resample.ml:105-163) ---------------------------------------

double kaiser_beta(double att) {
  if (att > 50.0) return 0.1102 * (att - 8.7);
  if (att > 21.0) return 0.5842 * std::pow(att - 21.0, 0.4) + 0.07886 * (att - 21.0);
  return 0.0;
}

double bessel_i0(double x) {
  const double hx2 = 0.25 * x * x;
  double term = 1.0, sum = 1.0;
  for (int k = 1;; ++k) {
    term = term * hx2 / (double)((int64_t)k * k);
    sum = sum + term;
    if (term <= std::numeric_limits<double>::epsilon() * sum || k > 1000) return sum;
  }
}

void design_lowpass(int64_t taps, double fc, double beta, double *h) {
  if (taps < 1)
    throw InvalidArgument(format("design_lowpass: cannot design %lld taps (taps must be at least 1)",
                                 (long long)taps));
  if (!(fc > 0.0 && fc <= 1.0))
    throw InvalidArgument(format(
        "design_lowpass: cannot place the cutoff at %g (cutoff must lie in (0, 1] Nyquist units)", fc));
  const double centre = (double)(taps - 1) / 2.0;
  const double i0_beta = bessel_i0(beta);
  double sum = 0.0;
  for (int64_t i = 0; i < taps; ++i) {
    const double z = (double)i - centre;
    const double s = z == 0.0 ? fc : std::sin(M_PI * fc * z) / (M_PI * z);
    const double r = centre > 0 ? z / centre : 0.0;
    const double inner = 1.0 - r * r;
    h[i] = s * (bessel_i0(beta * std::sqrt(inner > 0.0 ? inner : 0.0)) / i0_beta);
    sum += h[i];
  }
  const double gain = 1.0 / sum;
  for (int64_t i = 0; i < taps; ++i) h[i] = h[i] * gain;
}

// ---- Chroma.Config (chroma.ml:95-222) ------------------------------------------------------------------
namespace {
double round_half_even(double x) {   // chroma.ml:93-99: ties to the even neighbour
  const double below = std::floor(x), fraction = x - below;
  if (fraction > 0.5) return below + 1.0;
  if (fraction < 0.5) return below;
  return std::fmod(below, 2.0) == 0.0 ? below : below + 1.0;
}
}  // namespace

smx_chroma_config *chroma_config_create(int64_t n_chroma, double tuning, double ctroct, bool has_octwidth,
                                        double octwidth, bool base_c, int64_t sample_rate, int64_t fft_size) {
  if (n_chroma < 1)
    throw InvalidArgument(format("create: cannot build %lld chroma bands (n_chroma must be at least 1)",
                                 (long long)n_chroma));
  if (sample_rate < 1)
    throw InvalidArgument(format("create: cannot use a sample rate of %lld Hz (sample_rate must be at least 1)",
                                 (long long)sample_rate));
  if (fft_size < 1)
    throw InvalidArgument(format("create: cannot use an FFT of size %lld (fft_size must be at least 1)",
                                 (long long)fft_size));
  if (!std::isfinite(tuning))
    throw InvalidArgument(format("create: cannot shift the scale by %g bins (tuning must be finite)", tuning));
  if (!std::isfinite(ctroct))
    throw InvalidArgument(format("create: cannot centre the octave envelope at %g (ctroct must be finite)", ctroct));
  if (has_octwidth && !(std::isfinite(octwidth) && octwidth > 0.0))
    throw InvalidArgument(format(
        "create: cannot use an octave envelope of half-width %g (octwidth must be finite and positive)", octwidth));

  // chroma.ml:109-175 weights_of, scalar float64 in the reference's order
  const int64_t bins = fft_size / 2 + 1;
  const double chroma = (double)n_chroma;
  const double a440 = 440.0 * std::pow(2.0, tuning / chroma) / 16.0;
  const double step = (double)sample_rate / (double)fft_size;
  auto position = [&](int64_t j) { return chroma * std::log2((double)j * step / a440); };
  std::vector<double> positions((size_t)fft_size), widths((size_t)fft_size);
  for (int64_t j = 0; j < fft_size; ++j)
    positions[(size_t)j] = j == 0 ? position(1) - 1.5 * chroma : position(j);   // bin 0 carries no frequency
  for (int64_t j = 0; j < fft_size; ++j)
    widths[(size_t)j] = j == fft_size - 1 ? 1.0 : std::max(positions[(size_t)j + 1] - positions[(size_t)j], 1.0);
  const double half = round_half_even(chroma / 2.0);
  std::vector<double> w((size_t)(n_chroma * bins), 0.0);
  for (int64_t c = 0; c < n_chroma; ++c)
    for (int64_t j = 0; j < bins; ++j) {
      const double d = positions[(size_t)j] - (double)c;
      const double v = std::fmod(d + half + 10.0 * chroma, chroma);
      const double wrapped = (v < 0.0 ? v + chroma : v) - half;
      const double spread = 2.0 * wrapped / widths[(size_t)j];
      w[(size_t)(c * bins + j)] = std::exp(-0.5 * spread * spread);
    }
  for (int64_t j = 0; j < bins; ++j) {   // unit euclidean columns
    double sum = 0.0;
    for (int64_t c = 0; c < n_chroma; ++c) sum += w[(size_t)(c * bins + j)] * w[(size_t)(c * bins + j)];
    double length = std::sqrt(sum);
    if (length < 2.2250738585072014e-308) length = 1.0;
    for (int64_t c = 0; c < n_chroma; ++c) w[(size_t)(c * bins + j)] = w[(size_t)(c * bins + j)] / length;
  }
  if (has_octwidth)
    for (int64_t j = 0; j < bins; ++j) {
      const double offset = (positions[(size_t)j] / chroma - ctroct) / octwidth;
      const double envelope = std::exp(-0.5 * offset * offset);
      for (int64_t c = 0; c < n_chroma; ++c) w[(size_t)(c * bins + j)] = w[(size_t)(c * bins + j)] * envelope;
    }
  smx_chroma_config *cfg = new smx_chroma_config();
  cfg->n_chroma = n_chroma;
  cfg->tuning = tuning;
  cfg->ctroct = ctroct;
  cfg->has_octwidth = has_octwidth;
  cfg->octwidth = has_octwidth ? octwidth : 0.0;
  cfg->base_c = base_c;
  cfg->sample_rate = sample_rate;
  cfg->fft_size = fft_size;
  if (!base_c) {
    cfg->weights = std::move(w);
  } else {   // row 0 is C rather than A
    const int64_t shift = 3 * (n_chroma / 12);
    cfg->weights.resize(w.size());
    for (int64_t c = 0; c < n_chroma; ++c)
      std::copy_n(&w[(size_t)(((c + shift) % n_chroma) * bins)], (size_t)bins, &cfg->weights[(size_t)(c * bins)]);
  }
  return cfg;
}

}  // namespace smx
