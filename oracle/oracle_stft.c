/* CPU oracle, C restatement -- TEST INFRASTRUCTURE ONLY (see oracle/soundml_oracle.py).
 *
 * Restates the reference's STFT power path in plain C with the reference's
 * float64 interior so it can be timed on all host cores as the `cpu_baseline`
 * of bench.py (kind "port") and used as a second checker:
 *
 *   to_double samples                      stft.ml:345-346
 *   frame p = padded[p*hop, p*hop + fft)   stft.ml:356-364 (Nx.stft, third-party:
 *   x float64 window -> float64 rfft         pinned nx fork, dune-project:22-26; restated
 *   -> one rounding to the storage dtype     by its published semantics)
 *   boundary extension reflect/edge/const  stft.ml:300-338
 *   |z| (storage dtype) then ^power        stft.ml:670-674
 *   mel: W_f64 x S_f64, one rounding        mel.ml:231
 *
 * Pinned against the reference's golden vectors in tests/test_oracle_goldens.py
 * (through ctypes).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 *
 * Build: make -C oracle   (gcc -O3 -pthread, no external libraries).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
  int n;            /* real FFT length (power of two, >= 2) */
  int m;            /* n / 2 */
  int logm;
  double *tw_re, *tw_im;     /* exp(-2 pi i j / m), j < m/2 */
  double *pw_re, *pw_im;     /* exp(-2 pi i k / n), k <= m  */
  int *rev;                  /* bit reversal of m points */
} rfft_plan;

static rfft_plan *plan_create(int n) {
  rfft_plan *p = (rfft_plan *)calloc(1, sizeof(rfft_plan));
  p->n = n;
  p->m = n / 2;
  p->logm = 0;
  while ((1 << p->logm) < p->m) p->logm++;
  int half = p->m / 2 > 0 ? p->m / 2 : 1;
  p->tw_re = (double *)malloc(sizeof(double) * half);
  p->tw_im = (double *)malloc(sizeof(double) * half);
  for (int j = 0; j < half; ++j) {
    double a = -2.0 * M_PI * (double)j / (double)p->m;
    p->tw_re[j] = cos(a);
    p->tw_im[j] = sin(a);
  }
  p->pw_re = (double *)malloc(sizeof(double) * (p->m + 1));
  p->pw_im = (double *)malloc(sizeof(double) * (p->m + 1));
  for (int k = 0; k <= p->m; ++k) {
    double a = -2.0 * M_PI * (double)k / (double)n;
    p->pw_re[k] = cos(a);
    p->pw_im[k] = sin(a);
  }
  p->rev = (int *)malloc(sizeof(int) * p->m);
  for (int i = 0; i < p->m; ++i) {
    int r = 0;
    for (int b = 0; b < p->logm; ++b) r |= ((i >> b) & 1) << (p->logm - 1 - b);
    p->rev[i] = r;
  }
  return p;
}

static void plan_destroy(rfft_plan *p) {
  if (!p) return;
  free(p->tw_re); free(p->tw_im); free(p->pw_re); free(p->pw_im); free(p->rev); free(p);
}

/* real forward DFT of xw[n] -> re/im[0..m], kernel exp(-2 pi i k t / n); float64.
 * Even/odd packing into one m-point complex FFT (radix-2, decimation in time). */
static void rfft_forward(const rfft_plan *p, const double *xw, double *zr, double *zi, double *re, double *im) {
  const int m = p->m;
  for (int i = 0; i < m; ++i) {
    zr[p->rev[i]] = xw[2 * i];
    zi[p->rev[i]] = xw[2 * i + 1];
  }
  for (int half = 1; half < m; half <<= 1) {
    const int step = (m / 2) / half;
    for (int base = 0; base < m; base += 2 * half) {
      for (int j = 0; j < half; ++j) {
        const double wr = p->tw_re[j * step], wi = p->tw_im[j * step];
        const int i0 = base + j, i1 = i0 + half;
        const double tr = wr * zr[i1] - wi * zi[i1];
        const double ti = wr * zi[i1] + wi * zr[i1];
        zr[i1] = zr[i0] - tr; zi[i1] = zi[i0] - ti;
        zr[i0] = zr[i0] + tr; zi[i0] = zi[i0] + ti;
      }
    }
  }
  for (int k = 0; k <= m; ++k) {
    const int k0 = k % m, k1 = (m - k) % m;
    const double ar = zr[k0], ai = zi[k0], br = zr[k1], bi = -zi[k1];      /* B = conj Z[m-k] */
    const double er = ar + br, ei = ai + bi, dr = ar - br, di = ai - bi;
    const double wr = p->pw_re[k], wi = p->pw_im[k];
    /* X = (E - i w D) / 2 */
    re[k] = 0.5 * (er + wr * di + wi * dr);
    im[k] = 0.5 * (ei - wr * dr + wi * di);
  }
}

static double fetch(const float *x32, const double *x64, int64_t n, int64_t s, int pad, double pad_value) {
  if (s < 0 || s >= n) {
    if (pad == 0) {                       /* reflect, stft.ml:300-305 */
      if (n == 1) s = 0;
      else {
        const int64_t period = 2 * (n - 1);
        int64_t mm = s % period;
        if (mm < 0) mm += period;
        s = mm < n ? mm : period - mm;
      }
    } else if (pad == 2) {                /* edge */
      s = s < 0 ? 0 : n - 1;
    } else {
      return pad_value;
    }
  }
  return x32 ? (double)x32[s] : x64[s];
}

typedef struct {
  const float *x32; const double *x64;
  int64_t lead, n; int fft, hop; const double *window; int64_t left; int pad; double pad_value;
  double power; int64_t frames;
  float *out32; double *out64; int complex_out;
  int thread, threads;
} job_t;

static void *worker(void *arg) {
  job_t *j = (job_t *)arg;
  const int n = j->fft, m = n / 2, bins = m + 1;
  rfft_plan *p = plan_create(n);
  double *xw = (double *)malloc(sizeof(double) * n);
  double *zr = (double *)malloc(sizeof(double) * (m > 0 ? m : 1)), *zi = (double *)malloc(sizeof(double) * (m > 0 ? m : 1));
  double *re = (double *)malloc(sizeof(double) * bins), *im = (double *)malloc(sizeof(double) * bins);
  /* frames are produced FB at a time into a [bins][FB] block so the [bins; frames]
   * output is written in runs of FB values (plain cache blocking, same arithmetic) */
  enum { FB = 16 };
  double *bre = (double *)malloc(sizeof(double) * bins * FB), *bim = (double *)malloc(sizeof(double) * bins * FB);
  for (int64_t clip = j->thread; clip < j->lead; clip += j->threads) {
    const float *x32 = j->x32 ? j->x32 + clip * j->n : NULL;
    const double *x64 = j->x64 ? j->x64 + clip * j->n : NULL;
    for (int64_t f0 = 0; f0 < j->frames; f0 += FB) {
      const int nf = (int)(j->frames - f0 < FB ? j->frames - f0 : FB);
      for (int ff = 0; ff < nf; ++ff) {
        const int64_t s0 = (f0 + ff) * j->hop - j->left;
        if (s0 >= 0 && s0 + n <= j->n) {          /* interior frame: same arithmetic, no index mapping */
          if (x32) for (int i = 0; i < n; ++i) xw[i] = (double)x32[s0 + i] * j->window[i];
          else for (int i = 0; i < n; ++i) xw[i] = x64[s0 + i] * j->window[i];
        } else {
          for (int i = 0; i < n; ++i) xw[i] = fetch(x32, x64, j->n, s0 + i, j->pad, j->pad_value) * j->window[i];
        }
        rfft_forward(p, xw, zr, zi, re, im);
        for (int k = 0; k < bins; ++k) { bre[k * FB + ff] = re[k]; bim[k * FB + ff] = im[k]; }
      }
      for (int k = 0; k < bins; ++k) {
        for (int ff = 0; ff < nf; ++ff) {
          const int64_t o = (clip * bins + k) * j->frames + f0 + ff;
          const double rk = bre[k * FB + ff], ik = bim[k * FB + ff];
          if (j->out32) {
            const float r32 = (float)rk, i32 = (float)ik;   /* one rounding to complex64 */
            if (j->complex_out) { j->out32[2 * o] = r32; j->out32[2 * o + 1] = i32; continue; }
            const float mag = (float)sqrt((double)r32 * (double)r32 + (double)i32 * (double)i32);
            j->out32[o] = j->power == 2.0 ? mag * mag : (j->power == 1.0 ? mag : (float)pow((double)mag, j->power));
          } else {
            if (j->complex_out) { j->out64[2 * o] = rk; j->out64[2 * o + 1] = ik; continue; }
            const double mag = hypot(rk, ik);
            j->out64[o] = j->power == 2.0 ? mag * mag : (j->power == 1.0 ? mag : pow(mag, j->power));
          }
        }
      }
    }
  }
  free(bre); free(bim);
  free(xw); free(zr); free(zi); free(re); free(im);
  plan_destroy(p);
  return NULL;
}

/* Returns 0 on success, -1 for an unsupported fft (power of two >= 2 only). */
static int run(job_t base, int threads) {
  if (base.fft < 2 || (base.fft & (base.fft - 1))) return -1;
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t tid[256];
  job_t jobs[256];
  for (int t = 0; t < threads; ++t) {
    jobs[t] = base;
    jobs[t].thread = t;
    jobs[t].threads = threads;
    pthread_create(&tid[t], NULL, worker, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
  return 0;
}

int oracle_stft_f32(const float *x, int64_t lead, int64_t n, int fft, int hop, const double *window,
                    int64_t left, int pad, double pad_value, int64_t frames, int complex_out, double power,
                    float *out, int threads) {
  job_t j; memset(&j, 0, sizeof(j));
  j.x32 = x; j.lead = lead; j.n = n; j.fft = fft; j.hop = hop; j.window = window; j.left = left; j.pad = pad;
  j.pad_value = pad_value; j.power = power; j.frames = frames; j.out32 = out; j.complex_out = complex_out;
  return run(j, threads);
}

int oracle_stft_f64(const double *x, int64_t lead, int64_t n, int fft, int hop, const double *window,
                    int64_t left, int pad, double pad_value, int64_t frames, int complex_out, double power,
                    double *out, int threads) {
  job_t j; memset(&j, 0, sizeof(j));
  j.x64 = x; j.lead = lead; j.n = n; j.fft = fft; j.hop = hop; j.window = window; j.left = left; j.pad = pad;
  j.pad_value = pad_value; j.power = power; j.frames = frames; j.out64 = out; j.complex_out = complex_out;
  return run(j, threads);
}

/* mel.ml:231: out[l][m][t] = (float) sum_b W[m][b] * (double) S[l][b][t], bins in ascending order.
 * Clip-parallel; the inner loop runs along the frame axis (contiguous).  Exact-zero weights are skipped: for
 * finite spectrograms adding W*S = +-0 never changes a sum, so the values are those of the dense product. */
typedef struct {
  const double *w; int n_mels, bins; const float *s; int64_t lead, frames; float *out; int thread, threads;
} mel_job_t;

static void *mel_worker(void *arg) {
  mel_job_t *j = (mel_job_t *)arg;
  double *acc = (double *)malloc(sizeof(double) * (size_t)(j->frames > 0 ? j->frames : 1));
  for (int64_t u = j->thread; u < j->lead * j->n_mels; u += j->threads) {
    const int64_t l = u / j->n_mels;
    const int m = (int)(u % j->n_mels);
    for (int64_t t = 0; t < j->frames; ++t) acc[t] = 0.0;
    for (int b = 0; b < j->bins; ++b) {
      const double wv = j->w[(int64_t)m * j->bins + b];
      if (wv == 0.0) continue;
      const float *row = j->s + (l * j->bins + b) * j->frames;
      for (int64_t t = 0; t < j->frames; ++t) acc[t] += wv * (double)row[t];
    }
    float *o = j->out + (l * j->n_mels + m) * j->frames;
    for (int64_t t = 0; t < j->frames; ++t) o[t] = (float)acc[t];
  }
  free(acc);
  return NULL;
}

int oracle_mel_apply_f32_mt(const double *w, int n_mels, int bins, const float *s, int64_t lead, int64_t frames,
                            float *out, int threads) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t tid[256];
  mel_job_t jobs[256];
  for (int t = 0; t < threads; ++t) {
    mel_job_t j = {w, n_mels, bins, s, lead, frames, out, t, threads};
    jobs[t] = j;
    pthread_create(&tid[t], NULL, mel_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
  return 0;
}

int oracle_mel_apply_f32(const double *w, int n_mels, int bins, const float *s, int64_t lead, int64_t frames,
                         float *out) {
  return oracle_mel_apply_f32_mt(w, n_mels, bins, s, lead, frames, out, 1);
}

/* resample_stubs.c:329-372 `soundml_resample_shape_run`, restated (interleaved complex128 arrays): the block identity
 * between the two transforms of the overlap-save resample executor.  Built with -ffp-contract=off like the rest. */
static void cx_mul(const double *a, const double *b, double *r) {
  r[0] = (a[0] * b[0]) - (a[1] * b[1]);
  r[1] = (a[0] * b[1]) + (a[1] * b[0]);
}

int oracle_resample_shape(const double *x, const double *h, double *y, int64_t lines, int64_t n, int64_t sl, int64_t sm) {
  if (lines < 0 || n < 2 || (n % 2) != 0 || sl < 1 || sm < 1 || (sl > 1 && sm > 1)) return -1;
  const int64_t w = sl > 1 ? n * sl : (sm > 1 ? n / sm : n);
  if (w < 2 || (sm > 1 && (n % sm) != 0)) return -1;
  const int64_t bins = n / 2 + 1, obins = w / 2 + 1, half = n / 2;
  for (int64_t line = 0; line < lines; ++line) {
    const double *xs = x + 2 * line * bins;
    double *ys = y + 2 * line * obins;
    if (sl > 1) {
      int64_t k = 0;
      while (k < obins) {
        for (int64_t j = 0; j <= half && k < obins; ++j, ++k) cx_mul(xs + 2 * j, h + 2 * k, ys + 2 * k);
        for (int64_t j = half + 1; j < n && k < obins; ++j, ++k) {
          const double z[2] = {xs[2 * (n - j)], -xs[2 * (n - j) + 1]};
          cx_mul(z, h + 2 * k, ys + 2 * k);
        }
      }
    } else if (sm > 1) {
      for (int64_t k = 0; k < obins; ++k) {
        int64_t j = k;
        double acc[2], p[2];
        cx_mul(xs + 2 * k, h + 2 * k, acc);
        for (int64_t r = 1; r < sm; ++r) {
          j += w;
          if (j <= half) {
            cx_mul(xs + 2 * j, h + 2 * j, p);
          } else {
            cx_mul(xs + 2 * (n - j), h + 2 * (n - j), p);
            p[1] = -p[1];
          }
          acc[0] += p[0];
          acc[1] += p[1];
        }
        ys[2 * k] = acc[0];
        ys[2 * k + 1] = acc[1];
      }
    } else {
      for (int64_t k = 0; k < obins; ++k) cx_mul(xs + 2 * k, h + 2 * k, ys + 2 * k);
    }
  }
  return 0;
}
