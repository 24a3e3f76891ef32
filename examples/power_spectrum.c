/* The drop-in boundary from plain C (no Python, no torch): a 440 Hz tone through smx_stft_power_spectrum_f32 and
 * smx_mel_spectrogram_f32 with host buffers.  Build and run on a machine with a HIP device:
 *   gcc -O2 -I include examples/power_spectrum.c -L soundml_amd/lib -lsoundml_amd -Wl,-rpath,$PWD/soundml_amd/lib -lm -o /tmp/ps && /tmp/ps */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "soundml_amd.h"

#define CHECK(call)                                                         \
  do {                                                                      \
    if ((call) != SMX_OK) {                                                 \
      fprintf(stderr, "%s failed: %s\n", #call, smx_last_error());          \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main(void) {
  const int64_t sr = 44100, n = 441000, fft = 1024, hop = 256, n_mels = 128;
  float *x = (float *)malloc(sizeof(float) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) x[i] = (float)sin(2.0 * M_PI * 440.0 * (double)i / (double)sr);

  smx_stft_config *stft = NULL;
  smx_mel_config *mel = NULL;
  CHECK(smx_stft_config_create(fft, SMX_DEFAULT, hop, SMX_ALIGN_CENTERED, SMX_PAD_REFLECT, 0.0, SMX_SCALE_NONE, SMX_WINDOW_HANN,
                               NULL, &stft));
  CHECK(smx_mel_config_create(n_mels, sr, fft, 0.0, 0, 0.0, SMX_MEL_SLANEY, SMX_NORM_SLANEY, &mel));
  int64_t frames = 0;
  CHECK(smx_stft_frames(stft, n, &frames));
  const int64_t bins = smx_stft_config_bins(stft);

  float *power = (float *)malloc(sizeof(float) * (size_t)(bins * frames));
  float *mels = (float *)malloc(sizeof(float) * (size_t)(n_mels * frames));
  CHECK(smx_stft_power_spectrum_f32(stft, x, 1, n, 2.0, power));
  CHECK(smx_mel_spectrogram_f32(stft, mel, x, 1, n, 2.0, mels));

  int64_t peak = 0;
  for (int64_t k = 1; k < bins; ++k)
    if (power[k * frames + 800] > power[peak * frames + 800]) peak = k;
  printf("power spectrogram [%lld; %lld], peak bin %lld (%.1f Hz); mel [%lld; %lld], mel[13][800] = %g\n", (long long)bins,
         (long long)frames, (long long)peak, (double)peak * (double)sr / (double)fft, (long long)n_mels, (long long)frames,
         (double)mels[13 * frames + 800]);
  smx_mel_config_destroy(mel);
  smx_stft_config_destroy(stft);
  free(mels);
  free(power);
  free(x);
  return peak == 10 ? 0 : 2;
}
