/*
 * pafc_fbank.h -- C ABI of the Kaldi-compatible 80-bin log-mel filterbank on gfx950.
 *
 * Replaces the reference's call into a third-party CPU routine,
 *   torchaudio.compliance.kaldi.fbank(waveform, num_mel_bins=80, frame_length=25, frame_shift=10,
 *                                     dither=0|1, energy_floor=0.0, sample_frequency=16000)
 * at wenet/dataset/processor.py:363-369, wenet/bin/encoder-rtf.py:575-583, wenet/bin/recognize_wav2.py:510-518
 * (all other arguments at torchaudio's defaults: snip_edges, remove_dc_offset, preemphasis 0.97, povey window,
 * round_to_power_of_two -> 512-point FFT, power spectrum, low_freq 20 Hz, high_freq Nyquist, log with float-eps
 * floor).  Frame geometry is fixed to that configuration: 400-sample window, 160-sample shift.
 *
 * The constant tables are supplied by the caller (device pointers), so the library holds no state:
 *   window      (400)          povey window
 *   dft_table   (400, cols)    cols = pafc_fbank_tables_cols(); column 2k = cos(2 pi k n / 512),
 *                              column 2k+1 = -sin(2 pi k n / 512) for k = 0..256, remaining columns zero
 *   mel_weights (num_mel_bins, 257), mel_lo / mel_hi (num_mel_bins): filter b is non-zero on bins [lo, hi)
 * wave: (num_samples) float32 in int16 range; noise: (frames, 400) standard normal or NULL (dither off);
 * out: (frames, num_mel_bins) float32, frames = pafc_fbank_num_frames(num_samples) = 1 + (S - 400) / 160.
 */
#ifndef PAFC_FBANK_H
#define PAFC_FBANK_H

#include "pafc_wkv6.h"

#ifdef __cplusplus
extern "C" {
#endif

long pafc_fbank_num_frames(long num_samples);
int pafc_fbank_tables_cols(void);
int pafc_fbank_f32(const float *wave, long num_samples, const float *window, const float *dft_table,
                   const float *mel_weights, const int *mel_lo, const int *mel_hi, int num_mel_bins,
                   const float *noise, float dither, float preemph, float *out, pafc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PAFC_FBANK_H */
