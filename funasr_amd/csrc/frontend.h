// Kernel argument blocks of the fbank / LFR / CMVN frontend (see frontend.hip).
#pragma once
#include "common.h"

namespace pf {

struct FbankArgs {
    const float* wav;            // device [B, wav_stride] mono PCM in [-1, 1]
    size_t wav_stride;
    const int* n_frames;         // device int32 [B] fbank frames per utterance
    float* fbank;                // device [B, max_frames, n_mels]
    int max_frames;
    int frame_len, frame_shift, n_mels;
    float in_scale;              // 32768 (wav_frontend.py:168-169)
    float preemph;               // 0.97
    const float* window;         // device [frame_len]
    const float2* twiddle;       // device [256] (cos, -sin)(2 pi k / 512)
    const float* mel_weight;     // device dense [n_mels, 257]
    const int* mel_offset;       // device [n_mels] first non-zero fft bin
    const int* mel_len;          // device [n_mels] number of non-zero bins
    const float* mel_compact;    // device [mel_nnz]: the non-zero weights of all triangles back to back
    const int* mel_coff;         // device [n_mels] start of mel m inside mel_compact
    int mel_nnz;
};
int launch_fbank(const FbankArgs& a, int B, int max_frames_in_batch, hipStream_t stream);

struct LfrArgs {
    const float* fbank;          // device [B, max_frames, n_mels]
    int max_frames;
    const int* n_frames;         // device int32 [B]
    float* out;                  // device [B, T_out, n_mels * lfr_m]
    int T_out;
    int n_mels, lfr_m, lfr_n;
    const float* cmvn_shift;     // device [n_mels * lfr_m] or nullptr
    const float* cmvn_scale;
    int left;                    // frames of left context the row index is shifted by: (lfr_m-1)/2 offline, 0 online
    int rows_override;           // > 0: emit exactly this many rows (online LFR decides the count on the host)
};
int launch_lfr_cmvn(const LfrArgs& a, int B, hipStream_t stream);

}  // namespace pf
