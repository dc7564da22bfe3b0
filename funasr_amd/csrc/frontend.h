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
    const float2* twiddle;       // device [512] (cos, -sin)(2 pi k / 512)
    // the triangles cut into pieces of <= 8 consecutive bins (a lane owns <= 2 pieces, weights in registers)
    const float* piece_w;        // device [n_pieces, 8] weights (zero padded)
    const int* piece_k0;         // device [n_pieces] first fft bin of the piece (k0 + 7 <= 256: clamped, pad weights are 0)
    const int* mel_first;        // device [n_mels] first piece of mel m
    const int* mel_count;        // device [n_mels] pieces of mel m
    int n_pieces;                // <= 128
    // dither != 0: Gaussian noise of this standard deviation added to every sample OF EVERY FRAME (after the 2^15 scaling,
    // before DC removal) -- kaldi.fbank's `dither`, wav_frontend.py:106,171-181: independent draws per (frame, sample), so the
    // overlap of two frames gets two different noises. Counter-based generator (Philox4x32-10 keyed by seed, counted by
    // (call, frame, sample pair)): the same seed and call index reproduce the features bit for bit.
    float dither;
    unsigned long long seed;     // key
    unsigned int call;           // launch counter (a new noise field per forward, like the reference's advancing torch.randn)
    // verify != 0: every frame is evaluated (at least) twice from its samples in registers and re-evaluated until two consecutive
    // results agree; disagreements are counted in *faults (device, may be null). For GPUs shared with another process that runs
    // LDS-DMA-heavy kernels: there a frame's LDS exchange is disturbed about once per 10^4 frames (open issue).
    int verify; unsigned int* faults;
    // snip_edges = false (feature-window.cc:66-90): n_samples != null, frame f starts at f * frame_shift - first_offset with
    // first_offset = frame_len / 2 - frame_shift / 2, and samples outside [0, n) are mirrored at the ends (feature-window.cc:152-171)
    const int* n_samples;        // device int32 [B]; null = snip_edges (every frame lies inside the waveform)
    int first_offset;
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
