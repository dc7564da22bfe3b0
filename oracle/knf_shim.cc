// C shim around the kaldi-native-fbank sources vendored by the reference
// (/root/reference/runtime/onnxruntime/third_party/kaldi-native-fbank). Only this shim is ours; the library code
// is compiled where it lies (see oracle/Makefile) into oracle/_ref/libknf_ref.so. TEST INFRASTRUCTURE ONLY.
// Options mirror what the reference's C++ runtime sets for Paraformer (runtime/onnxruntime/src/paraformer.cpp:22-31).
#include <cstdint>
#include <vector>

#include "kaldi-native-fbank/csrc/online-feature.h"

// window_type / snip_edges: the FrameExtractionOptions fields kaldi.fbank(window_type=, snip_edges=) stands for
// (funasr/frontends/wav_frontend.py:178-180); InputFinished() releases the mirrored last frames of snip_edges = false
// frame_length_ms is a float: WavFrontend shrinks the window of a clip shorter than 25 ms to the clip itself (wav_frontend.py:176)
extern "C" int knf_fbank_opts(const float* wave_scaled, int64_t n, int n_mels, float frame_length_ms, int frame_shift_ms,
                              float sample_rate, const char* window_type, int snip_edges, float* out, int64_t max_frames) {
    knf::FbankOptions opts;
    opts.frame_opts.dither = 0.0f;
    opts.frame_opts.snip_edges = snip_edges != 0;
    opts.frame_opts.samp_freq = sample_rate;
    opts.frame_opts.window_type = window_type;
    opts.frame_opts.frame_shift_ms = (float)frame_shift_ms;
    opts.frame_opts.frame_length_ms = frame_length_ms;
    opts.mel_opts.num_bins = n_mels;
    opts.energy_floor = 0.0f;
    opts.mel_opts.debug_mel = false;
    knf::OnlineFbank fbank(opts);
    fbank.AcceptWaveform(sample_rate, wave_scaled, (int32_t)n);
    fbank.InputFinished();
    const int32_t frames = fbank.NumFramesReady();
    for (int32_t i = 0; i < frames && i < max_frames; ++i) {
        const float* f = fbank.GetFrame(i);
        for (int k = 0; k < n_mels; ++k) out[(int64_t)i * n_mels + k] = f[k];
    }
    return frames;
}

extern "C" int knf_fbank(const float* wave_scaled, int64_t n, int n_mels, int frame_length_ms, int frame_shift_ms,
                         float sample_rate, float* out, int64_t max_frames) {
    knf::FbankOptions opts;
    opts.frame_opts.dither = 0.0f;
    opts.frame_opts.snip_edges = true;
    opts.frame_opts.samp_freq = sample_rate;
    opts.frame_opts.window_type = "hamming";
    opts.frame_opts.frame_shift_ms = (float)frame_shift_ms;
    opts.frame_opts.frame_length_ms = (float)frame_length_ms;
    opts.mel_opts.num_bins = n_mels;
    opts.energy_floor = 0.0f;
    opts.mel_opts.debug_mel = false;
    knf::OnlineFbank fbank(opts);
    fbank.AcceptWaveform(sample_rate, wave_scaled, (int32_t)n);
    const int32_t frames = fbank.NumFramesReady();
    for (int32_t i = 0; i < frames && i < max_frames; ++i) {
        const float* f = fbank.GetFrame(i);
        for (int k = 0; k < n_mels; ++k) out[(int64_t)i * n_mels + k] = f[k];
    }
    return frames;
}
