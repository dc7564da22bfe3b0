/*
 * paraformer_hip.h -- C ABI of the MI355X-native (gfx950) Paraformer / SenseVoice inference hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / C++ types. Every module of the
 * reference's operator registry (funasr/register.py `tables`) that sits on the hot path has one opaque
 * handle here, with `create / set_tensor / forward / destroy` entry points whose tensor contract mirrors the
 * reference module's forward():
 *
 *   pf_frontend   <-> WavFrontend.forward                    funasr/frontends/wav_frontend.py:149-196
 *   pf_encoder    <-> SANMEncoder.forward                    funasr/models/sanm/encoder.py:392-461
 *                     SenseVoiceEncoderSmall.forward         funasr/models/sense_voice/model.py:623-655
 *   pf_predictor  <-> CifPredictorV2.forward                 funasr/models/paraformer/cif_predictor.py:253-314
 *                     CifPredictorV3.forward / get_upsample_timestamp (pf_predictor_create_v3, pf_predictor_timestamp)
 *                                                            funasr/models/bicif_paraformer/cif_predictor.py:215-352
 *   pf_decoder    <-> ParaformerSANMDecoder.forward          funasr/models/paraformer/decoder.py:397-449
 *                     (+ log_softmax/argmax of Paraformer.inference, funasr/models/paraformer/model.py:345,642;
 *                      kernel_size 21 / vocab_size 0 = SeACo's bias decoder, funasr/models/seaco_paraformer/model.py:98-108)
 *                     pf_decoder_asf_scores <-> ParaformerSANMDecoder.forward_asf6 as SeACo's hotword filter uses it
 *                                                            funasr/models/paraformer/decoder.py:485-513, seaco_paraformer/model.py:323-335
 *                     pf_decoder_create_contextual / pf_decoder_forward_contextual <-> ContextualParaformerDecoder.forward
 *                                                            funasr/models/contextual_paraformer/decoder.py:133-352
 *   pf_ctc        <-> CTC.log_softmax / argmax               funasr/models/ctc/ctc.py:192-216
 *   pf_k_log_softmax <-> the log_softmax feeding BeamSearchPara / CTCPrefixScorer
 *                                                            funasr/models/paraformer/model.py:345,629-637, transformer/scorers/ctc.py:46
 *   pf_stream     <-> ParaformerStreaming chunk step         funasr/models/paraformer_streaming/model.py
 *   pf_vad, pf_vad_decision <-> FsmnVADStreaming (network / state machine)   funasr/models/fsmn_vad_streaming/{encoder,model}.py
 *   pf_k_lstm     <-> torch.nn.LSTM layer (hotword encoder of SeACo, seaco_paraformer/model.py:388-424)
 *
 * The handle style (opaque pointer, int return codes, library-owned scratch) follows the reference's own C
 * API for the same path, runtime/onnxruntime/include/funasrruntime.h:21-24,58-73. Tensor names accepted by
 * `*_set_tensor` are the reference's state_dict keys relative to the module, e.g.
 * "encoders0.0.self_attn.linear_q_k_v.weight", so a real model.pt loads without a conversion step.
 *
 * Conventions
 *   - `*_dev` pointers are device (HBM) pointers, 16-byte aligned, row-major contiguous float32 unless stated.
 *   - `*_host` pointers are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream). All work is enqueued on it; a
 *     call only synchronises where it has to return a host value (documented per function).
 *   - return value: 0 = ok, -1 = invalid argument, -2 = HIP runtime error, -3 = missing tensor / not ready.
 *     pf_last_error() returns a thread-local message for the last failure.
 *   - the library never falls back to a CPU path: without a gfx950 device every create() fails with -2.
 */
#ifndef PARAFORMER_HIP_H_
#define PARAFORMER_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): everything added since the first release is covered by one number a dlsym-binding host can test -- pf_dp_*
 * (incl. pf_dp_broadcast_raw), pf_stream_step_begin / _end, pf_frontend_set_dither / _verify, pf_paraformer_forward, the
 * pf_k_* measurement entries. A library reporting 1 has none of them.
 * 3 (round 6): + pf_paraformer_begin / pf_paraformer_finish (the split-phase offline forward).
 * 4: + pf_predictor_alphas_begin / pf_predictor_embeds_slot.
 * 5: + pf_frontend_set_window / pf_frontend_set_snip_edges, pf_decoder_set_decoders2,
 *    pf_utterance_mvn / pf_global_mvn. */
#define PF_ABI_VERSION 5

const char* pf_last_error(void);
int pf_abi_version(void);
/* number of visible HIP devices (0 when there is no GPU); never fails */
int pf_device_count(void);

/* Process-wide fence: once on, every frontend handle runs fbank_kernel's cross-check (each frame evaluated twice until two runs
 * agree, pf_frontend_set_verify) whatever its own setting. The library switches it on itself at the first asynchronous streaming
 * step (pf_stream_step_begin); hosts that put several ranks or streams on one GPU call pf_set_concurrency_guard(1). */
int pf_set_concurrency_guard(int32_t on);
int pf_concurrency_guard(void);

/* ------------------------------------------------------------------------------------------------ frontend */
typedef struct pf_frontend pf_frontend;

typedef struct pf_frontend_config {
    int32_t sample_rate;      /* 16000 */
    int32_t frame_length;     /* samples per window: 400 (25 ms)      wav_frontend.py:100,174 */
    int32_t frame_shift;      /* samples per hop:    160 (10 ms)      wav_frontend.py:101 */
    int32_t n_mels;           /* 80                                   wav_frontend.py:99 */
    int32_t lfr_m;            /* 7 frames stacked                     wav_frontend.py:104 */
    int32_t lfr_n;            /* 6 frames hop                         wav_frontend.py:105 */
    float low_freq;           /* 20 Hz (Kaldi default) */
    float high_freq;          /* 0 -> Nyquist */
    float preemph;            /* 0.97 */
    float upscale;            /* 32768: waveform * (1 << 15)          wav_frontend.py:168-169 */
} pf_frontend_config;

pf_frontend* pf_frontend_create(const pf_frontend_config* cfg);
void pf_frontend_destroy(pf_frontend* f);
/* CMVN rows of am.mvn (<AddShift>, <Rescale>; wav_frontend.py:15-60): y = (x + shift) * scale. n = n_mels*lfr_m */
int pf_frontend_set_cmvn(pf_frontend* f, const float* shift_host, const float* scale_host, int32_t n);
/* kaldi.fbank's `dither` (wav_frontend.py:106,171-181; the reference default is 1.0): Gaussian noise of that standard deviation
 * on every sample of every frame, after the 2^15 scaling. 0 (default) = deterministic features. Noise comes from a counter-based
 * generator keyed by `seed` and the number of forwards since this call: the same seed gives the same feature sequence. Parity
 * with the reference is statistical by nature (its noise is torch.randn). */
int pf_frontend_set_dither(pf_frontend* f, float dither, uint64_t seed);
/* Cross-check of the fbank kernel (off by default): every frame is evaluated twice from its samples in registers and re-evaluated
 * until two consecutive results agree in every lane. It is how round 4 located the 'two-process frontend fault' (packed-fp32
 * VALU instructions next to waves of the 128 x 128 f16x2 GEMM; fixed at build level) and stays as a diagnostic. pf_frontend_faults:
 * disagreements seen since the handle was made. */
int pf_frontend_set_verify(pf_frontend* f, int32_t on);
int pf_frontend_faults(pf_frontend* f, uint32_t* count_host);
/* diagnostics: for the first 16 disagreements, 4 x uint32 each: shader-clock cycles of the two evaluations that disagreed, the
 * frame index, the retry number (an evaluation suspended in the middle -- another process's turn on the CU -- shows as a long one) */
int pf_frontend_fault_log(pf_frontend* f, uint32_t* log_host_64);
/* Optional: override the built-in Kaldi tables (float64 cos window, float32 mel triangles as in
 * kaldi-native-fbank feature-window.cc:25-47, mel-computations.cc:118-210) with caller-computed ones, e.g. the
 * float32 tables torchaudio.compliance.kaldi builds. window: [frame_length]; mel: dense [n_mels, 257]. */
int pf_frontend_set_tables(pf_frontend* f, const float* window_host, const float* mel_host);
/* The analysis window by name: hamming (default) | hanning | povey | rectangular | blackman | sine, evaluated as kaldi-native-fbank
 * does (float64 cosines, feature-window.cc:25-55; blackman_coeff 0.42 is Kaldi's default). The reference passes WavFrontend's
 * `window` to kaldi.fbank(window_type=...) (funasr/frontends/wav_frontend.py:178). */
int pf_frontend_set_window(pf_frontend* f, const char* window_type, float blackman_coeff);
/* snip_edges = 0: (n + shift / 2) / shift frames, frame i centred on sample i * shift + shift / 2, the waveform mirrored at both
 * ends (kaldi FrameExtractionOptions.snip_edges, feature-window.cc:66-90,152-171; funasr/frontends/wav_frontend.py:180). Default 1. */
int pf_frontend_set_snip_edges(pf_frontend* f, int32_t on);
/* frames after fbank and after LFR for an utterance of n_samples (by the handle's snip_edges setting) */
int32_t pf_frontend_num_fbank_frames(const pf_frontend* f, int64_t n_samples);
int32_t pf_frontend_num_frames(const pf_frontend* f, int64_t n_samples);
/* wav_dev: [B, wav_stride] float32 in [-1, 1]; n_samples_host: [B]; feats_dev: [B, T_out, n_mels*lfr_m] with
 * T_out >= max_b frames(b), rows past an utterance's length are zero-filled (pad_sequence, :195);
 * feat_lens_host: [B] output. Also accepts fbank_dev != NULL to export the raw [B, T_fb_max, n_mels] log-mel
 * (T_fb_max = max_b fbank frames). Does not synchronise. */
int pf_frontend_forward(pf_frontend* f, const float* wav_dev, int64_t wav_stride, const int32_t* n_samples_host,
                        int32_t B, float* feats_dev, int32_t T_out, int32_t* feat_lens_host, float* fbank_dev,
                        void* stream);

/* ---------------------------------------------------------------------------------- feature normalisation (normalize_classes)
 * The reference's `normalize` modules, applied between the frontend and the encoder (funasr/models/paraformer/model.py:305-306,
 * sense_voice/model.py:836-837). In place on x_dev [B, T, D]; lens_dev: device int32 [B]. No sync.
 * pf_utterance_mvn = UtteranceMVN (funasr/models/normalize/utterance_mvn.py:51-96), its arithmetic step by step -- including that
 * with norm_means the padded rows come out as -mean and, with norm_vars as well, take part in the variance and the divisor is
 * sqrt(std). pf_global_mvn = GlobalMVN.forward (global_mvn.py:66-92) with the mean / std vectors [D] the class derives from its
 * stats file: (x - mean), padded rows zero, / std -- bit-exact. */
int pf_utterance_mvn(float* x_dev, const int32_t* lens_dev, int32_t B, int32_t T, int32_t D, int32_t norm_means, int32_t norm_vars,
                     float eps, void* stream);
int pf_global_mvn(float* x_dev, const int32_t* lens_dev, int32_t B, int32_t T, int32_t D, const float* mean_dev, const float* std_dev,
                  int32_t norm_means, int32_t norm_vars, void* stream);

/* ------------------------------------------------------------------------------------------------- encoder */
typedef struct pf_encoder pf_encoder;

typedef struct pf_encoder_config {
    int32_t input_dim;        /* 560 = n_mels * lfr_m                         sanm/encoder.py:197 */
    int32_t d_model;          /* 512  output_size */
    int32_t n_heads;          /* 4    attention_heads (d_k must be 128) */
    int32_t ffn_dim;          /* 2048 linear_units */
    int32_t n_blocks;         /* 50   num_blocks: encoders0 (1) + encoders (n_blocks-1) */
    int32_t tp_blocks;        /* 0 (Paraformer) or 20 (SenseVoiceSmall tp_encoders, sense_voice/model.py:601-612) */
    int32_t kernel_size;      /* 11   FSMN taps */
    int32_t sanm_shift;       /* 0    sanm_shfit */
    float ln_eps;             /* 1e-12 (layer_norm.py:16) / 1e-5 (sense_voice/model.py:300-323) */
} pf_encoder_config;

pf_encoder* pf_encoder_create(const pf_encoder_config* cfg);
void pf_encoder_destroy(pf_encoder* e);
/* data may be a host or a device pointer; the tensor is copied (and repacked) into library-owned HBM */
int pf_encoder_set_tensor(pf_encoder* e, const char* name, const float* data, int64_t numel);
/* number of tensors still missing (0 = ready) */
int pf_encoder_missing(const pf_encoder* e);
/* Arithmetic mode. 3 = "f16x2", THE DEFAULT wherever its kernels exist (d_model / n_heads == 128, d_model % 256 == 0,
 * ffn_dim % 256 == 0; mode 0 otherwise): fp32-class results on the fp16 matrix cores -- every GEMM / attention operand is two
 * fp16 planes of the tensor times a power of two (x 2^e = hi + lo), three MFMA products per operand pair, fp32 accumulate,
 * fp32 residual stream / LayerNorm statistics / softmax / FSMN (gemm_f16x2.hip, gemm_f16x2_row.hip, attention_f16x2.hip).
 * Meets every parity bar of mode 0 (activations <= 1e-3, CIF indices equal to the CPU reference; tests/test_parity_gpu.py,
 * fixture f32_mode) and is the measured mode of bench.py. 0 = every product on the exact fp32 MFMA (the opt-out, 2.4x
 * slower). 1 = bf16 OPERANDS for the GEMMs and the attention with fp32 accumulation (bf16-class error; the reference's own
 * bf16=True casts the whole module, auto_model.py:665-668). 2 = fp32-class results from three bf16 planes per operand, six
 * bf16 MFMA products (gemm_split3.hip). */
int pf_encoder_set_precision(pf_encoder* e, int32_t mode);
/* Row packing (mode 3 only; other modes ignore it). The reference runs the encoder over every row of the padded batch
 * [B, T] (sanm/encoder.py:428-470: masks, not skipping); its consumers read only a prefix of each sequence: the CTC head
 * rows < len (sense_voice/model.py:1014), the CIF predictor rows <= len (the conv at the last valid frame and the tail
 * frame, cif_predictor.py:196-205,275-277), the decoder's cross-attention rows < len. extra_rows >= 0: sequence b gets
 * min(len_b + extra_rows, T) rows, laid out back to back (no alignment padding), identical in value to the same rows of
 * the padded computation up to the summation order of the attention's key tiles; the remaining rows of out_dev are
 * zero. extra_rows >= T keeps every row. extra_rows < 0 (default): padded layout, every row computed. */
int pf_encoder_set_row_packing(pf_encoder* e, int32_t extra_rows);
/* Options of mode 3 that do not change a result bit or only the kernel schedule: key "fuse_row" (1, default: linear_out / w_2
 * in their full-row form, residual adds + the following LayerNorm in the GEMM epilogue; 0: separate launches -- bitwise equal),
 * key "attn_variant" (attention_f16x2.hip: 3 lazy rescale, default; 1 pipelined; 0 plain), key "gemm_tile" (block shape of
 * the block's gemm_f16x2 launches: 0 by shape, default; 1 / 2 the 256 x 128 / 256 x 256 shapes; 5 the 128 x 256 shape with two
 * workgroups per CU -- all bitwise equal). Unknown keys return -1. */
int pf_encoder_set_option(pf_encoder* e, const char* key, int32_t value);
/* SANMVadEncoder (funasr/models/ct_transformer_streaming/encoder.py:175-430, the encoder of CTTransformerStreaming): the
 * SAN-M encoder whose self-attention is causal in every block and, in the last one, masked by the VAD corner of
 * transformer/utils/mask.py:38-52 (queries before vad_pos - 1 do not see keys from vad_pos on). vad_pos_host: one value
 * per sequence of the following forwards (copied), NULL switches the masks off again. fp32 mode, heads of d_k <= 64. */
int pf_encoder_set_vad_mask(pf_encoder* e, const int32_t* vad_pos_host, int32_t B);
/* xs_dev: [B, T, input_dim] (un-scaled features, exactly what SANMEncoder.forward receives), lens_host: [B],
 * pe_dev: [T, input_dim] sinusoidal table (embedding.py:396-420; NULL = library computes it with libm),
 * out_dev: [B, T, d_model]. run_blocks < 0 runs everything incl. the final norm(s); run_blocks = k >= 0 stops
 * after k encoder blocks and returns the raw residual stream (for per-layer parity checks). No sync. */
int pf_encoder_forward(pf_encoder* e, const float* xs_dev, const int32_t* lens_host, int32_t B, int32_t T,
                       const float* pe_dev, float* out_dev, int32_t run_blocks, void* stream);

/* ----------------------------------------------------------------------------------------------- predictor */
typedef struct pf_predictor pf_predictor;

typedef struct pf_predictor_config {
    int32_t d_model;          /* idim 512 */
    int32_t l_order;          /* 1 */
    int32_t r_order;          /* 1 */
    float threshold;          /* 1.0 (the prefix-sum/floor formulation of cif_v1 is only defined for 1.0) */
    float smooth_factor;      /* 1.0 */
    float noise_threshold;    /* 0.0 */
    float tail_threshold;     /* 0.45 */
    int32_t tail_mask;        /* 1 */
} pf_predictor_config;

pf_predictor* pf_predictor_create(const pf_predictor_config* cfg);
void pf_predictor_destroy(pf_predictor* p);
int pf_predictor_set_tensor(pf_predictor* p, const char* name, const float* data, int64_t numel);
int pf_predictor_missing(const pf_predictor* p);
/* Step 1: alphas + fire scan. hidden_dev: [B, T, d_model]; lens_host: [B];
 * alphas_dev / peaks_dev: [B, T+1] outputs (tail-extended alphas, cif_peak); token_num_host: [B] output =
 * number of fired tokens (= floor(sum alphas), cif_predictor.py:443-446). SYNCHRONISES the stream (the token
 * count decides the decoder's shape, like the .item() at cif_predictor.py:311). */
int pf_predictor_alphas(pf_predictor* p, const float* hidden_dev, const int32_t* lens_host, int32_t B, int32_t T,
                        float* alphas_dev, float* peaks_dev, int32_t* token_num_host, void* stream);
/* Step 2: acoustic embeddings for the scan of the last pf_predictor_alphas call: embeds_dev [B, N, d_model],
 * rows >= token_num[b] are zero. No sync. */
int pf_predictor_embeds(pf_predictor* p, const float* hidden_dev, int32_t B, int32_t T, int32_t N,
                        float* embeds_dev, void* stream);
/* The same two steps WITHOUT the synchronisation between them, for callers that keep two batches in flight (round 6; the
 * module-level counterpart of pf_paraformer_begin / _finish, used by the classes whose chain is not the plain offline one --
 * BiCifParaformer: funasr/models/bicif_paraformer/model.py:327-345 reads the counts between predictor and decoder).
 * pf_predictor_alphas_begin: everything of step 1 ENQUEUED into scan state `slot` (0 or 1); the token counts are copied by the
 * stream into counts_pinned_host (B int32 of PINNED host memory that stays alive until the caller has synchronised, e.g. on an
 * event recorded behind this call). pf_predictor_embeds_slot: step 2 for that slot -- before the slot's next _begin.
 * pf_predictor_alphas / _embeds are slot 0 of the same state. */
int pf_predictor_alphas_begin(pf_predictor* p, int32_t slot, const float* hidden_dev, const int32_t* lens_host, int32_t B, int32_t T,
                              float* alphas_dev, float* peaks_dev, int32_t* counts_pinned_host, void* stream);
int pf_predictor_embeds_slot(pf_predictor* p, int32_t slot, const float* hidden_dev, int32_t B, int32_t T, int32_t N,
                             float* embeds_dev, void* stream);

/* CifPredictorV3 (funasr/models/bicif_paraformer/cif_predictor.py:121-384), the predictor of BiCifParaformer and
 * SeACo-Paraformer ("paraformer-zh"): the same first head as V2 but integrated by the sequential fp32 loop `cif`
 * (:39-86), plus a second head on a `upsample_times` x finer time axis that gives the token timestamps
 * (`get_upsample_timestamp`, :301-352). A handle made by pf_predictor_create_v3 answers pf_predictor_alphas /
 * pf_predictor_embeds with V3's arithmetic (token_num = floor(sum alphas), :383) and accepts pf_predictor_timestamp.
 * Extra tensors: upsample_cnn.weight [d, d, U] / .bias [d]; cif_output2.weight [d or 2d] / .bias [1]; for
 * upsample_type 1 blstm.weight_ih_l0[_reverse] [4d, d], blstm.weight_hh_l0[_reverse] [4d, d], blstm.bias_ih_l0[_reverse],
 * blstm.bias_hh_l0[_reverse] [4d] (torch.nn.LSTM's state_dict, gates i, f, g, o). */
typedef struct pf_predictor_v3_config {
    int32_t upsample_times;   /* 3 */
    int32_t upsample_type;    /* 0 "cnn", 1 "cnn_blstm" ("cnn_attn" is not built) */
    int32_t use_cif1_cnn;     /* 0: the second head reads the encoder output, 1: relu(cif_conv1d(.)) */
    float smooth_factor2;     /* 0.25 */
    float noise_threshold2;   /* 0.01 */
} pf_predictor_v3_config;
pf_predictor* pf_predictor_create_v3(const pf_predictor_config* cfg, const pf_predictor_v3_config* cfg3);
/* hidden_dev [B, T, d_model], lens_host [B], token_num_host [B] (the rounded token counts of pf_predictor_alphas) ->
 * us_alphas_dev, us_peaks_dev [B, U * T]: the upsampled weights rescaled to sum to token_num per utterance and their
 * running integral at threshold 1 - 1e-4 (`cif_wo_hidden`, :89-118), the two inputs of ts_prediction_lfr6_standard
 * (funasr/utils/timestamp_tools.py:37). No sync. */
int pf_predictor_timestamp(pf_predictor* p, const float* hidden_dev, const int32_t* lens_host,
                           const int32_t* token_num_host, int32_t B, int32_t T, float* us_alphas_dev, float* us_peaks_dev,
                           void* stream);

/* ------------------------------------------------------------------------------------------------- decoder */
typedef struct pf_decoder pf_decoder;

typedef struct pf_decoder_config {
    int32_t vocab_size;       /* 8404; 0 = no output layer (SeACo's bias decoder): forward returns hidden states only */
    int32_t d_model;          /* 512 */
    int32_t n_heads;          /* 4 */
    int32_t ffn_dim;          /* 2048 */
    int32_t n_blocks;         /* 16 = att_layer_num (blocks with cross-attention; decoders2: pf_decoder_set_decoders2) */
    int32_t kernel_size;      /* 11; 21 (with sanm_shift 0) for SeACo's bias decoder */
    int32_t sanm_shift;       /* 0 offline / 5 streaming model */
    float ln_eps;             /* 1e-12 */
} pf_decoder_config;

pf_decoder* pf_decoder_create(const pf_decoder_config* cfg);
void pf_decoder_destroy(pf_decoder* d);
/* decoders2 (funasr/models/paraformer/decoder.py:363-380,436-437): n_layers = num_blocks - att_layer_num further blocks of
 * FFN + FSMN memory (sanm_shfit 0) WITHOUT cross-attention between `decoders` and `decoders3`. Registers the tensors
 * "decoders2.<i>.{norm1,norm2}.{weight,bias}", "decoders2.<i>.feed_forward.*", "decoders2.<i>.self_attn.fsmn_block.weight";
 * call once, right after create. The published recipes have none (att_layer_num == num_blocks). */
int pf_decoder_set_decoders2(pf_decoder* d, int32_t n_layers);
int pf_decoder_set_tensor(pf_decoder* d, const char* name, const float* data, int64_t numel);
int pf_decoder_missing(const pf_decoder* d);
/* 0 = fp32 (default), 1 = bf16 operands for the GEMMs and the cross-attention (see pf_encoder_set_precision; applies
 * to the fused arg-max route, logits_dev == NULL), 2 = w_1 and linear_k_v on the three-plane split GEMM (fp32 results) */
int pf_decoder_set_precision(pf_decoder* d, int32_t mode);
/* memory_dev: [B, T, d_model] encoder output, mem_lens_host: [B]; embeds_dev: [B, N, d_model] CIF output,
 * tok_lens_host: [B]. Outputs (each may be NULL): logits_dev [B, N, vocab] (pre-softmax, decoder.py:444),
 * ids_dev int32 [B, N] = argmax over the vocabulary (fused into the output GEMM when logits_dev is NULL),
 * hidden_dev [B, N, d_model] (after_norm output). No sync. */
int pf_decoder_forward(pf_decoder* d, const float* memory_dev, const int32_t* mem_lens_host,
                       const float* embeds_dev, const int32_t* tok_lens_host, int32_t B, int32_t T, int32_t N,
                       float* logits_dev, int32_t* ids_dev, float* hidden_dev, void* stream);
/* ------------------------------------------------------------------------------------------------ offline pipeline
 * The whole offline forward in one call: what Paraformer.inference runs between the frontend and the tokenizer
 * (funasr/models/paraformer/model.py:286-346, 614-616, 642), at the tensor boundary the reference exports for this path
 * (funasr/models/paraformer/export_meta.py:44-68: speech f32 [B, T, 560], speech_lengths i32 [B] -> logits, token_num; the
 * arg-max is fused here, so token ids come back). The object borrows the three module handles (they must outlive it; a
 * CifPredictorV2 / V3 predictor and a plain ParaformerSANMDecoder) and owns the intermediate buffers; precision, row packing
 * and schedule options are whatever the module handles are set to. Bitwise the chain pf_encoder_forward -> pf_predictor_alphas
 * -> pf_predictor_embeds -> pf_decoder_forward. */
typedef struct pf_paraformer pf_paraformer;
pf_paraformer* pf_paraformer_create(pf_encoder* e, pf_predictor* p, pf_decoder* d);
void pf_paraformer_destroy(pf_paraformer* m);
/* feats_dev [B, T, input_dim], lens_host [B], pe_dev as pf_encoder_forward. Outputs: token_num_host [B] (rounded CIF counts,
 * written after the call's ONE host synchronisation); ids_dev int32 [B, ids_ld] (may be NULL), row b valid in
 * [0, token_num[b]); alphas_dev / peaks_dev [B, T + 1] (may be NULL: kept internally). Returns N = the batch's largest token
 * count (0: nothing fired, ids untouched -- model.py:615-616), or < 0. Everything behind the token count is only ENQUEUED. */
int pf_paraformer_forward(pf_paraformer* m, const float* feats_dev, const int32_t* lens_host, int32_t B, int32_t T,
                          const float* pe_dev, int32_t* ids_dev, int32_t ids_ld, int32_t* token_num_host,
                          float* alphas_dev, float* peaks_dev, void* stream);
/* The same forward in its two phases, for serving loops (round 6). pf_paraformer_begin ENQUEUES encoder + predictor (alphas, scan)
 * and the token counts' copy to a pinned host buffer, and returns a ticket (0 / 1) without any host synchronisation;
 * pf_paraformer_finish(ticket) waits for THAT copy (an event of this batch, not the stream), fills token_num_host, enqueues
 * embeds + decoder + arg-max and returns N like pf_paraformer_forward. Two batches may be in flight: issue begin(i + 1) BEFORE
 * finish(i) and the stream holds batch i + 1's encoder while the host reads batch i's counts -- the decoder is still sized by the
 * exact CIF count (cif_predictor.py:311), but the GPU never waits for the host. Tickets are finished in the order they were begun;
 * lens_host is copied by begin. (A V3 predictor keeps one scan state: one batch in flight.) pf_paraformer_forward == begin + finish.
 * `finish` may be given another stream than `begin` (the host has already waited for the batch's scan; the library orders the reuse of
 * the slot with an event): batch i's decoder then runs beside batch i + 1's encoder. ids_dev is valid once finish's stream has reached
 * the end of the call's work. */
int pf_paraformer_begin(pf_paraformer* m, const float* feats_dev, const int32_t* lens_host, int32_t B, int32_t T,
                        const float* pe_dev, void* stream);
int pf_paraformer_finish(pf_paraformer* m, int32_t ticket, int32_t* ids_dev, int32_t ids_ld, int32_t* token_num_host,
                         float* alphas_dev, float* peaks_dev, void* stream);
/* the last finished forward's encoder output [B, T, d_model] / acoustic embeddings [B, N, d_model] (library-owned; valid until
 * the next begin into the same slot / the next finish of this object) */
const float* pf_paraformer_encoder_out(const pf_paraformer* m);
const float* pf_paraformer_embeds(const pf_paraformer* m);

/* SeACo attention-score filter (funasr/models/seaco_paraformer/model.py:323-335 over paraformer/decoder.py:485-513
 * `forward_asf6`): blocks 0 .. n_blocks_before - 1 of the (bias) decoder in full, then block n_blocks_before up to its
 * cross-attention probabilities over the T memory rows (the hotword embeddings); scores_dev [T] receives their sum over
 * heads and token positions for sequence 0 (attn_mat[0].sum(0).sum(0)). */
int pf_decoder_asf_scores(pf_decoder* d, const float* memory_dev, const int32_t* mem_lens_host, const float* embeds_dev,
                          const int32_t* tok_lens_host, int32_t B, int32_t T, int32_t N, int32_t n_blocks_before,
                          float* scores_dev, void* stream);
/* ContextualParaformerDecoder (funasr/models/contextual_paraformer/decoder.py:133-352): created like pf_decoder_create with
 * n_blocks = att_layer_num; tensor names "decoders.{0..n_blocks-2}.*", "last_decoder.*", "bias_decoder.norm3.*",
 * "bias_decoder.src_attn.linear_{q,k_v,out}.*", "bias_output.weight", "decoders3.0.*", "after_norm.*", "output_layer.*".
 * forward_contextual additionally takes the hotword embeddings contextual_dev [B, n_hot, d_model] and clas_scale. */
pf_decoder* pf_decoder_create_contextual(const pf_decoder_config* cfg);
int pf_decoder_forward_contextual(pf_decoder* d, const float* memory_dev, const int32_t* mem_lens_host, const float* embeds_dev,
                                  const int32_t* tok_lens_host, const float* contextual_dev, int32_t n_hot, float clas_scale,
                                  int32_t B, int32_t T, int32_t N, float* logits_dev, int32_t* ids_dev, float* hidden_dev,
                                  void* stream);

/* ---------------------------------------------------------------------------------------------------- ctc */
/* ---- FSMN-VAD network (funasr/models/fsmn_vad_streaming/encoder.py:288-378 FSMN.forward), reduced to what the decision
 * logic reads: the summed posterior of the silence pdfs per frame (model.py:783-795), and the frame energies
 * (ComputeDecibel, model.py:513-530). Tensor names as in the reference state_dict below `encoder.`:
 * in_linear1|in_linear2|out_linear1|out_linear2 ".linear.weight|bias", "fsmn.<i>.linear.linear.weight",
 * "fsmn.<i>.fsmn_block.conv_left.weight" ([proj, 1, lorder, 1]), "fsmn.<i>.affine.linear.weight|bias". */
typedef struct pf_vad pf_vad;
typedef struct pf_vad_config {
    int32_t input_dim, input_affine_dim, fsmn_layers, linear_dim, proj_dim, lorder, rorder, lstride, rstride,
        output_affine_dim, output_dim;
} pf_vad_config;
pf_vad* pf_vad_create(const pf_vad_config* cfg);
void pf_vad_destroy(pf_vad* v);
int pf_vad_set_tensor(pf_vad* v, const char* name, const float* data, int64_t numel);
int pf_vad_missing(const pf_vad* v);
/* feats_dev [B, T, input_dim]; cache_dev [B, fsmn_layers, (lorder-1)*lstride, proj_dim]: left context of the memory
 * blocks, read and updated in place (NULL = zero left context); p_sil_dev [B, T]; probs_dev [B, T, output_dim] or NULL;
 * small_m != 0: weight-streaming GEMMs (chunks of a few frames). No sync. */
int pf_vad_forward(pf_vad* v, const float* feats_dev, int32_t B, int32_t T, float* cache_dev, const int32_t* sil_ids_host,
                   int32_t n_sil, float* p_sil_dev, float* probs_dev, int32_t small_m, void* stream);
int pf_vad_frame_decibel(const float* wav_dev, int64_t n_samples, int32_t n_frames, int32_t frame_len, int32_t frame_shift,
                         float* out_dev, void* stream);

/* ---- FSMN-VAD decision logic in native host code (no device work): silence posteriors + frame energies -> speech
 * segments; restates what FsmnVADStreaming.forward does after the network (model.py:552-905,1117-1302). Options = the
 * post-processing section of the model's config.yaml (VADXOptions, model.py:71-173). */
typedef struct pf_vad_decision pf_vad_decision;
typedef struct pf_vad_options {
    int32_t sample_rate, detect_mode, max_end_silence_time, max_start_silence_time, window_size_ms,
        sil_to_speech_time_thres, speech_to_sil_time_thres, do_extend, lookback_time_start_point,
        lookahead_time_end_point, max_single_segment_time, noise_frame_num_used_for_snr, frame_in_ms, frame_length_ms;
    double speech_2_noise_ratio, snr_thres, decibel_thres, speech_noise_thres, fe_prior_thres;
} pf_vad_options;
pf_vad_decision* pf_vad_decision_create(const pf_vad_options* opts);
void pf_vad_decision_destroy(pf_vad_decision* d);
/* the per-block dynamic end-silence schedule of FsmnVADStreaming.inference (model.py:1005-1019) */
void pf_vad_decision_set_thresholds(pf_vad_decision* d, double max_end_sil_ms, double speech_noise_thres);
int pf_vad_decision_state(const pf_vad_decision* d);   /* 1 start point not detected, 2 in speech, 3 end point detected */
/* next block of frames (host arrays). Returns the number of reportable segments (written as [beg_ms, end_ms] pairs up to
 * `capacity`; with streaming_events a started segment is [beg, -1] and closed later by [-1, end]); -1 on error. */
int pf_vad_decision_push(pf_vad_decision* d, const float* sil_scores, const float* decibels, int32_t n_frames,
                         int32_t is_final, int32_t streaming_events, int32_t* segments_out, int32_t capacity);

typedef struct pf_ctc pf_ctc;
pf_ctc* pf_ctc_create(int32_t d_model, int32_t vocab_size);
void pf_ctc_destroy(pf_ctc* c);
int pf_ctc_set_tensor(pf_ctc* c, const char* name, const float* data, int64_t numel);   /* "ctc_lo.weight|bias" */
int pf_ctc_missing(const pf_ctc* c);
/* hidden_dev: [M, d_model]; ids_dev: int32 [M] frame-wise argmax; logits_dev (nullable): [M, vocab]. No sync. */
int pf_ctc_greedy(pf_ctc* c, const float* hidden_dev, int32_t M, int32_t* ids_dev, float* logits_dev, void* stream);
/* 0 = fp32 MFMA (default), 3 = the arg-max route (logits_dev == NULL) from two-plane fp16 operands on the fp16 MFMA
 * (fp32-class logits; the modes of pf_encoder_set_precision) */
int pf_ctc_set_precision(pf_ctc* c, int32_t mode);

/* ----------------------------------------------------------------------------------------------- streaming
 * Chunked online decoding (ParaformerStreaming, funasr/models/paraformer_streaming/model.py:552-763): one handle =
 * a lock-step batch of n_streams independent streams (the reference: exactly 1) with all caches in HBM
 * (encoder K/V rings funasr/models/sanm/attention.py:343-361, overlap window funasr/models/scama/encoder.py:480-494,
 * CIF remainder cif_predictor.py:347-392, decoder FSMN / cross-attention caches attention.py:606-625,829-841).
 * The steady-state step is captured in a hipGraph per (n_frames, is_final, tail_chunk) and replayed. */
typedef struct pf_stream pf_stream;

typedef struct pf_stream_config {
    int32_t n_streams;        /* streams advanced together (>= 1) */
    int32_t chunk_left;       /* chunk_size[0] = 0 */
    int32_t chunk_cur;        /* chunk_size[1] = 10 LFR frames = 600 ms */
    int32_t chunk_right;      /* chunk_size[2] = 5 look-ahead frames */
    int32_t enc_look_back;    /* encoder_chunk_look_back (chunks), 4 */
    int32_t dec_look_back;    /* decoder_chunk_look_back (chunks), 1 */
    int32_t max_frames;       /* most feature frames a step may bring (>= chunk_cur) */
    int32_t max_tokens;       /* token rows decoded per stream per step (<= 96) */
    int32_t use_graph;        /* 1: capture + replay the step as a hipGraph */
} pf_stream_config;

/* the three handles must outlive the stream and carry all their tensors */
pf_stream* pf_stream_create(pf_encoder* e, pf_predictor* p, pf_decoder* d, const pf_stream_config* cfg);
void pf_stream_destroy(pf_stream* s);
/* sinusoidal position table, row r = position r+1 (StreamSinusoidalPositionEncoder, embedding.py:468-482); host or
 * device pointer, [rows, input_dim]. Default: 4096 rows computed with libm. */
int pf_stream_set_pe(pf_stream* s, const float* pe, int32_t rows);
int pf_stream_reset(pf_stream* s, void* stream);
/* "gemm_mode": 0 (default) = the step's GEMMs on the fp32 kernels (weight-streaming GEMM for a few rows: the latency path of
 * one or a few streams); 3 = on the fp16 matrix cores with two-plane fp16 operands and fp32 results (the offline f16x2 mode's
 * arithmetic, pf_encoder_set_precision 3: the throughput path of many lock-step streams). Attention, FSMN, CIF and the caches
 * stay fp32 in both. The reference has one arithmetic (torch fp32, paraformer_streaming/model.py:552-763); both modes meet
 * its parity bars. Synchronises, prepares the weight planes, drops captured graphs.
 * A/B switches of the step's launch fusions (defaults in brackets; all but "ln_carry" return the bits of the
 * separate launches): "ln_carry" [2] fp32 step: LayerNorms carried between the small-M GEMMs (0 never, 1 steps of <= 32 rows,
 * 2 always; one-pass variance: fp32-class, not the bits of the stand-alone LayerNorm); "fsmn_rides" [1] the encoder's FSMN memory
 * block inside the attention launch; "kv_batched" [1] fp32 step: the decoder's key/value projections of the step's encoder rows
 * and the ring appends as one launch each; "ln_folded" [1] f16x2 step: LayerNorms in the second launch of the split-K
 * projections, attention writing the out-projection's operand planes; "short_k" [3] f16x2 step of a small handle (<= 2048 rows):
 * linear_out in the four-slice split-K form with norm2 in its second launch (0 never, 1 / 2 every K = d_model projection: break-even);
 * "wide_k" [0] four workgroups per tile for the long-K
 * projections of a <= 32-row step (bitwise, slower: measured and kept off). */
int pf_stream_set_option(pf_stream* s, const char* key, int32_t value);
/* One chunk for every stream. feats_dev: [n_streams, n_frames, input_dim] un-scaled online features (ignored for a
 * tail chunk, which re-feeds the cached window, model.py:715-720). Outputs: ids_host int32 [n_streams, max_tokens]
 * (raw arg-max ids incl. sos/eos/blank), n_tokens_host int32 [n_streams]; optional enc_out_dev
 * [n_streams, W, d_model] (W = chunk_left + chunk_right + n_frames, or chunk_left + chunk_right for a tail chunk).
 * SYNCHRONISES (returns host values). */
int pf_stream_step(pf_stream* s, const float* feats_dev, int32_t n_frames, int32_t is_final, int32_t tail_chunk,
                   int32_t* ids_host, int32_t* n_tokens_host, float* enc_out_dev, void* stream);
/* pf_stream_step in two halves: _begin enqueues the step on the handle's own HIP stream and returns at once, _end waits for it and
 * fills ids_host / n_tokens_host. Handles built over DIFFERENT encoder / predictor / decoder handles may each have a step in
 * flight: steps of a few dozen streams leave most of the chip idle and overlap there (tools/bench_streaming.py --replicas). */
int pf_stream_step_begin(pf_stream* s, const float* feats_dev, int32_t n_frames, int32_t is_final, int32_t tail_chunk,
                         float* enc_out_dev, void* stream);
int pf_stream_step_end(pf_stream* s, int32_t* ids_host, int32_t* n_tokens_host);
/* debug / parity: carried CIF state and position counter (any pointer may be NULL) */
int pf_stream_peek(pf_stream* s, float* cif_alpha_host, float* cif_hidden_host, int32_t* start_idx_host);

/* ------------------------------------------------------------------------------------------------------------------
 * Utterance-level data parallelism over the GPUs of one node (dp_rccl.hip): ONE PROCESS PER GPU, RCCL over xGMI.
 * Replaces the reference's multi-GPU recipe -- one inference process per GPU over a split wav.scp, the outputs concatenated
 * (examples/aishell/paraformer/run.sh:135-190) -- for hosts that bind this header directly; funasr_amd/dp.py is the same
 * split on torch.distributed. The path shards over independent clips: there is NO collective on the data path. The two
 * exchanges: the weights once (rank `root` -> every rank, written straight into the handles' device storage) and a gather of
 * fixed-stride int32 hypotheses per batch.
 *   rank 0:      n = pf_dp_unique_id(id, 128)            -> ship the 128 bytes to the other ranks (any channel: a file, MPI, a socket)
 *   every rank:  hipSetDevice(local_rank); dp = pf_dp_create(id, 128, world, rank)     (collective: all ranks call it)
 *                pf_dp_broadcast_encoder(dp, enc, 0, stream) ... _predictor / _decoder / _ctc   (collective; handles built from
 *                                                                  the same config on every rank, tensors set on the root only)
 *                per batch: pf_dp_gather_ids(dp, my_ids_dev, n, all_ids_dev_on_root, 0, stream)
 * RCCL is loaded on first use (dlopen): without it every pf_dp_* call fails with pf_last_error() and nothing else changes. */
typedef struct pf_dp pf_dp;
/* rank 0 only: fills the 128-byte RCCL unique id; returns 128, or < 0 */
int pf_dp_unique_id(void* id_out, int32_t cap);
/* communicator of `world` ranks on the CURRENT device (one rank per device; collective). NULL on failure. */
pf_dp* pf_dp_create(const void* unique_id, int32_t id_bytes, int32_t world, int32_t rank);
int pf_dp_destroy(pf_dp* dp);
int pf_dp_world(const pf_dp* dp);
int pf_dp_rank(const pf_dp* dp);
/* rank `root`'s weights -> the same handle on every rank, in place in library-owned HBM (one grouped broadcast over the
 * handle's tensors); every tensor must be set on the root; on return (stream synchronised) the handle is ready on all ranks */
int pf_dp_broadcast_encoder(pf_dp* dp, pf_encoder* e, int32_t root, void* stream);
int pf_dp_broadcast_predictor(pf_dp* dp, pf_predictor* p, int32_t root, void* stream);
int pf_dp_broadcast_decoder(pf_dp* dp, pf_decoder* d, int32_t root, void* stream);
int pf_dp_broadcast_ctc(pf_dp* dp, pf_ctc* c, int32_t root, void* stream);
/* any device buffer (parameters that live outside the module handles): rank `root`'s bytes -> every rank, in place. Enqueued on
 * `stream`; does not synchronise. */
int pf_dp_broadcast_raw(pf_dp* dp, void* buf_dev, int64_t bytes, int32_t root, void* stream);
/* ids_dev: `count` int32 on this rank's device (e.g. [B, 1 + n_pad]: token count, then ids); out_dev (root only):
 * world * count int32, rank r's block at r * count. Enqueued on `stream`; does not synchronise. */
int pf_dp_gather_ids(pf_dp* dp, const int32_t* ids_dev, int64_t count, int32_t* out_dev, int32_t root, void* stream);

/* online frontend pieces (WavFrontendOnline, wav_frontend.py:395-505): log-mel of one buffer, and LFR + CMVN over a
 * frame buffer that already carries its left context */
int pf_frontend_fbank(pf_frontend* f, const float* wav_dev, int64_t n_samples, float* fbank_dev, void* stream);
int pf_frontend_lfr_cmvn(pf_frontend* f, const float* frames_dev, int32_t T, int32_t rows, float* out_dev, void* stream);

/* Test hooks: fill every activation workspace of the handle with `byte` (use 0x7B: huge finite fp16 / fp32 patterns). A
 * forward must not depend on what earlier batches left there (tests/test_stateless_gpu.py). Synchronise. */
int pf_encoder_debug_poison(pf_encoder* e, int32_t byte);
int pf_decoder_debug_poison(pf_decoder* d, int32_t byte);
int pf_predictor_debug_poison(pf_predictor* p, int32_t byte);
/* -------------------------------------------------------------------------------- single kernels (tests / bench)
 * Thin wrappers over the individual gfx950 kernels so that parity tests and the roofline bench can drive one
 * kernel at a time through the same ABI. All pointers are device pointers. */
/* small-M GEMM with a LayerNorm carried between GEMMs (the streaming step's fused LayerNorms): see gemm_skinny.hip.
 * stats_out [M][N / 16][2]: the block partials (sum, sum of squares) of the finished outputs; out_gamma [N] (with stats_out): the
 * stored values are C * out_gamma. stats_in [M][K / 16][2] + ln_c (2 N floats from pf_k_ln_consts): C = W LayerNorm(A) + bias
 * evaluated as rstd (W (gamma A) - mean c1) + c2; ln_g = gamma, or NULL when A already holds gamma A.
 * ws_part (>= tiles * 16 * 512 floats) + ws_count (one zeroed int32 per tile; tiles = ceil(N / 16) * ceil(M / 16 or 32)), M <= 32:
 * the four-workgroups-per-tile form for long K, the same bits */
int pf_k_gemm_skinny_ln(const float* A, int32_t lda, const float* W, int32_t ldw, const float* bias, const float* R2, int32_t ldr2,
                        float* Cout, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t relu, float* stats_out,
                        const float* stats_in, const float* ln_g, const float* ln_c, float ln_eps, const float* out_gamma, float* ws_part,
                        int32_t* ws_count, void* stream);
int pf_k_ln_consts(const float* W, int32_t ldw, int32_t N, int32_t K, const float* gamma, const float* beta, const float* bias, float* c,
                   void* stream);
/* test hook: pf_k_gemm_f32 takes the small-M weight-streaming kernel (the streaming step's GEMM) for M <= m;
 * default 0 = the 128x128 tile kernel (the offline path's GEMM) */
int pf_set_skinny_max_m(int32_t m);
int pf_k_gemm_f32(const float* A, int32_t lda, const float* W, int32_t ldw, const float* bias, const float* R1,
                  int32_t ldr1, const float* R2, int32_t ldr2, float* C, int32_t ldc, int32_t M, int32_t N,
                  int32_t K, int32_t relu, void* stream);
/* 1 if the library holds the measured-and-off kernel shapes and ablation builds (libparaformer_hip_measure.so, `make -C funasr_amd/csrc
 * measure`: what tools/bench_*.py and the records under profiles/ were made with); 0 for the product library, in which the option keys
 * and tile ids of those shapes return an error. */
int pf_measurement_build(void);
/* Conv1d over time as one exact-fp32 GEMM whose im2col is gathered by the operand loads (the predictor's cif_conv1d,
 * funasr/models/paraformer/cif_predictor.py:275-278): hidden [B, T, D], W [N, taps * D] (column tap * D + c = weight[n, c, tap]),
 * C [B * T, N]; rows t + tap - left outside [0, T) read as zero; zero_dev: >= 128 B of zeros; D % 32 == 0. Bitwise pf_k_gemm_f32
 * on the materialised column matrix. */
int pf_k_conv1d_gemm_f32(const float* hidden, const float* W, const float* bias, float* C, int32_t B, int32_t T, int32_t D,
                         int32_t N, int32_t taps, int32_t left, int32_t relu, const float* zero_dev, void* stream);
/* bf16-operand GEMM of the throughput mode: A [M,K] bf16, W [N,K] bf16 (strides in elements), fp32 accumulate on
 * v_mfma_f32_32x32x16_bf16, fp32 bias / residuals, C fp32 (c_bf16 = 0) or bf16 (1). K % 64 == 0. */
int pf_k_gemm_bf16(const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias, const float* R1,
                   int32_t ldr1, const float* R2, int32_t ldr2, void* C, int32_t ldc, int32_t M, int32_t N, int32_t K,
                   int32_t relu, int32_t c_bf16, void* stream);
int pf_k_gemm_bf16_time(const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias, void* C, int32_t ldc,
                        int32_t M, int32_t N, int32_t K, int32_t c_bf16, int32_t iters, float* ms_out, void* stream);
/* fp32-accurate GEMM on the bf16 matrix cores (gemm_split3.hip): both operands as three bf16 planes
 * (x = hi + mid + lo exactly; plane p of a matrix starts `*_plane` elements after plane p-1), six bf16 MFMA products
 * per operand pair, fp32 accumulate / bias / residuals. Output fp32 C, or (C3 != NULL) the three planes of the result.
 * K % 32 == 0 (planes zero-padded), N % 4 == 0, strides in elements. */
int pf_k_split3(const float* x, int32_t ldx, void* y3, int32_t ldy, int64_t plane, int32_t M, int32_t N, void* stream);
int pf_k_gemm_split3(const void* A3, int32_t lda, int64_t a_plane, const void* W3, int32_t ldw, int64_t w_plane,
                     const float* bias, const float* R1, int32_t ldr1, const float* R2, int32_t ldr2, float* C,
                     int32_t ldc, void* C3, int32_t ldc3, int64_t c_plane, int32_t M, int32_t N, int32_t K,
                     int32_t relu, int32_t iters, float* ms_out, void* stream);
/* the kernels of the headline mode, one at a time (tests/test_kernels_f16x2_gpu.py):
 * full-row form of the N = 512 projections with the residual adds and the following LayerNorm in the epilogue
 * (gemm_f16x2_row.hip; replaces linear_out / w_2 + LayerNorm of funasr/models/sanm/encoder.py:120-146) */
int pf_k_gemm_f16x2_row(const void* A2, int32_t lda, int64_t a_plane, const void* W2, int32_t ldw, int64_t w_plane, float oscale,
                        const float* bias, const float* R1, int32_t ldr1, const float* R2, int32_t ldr2, float* C, int32_t ldc,
                        const float* ln_g, const float* ln_b, float ln_eps, void* Y2, int64_t y_plane, float yscale, float* Yf,
                        int32_t M, int32_t K, int32_t relu, int32_t a_nt, int32_t iters, float* ms_out, void* stream);
/* The encoder block's position-wise feed-forward in ONE launch (gemm_f16x2_ffn.hip; funasr/models/transformer/
 * positionwise_feed_forward.py:14-34 with the residual add of sanm/encoder.py:141-146 and, optionally, the next LayerNorm):
 * Cout = R + (relu(X W1^T + b1) W2^T + b2); with ln_g: Y2 (planes of LayerNorm(Cout) * yscale) or Yf (fp32). X2 [2][M, 512],
 * W1 [2][F, 512], W2 [2][512, F] are two-plane fp16 operands (hi plane, then lo plane); oscale1 = 2^-(e_x + e_w1), hscale = the
 * hidden activations' plane scale 2^e_h, oscale2 = 2^-(e_h + e_w2). F % 128 == 0. Cout may alias R. */
int pf_k_ffn_f16x2(const void* X2, const void* W1, const void* W2, const float* b1, const float* b2, float oscale1, float hscale,
                   float oscale2, const float* R, float* Cout, const float* ln_g, const float* ln_b, float ln_eps, void* Y2,
                   float yscale, float* Yf, int32_t M, int32_t F, int32_t iters, float* ms_out, void* stream);
/* FSMN form of the full-row kernel: the first addend is the FSMN memory block (11 taps [512, 11], left padding 5;
 * funasr/models/sanm/attention.py:216-239) of the fp32 rows fs_v [M, 512], computed in the epilogue; fs_lo / fs_hi (device
 * int32 [M / 16]): valid input rows [lo, hi) of the sequence that owns each 16-row group. M % 16 == 0; ln_g / ln_b required */
int pf_k_gemm_f16x2_row_fsmn(const void* A2, int32_t lda, int64_t a_plane, const void* W2, int32_t ldw, int64_t w_plane, float oscale,
                             const float* bias, const float* fs_v, int32_t ldfv, const float* fs_w, const int32_t* fs_lo,
                             const int32_t* fs_hi, const float* R2, int32_t ldr2, float* C, int32_t ldc, const float* ln_g,
                             const float* ln_b, float ln_eps, void* Y2, int64_t y_plane, float yscale, float* Yf, int32_t M, int32_t K,
                             int32_t a_nt, int32_t iters, float* ms_out, void* stream);
/* LayerNorm writing the two fp16 planes of y * scale (what the f16x2 GEMMs read) */
int pf_k_layernorm_planes(const float* x, int32_t ldx, const float* gamma, const float* beta, void* y2, int32_t ldy, int64_t plane,
                          float scale, int32_t M, int32_t D, float eps, int32_t iters, float* ms_out, void* stream);
/* QKV form (kv_form 0, N = 3 D: Q planes, K planes, fp32 V, V^T planes) / KV form (kv_form 1, N = 2 D: K planes, V^T planes) */
int pf_k_gemm_f16x2_qkv(const void* A2, int32_t lda, int64_t a_plane, const void* W2, int32_t ldw, int64_t w_plane, float oscale,
                        const float* bias, int32_t M, int32_t D, int32_t K, int32_t kv_form, void* Qp, void* Kp, int64_t qk_plane,
                        float* Vf, void* VT, int32_t ldvt, int64_t vt_plane, float q_mul, float k_mul, float v_mul, int32_t tile,
                        int32_t iters, float* ms_out, void* stream);
/* fused arg-max form (vocabulary / CTC projection): ids[row] = argmax_n, lowest index on ties; scratch [M, 2 ceil(N / 256)] */
int pf_k_gemm_f16x2_argmax(const void* A2, int32_t lda, int64_t a_plane, const void* W2, int32_t ldw, int64_t w_plane, float oscale,
                           const float* bias, int32_t M, int32_t N, int32_t K, int32_t* ids, float* scratch_val,
                           int32_t* scratch_idx, void* stream);
/* two-plane fp16 split operands (x * scale = hi + lo) and the fp32-accurate three-product GEMM on them (gemm_f16x2.hip) */
int pf_k_split2(const float* x, int32_t ldx, void* y2, int32_t ldy, int64_t plane, int32_t M, int32_t N, float scale,
                void* stream);
int pf_k_gemm_f16x2(const void* A2, int32_t lda, int64_t a_plane, const void* W2, int32_t ldw, int64_t w_plane,
                    float oscale, const float* bias, const float* R1, int32_t ldr1, const float* R2, int32_t ldr2,
                    float* C, int32_t ldc, void* C2, int32_t ldc2, int64_t c_plane, float cscale, int32_t M, int32_t N,
                    int32_t K, int32_t relu, int32_t tile, int32_t iters, float* ms_out, void* stream);
/* attention on two-plane fp16 operands (attention_f16x2.hip; the layouts the QKV / KV forms of gemm_f16x2.hip write) */
int pf_k_attention_f16x2(const void* Q2, int64_t q_plane, const void* K2, int64_t k_plane, const void* VT2, int32_t ldvt,
                         int64_t vt_plane, void* O2, int64_t o_plane, const int32_t* klens_dev, int32_t B, int32_t H,
                         int32_t Tp, int32_t Tq, float sscale, float oscale, int32_t variant, int32_t iters, float* ms_out,
                         void* stream);
/* fp32 -> bf16 (round to nearest even), n % 4 == 0 */
int pf_k_cast_bf16(const float* x, void* y, int64_t n, void* stream);
int pf_k_gemm_argmax_f32(const float* A, int32_t lda, const float* W, int32_t ldw, const float* bias, int32_t M,
                         int32_t N, int32_t K, int32_t* ids, float* scratch_val, int32_t* scratch_idx, void* stream);
/* row-wise log-softmax over N columns (the scores the beam search of paraformer/search.py consumes); in place allowed */
int pf_k_log_softmax(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t M, int32_t N, void* stream);
int pf_k_layernorm(const float* x, int32_t ldx, const float* gamma, const float* beta, float* y, int32_t ldy,
                   int32_t M, int32_t D, int32_t Dpad, float eps, void* stream);
int pf_k_fsmn(const float* in, int32_t ldin, const float* w, const float* R, int32_t ldr, float* out, int32_t ldo,
              const int32_t* lens_dev, int32_t B, int32_t T, int32_t C, int32_t K, int32_t left_pad, void* stream);
int pf_k_attention_f32(const float* Q, int32_t ldq, const float* K, int32_t ldk, const float* V, int32_t ldv,
                       float* O, int32_t ldo, const int32_t* klens_dev, int32_t B, int32_t H, int32_t Tq,
                       int32_t Tk, float scale, void* stream);
/* small heads (d_k <= 64, multiple of 4; the CT-Transformer's 8 x 32), Tk <= 1024: rows hold H heads of d_k columns */
int pf_k_attention_small(const float* Q, int32_t ldq, const float* K, int32_t ldk, const float* V, int32_t ldv,
                         float* O, int32_t ldo, const int32_t* klens_dev, int32_t B, int32_t H, int32_t d_k,
                         int32_t Tq, int32_t Tk, float scale, void* stream);
/* embedding lookup: out[i, :] = table[ids_dev[i], :] (ids clamped to [0, rows)); D % 4 == 0 */
int pf_k_gather_rows(const float* table, int32_t ld, int32_t rows, const int32_t* ids_dev, float* out, int32_t n,
                     int32_t D, void* stream);
/* same contract, both products on the bf16 MFMA from three-plane split operands (fp32-class results; mode bf16x3) */
int pf_k_attention_split3(const float* Q, int32_t ldq, const float* K, int32_t ldk, const float* V, int32_t ldv,
                          float* O, int32_t ldo, const int32_t* klens_dev, int32_t B, int32_t H, int32_t Tq,
                          int32_t Tk, float scale, void* stream);
/* bf16 twin of pf_k_attention_f32: Q/K/V/O bf16, strides in elements */
int pf_k_attention_bf16(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, void* O,
                        int32_t ldo, const int32_t* klens_dev, int32_t B, int32_t H, int32_t Tq, int32_t Tk, float scale,
                        void* stream);
/* torch.nn.LSTM (one layer, ndir = 1 | 2, zero initial state) on device tensors in torch's layouts: x [B, T, D],
 * w_ih [ndir][4H][D], w_hh [ndir][4H][H], b_ih / b_hh [ndir][4H] -> out [B, T, ndir * H]; H % 16 == 0, H <= 512, D % 32 == 0.
 * Synchronises (lstm.hip). */
int pf_k_lstm(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int32_t B, int32_t T,
              int32_t D, int32_t H, int32_t ndir, float* out, void* stream);
/* CIF integrate-and-fire on caller-provided weights (cif_v1, cif_predictor.py:853-908): alphas [B, T], hidden
 * [B, T, D] -> peaks [B, T] (= "fires"), n_fires int32 [B], embeds [B, N, D] (rows >= n_fires[b] zero). */
int pf_k_cif(const float* alphas, const float* hidden, int32_t B, int32_t T, int32_t D, int32_t N, float* peaks,
             int32_t* n_fires, float* embeds, void* stream);
/* average kernel time in milliseconds of `iters` back-to-back launches of the GEMM above, measured with
 * hipEvents on `stream` (used by bench.py for the roofline line) */
int pf_k_gemm_f32_time(const float* A, int32_t lda, const float* W, int32_t ldw, const float* bias, float* C,
                       int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t iters, float* ms_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PARAFORMER_HIP_H_ */
