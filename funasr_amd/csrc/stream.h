// Kernel argument blocks of the streaming (chunked) Paraformer step (see stream.hip).
#pragma once
#include "common.h"

namespace pf {

// device-resident step state shared by all streams of a lock-step batch; advanced ON DEVICE so that a captured
// hipGraph of the steady-state step can be replayed without patching kernel arguments
struct StreamDev {
    int start_idx;      // absolute position of the next feature frame (StreamSinusoidalPositionEncoder cache)
    int enc_valid;      // valid rows in the encoder K/V rings (<= capacity)
    int enc_wp;         // next write row of the encoder K/V rings
    int step;           // chunks processed
};

struct StreamEmbedArgs {
    const float* feats;        // [S, n, Din] un-scaled online features (nullptr for a tail chunk)
    const float* pe;           // [pe_rows, Din] sinusoidal table, row = absolute position (0-based)
    float* cache_feats;        // [S, keep, Din] last `keep` rows of the previous window (in/out)
    float* win;                // [S, W, Din] output window
    const StreamDev* st;
    int S, n, keep, Din, pe_rows, tail;
    float scale;
};
int launch_stream_embed(const StreamEmbedArgs& a, hipStream_t stream);

struct RingAppendArgs {
    const float* src; int ldsrc;     // rows (s * src_T + r0 + i), `cols` floats each
    int src_T, r0, rows, cols;
    float* ring;                     // [S, cap, cols]
    int cap, S;
    const StreamDev* st;             // uniform mode: write pointer = st->enc_wp
    const int* wp_dev;               // per-stream mode (st == nullptr): write pointer wp_dev[s]
    const int* gate_dev;             // per-stream mode: stream s is skipped when gate_dev[s] < 1
    // the same append for n_layers (src, ring) pairs src_layer / ring_layer floats apart (0 / 1: one pair): the decoder's sixteen
    // rings share the write pointers, and with the batched key/value projection every layer's rows outlive the token chain
    int n_layers = 0; size_t src_layer = 0, ring_layer = 0;
    // uniform mode, mod > 0: an encoder ring of `cap` rows whose first append (st->enc_valid == 0) is linear and untrimmed and whose
    // later appends run modulo `mod` = look_back * chunk_cur (AttnArgs.app_mod; sanm/attention.py:353-361)
    int mod = 0;
};
int launch_ring_append(const RingAppendArgs& a, hipStream_t stream);

// The one step after an untrimmed first append (st->enc_valid > mod): the rows of the first chunk that survive this step's trim and sit
// behind position `mod` move to their ring positions (global row index modulo `mod`), so that rows [0, mod) are again exactly the last
// `mod` rows appended. new_rows = this step's append. Does nothing in every other step.
struct RingFoldArgs {
    float* ring; size_t ring_layer; int n_layers, S, cap, mod, cols, new_rows;
    const StreamDev* st;
};
int launch_ring_fold(const RingFoldArgs& a, hipStream_t stream);

struct StreamAdvanceArgs {
    StreamDev* st; int n_frames, enc_rows, enc_cap;
    int enc_mod = 0;           // > 0: the ring's trim size (RingAppendArgs.mod); enc_cap is then its row count
    int* dec_valid; int* dec_wp; const int* gate; int S, dec_rows, dec_cap;
};
int launch_stream_advance_enc(const StreamAdvanceArgs& a, hipStream_t stream);
int launch_stream_advance_dec(const StreamAdvanceArgs& a, hipStream_t stream);

struct CifChunkArgs {
    const float* hidden;       // [S, W, D] encoder window output
    const float* alphas; int ld_alpha;   // [S, ld_alpha] raw alphas of the window (alpha_kernel)
    float* cif_hidden;         // [S, D] carried un-fired remainder (in/out)
    float* cif_alpha;          // [S]    carried weight (in/out)
    float* embeds;             // [S, Nmax, D] fired frames, rows >= n zero-filled
    int* n_fired;              // [S]
    int S, W, D, Nmax;
    int lo, hi;                // window frames outside [lo, hi) get alpha 0 (cif_predictor.py:343-346)
    int is_final; float tail_threshold, threshold;
};
int launch_cif_chunk(const CifChunkArgs& a, hipStream_t stream);

struct DecFsmnChunkArgs {
    const float* in;           // [S, N, C] LN2 output of this chunk's tokens
    const float* resid;        // [S, N, C] layer input (residual)
    float* out;                // [S, N, C]
    const float* w;            // [C, K]
    float* state;              // [S, K-1, C] left context carried across chunks (in/out)
    const int* n_valid;        // [S] tokens fired this chunk
    int S, N, C;
    // LayerNorms carried with the neighbouring small-M GEMMs (GemmArgs.ln_stats_*, gemm_skinny.hip): ln_stats_in [S * N][C / 16][2]
    // = the block partials of `in`'s rows, which is then the LayerNorm's INPUT (norm2 applied on the fetch); ln_stats_out receives
    // the partials of the output rows (for norm3 on the fetch of the query projection)
    const float* ln_stats_in = nullptr; const float* ln_g = nullptr; const float* ln_b = nullptr; float ln_eps = 0.f;
    float* ln_stats_out = nullptr;
};
int launch_dec_fsmn_chunk(const DecFsmnChunkArgs& a, hipStream_t stream);

int launch_fill_int(int* p, int n, int value, hipStream_t stream);

}  // namespace pf
