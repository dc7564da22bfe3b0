// LSTM recurrence + CifPredictorV3's upsampled head (lstm.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace pf {

struct LstmStepArgs {
    const float* pre;        // device [ndir * 4H][ld_pre] gates-major input projections (W_ih x, no bias), column t * B + b
    const float* whh;        // device [ndir][4H][H]       recurrent weights in torch's layout (gates i, f, g, o stacked)
    const float* b_ih;       // device [ndir * 4H]
    const float* b_hh;       // device [ndir * 4H]
    float* h_a;              // device ndir * H * Bs floats: state ping, in MFMA B-fragment order (lstm.hip)
    float* h_b;              // state pong
    float* c;                // device [ndir][H][Bs] cell state
    float* out;              // layout 0: [B][T][ndir * H] row-major; layout 1: [T][ndir * H][Bs] (unit-major, for the head)
    size_t ld_pre;
    int T, B, Bs, H, ndir, out_layout;
    // filled by the launcher per step
    int step;
    const float* h_prev;
    float* h_next;
};
// zeroes the state and runs the T steps (one launch per step) on `stream`
int launch_lstm_steps(const LstmStepArgs& a, hipStream_t stream);

int launch_rows_bt_to_tb(const float* in, float* out, int B, int T, int D, hipStream_t stream);

struct UsAlphaArgs {
    const float* out_t;      // device [T][C][Bs]
    const float* w;          // device [C]
    const float* bias;       // device [1]
    const int* lens;         // device [B] encoder frames per utterance (not upsampled)
    float* alphas;           // device [B][T]
    int B, Bs, T, C, U;      // T = U * encoder frames
    float smooth, noise;
};
int launch_us_alpha_t(const UsAlphaArgs& a, hipStream_t stream);

// alphas [B, T] *= token_num[b] / sum_t alphas[b, t]; peaks = cif_wo_hidden(alphas, threshold)
int launch_us_scale_scan(float* alphas, float* peaks, const int* token_num, int B, int T, float threshold,
                         hipStream_t stream);

}  // namespace pf
