// Kernel argument blocks of the CIF predictor and the greedy arg-max tail (see cif.hip).
#pragma once
#include "common.h"

namespace pf {

// [B*T, taps*D] im2col view of hidden for the dense Conv1d(D, D, l+r+1) of CifPredictorV2
int launch_im2col(const float* hidden, float* out, int B, int T, int D, int l_order, int r_order, hipStream_t stream);

struct AlphaArgs {
    const float* conv;      // device [B*T, D] relu(conv1d(hidden))
    const float* w;         // device [D]   cif_output.weight
    const float* bias;      // device [1]   cif_output.bias
    const int* lens;        // device [B]
    float* alphas;          // device [B, T_ext] (row stride T_ext = T + 1); columns [0, T) written here
    int B, T, D, T_ext;
    float smooth, noise;
};
int launch_alpha(const AlphaArgs& a, hipStream_t stream);

struct CifScanArgs {
    float* alphas;          // device [B, T+1] in/out: the tail threshold is folded in and column T written
    float* peaks;           // device [B, T+1] "fires" = cif_peak
    float* rems;            // device [B, T+1] fires - floor(fires)
    int* fire_flag;         // device [B, T+1]
    int* n_fires;           // device [B]
    const int* lens;        // device [B]
    int B, T;
    float tail_threshold;
    int tail_mask;
};
int launch_cif_scan(const CifScanArgs& a, hipStream_t stream);

struct CifEmitArgs {
    const float* hidden;    // device [B, T, D]
    const float* alphas;    // device [B, T+1] (tail-extended)
    const float* rems;      // device [B, T+1]
    const int* fire_flag;   // device [B, T+1]
    float* embeds;          // device [B, N, D]
    int B, T, D, N;
};
int launch_cif_emit(const CifEmitArgs& a, hipStream_t stream);

// CifPredictorV3's sequential fp32 integrate-and-fire (`cif`, funasr/models/bicif_paraformer/cif_predictor.py:39-86).
// Scan: same arguments as launch_cif_scan plus `curs` [B, T+1] (the weight a frame contributes to the token being
// integrated: alpha, or 1 - integrate on a fire), `rems` = alpha - cur, `peaks` = the running integral before the
// threshold is taken off, `n_tok` [B] = floor(sum alphas) (the reference's token_num; n_fires counts the fires).
// Needs T + 1 <= 4096.
int launch_cif_scan_loop(const CifScanArgs& a, float* curs, int* n_tok, hipStream_t stream);
// Emit: frame += cur * hidden (product and sum rounded separately, like the reference's two tensor ops); a fire emits the
// frame and restarts it at rem * hidden. `a.alphas` = curs of the scan.
int launch_cif_emit_loop(const CifEmitArgs& a, hipStream_t stream);

// reduce per-row partial (max, argmax) pairs to token ids; rows >= n_valid[b] get `pad_id`
int launch_argmax_reduce(const float* pval, const int* pidx, int ld, int nparts, int* ids, float* best,
                         int M, hipStream_t stream);
// plain arg-max over materialised rows [M, N] (first maximum wins)
int launch_argmax_rows(const float* x, int ldx, int M, int N, int* ids, hipStream_t stream);

}  // namespace pf
