#!/usr/bin/env python3
"""CPU emulation of `lstm_step_kernel` (funasr_amd/csrc/lstm.hip): the same index arithmetic -- W_hh rows staged to LDS in
MFMA row order, the state in B-fragment order, the documented lane maps of `v_mfma_f32_16x16x4_f32`
(A[l & 15][l >> 4], B[l >> 4][l & 15], D[4 (l >> 4) + reg][l & 15]) and the lane-local cell update -- in numpy, compared
with a plain LSTM. Used to check the kernel's layouts before the one GPU run that was left to validate it.

    python tools/emulate_lstm_kernel.py        # prints the largest |difference| for three shapes (~1e-6)
"""
import numpy as np


def emulate(H, B, ndir=2, T=3, seed=0):
    rng = np.random.default_rng(seed)
    NM, RS, Q4, Bs = H // 16, H + 4, H // 4, (B + 63) // 64 * 64
    whh = rng.standard_normal((ndir, 4 * H, H)).astype(np.float32)
    pre = rng.standard_normal((ndir * 4 * H, T * B)).astype(np.float32)
    bi = rng.standard_normal(ndir * 4 * H).astype(np.float32)
    bh = rng.standard_normal(ndir * 4 * H).astype(np.float32)
    sig = lambda x: 1 / (1 + np.exp(-x))  # noqa: E731

    def state_index(d, k, b):
        q, s = divmod(k, H >> 2)
        return ((((d * (Bs >> 4) + (b >> 4)) * (H >> 4) + (s >> 2)) * 64 + q * 16 + (b & 15)) * 4 + (s & 3))

    ref = np.zeros((B, T, ndir * H), np.float32)
    for d in range(ndir):
        h, c = np.zeros((B, H), np.float32), np.zeros((B, H), np.float32)
        for step in range(T):
            t = step if d == 0 else T - 1 - step
            g = pre[d * 4 * H:(d + 1) * 4 * H, t * B:(t + 1) * B].T + bi[d * 4 * H:(d + 1) * 4 * H] + bh[d * 4 * H:(d + 1) * 4 * H] + h @ whh[d].T
            c = sig(g[:, H:2 * H]) * c + sig(g[:, :H]) * np.tanh(g[:, 2 * H:3 * H])
            h = sig(g[:, 3 * H:]) * np.tanh(c)
            ref[:, t, d * H:(d + 1) * H] = h

    out = np.zeros_like(ref)
    h_a, h_b, cst = (np.zeros(ndir * H * Bs, np.float32) for _ in range(3))
    for step in range(T):
        h_prev, h_next = (h_a, h_b) if step % 2 == 0 else (h_b, h_a)
        for d in range(ndir):
            t = step if d == 0 else T - 1 - step
            for bx in range(H // 4):                                   # blockIdx.x: 4 hidden units
                u0 = bx * 4
                s_w = np.zeros(16 * RS, np.float32)
                for f in range(16 * Q4):                               # cooperative float4 staging of the W slice
                    r, c4 = divmod(f, Q4)
                    src = ((r >> 2) * H + u0 + (r & 3)) * H + 4 * c4
                    dst = ((r & 3) * 4 + (r >> 2)) * RS + 4 * c4
                    s_w[dst:dst + 4] = whh[d].reshape(-1)[src:src + 4]
                for btile in range(Bs // 16):                          # blockIdx.z * 4 + wave
                    A = np.zeros((64, NM, 4), np.float32)
                    Bf = np.zeros((64, NM, 4), np.float32)
                    acc = np.zeros((64, 4), np.float32)
                    for lane in range(64):
                        for m in range(NM):
                            base = (((d * (Bs >> 4) + btile) * NM + m) * 64 + lane) * 4
                            Bf[lane, m] = h_prev[base:base + 4]
                            wr = (lane & 15) * RS + (lane >> 4) * Q4 + 4 * m
                            A[lane, m] = s_w[wr:wr + 4]
                        u, b = u0 + (lane >> 4), btile * 16 + (lane & 15)
                        for g in range(4):
                            n = d * 4 * H + g * H + u
                            acc[lane, g] = (pre[n, t * B + b] if b < B else 0.0) + (bi[n] + bh[n])
                    for m in range(NM):
                        for e in range(4):                             # one v_mfma_f32_16x16x4_f32
                            Am, Bm = np.zeros((16, 4), np.float32), np.zeros((4, 16), np.float32)
                            for lane in range(64):
                                Am[lane & 15, lane >> 4] = A[lane, m, e]
                                Bm[lane >> 4, lane & 15] = Bf[lane, m, e]
                            D = Am @ Bm
                            for lane in range(64):
                                for reg in range(4):
                                    acc[lane, reg] += D[(lane >> 4) * 4 + reg, lane & 15]
                    for lane in range(64):
                        u, b = u0 + (lane >> 4), btile * 16 + (lane & 15)
                        ci = (d * H + u) * Bs + b
                        cn = sig(acc[lane, 1]) * cst[ci] + sig(acc[lane, 0]) * np.tanh(acc[lane, 2])
                        hn = sig(acc[lane, 3]) * np.tanh(cn)
                        cst[ci] = cn
                        h_next[state_index(d, u, b)] = hn
                        if b < B:
                            out[b, t, d * H + u] = hn
    return float(np.abs(out - ref).max())


if __name__ == "__main__":
    print(emulate(32, 3), emulate(48, 70, ndir=1), emulate(64, 17))
