// canon_layout.h -- offsets of the canonical (Keras-orientation, [in][out]) encoder weight blob;
// see qbold_encoder_num_params in include/qbold_hip.h.
#pragma once

#include <hip/hip_runtime.h>

namespace qb {

// Canonical blob offsets (floats); see qbold_encoder_num_params in include/qbold_hip.h.
struct CanonLayout {
    int T, U, L, G, taps;
    int W0, b0, blk0, blk_stride, Wf, bf, Ws, bs, total;
    int ln;   // GroupNormalization parameters behind the heads: [L][4][U] = gamma1, beta1, gamma2, beta2 per block (0: none)
    // inside a block
    int Wc, bc, Wr1, br1, Wr2, br2, Wg, bg;
};
__host__ __device__ inline CanonLayout make_canon(int T, int U, int L, int cw, int taps = 1, int layer_norm = 0) {
    CanonLayout c;
    c.T = T; c.U = U; c.L = L; c.G = cw ? U : 1;
    c.taps = taps == 9 ? 9 : 1;
    c.W0 = 0;
    c.b0 = T * U;
    c.blk0 = c.b0 + U;
    c.Wc = 0;
    c.bc = U * U;
    c.Wr1 = c.bc + U;
    c.br1 = c.Wr1 + c.taps * U * U;
    c.Wr2 = c.br1 + U;
    c.br2 = c.Wr2 + c.taps * U * U;
    c.Wg = c.br2 + U;
    c.bg = c.Wg + U * c.G;
    c.blk_stride = c.bg + c.G;
    c.Wf = c.blk0 + L * c.blk_stride;
    c.bf = c.Wf + U * 5;
    c.Ws = c.bf + 5;
    c.bs = c.Ws + U * T;
    c.total = c.bs + T;
    c.ln = 0;
    if (layer_norm) {   // model.py:139: two normalizers per block (model.py:150, 154), per-channel scale and offset
        c.ln = c.total;
        c.total += L * 4 * U;
    }
    return c;
}

}  // namespace qb
