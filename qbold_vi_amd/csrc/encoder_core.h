// encoder_core.h -- the voxel-wise encoder MLP on the CDNA4 matrix cores (exact-f32 MFMA).
//
// Reference: EncoderTrainer.normalise_data (model.py:97-113) and create_encoder (model.py:122-223)
// for (N,1,1,1,T) voxel batches, where the 3x3x1 convolutions of stream 2 act through their centre
// tap.  Per voxel:  n = log(clip(x)/clip(x)[se]);  h = relu(W0 n + b0);
//   stream 1:  a <- relu(Wc a + bc)
//   stream 2:  skip = relu(Wc b + bc);  r = Wr2 relu(Wr1 relu(b) + br1) + br2;
//              g = sigmoid(Wg r + bg + gate_offset);  b <- skip (1-g) + r g
//   heads:     out = Wf . + bf  (5),   sigma = exp(Ws b + bs)  (T)
//
// Mapping.  One wave owns a tile of 32 voxels.  Every dense layer is computed TRANSPOSED,
// Y^T[unit][voxel] = W^T[unit][k] X^T[k][voxel], with v_mfma_f32_32x32x2_f32: the weights are the A
// operand (one LDS read per k-step), the activations the B operand.  The 32x32 accumulator puts
// the voxel on the lane (col = lane & 31) and 16 units in the registers of each half-wave
// (row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)), which is exactly what the next layer's B
// operand wants for k-step `reg` if the weight image is stored in that k order -- so activations
// never leave the register file and never touch LDS.  U <= 64 is padded to two 32-unit tiles.
//
// LDS weight image (built by pack_kernel in encoder_kernels.hip), per dense op with MT output
// tiles:  A[kstep][half][i = 0..31][m_out = 0..MT-1]  (lane (half, i) reads MT consecutive
// floats: conflict-free ds_read_b64 / b32),  bias[m_out][half][reg = 0..15].
#pragma once

#include "qbold_dev.h"

namespace qb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define QB_ENC_MAX_L 8

// Offsets (in floats) of every piece of the packed image; wave-uniform kernel argument.
struct EncLayout {
    int T, U, L;
    int ksteps_first;   // ceil(T / 2)
    int n_head;         // 5 + T outputs of the merged head tile
    int first_A, first_b;
    int blk0;           // offset of block 0
    int blk_stride;     // floats per block
    int head_A, head_b;
    int total;
};
// inside one block
enum { BLK_WC_A = 0, BLK_WC_B = 4096, BLK_R1_A = 4160, BLK_R1_B = 8256, BLK_R2_A = 8320,
       BLK_R2_B = 12416, BLK_G_A = 12480, BLK_G_B = 16576, BLK_FLOATS = 16640 };

__host__ __device__ inline EncLayout make_enc_layout(int T, int U, int L) {
    EncLayout e;
    e.T = T; e.U = U; e.L = L;
    e.ksteps_first = (T + 1) / 2;
    e.n_head = 5 + T;
    e.first_A = 0;
    e.first_b = e.ksteps_first * 128;
    e.blk0 = e.first_b + 64;
    e.blk_stride = BLK_FLOATS;
    e.head_A = e.blk0 + L * BLK_FLOATS;
    e.head_b = e.head_A + 2048;
    e.total = e.head_b + 32;
    e.total = (e.total + 3) & ~3;
    return e;
}

// unit held by accumulator register r of half h in tile m
__host__ __device__ inline int acc_unit(int m, int r, int h) {
    return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
}

__device__ __forceinline__ f32x16 relu16(f32x16 v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.0f);
    return v;
}

__device__ __forceinline__ f32x16 load_bias16(const float* __restrict__ b) {
    const float4* p = reinterpret_cast<const float4*>(b);
    float4 a = p[0], c = p[1], d = p[2], e = p[3];
    f32x16 v;
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    v[8] = d.x; v[9] = d.y; v[10] = d.z; v[11] = d.w;
    v[12] = e.x; v[13] = e.y; v[14] = e.z; v[15] = e.w;
    return v;
}

// out[0..1] = W in + bias for a 64 -> 64 layer.  A: LDS image, bias: LDS [2][2][16].
__device__ __forceinline__ void dense64(const float* __restrict__ A, const float* __restrict__ bias,
                                        const f32x16 (&in)[2], f32x16 (&out)[2], int h, int i) {
    out[0] = load_bias16(bias + h * 16);
    out[1] = load_bias16(bias + 32 + h * 16);
    const float2* Ap = reinterpret_cast<const float2*>(A) + h * 32 + i;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float2 a = Ap[(m * 16 + r) * 64];
            out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, in[m][r], out[0], 0, 0, 0);
            out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, in[m][r], out[1], 0, 0, 0);
        }
    }
}

// head: one 32-row output tile from a 64-unit input.  A image has MT = 1.
__device__ __forceinline__ f32x16 dense_head(const float* __restrict__ A,
                                             const float* __restrict__ bias,
                                             const f32x16 (&in)[2], int h, int i) {
    f32x16 out = load_bias16(bias + h * 16);
    const float* Ap = A + h * 32 + i;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap[(m * 16 + r) * 64], in[m][r], out, 0, 0, 0);
    }
    return out;
}

// normalise_data -- model.py:97-113; n[t] for this lane's voxel.
template <int T>
__device__ __forceinline__ void normalise(const QbDev& c, const float (&x)[T], float (&n)[T]) {
    float cl[T];
#pragma unroll
    for (int t = 0; t < T; ++t) cl[t] = clampf_(x[t], 1e-2f, 1e8f);  // model.py:101
    const int se = c.se_idx;
    float a = 0.0f, b = 0.0f, d = 0.0f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        a = (t == se - 1) ? cl[t] : a;
        b = (t == se) ? cl[t] : b;
        d = (t == se + 1) ? cl[t] : d;
    }
    const float den = c.multi_norm ? (a + b + d) / 3.0f : b;  // model.py:104 / :106
#pragma unroll
    for (int t = 0; t < T; ++t) n[t] = __logf(cl[t] / den);  // model.py:108
}

// First layer: T -> 64 with relu.  The B operand of k-step s is n[2s + half].
template <int T>
__device__ __forceinline__ void dense_first(const float* __restrict__ A,
                                            const float* __restrict__ bias, const float (&n)[T],
                                            f32x16 (&out)[2], int h, int i) {
    out[0] = load_bias16(bias + h * 16);
    out[1] = load_bias16(bias + 32 + h * 16);
    const float2* Ap = reinterpret_cast<const float2*>(A) + h * 32 + i;
    constexpr int KS = (T + 1) / 2;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const float lo = n[2 * s];
        const float hi = (2 * s + 1 < T) ? n[(2 * s + 1 < T) ? 2 * s + 1 : 0] : 0.0f;
        const float b = h ? hi : lo;
        const float2 a = Ap[s * 64];
        out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, out[0], 0, 0, 0);
        out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, out[1], 0, 0, 0);
    }
    out[0] = relu16(out[0]);
    out[1] = relu16(out[1]);
}

// One create_block step of stream 2 (gated residual), in place -- model.py:147-172.
__device__ __forceinline__ void block_stream2(const float* __restrict__ W, f32x16 (&b)[2], int h,
                                              int i) {
    f32x16 skip[2], t[2], r[2];
    dense64(W + BLK_WC_A, W + BLK_WC_B, b, skip, h, i);  // shared 1x1x1 conv as skip, :148
    skip[0] = relu16(skip[0]);
    skip[1] = relu16(skip[1]);
    b[0] = relu16(b[0]);  // Activation before the first 3x3x1 conv, :151
    b[1] = relu16(b[1]);
    dense64(W + BLK_R1_A, W + BLK_R1_B, b, t, h, i);  // :152
    t[0] = relu16(t[0]);                              // :155
    t[1] = relu16(t[1]);
    dense64(W + BLK_R2_A, W + BLK_R2_B, t, r, h, i);  // :156
    dense64(W + BLK_G_A, W + BLK_G_B, r, t, h, i);    // gating logits (+ gate_offset in bias), :164
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float g = sigmoidf_(t[m][k]);                    // :169
            b[m][k] = skip[m][k] * (1.0f - g) + r[m][k] * g;       // :170
        }
    }
}

// One create_block step of stream 1 -- model.py:144-145.
__device__ __forceinline__ void block_stream1(const float* __restrict__ W, f32x16 (&a)[2], int h,
                                              int i) {
    f32x16 o[2];
    dense64(W + BLK_WC_A, W + BLK_WC_B, a, o, h, i);
    a[0] = relu16(o[0]);
    a[1] = relu16(o[1]);
}

// Broadcast both half-waves' copies of an accumulator register to every lane:
// lo = value held by lanes 0-31, hi = value held by lanes 32-63 (same voxel = lane & 31).
__device__ __forceinline__ void both_halves(float v, float& lo, float& hi) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}

// Head outputs of this lane's voxel, gathered to every lane: o[k], k < NOUT (<= 32), where head
// row k sits in register (k & 3) + 4 (k >> 3) of half (k >> 2) & 1.
template <int NOUT>
__device__ __forceinline__ void gather_head(const f32x16& acc, float (&o)[NOUT]) {
    constexpr int NREG = ((NOUT + 7) / 8) * 4;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        float lo, hi;
        both_halves(acc[r], lo, hi);
        const int k_lo = (r & 3) + 8 * (r >> 2), k_hi = k_lo + 4;
        if (k_lo < NOUT) o[k_lo < NOUT ? k_lo : 0] = lo;
        if (k_hi < NOUT) o[k_hi < NOUT ? k_hi : 0] = hi;
    }
}

}  // namespace qb
