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
// Mapping.  One wave owns a tile of 16 voxels.  Every dense layer is computed TRANSPOSED,
// Y^T[unit][voxel] = W^T[unit][k] X^T[k][voxel], with v_mfma_f32_16x16x4_f32: the weights are the A
// operand (one 16-byte LDS read per k-step feeds the four 16-unit output tiles), the activations
// the B operand.  The 16x16 accumulator puts the voxel on the lane (col = lane & 15) and four
// units in the registers of each 16-lane group (row = 4 (lane >> 4) + reg), which is exactly what
// the next layer's B operand wants for k-step (tile, reg) if the weight image is stored in that k
// order -- so activations never leave the register file and never touch LDS.  A 64-unit
// activation tensor costs 16 VGPRs per lane, which keeps the whole fused kernel under 128 VGPRs
// (four waves per SIMD).  U <= 64 is padded to four 16-unit tiles.
//
// LDS weight image (built by pack_kernel in encoder_kernels.hip), per dense op with MT output
// tiles:  A[kstep][group = 0..3][i = 0..15][m_out = 0..MT-1]  (lane (group, i) reads MT
// consecutive floats; a wave reads one contiguous KiB: conflict-free),
// bias[m_out][group][reg = 0..3].
#pragma once

#include "qbold_dev.h"

namespace qb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Offsets (in floats) of every piece of the packed image; wave-uniform kernel argument.
struct EncLayout {
    int T, U, L;
    int ksteps_first;   // ceil(T / 4)
    int n_head;         // 5 + T outputs of the merged head
    int head_tiles;     // ceil(n_head / 16): 1 or 2
    int first_A, first_b;
    int blk0;           // offset of block 0
    int blk_stride;     // floats per block
    int head_A, head_b;
    int total;
};
// inside one block: four dense ops, each 4096 (A) + 64 (bias) floats
enum { BLK_WC_A = 0, BLK_WC_B = 4096, BLK_R1_A = 4160, BLK_R1_B = 8256, BLK_R2_A = 8320,
       BLK_R2_B = 12416, BLK_G_A = 12480, BLK_G_B = 16576, BLK_FLOATS = 16640 };

__host__ __device__ inline EncLayout make_enc_layout(int T, int U, int L) {
    EncLayout e;
    e.T = T; e.U = U; e.L = L;
    e.ksteps_first = (T + 3) / 4;
    e.n_head = 5 + T;
    e.head_tiles = (e.n_head + 15) / 16;
    e.first_A = 0;
    e.first_b = e.ksteps_first * 256;
    e.blk0 = e.first_b + 64;
    e.blk_stride = BLK_FLOATS;
    e.head_A = e.blk0 + L * BLK_FLOATS;
    e.head_b = e.head_A + 1024 * e.head_tiles;
    e.total = e.head_b + 16 * e.head_tiles;
    e.total = (e.total + 3) & ~3;
    return e;
}

// unit held by accumulator register r of lane group g in 16-unit tile m
__host__ __device__ inline int acc_unit(int m, int r, int g) { return 16 * m + 4 * g + r; }

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
    return v;
}

__device__ __forceinline__ f32x4 load4(const float* __restrict__ p) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    f32x4 v;
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    return v;
}

#define QB_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// out[0..3] = W in + bias for a 64 -> 64 layer.  A: LDS image, bias: LDS [4][4][4].
__device__ __forceinline__ void dense64(const float* __restrict__ A, const float* __restrict__ bias,
                                        const f32x4 (&in)[4], f32x4 (&out)[4], int g, int i) {
#pragma unroll
    for (int m = 0; m < 4; ++m) out[m] = load4(bias + m * 16 + g * 4);
    const float4* Ap = reinterpret_cast<const float4*>(A) + g * 16 + i;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 a = Ap[(m * 4 + r) * 64];
            out[0] = QB_MFMA16(a.x, in[m][r], out[0]);
            out[1] = QB_MFMA16(a.y, in[m][r], out[1]);
            out[2] = QB_MFMA16(a.z, in[m][r], out[2]);
            out[3] = QB_MFMA16(a.w, in[m][r], out[3]);
        }
    }
}

// heads: HT (1 or 2) 16-row output tiles from a 64-unit input.
template <int HT>
__device__ __forceinline__ void dense_head(const float* __restrict__ A,
                                           const float* __restrict__ bias, const f32x4 (&in)[4],
                                           f32x4 (&out)[HT], int g, int i) {
#pragma unroll
    for (int m = 0; m < HT; ++m) out[m] = load4(bias + m * 16 + g * 4);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* ap = A + (((m * 4 + r) * 4 + g) * 16 + i) * HT;
            if (HT == 1) {
                out[0] = QB_MFMA16(ap[0], in[m][r], out[0]);
            } else {
                const float2 a = *reinterpret_cast<const float2*>(ap);
                out[0] = QB_MFMA16(a.x, in[m][r], out[0]);
                out[HT - 1] = QB_MFMA16(a.y, in[m][r], out[HT - 1]);
            }
        }
    }
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
    const float inv_den = 1.0f / den;
#pragma unroll
    for (int t = 0; t < T; ++t) n[t] = QB_LN2 * log2f_(cl[t] * inv_den);  // model.py:108
}

// First layer: T -> 64 with relu.  The B operand of k-step s is n[4s + group].
template <int T>
__device__ __forceinline__ void dense_first(const float* __restrict__ A,
                                            const float* __restrict__ bias, const float (&n)[T],
                                            f32x4 (&out)[4], int g, int i) {
#pragma unroll
    for (int m = 0; m < 4; ++m) out[m] = load4(bias + m * 16 + g * 4);
    const float4* Ap = reinterpret_cast<const float4*>(A) + g * 16 + i;
    constexpr int KS = (T + 3) / 4;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float b = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * s + k < T) b = (g == k) ? n[(4 * s + k < T) ? 4 * s + k : 0] : b;
        const float4 a = Ap[s * 64];
        out[0] = QB_MFMA16(a.x, b, out[0]);
        out[1] = QB_MFMA16(a.y, b, out[1]);
        out[2] = QB_MFMA16(a.z, b, out[2]);
        out[3] = QB_MFMA16(a.w, b, out[3]);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) out[m] = relu4(out[m]);
}

// One create_block step of stream 2 (gated residual), in place -- model.py:147-172.
__device__ __forceinline__ void block_stream2(const float* __restrict__ W, f32x4 (&b)[4], int g,
                                              int i) {
    f32x4 skip[4], t[4], r[4];
    dense64(W + BLK_WC_A, W + BLK_WC_B, b, skip, g, i);  // shared 1x1x1 conv as skip, :148
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        skip[m] = relu4(skip[m]);
        b[m] = relu4(b[m]);  // Activation before the first 3x3x1 conv, :151
    }
    dense64(W + BLK_R1_A, W + BLK_R1_B, b, t, g, i);  // :152
#pragma unroll
    for (int m = 0; m < 4; ++m) t[m] = relu4(t[m]);   // :155
    dense64(W + BLK_R2_A, W + BLK_R2_B, t, r, g, i);  // :156
    dense64(W + BLK_G_A, W + BLK_G_B, r, t, g, i);    // gating logits (+ gate_offset in bias), :164
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gate = sigmoidf_(t[m][k]);                       // :169
            b[m][k] = skip[m][k] * (1.0f - gate) + r[m][k] * gate;       // :170
        }
    }
}

// One create_block step of stream 1 -- model.py:144-145.
__device__ __forceinline__ void block_stream1(const float* __restrict__ W, f32x4 (&a)[4], int g,
                                              int i) {
    f32x4 o[4];
    dense64(W + BLK_WC_A, W + BLK_WC_B, a, o, g, i);
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = relu4(o[m]);
}

// Give every lane the copies of an accumulator register held by all four lane groups of its
// voxel column (lane & 15): v[g] = value held by group g.  v_permlane16_swap / v_permlane32_swap
// with both operands equal broadcast the even/odd 16-lane row of each pair and the lower/upper
// half-wave respectively.
__device__ __forceinline__ void all_groups(float x, float (&v)[4]) {
    const unsigned u = __float_as_uint(x);
    const auto p = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // p[0]: rows 0,2; p[1]: rows 1,3
    const auto e = __builtin_amdgcn_permlane32_swap(p[0], p[0], false, false);  // row 0 | row 2
    const auto o = __builtin_amdgcn_permlane32_swap(p[1], p[1], false, false);  // row 1 | row 3
    v[0] = __uint_as_float(e[0]);
    v[1] = __uint_as_float(o[0]);
    v[2] = __uint_as_float(e[1]);
    v[3] = __uint_as_float(o[1]);
}

// Head outputs of this lane's voxel, gathered to every lane: o[k], k < NOUT (<= 16 HT), where head
// row k sits in register (k & 3) of group (k >> 2) & 3 of tile k >> 4.
template <int NOUT, int HT>
__device__ __forceinline__ void gather_head(const f32x4 (&acc)[HT], float (&o)[NOUT]) {
#pragma unroll
    for (int m = 0; m < HT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (16 * m + r < NOUT) {  // some group holds a live row in this register
                float v[4];
                all_groups(acc[m][r], v);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int k = 16 * m + 4 * g + r;
                    if (k < NOUT) o[k < NOUT ? k : 0] = v[g];
                }
            }
        }
    }
}

}  // namespace qb
