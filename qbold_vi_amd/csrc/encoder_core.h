// encoder_core.h -- the voxel-wise encoder MLP on the CDNA4 matrix cores.
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
// Y^T[unit][voxel] = W^T[unit][k] X^T[k][voxel]: the weights are the MFMA A operand, the
// activations the B operand.  The 16x16 accumulator puts the voxel on the lane (col = lane & 15)
// and four units in the registers of each 16-lane group (row = 4 (lane >> 4) + reg), which is what
// the next layer's B operand wants if the weight image is stored in that k order -- so activations
// never leave the register file and never touch LDS.  A 64-unit activation tensor costs 16 VGPRs
// per lane, which keeps the whole fused kernel within 128 VGPRs (four waves per SIMD).  U <= 64 is
// padded to four 16-unit tiles.
//
// Arithmetic: float32 in, float32 accumulate, through v_mfma_f32_16x16x32_f16 with every operand
// split in two halves, x = hi + 2^-11 lo (hi = f16(x), lo = f16((x - hi) * 2^11); the scaling keeps
// lo out of the f16 subnormals).  Three MFMAs per tile -- hi.hi into one accumulator, hi.lo and
// lo.hi into a second that is folded in with 2^-11 -- drop only the lo.lo term: per-product
// relative error <= ~3 * 2^-22 (7e-7), i.e. float32-grade, at 3/16 of the cycles of the exact
// v_mfma_f32_16x16x4_f32 path.  The motive is not only the cycles: on gfx950 the f32-input MFMA runs
// at the vector rate and, measured here, does NOT overlap with f32 VALU work from other waves
// (profiles/, MEASUREMENTS.md 4.4), while the f16 matrix pipe does.
//
// LDS weight image (built by pack_kernel in encoder_kernels.hip), per dense op:
//   A[kstep s][m_out][part = hi, lo][lane 0..63][j = 0..7] f16  (one ds_read_b128 per fragment,
//   a wave reads one contiguous KiB: conflict-free), element = W[in = unit(s, lane>>4, j)]
//   [out = 16 m_out + (lane & 15)], unit(s, g, j) = 16 (2s + (j >> 2)) + 4g + (j & 3);
//   bias[m_out][group][reg = 0..3] f32.
#pragma once

#include "qbold_dev.h"

namespace qb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Offsets (in floats) of every piece of the packed image; wave-uniform kernel argument.
struct EncLayout {
    int T, U, L;
    int ksteps_first;   // 1 (K = 32 >= T)
    int n_head;         // 5 + T outputs of the merged head
    int head_tiles;     // ceil(n_head / 16): 1 or 2
    int first_A, first_b;
    int blk0;           // offset of block 0
    int blk_stride;     // floats per block
    int head_A, head_b;
    int flag;           // one float: max |w| over the weights beyond the f16 range (0: none), see split_part
    int total;
};
// inside one block: four dense ops, each 4096 (A) + 64 (bias) floats
enum { BLK_WC_A = 0, BLK_WC_B = 4096, BLK_R1_A = 4160, BLK_R1_B = 8256, BLK_R2_A = 8320,
       BLK_R2_B = 12416, BLK_G_A = 12480, BLK_G_B = 16576, BLK_FLOATS = 16640 };

__host__ __device__ constexpr inline EncLayout make_enc_layout(int T, int U, int L) {
    EncLayout e{};
    e.T = T; e.U = U; e.L = L;
    e.ksteps_first = 1;
    e.n_head = 5 + T;
    e.head_tiles = (e.n_head + 15) / 16;
    e.first_A = 0;
    e.first_b = 2048;  // [m_out 4][part 2][lane 64][8 halves] = 4096 halves
    e.blk0 = e.first_b + 64;
    e.blk_stride = BLK_FLOATS;
    e.head_A = e.blk0 + L * BLK_FLOATS;
    e.head_b = e.head_A + 1024 * e.head_tiles;
    e.flag = e.head_b + 16 * e.head_tiles;
    e.total = (e.flag + 1 + 3) & ~3;
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

// global -> LDS copy of n4 float4 with a thread's loads in flight eight at a time.  Written as `dst[p] = src[p]` per
// trip the compiler issues one load, waits for it and stores it: a kernel's weight image took ten dependent L2 round
// trips per thread (more where the block is smaller) before its first tile.
template <int THREADS>
__device__ __forceinline__ void copy_to_lds(float* __restrict__ dst, const float* __restrict__ src, int n4) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int base = 0; base < n4; base += 8 * THREADS) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = base + u * THREADS + (int)threadIdx.x;
            v[u] = s4[p < n4 ? p : n4 - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = base + u * THREADS + (int)threadIdx.x;
            if (p < n4) d4[p] = v[u];
        }
    }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define QB_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define QB_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#ifndef QB_ENC_BASE_PRIO
#define QB_ENC_BASE_PRIO 0
#endif
#define QB_LO_SCALE 2048.0f
#ifndef QB_SPLIT_MIX
#define QB_SPLIT_MIX 1
#endif
#define QB_LO_UNSCALE (1.0f / 2048.0f)

// Reduced-precision mode (qbold_encoder_shape.precision = QBOLD_ENC_BF16; BASELINE config 5, "bf16
// forward / fp32 ELBO accum"): BF = true everywhere below means ONE v_mfma_f32_16x16x32_bf16 per
// tile on bfloat16-rounded operands (round-to-nearest-even) with float32 accumulation, instead of
// the three split-f16 MFMAs.  The `hi` slots of the weight image then hold bf16 bit patterns and the
// `lo` slots are unused; fragments travel as f16x8 (16 raw bytes) and are reinterpreted at the MFMA.
__device__ __forceinline__ bf16x8 as_bf16(const f16x8& v) { return __builtin_bit_cast(bf16x8, v); }

// hi / lo halves of eight float32 values as the B fragment of one K = 32 step.
template <bool BF = false>
__device__ __forceinline__ void split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
    if constexpr (BF) {
        bf16x8 b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
        hi = __builtin_bit_cast(f16x8, b);
        lo = hi;  // unused
    } else {
#if QB_SPLIT_MIX
        // lo = f16((x - hi) 2^11) = f16(fma(hi, -2^11, 2^11 x)): one mixed-precision FMA per value reads the f16 hi half
        // in place and writes the f16 lo half in place (the FMA's float32 result is exact -- hi is x rounded -- so its
        // one rounding is the cast's; bit-identical to the three-instruction form below).  Two instructions per
        // value (packed hi conversion, packed 2^11 x, the FMA) instead of three (conversion back, subtraction,
        // scale, conversion) on kernels bound by vector-pipe issue.
        typedef uint32_t u32x4s8 __attribute__((ext_vector_type(4)));
        u32x4s8 hp, lp;
        const float nscale = -QB_LO_SCALE;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float a = v[2 * p], b = v[2 * p + 1];
            uint32_t h = __builtin_bit_cast(uint32_t, f16x2{(_Float16)a, (_Float16)b}), l;
            typedef float f32x2s8 __attribute__((ext_vector_type(2)));
            const f32x2s8 sc = f32x2s8{a, b} * QB_LO_SCALE;   // one v_pk_mul_f32
            asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "s"(nscale), "v"(sc[0]));
            asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "s"(nscale), "v"(sc[1]));
            hp[p] = h;
            lp[p] = l;
        }
        hi = __builtin_bit_cast(f16x8, hp);
        lo = __builtin_bit_cast(f16x8, lp);
#else
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const _Float16 h = (_Float16)v[j];
            hi[j] = h;
            lo[j] = (_Float16)((v[j] - (float)h) * QB_LO_SCALE);
        }
#endif
    }
}

// B fragments (k-steps 0, 1) of a 64-unit activation tensor held as four accumulator tiles.
struct ActFrag {
    f16x8 hi[2], lo[2];
};
// The largest magnitude an f16 hi half can hold; an activation beyond it overflows the operand split to inf,
// and what follows is NOT reliably non-finite (inf - inf = NaN in the accumulators, and relu's v_max_f32 turns a
// NaN into 0).  The kernels therefore track max |activation| over everything they split (amax, one v_max3_f32
// per pair of values) and poison the voxel's outputs with NaN when it passes the limit: non-finite outputs /
// sums are the status channel of include/qbold_hip.h.
#ifndef QB_AMAX_ASM
#define QB_AMAX_ASM 1
#endif
#define QB_SPLIT_MAX 65504.0f
__device__ __forceinline__ bool split_overflowed(float amax) { return !(amax <= QB_SPLIT_MAX); }

template <bool BF = false>
__device__ __forceinline__ ActFrag split_act(const f32x4 (&in)[4], float* amax = nullptr) {
    ActFrag f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float v[8] = {in[2 * s][0], in[2 * s][1], in[2 * s][2], in[2 * s][3],
                            in[2 * s + 1][0], in[2 * s + 1][1], in[2 * s + 1][2], in[2 * s + 1][3]};
        if (!BF && amax) {
            // one v_max3_f32 with |.| source modifiers per pair.  Written as fmaxf(amax, fmaxf(|a|, |b|)) the compiler
            // first canonicalises each MFMA result (v_max_f32 x, |a|, |a|): four instructions per pair instead of one
            // (QB_AMAX_ASM=0 restores that form)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
#if QB_AMAX_ASM
                asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(*amax) : "v"(v[j]), "v"(v[j + 1]));
#else
                *amax = fmaxf(*amax, fmaxf(fabsf(v[j]), fabsf(v[j + 1])));
#endif
            }
        }
        split8<BF>(v, f.hi[s], f.lo[s]);
    }
    return f;
}

__device__ __forceinline__ f16x8 lds_frag(const float* __restrict__ A, int idx, int lane) {
    // fragment idx of a dense op: 64 lanes x 16 bytes each
    return *reinterpret_cast<const f16x8*>(reinterpret_cast<const unsigned char*>(A) + idx * 1024 +
                                           lane * 16);
}

// out[0..MT-1] = W in + bias over KS k-steps of 32.  A: LDS image [s][m][part][lane][8], bias: f32.
template <int MT, int KS, bool BF = false>
__device__ __forceinline__ void dense_f16x3(const float* __restrict__ A,
                                            const float* __restrict__ bias, const f16x8 (&bhi)[KS],
                                            const f16x8 (&blo)[KS], f32x4 (&out)[MT], int lane) {
    const int g = lane >> 4;
    f32x4 cross[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        out[m] = load4(bias + m * 16 + g * 4);
        cross[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    // Fragments are fetched one (k-step, tile) ahead of the MFMAs that consume them; the
    // sched_barrier keeps the scheduler from hoisting all 2*KS*MT reads to the top (64 VGPRs).
    if constexpr (BF) {
        f16x8 whi = lds_frag(A, 0, lane);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int nxt = s * MT + m + 1;
                f16x8 nhi = whi;
                if (nxt < KS * MT) nhi = lds_frag(A, nxt * 2, lane);
                out[m] = QB_MFMA_BF16(as_bf16(whi), as_bf16(bhi[s]), out[m]);
                whi = nhi;
            }
        }
        return;
    }
    f16x8 whi = lds_frag(A, 0, lane), wlo = lds_frag(A, 1, lane);
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int nxt = s * MT + m + 1;
            f16x8 nhi = whi, nlo = wlo;
            if (nxt < KS * MT) {
                nhi = lds_frag(A, nxt * 2 + 0, lane);
                nlo = lds_frag(A, nxt * 2 + 1, lane);
            }
            out[m] = QB_MFMA_F16(whi, bhi[s], out[m]);
            cross[m] = QB_MFMA_F16(whi, blo[s], cross[m]);
            cross[m] = QB_MFMA_F16(wlo, bhi[s], cross[m]);
            whi = nhi;
            wlo = nlo;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(QB_ENC_BASE_PRIO);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[m][r] = fmaf(cross[m][r], QB_LO_UNSCALE, out[m][r]);
    }
}

// 64 -> 64 layer on an activation tensor
template <bool BF = false>
__device__ __forceinline__ void dense64(const float* __restrict__ A, const float* __restrict__ bias,
                                        const f32x4 (&in)[4], f32x4 (&out)[4], int lane, float* amax = nullptr) {
    const ActFrag f = split_act<BF>(in, amax);
    dense_f16x3<4, 2, BF>(A, bias, f.hi, f.lo, out, lane);
}

// heads: HT (1 or 2) 16-row output tiles from a 64-unit input.
template <int HT, bool BF = false>
__device__ __forceinline__ void dense_head(const float* __restrict__ A,
                                           const float* __restrict__ bias, const f32x4 (&in)[4],
                                           f32x4 (&out)[HT], int lane, float* amax = nullptr) {
    const ActFrag f = split_act<BF>(in, amax);
    dense_f16x3<HT, 2, BF>(A, bias, f.hi, f.lo, out, lane);
}

// normalise_data -- model.py:97-113; n[t] for this lane's voxel.  SE >= 0: the caller dispatched on
// se_idx == SE && !multi_norm, and the spin-echo image is picked at compile time (same arithmetic; the
// run-time form costs 3 T selects whose lane-uniform masks the register allocator spills).
template <int T, int SE = -1>
__device__ __forceinline__ void normalise(const QbDev& c, const float (&x)[T], float (&n)[T]) {
    float cl[T];
#pragma unroll
    for (int t = 0; t < T; ++t) cl[t] = clampf_(x[t], 1e-2f, 1e8f);  // model.py:101
    float den;
    if (SE >= 0) {
        den = cl[SE >= 0 ? SE : 0];  // model.py:106
    } else {
        const int se = c.se_idx;
        float a = 0.0f, b = 0.0f, d = 0.0f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            a = (t == se - 1) ? cl[t] : a;
            b = (t == se) ? cl[t] : b;
            d = (t == se + 1) ? cl[t] : d;
        }
        den = c.multi_norm ? (a + b + d) / 3.0f : b;  // model.py:104 / :106
    }
    const float inv_den = 1.0f / den;
#pragma unroll
    for (int t = 0; t < T; ++t) n[t] = QB_LN2 * log2f_(cl[t] * inv_den);  // model.py:108
}

// First layer: T -> 64 with relu.  One K = 32 step; k-slot 8 group + j carries n[8 group + j].
template <int T, bool BF = false>
__device__ __forceinline__ void dense_first(const float* __restrict__ A,
                                            const float* __restrict__ bias, const float (&n)[T],
                                            f32x4 (&out)[4], int lane) {
    static_assert(T <= 32, "first layer is a single K = 32 step");
    const int g = lane >> 4;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = 0.0f;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
            if (8 * gg + j < T) x = (g == gg) ? n[(8 * gg + j < T) ? 8 * gg + j : 0] : x;
        v[j] = x;
    }
    f16x8 hi[1], lo[1];
    split8<BF>(v, hi[0], lo[0]);
    dense_f16x3<4, 1, BF>(A, bias, hi, lo, out, lane);
#pragma unroll
    for (int m = 0; m < 4; ++m) out[m] = relu4(out[m]);
}

// One create_block step of stream 2 (gated residual), in place -- model.py:147-172.
template <bool BF = false>
__device__ __forceinline__ void block_stream2(const float* __restrict__ W, f32x4 (&b)[4],
                                              int lane, float* amax = nullptr) {
    f32x4 skip[4], t[4], r[4];
    dense64<BF>(W + BLK_WC_A, W + BLK_WC_B, b, skip, lane, amax);  // shared 1x1x1 conv as skip, :148
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        skip[m] = relu4(skip[m]);
        b[m] = relu4(b[m]);  // Activation before the first 3x3x1 conv, :151
    }
    dense64<BF>(W + BLK_R1_A, W + BLK_R1_B, b, t, lane);  // :152  (|relu b| <= |b|: already tracked)
#pragma unroll
    for (int m = 0; m < 4; ++m) t[m] = relu4(t[m]);   // :155
    dense64<BF>(W + BLK_R2_A, W + BLK_R2_B, t, r, lane, amax);  // :156
    dense64<BF>(W + BLK_G_A, W + BLK_G_B, r, t, lane, amax);    // gating logits (+ gate_offset in bias), :164
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
template <bool BF = false>
__device__ __forceinline__ void block_stream1(const float* __restrict__ W, f32x4 (&a)[4],
                                              int lane, float* amax = nullptr) {
    f32x4 o[4];
    dense64<BF>(W + BLK_WC_A, W + BLK_WC_B, a, o, lane, amax);
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
