// encoder_kernels.hip -- weight packing and the stand-alone encoder forward kernel.
//
// qbold_encoder_fwd replaces the Keras encoder model built by EncoderTrainer.create_encoder
// (model.py:122-223) for voxel batches: x [N][T] -> stream-1 parameters out1 [N][5] (the
// pre-training / prior output, model.py:199), stream-2 parameters out2 [N][5] (model.py:208) and
// the heteroscedastic sigma [N][T] (model.py:211-220).
//
// Roofline: 2 * 30,420 MAC = 60.8 kFLOP per voxel (U=60, L=2, T=11, stream 2) against 44 B read
// + 64 B written -> ~560 flop/B: compute-bound (split-f16 MFMA, see encoder_core.h), not HBM.
#include "canon_layout.h"
#include "encoder_core.h"
#include "qbold_ctx.h"


namespace {

using qb::EncLayout;
using qb::f32x4;

// value of W[in][out] of a canonical [nin][nout] matrix, zero outside
__device__ __forceinline__ float wval(const float* W, int nin, int nout, int in, int out) {
    return (in < nin && out < nout) ? W[in * nout + out] : 0.0f;
}

// One element of a dense op's f16 image: fragment (s, m_out, part), lane, j.
// bf16 mode: the hi slot carries the bfloat16 bit pattern of w, the lo slot is unused.
__device__ __forceinline__ _Float16 split_part(float w, int part, bool bf, float* flag = nullptr) {
    // a weight beyond f16's 65504 (or non-finite) cannot be split: recorded in the image's flag slot, which
    // the kernels fold into their activation range guard (outputs become NaN, encoder_core.h)
    if (!bf && flag && part == 0 && !(fabsf(w) <= QB_SPLIT_MAX))
        atomicMax(reinterpret_cast<int*>(flag), __float_as_int(fminf(fabsf(w), 3.0e38f)));
    if (bf) {
        const __bf16 b = (__bf16)w;
        return part == 0 ? __builtin_bit_cast(_Float16, b) : (_Float16)0.0f;
    }
    const _Float16 hi = (_Float16)w;
    return part == 0 ? hi : (_Float16)((w - (float)hi) * QB_LO_SCALE);
}
__device__ __forceinline__ int frag_unit(int s, int g, int j) { return 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3); }

// packed image, addressed in HALVES for the A pieces (2 per float slot) and floats for biases
__global__ void pack_kernel(EncLayout e, qb::CanonLayout c, float gate_offset, bool bf,
                            const float* __restrict__ w, float* __restrict__ packed) {
    const int U = c.U, T = c.T, G = c.G;
    _Float16* ph = reinterpret_cast<_Float16*>(packed);
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < 2 * e.total; p += gridDim.x * blockDim.x) {
        const int pf = p >> 1;  // float slot this half lives in
        if (pf < e.first_b) {   // first-layer A: [m][part][lane][j], k-slot 8g + j = input 8g + j
            const int j = p & 7, lane = (p >> 3) & 63, part = (p >> 9) & 1, m = p >> 10;
            const int in = 8 * (lane >> 4) + j;
            ph[p] = split_part(wval(w + c.W0, T, U, in, 16 * m + (lane & 15)), part, bf, packed + e.flag);
        } else if (pf < e.blk0) {  // first-layer bias [m][g][r] (float: written by the even half)
            if (p & 1) continue;
            const int q = pf - e.first_b, r = q & 3, g = (q >> 2) & 3, m = q >> 4;
            const int u = qb::acc_unit(m, r, g);
            packed[pf] = u < U ? w[c.b0 + u] : 0.0f;
        } else if (pf < e.head_A) {
            const int q = pf - e.blk0, l = q / qb::BLK_FLOATS, o = q % qb::BLK_FLOATS;
            const float* wb = w + c.blk0 + l * c.blk_stride;
            const int piece = o / 4160, oo = o % 4160;  // 4 x (4096 A + 64 bias)
            // voxel batches see only the centre tap (index 4 of 3x3) of the residual convolutions
            const int ctr = c.taps == 9 ? 4 * U * U : 0;
            const int Aoff = piece == 0 ? c.Wc : piece == 1 ? c.Wr1 + ctr : piece == 2 ? c.Wr2 + ctr : c.Wg;
            const int boff = piece == 0 ? c.bc : piece == 1 ? c.br1 : piece == 2 ? c.br2 : c.bg;
            const int nout = piece == 3 ? G : U;
            if (oo < 4096) {  // A: [s][m_out][part][lane][j] in halves
                const int h = 2 * oo + (p & 1);
                const int j = h & 7, lane = (h >> 3) & 63, part = (h >> 9) & 1, m = (h >> 10) & 3, s = h >> 12;
                const int in = frag_unit(s, lane >> 4, j);
                int out = 16 * m + (lane & 15);
                if (piece == 3 && G == 1) out = out < U ? 0 : U;  // shared gate broadcast to all units
                ph[p] = split_part(wval(wb + Aoff, U, nout, in, out), part, bf, packed + e.flag);
            } else {
                if (p & 1) continue;
                const int qq = oo - 4096, r = qq & 3, g = (qq >> 2) & 3, m = qq >> 4;
                const int u = qb::acc_unit(m, r, g);
                float v = 0.0f;
                if (u < U) {
                    v = wb[boff + ((piece == 3 && G == 1) ? 0 : u)];
                    if (piece == 3) v += gate_offset;
                }
                packed[pf] = v;
            }
        } else if (pf < e.head_b) {  // head A: [s][mh][part][lane][j], rows: 0-4 = Wf, 5.. = Ws
            const int HT = e.head_tiles;
            const int h = p - 2 * e.head_A;
            const int j = h & 7, lane = (h >> 3) & 63, part = (h >> 9) & 1, rest = h >> 10;
            const int mh = rest % HT, s = rest / HT;
            const int in = frag_unit(s, lane >> 4, j);
            const int row = 16 * mh + (lane & 15);
            float v = 0.0f;
            if (row < 5) v = wval(w + c.Wf, U, 5, in, row);
            else if (row < 5 + T) v = wval(w + c.Ws, U, T, in, row - 5);
            ph[p] = split_part(v, part, bf, packed + e.flag);
        } else if (pf < e.head_b + 16 * e.head_tiles) {  // head bias [mh][g][r]
            if (p & 1) continue;
            const int q = pf - e.head_b, r = q & 3, g = (q >> 2) & 3, mh = q >> 4;
            const int row = qb::acc_unit(mh, r, g);
            float v = 0.0f;
            if (row < 5) v = w[c.bf + row];
            else if (row < 5 + T) v = w[c.bs + row - 5];
            packed[pf] = v;
        } else if (!(p & 1) && pf != e.flag) {   // (the flag slot is zeroed by the host before this launch)
            packed[pf] = 0.0f;
        }
    }
}

constexpr int kEncBlock = 1024;

template <int T, int NL, bool BF>
__global__ __launch_bounds__(kEncBlock) void encoder_fwd_kernel(
    QbDev c, const float* __restrict__ packed, const float* __restrict__ x,
    float* __restrict__ out1, float* __restrict__ out2, float* __restrict__ sigma, int64_t N) {
    constexpr EncLayout e = qb::make_enc_layout(T, 64, NL);
    extern __shared__ __align__(16) float lds_w[];
    qb::copy_to_lds<kEncBlock>(lds_w, packed, e.total / 4);
    __syncthreads();

    constexpr int HT = (5 + T + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    constexpr int NW = kEncBlock / 64;
    const int64_t ntile = (N + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntile; tile += (int64_t)gridDim.x * NW) {
        const int64_t v = tile * 16 + i;
        const int64_t vc = v < N ? v : N - 1;  // clamp: every lane takes part in the MFMAs
        float xv[T], nv[T];
#pragma unroll
        for (int t = 0; t < T; ++t) xv[t] = x[vc * T + t];
        qb::normalise<T>(c, xv, nv);
        f32x4 a[4];
        qb::dense_first<T, BF>(lds_w + e.first_A, lds_w + e.first_b, nv, a, lane);
        // operand range guard (encoder_core.h): the largest activation split by any of the voxel's four lanes;
        // beyond f16's 65504 the voxel's outputs are written as NaN, never a silently clamped number
        auto voxel_max = [](float m) {
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            return fmaxf(m, __shfl_xor(m, 32, 64));
        };
        if (out2 || sigma) {
            f32x4 b[4] = {a[0], a[1], a[2], a[3]};  // net2 = net1, model.py:185
            float amax = lds_w[e.flag];   // 0, or the largest unsplittable weight
#pragma unroll
            for (int l = 0; l < NL; ++l) qb::block_stream2<BF>(lds_w + e.blk0 + l * e.blk_stride, b, lane, &amax);
            f32x4 hd[HT];
            qb::dense_head<HT, BF>(lds_w + e.head_A, lds_w + e.head_b, b, hd, lane, &amax);
            float o[5 + T];
            qb::gather_head<5 + T, HT>(hd, o);
            if (!BF && qb::split_overflowed(voxel_max(amax))) {
#pragma unroll
                for (int k = 0; k < 5 + T; ++k) o[k] = __builtin_nanf("");
            }
            if (v < N) {
                if (out2 && g == 0) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) out2[v * 5 + k] = o[k];
                }
                if (sigma && g == 1) {
#pragma unroll
                    for (int t = 0; t < T; ++t) sigma[v * T + t] = __expf(o[5 + t]);  // model.py:214
                }
            }
        }
        if (out1) {
            float amax = lds_w[e.flag];
#pragma unroll
            for (int l = 0; l < NL; ++l) qb::block_stream1<BF>(lds_w + e.blk0 + l * e.blk_stride, a, lane, &amax);
            f32x4 hd[HT];
            qb::dense_head<HT, BF>(lds_w + e.head_A, lds_w + e.head_b, a, hd, lane, &amax);
            float o[5];
            qb::gather_head<5, 1>(reinterpret_cast<f32x4(&)[1]>(hd[0]), o);
            if (!BF && qb::split_overflowed(voxel_max(amax))) {
#pragma unroll
                for (int k = 0; k < 5; ++k) o[k] = __builtin_nanf("");
            }
            if (v < N && g == 2) {
#pragma unroll
                for (int k = 0; k < 5; ++k) out1[v * 5 + k] = o[k];
            }
        }
    }
}


// ---- training forward in one launch ------------------------------------------------------------------
// Stream 2 with every tensor the layer-wise backward (train_kernels.hip, qbold_encoder_train_bwd) reads saved on
// the way: the workspace slots of qbold_encoder_train_fwd, [N][64] float32 each -- 0: n (normalised signals,
// columns T .. round-up-to-4 zeroed), 1: h, then per block skip, t, r, gate logits (without gate_offset: the
// backward adds it), b_out.  Same arithmetic as encoder_fwd_kernel (activations in registers, split-f16 MFMA
// products); the accumulator layout hands each lane four consecutive units of its voxel per 16-row tile, one
// 16-byte store.  The layer-wise forward moves each tensor through HBM two to three times (1.70 ms per 1 M
// voxels); this kernel writes each once.
// (wave-uniform slot base + a 32-bit lane offset: the scalar-base store form, no 64-bit address per store; lanes
// beyond the batch repeat the last voxel's row -- same values to the same address)
__device__ __forceinline__ void save_rows(float* __restrict__ slot, uint32_t voff, const f32x4 (&a)[4]) {
    char* sb = reinterpret_cast<char*>(slot);
#pragma unroll
    for (int m = 0; m < 4; ++m)
        *reinterpret_cast<float4*>(sb + voff + 64 * m) = make_float4(a[m][0], a[m][1], a[m][2], a[m][3]);
}

// The same four store instructions into a buffer resource of zero records: the hardware's range check drops them,
// nothing reaches memory.  Why issue them at all: with the stores of t and r simply removed (SAVE = 0) the compiler
// merges the block's stages and the 128-VGPR kernel spills 200 bytes per lane -- 0.66 ms against 0.58 ms WITH the
// stores; scheduling and memory barriers in their place did not change that, stores nobody receives do.
typedef uint32_t u32x4e __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void drop_rows(__amdgpu_buffer_rsrc_t nowhere, const f32x4 (&a)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4e, a[m]), nowhere, 64u * m, 0, 0);
}

// SAVE: 2 = every tensor; 1 leaves out skip and the gate logits, which block_bwd_kernel (train_kernels.hip)
// recomputes: eight tensors instead of twelve; 0 leaves out t and r as well, which block_bwd_dw_kernel also
// recomputes (its weight gradients read them from registers): n, h and each block's output, four tensors.
template <int T, int NL, int SAVE>
__global__ __launch_bounds__(kEncBlock) void encoder_train_fwd_kernel(
    QbDev c, const float* __restrict__ packed, const float* __restrict__ x, float gate_offset,
    float* __restrict__ ws, float* __restrict__ out_q, float* __restrict__ out_ls, int64_t N) {
    constexpr EncLayout e = qb::make_enc_layout(T, 64, NL);
    extern __shared__ __align__(16) float lds_w[];
    qb::copy_to_lds<kEncBlock>(lds_w, packed, e.total / 4);
    __syncthreads();

    constexpr int HT = (5 + T + 15) / 16;
    constexpr int WC = (T + 3) & ~3;   // (whole 64-byte row heads instead of 48 bytes: measured, 2.27 against 2.25 ms per step)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    constexpr int NW = kEncBlock / 64;
    const int64_t ntile = (N + 15) / 16;
    const int64_t slot_floats = N * 64;
    const __amdgpu_buffer_rsrc_t nowhere = __builtin_amdgcn_make_buffer_rsrc(ws, 0, 0, 0x00020000);
#ifndef QB_TRAIN_FWD_PREFETCH
#define QB_TRAIN_FWD_PREFETCH 1
#endif
    // The next tile's signals are requested BEFORE this tile's stores: the memory counter retires in order, so a load
    // issued behind a tile's 0.9 KB of stores waits for their write acknowledgements -- every wave would idle through
    // the HBM write latency once per tile.
    constexpr bool PF = QB_TRAIN_FWD_PREFETCH && T <= 16;   // (24 more live registers spill)
    float xn[PF ? T : 1];
    if constexpr (PF) {
        const int64_t v0 = ((int64_t)blockIdx.x * NW + wave) * 16 + i;
        const int64_t vc0 = v0 < N ? v0 : N - 1;
#pragma unroll
        for (int t = 0; t < T; ++t) xn[t] = x[vc0 * T + t];
    }
    for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntile; tile += (int64_t)gridDim.x * NW) {
        const int64_t v = tile * 16 + i;
        const bool live = v < N;
        const int64_t vc = live ? v : N - 1;  // clamp: every lane takes part in the MFMAs
        float xv[T], nv[T];
        if constexpr (PF) {
#pragma unroll
            for (int t = 0; t < T; ++t) xv[t] = xn[t];
            const int64_t vn = (tile + (int64_t)gridDim.x * NW) * 16 + i;
            const int64_t vcn = vn < N ? vn : N - 1;
#pragma unroll
            for (int t = 0; t < T; ++t) xn[t] = x[vcn * T + t];
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) xv[t] = x[vc * T + t];
        }
        qb::normalise<T>(c, xv, nv);
        const uint32_t voff = (uint32_t)vc * 256u + 16u * (uint32_t)g;  // N < 2^23 voxels (checked by the host)
        if (g == 0) {
#pragma unroll
            for (int t = 0; t < WC; ++t) ws[vc * 64 + t] = t < T ? nv[t < T ? t : 0] : 0.0f;
        }
        f32x4 b[4];
        qb::dense_first<T, false>(lds_w + e.first_A, lds_w + e.first_b, nv, b, lane);
        save_rows(ws + slot_floats, voff, b);
        float amax = lds_w[e.flag];
#pragma unroll
        for (int l = 0; l < NL; ++l) {   // block_stream2 (encoder_core.h) with its tensors saved
            const float* W = lds_w + e.blk0 + l * e.blk_stride;
            float* base = ws + (int64_t)(2 + 5 * l) * slot_floats;
            f32x4 skip[4], t[4], r[4];
            qb::dense64<false>(W + qb::BLK_WC_A, W + qb::BLK_WC_B, b, skip, lane, &amax);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                skip[m] = qb::relu4(skip[m]);
                b[m] = qb::relu4(b[m]);
            }
            if constexpr (SAVE >= 2) save_rows(base, voff, skip);
            qb::dense64<false>(W + qb::BLK_R1_A, W + qb::BLK_R1_B, b, t, lane);
#pragma unroll
            for (int m = 0; m < 4; ++m) t[m] = qb::relu4(t[m]);
            if constexpr (SAVE >= 1) save_rows(base + slot_floats, voff, t);
            else drop_rows(nowhere, t);   // (see drop_rows)
            qb::dense64<false>(W + qb::BLK_R2_A, W + qb::BLK_R2_B, t, r, lane, &amax);
            if constexpr (SAVE >= 1) save_rows(base + 2 * slot_floats, voff, r);
            else drop_rows(nowhere, r);
            qb::dense64<false>(W + qb::BLK_G_A, W + qb::BLK_G_B, r, t, lane, &amax);  // logits + gate_offset
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float gate = qb::sigmoidf_(t[m][k]);
                    b[m][k] = skip[m][k] * (1.0f - gate) + r[m][k] * gate;
                    if constexpr (SAVE >= 2) t[m][k] -= gate_offset;
                }
            }
            if constexpr (SAVE >= 2) save_rows(base + 3 * slot_floats, voff, t);
            save_rows(base + 4 * slot_floats, voff, b);
        }
        f32x4 hd[HT];
        qb::dense_head<HT, false>(lds_w + e.head_A, lds_w + e.head_b, b, hd, lane, &amax);
        float o[5 + T];
        qb::gather_head<5 + T, HT>(hd, o);
        amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        if (qb::split_overflowed(amax)) {   // operand range guard: NaN heads, never a clamped number
#pragma unroll
            for (int k = 0; k < 5 + T; ++k) o[k] = __builtin_nanf("");
        }
        if (live && g == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) out_q[v * 5 + k] = o[k];
        }
        if (live && g == 1 && out_ls) {
#pragma unroll
            for (int t = 0; t < T; ++t) out_ls[v * T + t] = o[5 + t];
        }
    }
}

}  // namespace

namespace qb {
int check_encoder_shape(const qbold_ctx* ctx, const qbold_encoder_shape* s) {
    if (!s) { set_error("encoder shape is null"); return QBOLD_ERR_INVALID; }
    if (s->T != ctx->dev.T) { set_error("encoder shape T differs from the context's tau grid"); return QBOLD_ERR_INVALID; }
    if (s->precision != QBOLD_ENC_F32 && s->precision != QBOLD_ENC_BF16) {
        set_error("encoder shape: precision must be QBOLD_ENC_F32 or QBOLD_ENC_BF16");
        return QBOLD_ERR_INVALID;
    }
    if (s->U < 1 || s->U > 64 || s->L < 1 || s->L > 2 || s->T > 27) {
        set_error("encoder kernels are built for U <= 64, L <= 2, T <= 27 (LDS-resident weights)");
        return QBOLD_ERR_UNSUPPORTED;
    }
    return QBOLD_OK;
}
}  // namespace qb

extern "C" int64_t qbold_encoder_num_params(const qbold_encoder_shape* s) {
    if (!s) return QBOLD_ERR_INVALID;
    return qb::make_canon(s->T, s->U, s->L, s->channelwise_gating, s->spatial_taps, s->layer_norm).total;
}

extern "C" int64_t qbold_encoder_packed_floats(const qbold_encoder_shape* s) {
    if (!s) return QBOLD_ERR_INVALID;
    return qb::make_enc_layout(s->T, s->U, s->L).total;
}

extern "C" int qbold_encoder_pack(const qbold_ctx* ctx, const qbold_encoder_shape* shape,
                                  const float* weights, float* packed, void* stream) {
    QB_NEED_DEVICE(ctx);
    int rc = qb::check_encoder_shape(ctx, shape);
    if (rc) return rc;
    QB_REQUIRE(weights && packed, "qbold_encoder_pack: null buffer");
    const EncLayout e = qb::make_enc_layout(shape->T, shape->U, shape->L);
    const qb::CanonLayout c = qb::make_canon(shape->T, shape->U, shape->L, shape->channelwise_gating,
                                             shape->spatial_taps);
    QB_HIP(hipMemsetAsync(packed + e.flag, 0, sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(pack_kernel, dim3((2 * e.total + 255) / 256), dim3(256), 0, (hipStream_t)stream, e,
                       c, shape->gate_offset, shape->precision == QBOLD_ENC_BF16, weights, packed);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_encoder_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape,
                                 const float* packed, const float* x, float* out1, float* out2,
                                 float* sigma, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_RELU_ONLY(shape, "qbold_encoder_fwd");
    int rc = qb::check_encoder_shape(ctx, shape);
    if (rc) return rc;
    QB_REQUIRE(N >= 0, "qbold_encoder_fwd: negative N");
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(packed && x, "qbold_encoder_fwd: null buffer");
    const EncLayout e = qb::make_enc_layout(shape->T, shape->U, shape->L);
    const size_t smem = sizeof(float) * e.total;
    const int64_t ntile = (N + 15) / 16;
    const int64_t nblk = (ntile + kEncBlock / 64 - 1) / (kEncBlock / 64);
    const int grid = (int)(nblk < ctx->num_cus ? nblk : ctx->num_cus);
#define QB_LAUNCH_ENC(TT, NL)                                                                      \
    do {                                                                                           \
        auto k = shape->precision == QBOLD_ENC_BF16 ? encoder_fwd_kernel<TT, NL, true>             \
                                                    : encoder_fwd_kernel<TT, NL, false>;           \
        QB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k),                               \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));        \
        hipLaunchKernelGGL(k, dim3(grid), dim3(kEncBlock), smem, (hipStream_t)stream, ctx->dev,    \
                           packed, x, out1, out2, sigma, N);                                       \
    } while (0)
    if (shape->T == 11 && shape->L == 1) QB_LAUNCH_ENC(11, 1);
    else if (shape->T == 11 && shape->L == 2) QB_LAUNCH_ENC(11, 2);
    else if (shape->T == 24 && shape->L == 1) QB_LAUNCH_ENC(24, 1);
    else if (shape->T == 24 && shape->L == 2) QB_LAUNCH_ENC(24, 2);
    else {
        qb::set_error("qbold_encoder_fwd: kernels are built for T = 11 or 24 taus, L = 1 or 2");
        return QBOLD_ERR_UNSUPPORTED;
    }
#undef QB_LAUNCH_ENC
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_encoder_train_fwd_fused(const qbold_ctx* ctx, const qbold_encoder_shape* shape,
                                             const float* packed, const float* x, int save_all, float* ws,
                                             float* out_q, float* out_log_sigma, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_RELU_ONLY(shape, "qbold_encoder_train_fwd_fused");
    int rc = qb::check_encoder_shape(ctx, shape);
    if (rc) return rc;
    if (!shape->channelwise_gating || shape->precision != QBOLD_ENC_F32) {
        qb::set_error("qbold_encoder_train_fwd_fused: built for channel-wise gating, QBOLD_ENC_F32");
        return QBOLD_ERR_UNSUPPORTED;
    }
    QB_REQUIRE(N > 0 && N < ((int64_t)1 << 23) && packed && x && ws && out_q,
               "qbold_encoder_train_fwd_fused: bad argument (0 < N < 2^23 voxels per call)");
    QB_REQUIRE(reinterpret_cast<uintptr_t>(ws) % 16 == 0 && reinterpret_cast<uintptr_t>(packed) % 16 == 0,
               "qbold_encoder_train_fwd_fused: packed image and workspace must be 16-byte aligned");
    const EncLayout e = qb::make_enc_layout(shape->T, shape->U, shape->L);
    const size_t smem = sizeof(float) * e.total;
    const int64_t ntile = (N + 15) / 16;
    const int64_t nblk = (ntile + kEncBlock / 64 - 1) / (kEncBlock / 64);
    const int grid = (int)(nblk < ctx->num_cus ? nblk : ctx->num_cus);
#define QB_LAUNCH_TRAIN_FWD(TT, NL)                                                                  \
    do {                                                                                             \
        auto k = save_all >= 2 ? encoder_train_fwd_kernel<TT, NL, 2>                                         \
                 : save_all == 1 ? encoder_train_fwd_kernel<TT, NL, 1> : encoder_train_fwd_kernel<TT, NL, 0>; \
        QB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                 \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));          \
        hipLaunchKernelGGL(k, dim3(grid), dim3(kEncBlock), smem, (hipStream_t)stream, ctx->dev,      \
                           packed, x, shape->gate_offset, ws, out_q, out_log_sigma, N);              \
    } while (0)
    if (shape->T == 11 && shape->L == 1) QB_LAUNCH_TRAIN_FWD(11, 1);
    else if (shape->T == 11 && shape->L == 2) QB_LAUNCH_TRAIN_FWD(11, 2);
    else if (shape->T == 24 && shape->L == 1) QB_LAUNCH_TRAIN_FWD(24, 1);
    else if (shape->T == 24 && shape->L == 2) QB_LAUNCH_TRAIN_FWD(24, 2);
    else {
        qb::set_error("qbold_encoder_train_fwd_fused: kernels are built for T = 11 or 24 taus, L = 1 or 2");
        return QBOLD_ERR_UNSUPPORTED;
    }
#undef QB_LAUNCH_TRAIN_FWD
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}
