// wide_kernels.hip -- the voxel-wise encoder for widths that do not fit the LDS-resident kernel
// (BASELINE config 3: 64 taus, no_units = 256, "MFMA path stressed").
//
// Reference: create_encoder (model.py:122-223) on (N,1,1,1,T) voxel batches, as encoder_core.h.
// With U = 256 one dense layer is 256 KiB of split-f16 weights: no longer LDS-resident, and a
// 256-wide activation row no longer fits next to its accumulators.  So the layers run one launch
// each as Y = act(X W + b) over activations [N][U] float32 in HBM, and each launch is a
// weight-streaming GEMM on the f16 matrix cores with the same float32-grade operand split as the
// fused kernel (x = hi + 2^-11 lo, three v_mfma_f32_16x16x32_f16 per tile: hi.hi, hi.lo, lo.hi):
//
//   * a wave owns 32 voxels (two 16-voxel MFMA column tiles) x 128 outputs (eight 16-row tiles):
//     2 x 8 x (out, cross) accumulators = 128 VGPRs.  Every weight fragment read from LDS feeds
//     two voxel tiles, which keeps the LDS read rate (2 KiB per six MFMAs) under the 128 B/clk port;
//   * a 512-thread block = 4 voxel groups x 2 output halves = 128 voxels x 256 outputs per pass;
//   * K runs in steps of 32.  Everything reaches LDS by LDS-direct loads (global_load_lds_dwordx4,
//     1 KiB per wave-instruction, no staging registers): the step's weight image (32 KiB:
//     [out tile][hi, lo][lane][8 halves], one conflict-free ds_read_b128 per fragment, L2-resident)
//     double-buffered one step ahead; the step's activations (16 KiB: 128 voxels x 32 k float32, full
//     128-byte lines, shared by the two output-half waves) in a ring of five slots, four steps
//     ahead -- HBM latency under load is several thousand cycles, one k-step of MFMAs is ~1500;
//   * the two streams are issued by different waves so that each wave's in-order vmcnt counter follows
//     one prefetch distance; steps are separated by a raw s_barrier after a counted s_waitcnt, so the
//     activation stream is never drained (a __syncthreads would wait vmcnt(0));
//   * bias, relu / sigmoid-gate / head scatter are fused into the epilogue, relu on the input side
//     into the operand load -- no separate element-wise passes over [N][U].
//
// Roofline per layer at N = 1 M, U = 256: 2 * 65,536 MAC * 3 passes = 0.41 TFLOP of f16 MFMA
// (0.17 ms at the 2.5 PFLOP/s dense peak) against 1 GiB read + 1 GiB written (0.27 ms at 8 TB/s):
// HBM-bound, by the float32 activations between layers.
#include "canon_layout.h"
#include "encoder_core.h"
#include "qbold_ctx.h"
#include "wide_common.h"

namespace {

using qb::f16x8;
using qb::f32x4;
using namespace qbw;

constexpr int kWB = 512;  // threads per block: 2 waves per SIMD at <= 256 registers
enum { EPI_LINEAR = 0, EPI_RELU = 1, EPI_GATE = 2, EPI_HEAD = 3 };

struct WideArgs {
    const float* X;   // [N][ldx] activations (K = 32 KS leading columns used)
    int ldx;
    const uint4* W;   // weight image of this dense op: [KS][MT][hi, lo][64 lanes] x 16 bytes
    const float* bias;  // [16 MT]
    float* Y;         // [N][ldy]
    int ldy;
    const float* skip;  // EPI_GATE: y = skip (1 - g) + r g, g = sigmoid(acc)   (model.py:167-170)
    const float* r;
    float* q;         // EPI_HEAD: rows 0-4 -> q [N][5], rows 5..5+T-1 -> log sigma [N][T]
    float* ls;
    int T;
    int dbg;          // QBOLD_DEBUG_SKIP ablation bits (timing experiments only): 16 no MFMA, 32 no activation
                      // loads, 64 no stores, 128 no weight loads
    int64_t N;
};

// Image element order inside a fragment: lane (g = lane >> 4, i = lane & 15), j = 0..7 holds
// W[in = 32 s + 8 g + j][out = 16 m + i] -- the A operand of v_mfma_f32_16x16x32_f16 for output
// rows 16 m .. 16 m + 15; the B operand of lane (g, i) is X[voxel i][32 s + 8 g + j], eight
// contiguous floats of the activation row.
struct OpImage {
    int KS, MT;      // k-chunks of 32, output tiles of 16
    int64_t A;       // offset (floats) of the image
    int64_t b;       // offset (floats) of the bias [16 MT]
};
struct WideLayout {
    int T, U, L, KS1, KSU, MTU, HT;
    OpImage first, blk[8][4], head;  // blk[l][0..3] = Wc, Wr1, Wr2, Wg
    int64_t total;
};
__host__ __device__ inline int64_t image_floats(int KS, int MT) { return (int64_t)KS * MT * 2 * 64 * 4; }

inline bool wide_supported(const qbold_encoder_shape* s) {
    return s && (s->U == 128 || s->U == 256) && s->L >= 1 && s->L <= 8 && s->T >= 1 && s->T <= 64 &&
           s->channelwise_gating && s->precision == QBOLD_ENC_F32;
}

inline WideLayout make_wide_layout(int T, int U, int L) {
    WideLayout w{};
    w.T = T; w.U = U; w.L = L;
    w.KS1 = (T + 31) / 32;
    w.KSU = U / 32;
    w.MTU = U / 16;
    w.HT = (5 + T + 15) / 16;
    int64_t off = 0;
    auto op = [&](int KS, int MT) {
        OpImage o{KS, MT, off, 0};
        off += image_floats(KS, MT);
        o.b = off;
        off += 16 * MT;
        return o;
    };
    w.first = op(w.KS1, w.MTU);
    for (int l = 0; l < L; ++l)
        for (int p = 0; p < 4; ++p) w.blk[l][p] = op(w.KSU, w.MTU);
    w.head = op(w.KSU, w.HT);
    w.total = (off + 3) & ~(int64_t)3;
    return w;
}

// One dense op's image + bias from the canonical blob.  W: [nin][nout] row-major (Keras), rows /
// columns beyond (nin, nout) are zero.  Head: output row k < 5 -> Wf[:, k], 5 <= k < 5 + T -> Ws.
__global__ void wide_pack_kernel(OpImage o, const float* __restrict__ W, const float* __restrict__ b,
                                 int nin, int nout, const float* __restrict__ W2,
                                 const float* __restrict__ b2, int nout2, float bias_add,
                                 float* __restrict__ packed) {
    _Float16* ph = reinterpret_cast<_Float16*>(packed + o.A);
    const int64_t halves = image_floats(o.KS, o.MT) * 2;
    for (int64_t h = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; h < halves + 16 * o.MT;
         h += (int64_t)gridDim.x * blockDim.x) {
        if (h >= halves) {  // bias
            const int row = (int)(h - halves);
            float v = 0.0f;
            if (row < nout) v = b[row] + bias_add;
            else if (row < nout + nout2) v = b2[row - nout];
            packed[o.b + row] = v;
            continue;
        }
        const int j = (int)(h & 7), lane = (int)((h >> 3) & 63), part = (int)((h >> 9) & 1);
        const int64_t frag = h >> 10;  // s * MT + m
        const int m = (int)(frag % o.MT), s = (int)(frag / o.MT);
        const int in = 32 * s + 8 * (lane >> 4) + j, out = 16 * m + (lane & 15);
        float w = 0.0f;
        if (in < nin) {
            if (out < nout) w = W[(int64_t)in * nout + out];
            else if (out < nout + nout2) w = W2[(int64_t)in * nout2 + (out - nout)];
        }
        const _Float16 hi = (_Float16)w;
        ph[h] = part == 0 ? hi : (_Float16)((w - (float)hi) * QB_LO_SCALE);
    }
}

// s_waitcnt vmcnt(4 n): at most n steps' worth of this wave's four-instruction groups still in flight
template <int PER>
__device__ __forceinline__ void wait_groups(int n) {
    static_assert(PER * 4 <= 63, "vmcnt is a 6-bit counter");
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PER) : "memory"); break;
    }
}

// Output tiles M .. MTW-1 of one k-step: the fragments of tile M+1 are requested before tile M's six
// MFMAs and waited for after them.  (Recursive template: the LDS offsets are instruction immediates.)
template <int M, int MTW>
__device__ __forceinline__ void mfma_tiles(uint32_t buf, u32x4 whi, u32x4 wlo, const f16x8& ah, const f16x8& al,
                                           const f16x8& bh, const f16x8& bl, f32x4 (&out)[2][MTW],
                                           f32x4 (&cross)[2][MTW]) {
    if constexpr (M < MTW) {
        u32x4 nhi = whi, nlo = wlo;
        if constexpr (M + 1 < MTW) {
            nhi = lds_read16<((M + 1) * 2 + 0) * 1024>(buf);
            nlo = lds_read16<((M + 1) * 2 + 1) * 1024>(buf);
        }
        const f16x8 h = __builtin_bit_cast(f16x8, whi), l = __builtin_bit_cast(f16x8, wlo);
        out[0][M] = QB_MFMA_F16(h, ah, out[0][M]);
        out[1][M] = QB_MFMA_F16(h, bh, out[1][M]);
        cross[0][M] = QB_MFMA_F16(h, al, cross[0][M]);
        cross[1][M] = QB_MFMA_F16(h, bl, cross[1][M]);
        cross[0][M] = QB_MFMA_F16(l, ah, cross[0][M]);
        cross[1][M] = QB_MFMA_F16(l, bh, cross[1][M]);
        if constexpr (M + 1 < MTW) lds_wait(nhi, nlo);
        mfma_tiles<M + 1, MTW>(buf, nhi, nlo, ah, al, bh, bl, out, cross);
    }
}

// One dense layer.  The block walks a flat sequence of k-steps q = pass * KS + s over its voxel passes.
// Waves 0-3 also stage the weights of step q+1 (L2-resident, one step ahead, two LDS buffers); waves
// 4-7 also stage the activations of step q+D (HBM, D = R-1 steps ahead, a ring of R LDS slots) --
// two loader roles so that each wave's in-order vmcnt counter tracks ONE prefetch distance, and the
// activation stream stays D steps deep in flight across the per-step barrier (raw s_barrier with a
// counted s_waitcnt, never vmcnt(0) on the activation waves), including across pass boundaries and
// under the epilogue's stores.
template <int KS, int MTW, int NSPLIT, bool RELU_IN, int EPI>
__global__ __launch_bounds__(kWB) void wide_dense_kernel(WideArgs a) {
    extern __shared__ __align__(16) uint4 wlds[];
    constexpr int MT = MTW * NSPLIT;
    constexpr int NVG = (kWB / 64) / NSPLIT;       // 32-voxel groups per block pass
    constexpr int VPB = NVG * 32;                  // voxels per block pass
    constexpr int WFR = MT * 2;                    // 1 KiB weight fragments per k-step
    constexpr int AFR = NVG * 4;                   // 1 KiB activation fragments per k-step
    constexpr int R = NSPLIT == 2 ? 5 : 3;         // activation ring slots
    constexpr int D = R - 1;                       // activation prefetch distance (k-steps)
    constexpr int APW = AFR / 4;                   // activation fragments per loader wave per step
    constexpr int VGW = NVG / 4;                   // voxel groups per activation-loader wave
    uint4* wbuf = wlds;                            // [2][WFR][64]
    uint4* abuf = wlds + 2 * WFR * 64;             // [R][AFR][64]
    float* lbias = reinterpret_cast<float*>(abuf + R * AFR * 64);  // [16 MT]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int osplit = wave % NSPLIT, vgrp = wave / NSPLIT;
    const bool wloader = wave < 4;
    const int64_t nblk = (a.N + VPB - 1) / VPB;
    const int np = (int)((nblk - blockIdx.x + gridDim.x - 1) / gridDim.x);  // passes of this block (>= 1)
    const int Q = np * KS;

    for (int k = threadIdx.x; k < 16 * MT; k += kWB) lbias[k] = a.bias[k];

    auto issue_w = [&](int q) {
        const int s = q % KS;
        const uint4* src = a.W + (int64_t)s * WFR * 64 + lane;
        uint4* dst = wbuf + (q & 1) * WFR * 64;
#pragma unroll
        for (int f0 = 0; f0 < WFR; f0 += 4) {
            const int f = f0 + wave;
            if (WFR % 4 == 0 || f < WFR) glds16(src + f * 64, dst + f * 64);
        }
    };
    auto issue_a = [&](int q) {
        const int s = q % KS;
        const int64_t blk = blockIdx.x + (int64_t)(q / KS) * gridDim.x;
        uint4* slot = abuf + (q % R) * AFR * 64;
#pragma unroll
        for (int k = 0; k < VGW; ++k) {
            const int vg = (wave - 4) * VGW + k;
#pragma unroll
            for (int vt = 0; vt < 2; ++vt) {
                int64_t v = blk * VPB + vg * 32 + vt * 16 + i;
                v = v < a.N ? v : a.N - 1;  // clamp: every lane takes part in the MFMAs; stores are predicated
                const float* src = a.X + v * a.ldx + 32 * s + 8 * g;
                glds16(src, slot + ((vg * 2 + vt) * 2 + 0) * 64);
                glds16(src + 4, slot + ((vg * 2 + vt) * 2 + 1) * 64);
            }
        }
    };

    if (wloader) {
        issue_w(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        const int pre = D < Q ? D : Q;
        for (int q = 0; q < pre; ++q) issue_a(q);
        wait_groups<APW>(pre - 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the bias copy
    __builtin_amdgcn_s_barrier();

    f32x4 out[2][MTW], cross[2][MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        out[0][m] = out[1][m] = cross[0][m] = cross[1][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll 1
    for (int q = 0; q < Q; ++q) {
        // slots / buffers written here were last read in step q-1, which every wave left through a barrier
        if (wloader) {
            if (q + 1 < Q && !(a.dbg & 128)) issue_w(q + 1);
        } else if (q + D < Q && !(a.dbg & 32)) {
            issue_a(q + D);
        }
        const uint32_t as = lds_addr(abuf + (q % R) * AFR * 64 + (vgrp * 4) * 64 + lane);
        u32x4 a0 = lds_read16<0>(as), a1 = lds_read16<1024>(as), b0 = lds_read16<2048>(as), b1 = lds_read16<3072>(as);
        const uint32_t buf = lds_addr(wbuf + (q & 1) * WFR * 64 + (osplit * MTW * 2) * 64 + lane);
        // weight fragments are fetched one output tile ahead of the MFMAs that consume them
        u32x4 whi = lds_read16<0>(buf), wlo = lds_read16<1024>(buf);
        lds_wait(a0, a1, b0, b1);
        float fa[8], fb[8];
        {
            const float4 x0 = __builtin_bit_cast(float4, a0), x1 = __builtin_bit_cast(float4, a1);
            const float4 y0 = __builtin_bit_cast(float4, b0), y1 = __builtin_bit_cast(float4, b1);
            fa[0] = x0.x; fa[1] = x0.y; fa[2] = x0.z; fa[3] = x0.w; fa[4] = x1.x; fa[5] = x1.y; fa[6] = x1.z; fa[7] = x1.w;
            fb[0] = y0.x; fb[1] = y0.y; fb[2] = y0.z; fb[3] = y0.w; fb[4] = y1.x; fb[5] = y1.y; fb[6] = y1.z; fb[7] = y1.w;
        }
        if (RELU_IN) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                fa[j] = fmaxf(fa[j], 0.0f);
                fb[j] = fmaxf(fb[j], 0.0f);
            }
        }
        f16x8 ah, al, bh, bl;
        qb::split8(fa, ah, al);
        qb::split8(fb, bh, bl);
        lds_wait(whi, wlo);
        if (!(a.dbg & 16)) mfma_tiles<0, MTW>(buf, whi, wlo, ah, al, bh, bl, out, cross);
        if (q % KS == KS - 1) {
            // epilogue: lane (g, i) holds output rows 16 t + 4 g + 0..3 of voxels va (tile 0), vb (tile 1)
            const int64_t blk = blockIdx.x + (int64_t)(q / KS) * gridDim.x;
            const int64_t va = blk * VPB + vgrp * 32 + i, vb = va + 16;
#pragma unroll
            for (int vt = 0; vt < 2; ++vt) {
                const int64_t v = vt ? vb : va;
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const int col = 16 * (osplit * MTW + m) + 4 * g;
                    u32x4 braw = lds_read16<0>(lds_addr(lbias + col));
                    lds_wait(braw);   // (no copy of braw before this: its registers are still being written)
                    const float4 bi = __builtin_bit_cast(float4, braw);
                    float y[4] = {fmaf(cross[vt][m][0], QB_LO_UNSCALE, out[vt][m][0]) + bi.x,
                                  fmaf(cross[vt][m][1], QB_LO_UNSCALE, out[vt][m][1]) + bi.y,
                                  fmaf(cross[vt][m][2], QB_LO_UNSCALE, out[vt][m][2]) + bi.z,
                                  fmaf(cross[vt][m][3], QB_LO_UNSCALE, out[vt][m][3]) + bi.w};
                    out[vt][m] = cross[vt][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    if (v >= a.N || (a.dbg & 64)) continue;
                    if (EPI == EPI_HEAD) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = col + r;
                            if (row < 5) a.q[v * 5 + row] = y[r];
                            else if (row < 5 + a.T && a.ls) a.ls[v * a.T + (row - 5)] = y[r];
                        }
                        continue;
                    }
                    if (EPI == EPI_RELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.0f);
                    }
                    if (EPI == EPI_GATE) {
                        const float4 sk = *reinterpret_cast<const float4*>(a.skip + v * a.ldy + col);
                        const float4 rr = *reinterpret_cast<const float4*>(a.r + v * a.ldy + col);
                        const float s4[4] = {sk.x, sk.y, sk.z, sk.w}, r4[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float gate = qb::sigmoidf_(y[r]);                  // model.py:169
                            y[r] = s4[r] * (1.0f - gate) + r4[r] * gate;             // model.py:170
                        }
                    }
                    *reinterpret_cast<float4*>(a.Y + v * a.ldy + col) = make_float4(y[0], y[1], y[2], y[3]);
                }
            }
        }
        // publish step q+1: each loader waits for ITS loads of that step, then everyone meets
        if (wloader) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int ahead = Q - 2 - q;  // steps beyond q+1 already requested (capped at D-1)
            wait_groups<APW>(ahead < 0 ? 0 : (ahead < D - 1 ? ahead : D - 1));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// normalise_data (model.py:97-113) into rows of ld floats, zero beyond T
__global__ void wide_normalise_kernel(QbDev c, const float* __restrict__ x, float* __restrict__ out,
                                      int ld, int64_t N) {
    const int T = c.T, se = c.se_idx;
    const int64_t total = N * ld;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(e % ld);
        const int64_t v = e / ld;
        float o = 0.0f;
        if (t < T) {
            const float* xv = x + v * T;
            float den;
            if (c.multi_norm)
                den = (qb::clampf_(xv[se - 1], 1e-2f, 1e8f) + qb::clampf_(xv[se], 1e-2f, 1e8f) +
                       qb::clampf_(xv[se + 1], 1e-2f, 1e8f)) / 3.0f;
            else
                den = qb::clampf_(xv[se], 1e-2f, 1e8f);
            o = logf(qb::clampf_(xv[t], 1e-2f, 1e8f) / den);
        }
        out[e] = o;
    }
}

template <int KS, int MTW, int NSPLIT, bool RELU_IN, int EPI>
int launch_one(const qbold_ctx* ctx, const WideArgs& a, hipStream_t s) {
    constexpr int MT = MTW * NSPLIT, NVG = (kWB / 64) / NSPLIT, R = NSPLIT == 2 ? 5 : 3;
    constexpr int VPB = NVG * 32;
    const size_t smem = sizeof(uint4) * (2 * MT * 2 * 64 + R * NVG * 4 * 64) + sizeof(float) * 16 * MT;
    auto k = wide_dense_kernel<KS, MTW, NSPLIT, RELU_IN, EPI>;
    QB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem));
    const int64_t nblk = (a.N + VPB - 1) / VPB;
    const int grid = (int)(nblk < ctx->num_cus ? nblk : ctx->num_cus);
    hipLaunchKernelGGL(k, dim3(grid), dim3(kWB), smem, s, a);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

// body layers: K = U
template <bool RELU_IN, int EPI>
int launch_body(const qbold_ctx* ctx, int U, const WideArgs& a, hipStream_t s) {
    if (U == 256) return launch_one<8, 8, 2, RELU_IN, EPI>(ctx, a, s);
    return launch_one<4, 8, 1, RELU_IN, EPI>(ctx, a, s);
}
int launch_first(const qbold_ctx* ctx, int U, int KS1, const WideArgs& a, hipStream_t s) {
    if (U == 256) return KS1 == 1 ? launch_one<1, 8, 2, false, EPI_RELU>(ctx, a, s)
                                  : launch_one<2, 8, 2, false, EPI_RELU>(ctx, a, s);
    return KS1 == 1 ? launch_one<1, 8, 1, false, EPI_RELU>(ctx, a, s)
                    : launch_one<2, 8, 1, false, EPI_RELU>(ctx, a, s);
}
template <int KS>
int launch_head_ks(const qbold_ctx* ctx, int HT, const WideArgs& a, hipStream_t s) {
    switch (HT) {
        case 1: return launch_one<KS, 1, 1, false, EPI_HEAD>(ctx, a, s);
        case 2: return launch_one<KS, 2, 1, false, EPI_HEAD>(ctx, a, s);
        case 3: return launch_one<KS, 3, 1, false, EPI_HEAD>(ctx, a, s);
        case 4: return launch_one<KS, 4, 1, false, EPI_HEAD>(ctx, a, s);
        default: return launch_one<KS, 5, 1, false, EPI_HEAD>(ctx, a, s);
    }
}
int launch_head(const qbold_ctx* ctx, int U, int HT, const WideArgs& a, hipStream_t s) {
    return U == 256 ? launch_head_ks<8>(ctx, HT, a, s) : launch_head_ks<4>(ctx, HT, a, s);
}

int check_wide(const qbold_ctx* ctx, const qbold_encoder_shape* s, const char* who) {
    if (!wide_supported(s)) {
        qb::set_error(std::string(who) + ": the weight-streaming encoder is built for U = 128 or 256, "
                      "L <= 8, T <= 64, channel-wise gating, QBOLD_ENC_F32");
        return QBOLD_ERR_UNSUPPORTED;
    }
    if (ctx && s->T != ctx->dev.T) {
        qb::set_error(std::string(who) + ": encoder shape T differs from the context's tau grid");
        return QBOLD_ERR_INVALID;
    }
    return QBOLD_OK;
}

}  // namespace

extern "C" int64_t qbold_encoder_wide_packed_floats(const qbold_encoder_shape* s) {
    if (!wide_supported(s)) return QBOLD_ERR_UNSUPPORTED;
    return make_wide_layout(s->T, s->U, s->L).total;
}

extern "C" int64_t qbold_encoder_wide_workspace_floats(const qbold_encoder_shape* s, int64_t N) {
    if (!wide_supported(s) || N < 0) return QBOLD_ERR_UNSUPPORTED;
    const int K1 = 32 * ((s->T + 31) / 32);
    return N * (int64_t)(K1 + 4 * s->U);  // normalised input, b, skip, t, r
}

extern "C" int qbold_encoder_wide_pack(const qbold_ctx* ctx, const qbold_encoder_shape* shape,
                                       const float* weights, float* packed, void* stream) {
    QB_NEED_DEVICE(ctx);
    int rc = check_wide(ctx, shape, "qbold_encoder_wide_pack");
    if (rc) return rc;
    QB_REQUIRE(weights && packed, "qbold_encoder_wide_pack: null buffer");
    const int T = shape->T, U = shape->U, L = shape->L;
    const WideLayout wl = make_wide_layout(T, U, L);
    const qb::CanonLayout c = qb::make_canon(T, U, L, shape->channelwise_gating, shape->spatial_taps);
    hipStream_t s = (hipStream_t)stream;
    auto pack = [&](const OpImage& o, const float* W, const float* b, int nin, int nout, const float* W2,
                    const float* b2, int nout2, float add) {
        const int64_t n = image_floats(o.KS, o.MT) * 2 + 16 * o.MT;
        hipLaunchKernelGGL(wide_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, o, W, b, nin,
                           nout, W2, b2, nout2, add, packed);
    };
    pack(wl.first, weights + c.W0, weights + c.b0, T, U, nullptr, nullptr, 0, 0.0f);
    const int ctr = c.taps == 9 ? 4 * U * U : 0;  // voxel batches see the centre tap of the 3x3x1 kernels
    for (int l = 0; l < L; ++l) {
        const float* wb = weights + c.blk0 + (int64_t)l * c.blk_stride;
        pack(wl.blk[l][0], wb + c.Wc, wb + c.bc, U, U, nullptr, nullptr, 0, 0.0f);
        pack(wl.blk[l][1], wb + c.Wr1 + ctr, wb + c.br1, U, U, nullptr, nullptr, 0, 0.0f);
        pack(wl.blk[l][2], wb + c.Wr2 + ctr, wb + c.br2, U, U, nullptr, nullptr, 0, 0.0f);
        pack(wl.blk[l][3], wb + c.Wg, wb + c.bg, U, U, nullptr, nullptr, 0, shape->gate_offset);
    }
    pack(wl.head, weights + c.Wf, weights + c.bf, U, 5, weights + c.Ws, weights + c.bs, T, 0.0f);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_encoder_wide_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape,
                                      const float* packed, const float* x, int stream_sel,
                                      float* workspace, float* out_q, float* out_log_sigma, int64_t N,
                                      void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_RELU_ONLY(shape, "qbold_encoder_wide_fwd");
    int rc = check_wide(ctx, shape, "qbold_encoder_wide_fwd");
    if (rc) return rc;
    QB_REQUIRE(N >= 0, "qbold_encoder_wide_fwd: negative N");
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(packed && x && workspace && out_q, "qbold_encoder_wide_fwd: null buffer");
    QB_REQUIRE(stream_sel == 1 || stream_sel == 2, "qbold_encoder_wide_fwd: stream_sel must be 1 or 2");
    QB_REQUIRE(reinterpret_cast<uintptr_t>(workspace) % 16 == 0 && reinterpret_cast<uintptr_t>(packed) % 16 == 0,
               "qbold_encoder_wide_fwd: workspace and packed image must be 16-byte aligned");
    const int T = shape->T, U = shape->U, L = shape->L;
    const WideLayout wl = make_wide_layout(T, U, L);
    hipStream_t s = (hipStream_t)stream;
    const int K1 = 32 * wl.KS1;
    float* xn = workspace;
    float* bufs[4];
    for (int k = 0; k < 4; ++k) bufs[k] = workspace + N * (int64_t)K1 + (int64_t)k * N * U;
    {
        const int64_t total = N * K1;
        const int64_t nb = (total + 255) / 256;
        const int64_t cap = (int64_t)ctx->num_cus * 8;
        hipLaunchKernelGGL(wide_normalise_kernel, dim3((unsigned)(nb < cap ? nb : cap)), dim3(256), 0, s, ctx->dev,
                           x, xn, K1, N);
    }
    auto args = [&](const float* X, int ldx, const OpImage& o, float* Y) {
        WideArgs a{};
        a.X = X; a.ldx = ldx;
        a.W = reinterpret_cast<const uint4*>(packed + o.A);
        a.bias = packed + o.b;
        a.Y = Y; a.ldy = U;
        a.T = T; a.N = N;
        a.dbg = QB_ABLATE_MASK(ctx->dev);
        return a;
    };
    float* cur = bufs[0];
    rc = launch_first(ctx, U, wl.KS1, args(xn, K1, wl.first, cur), s);
    if (rc) return rc;
    if (stream_sel == 1) {  // a <- relu(Wc a + bc), model.py:144-145
        for (int l = 0; l < L; ++l) {
            float* nxt = cur == bufs[0] ? bufs[1] : bufs[0];
            rc = launch_body<false, EPI_RELU>(ctx, U, args(cur, U, wl.blk[l][0], nxt), s);
            if (rc) return rc;
            cur = nxt;
        }
    } else {  // gated residual block, model.py:147-172
        for (int l = 0; l < L; ++l) {
            float* skip = bufs[1], *t = bufs[2], *r = bufs[3];
            rc = launch_body<false, EPI_RELU>(ctx, U, args(cur, U, wl.blk[l][0], skip), s);
            if (rc) return rc;
            rc = launch_body<true, EPI_RELU>(ctx, U, args(cur, U, wl.blk[l][1], t), s);     // relu(b) in, :151-155
            if (rc) return rc;
            rc = launch_body<false, EPI_LINEAR>(ctx, U, args(t, U, wl.blk[l][2], r), s);    // :156
            if (rc) return rc;
            WideArgs g = args(r, U, wl.blk[l][3], t);  // b_new written over t
            g.skip = skip;
            g.r = r;
            rc = launch_body<false, EPI_GATE>(ctx, U, g, s);
            if (rc) return rc;
            // rotate: t holds the new b; the old b's buffer becomes the next t
            float* old = cur;
            cur = t;
            bufs[2] = old;
            if (old == bufs[0]) bufs[0] = nullptr;  // bufs[0] is only the first layer's output
        }
    }
    WideArgs h = args(cur, U, wl.head, nullptr);
    h.q = out_q;
    h.ls = stream_sel == 2 ? out_log_sigma : nullptr;
    return launch_head(ctx, U, wl.HT, h, s);
}
