// wide_fused_kernels.hip -- the whole stream-2 encoder for widths beyond the LDS-resident kernel
// (BASELINE config 3: 64 taus, no_units = 256) in ONE launch, activations never leaving the register file.
//
// Reference: normalise_data (model.py:97-113) and create_encoder (model.py:122-223) on (N,1,1,1,T) voxel
// batches, as encoder_core.h:  h = relu(W0 n + b0);  per block  skip = relu(Wc b + bc),
// t = relu(Wr1 relu(b) + br1), r = Wr2 t + br2, g = sigmoid(Wg r + bg + gate_offset), b <- skip (1 - g) + r g;
// heads q = Wf b + bf, log sigma = Ws b + bs.
//
// Why one launch.  wide_kernels.hip runs one weight-streaming GEMM per layer over float32 activations in HBM:
// at N = 1 M, U = 256 that is ~21 GB of activation traffic per evaluation against 0.32 GB of algorithmic bytes
// (signals in, heads out) and 6 ms of HBM-bound launches.  Here a wave keeps its voxels' activations in
// registers from the first layer to the heads; HBM sees the signals once and the heads once.
//
// Mapping.  256-thread workgroups, one per CU, ONE wave per SIMD, so a wave may use the whole 512-entry
// register file.  A wave owns 32 voxels (two 16-voxel MFMA column tiles); every dense layer is computed
// transposed (weights = A operand, activations = B operand) so that the 16x16 accumulator layout (voxel on
// lane & 15, units 4 (lane >> 4) + reg) IS the next layer's B operand if the weight image is stored in that k
// order (encoder_core.h).  A 256-unit activation panel of 32 voxels is kept as ready-made B operands -- split
// f16 halves hi / lo of every value, 8 k-steps x 2 voxel tiles x (4 + 4) VGPRs = 128 VGPRs -- and a gated
// block needs three panels (b / relu b, skip, t -> r), 384 VGPRs, plus one output tile's accumulators.
// The loop nest is output tile outermost: the B operands of all k-steps are register-resident, so a tile's
// accumulators live for 48 MFMAs only and its result goes straight into the next panel (bias, activation,
// split).  Arithmetic: x = hi + lo with hi = f16(x), lo = f16(x - hi) UNSCALED, so that all three products
// (hi.hi, hi.lo, lo.hi) add into ONE accumulator per voxel tile and the epilogue has nothing to combine;
// weights are pre-multiplied per dense op by a power of two that puts max |w| in [2^12, 2^13) (their lo halves
// stay normal f16 numbers down to |w| = 2^-15 max |w|), undone for free in the epilogue's y = acc 2^-e + bias.
// An activation's lo half is a normal number for |x| >= 2^-3 and carries an absolute error <= 2^-25 below.
//
// Weights.  All dense ops of one pass over 128 voxels form one flat stream of 1 KiB MFMA fragments in
// consumption order ([op][out tile][k-step][hi, lo]; 2.2 MB for config 3, L2-resident).  Stages of 16
// fragments travel L2 -> LDS by LDS-direct loads (global_load_lds_dwordx4, each wave issues a quarter) into a
// ring of four 16 KiB slots, three stages ahead; the handshake sits in the MIDDLE of a stage -- counted
// s_waitcnt vmcnt, raw s_barrier, issue of the stage that reuses the slot everyone has just left -- so the
// first fragments of the next stage are known to have landed before the current stage ends and the fragment
// reads run one pair ahead of the MFMAs without a bubble at stage boundaries.  The stream wraps from one
// pass to the next (same weights), so the ring never drains.  Biases live in LDS for the whole launch.
//
// Per pass and wave: 6,576 MFMAs (16 cycles each) = 105 k cycles of matrix pipe for 32 voxels; at N = 1 M that
// is 32 passes per CU = 1.40 ms at 2.4 GHz -- the floor this kernel is measured against.
#include "canon_layout.h"
#include "encoder_core.h"
#include "qbold_ctx.h"
#include "wide_common.h"

namespace {

using namespace qbw;

#ifndef QB_FUSED_VALU_PER_MFMA
#define QB_FUSED_VALU_PER_MFMA 3
#endif
constexpr int kFB = 256;          // threads per block: one wave per SIMD
constexpr int kRing = 8;          // LDS ring slots of 16 fragments (128 KiB)
constexpr int kAhead = 7;         // stages requested beyond the one being read
constexpr int kStageFrags = 16;   // 1 KiB fragments per stage
constexpr int kVoxPerPass = 128;  // 4 waves x 32 voxels

// ---- fused image layout (host + device) ------------------------------------------------------------
struct FusedLayout {
    int T, U, L;
    int KS1, KS, MT, TT, HT;  // first-layer k-steps, body k-steps, body tiles, log-sigma tiles, head tiles
    int frags_first, frags_op, frags_head, frags_pass, stages_pass;
    int bias_first, bias_blk0, bias_head, bias_total;  // float offsets inside the bias image
    int scale_off, n_ops, aux_floats;                   // per-op (2^e, 2^-e) pairs behind the biases
    int64_t img_floats, total_floats;                   // weight fragments, + biases and scales
};
__host__ __device__ constexpr inline FusedLayout make_fused_layout(int T, int U, int L) {
    FusedLayout f{};
    f.T = T; f.U = U; f.L = L;
    f.KS1 = (T + 31) / 32;
    f.KS = U / 32;
    f.MT = U / 16;
    f.TT = (T + 15) / 16;
    f.HT = f.TT + 1;
    f.frags_first = f.MT * f.KS1 * 2;
    f.frags_op = f.MT * f.KS * 2;
    f.frags_head = f.HT * f.KS * 2;
    f.frags_pass = f.frags_first + 4 * L * f.frags_op + f.frags_head;
    f.stages_pass = (f.frags_pass + kStageFrags - 1) / kStageFrags;
    f.bias_first = 0;
    f.bias_blk0 = U;
    f.bias_head = U + 4 * L * U;
    f.bias_total = f.bias_head + 16 * f.HT;
    f.scale_off = (f.bias_total + 3) & ~3;
    f.n_ops = 2 + 4 * L;  // first layer, four per block, heads
    f.aux_floats = (f.scale_off + 2 * f.n_ops + 3) & ~3;
    f.img_floats = (int64_t)f.stages_pass * kStageFrags * 256;
    f.total_floats = f.img_floats + f.aux_floats;
    return f;
}
inline bool fused_supported(const qbold_encoder_shape* s) {
    // the unrolled kernels below are instantiated for these shapes (T <= 16 or 49 .. 64 taus: one or four
    // log-sigma tiles); anything else keeps the layer-wise path
    return s && s->U == 256 && (s->L == 1 || s->L == 2) && ((s->T >= 1 && s->T <= 16) || (s->T >= 49 && s->T <= 64)) &&
           s->channelwise_gating && s->precision == QBOLD_ENC_F32;
}

// Power-of-two scale of one dense op: 2^e with max |w| 2^e in [2^12, 2^13); (2^e, 2^-e) -> scale[0..1].
__global__ void fused_scale_kernel(const float* __restrict__ W, int64_t n, const float* __restrict__ W2, int64_t n2,
                                   float* __restrict__ scale) {
    __shared__ float red[256];
    float m = 0.0f;
    for (int64_t k = threadIdx.x; k < n; k += 256) m = fmaxf(m, fabsf(W[k]));
    for (int64_t k = threadIdx.x; k < n2; k += 256) m = fmaxf(m, fabsf(W2[k]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mx = red[0];
        int e = 0;
        if (mx > 0.0f && mx < 3.0e38f) e = 12 - ilogbf(mx);
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        scale[0] = ldexpf(1.0f, e);
        scale[1] = ldexpf(1.0f, -e);
    }
}

// One dense op into the fused image.  korder 0: in = 32 s + 8 g + j (first layer: rows of x); 1: the
// accumulator order unit(s, g, j) = 16 (2 s + (j >> 2)) + 4 g + (j & 3) of encoder_core.h.  Head (W2 != null):
// output row < 16 TT -> log-sigma row of W (Ws), else q row (row - 16 TT) of W2 (Wf).  Weights are stored
// times scale[0] as hi = f16(w'), lo = f16(w' - hi); biases unscaled.
__global__ void fused_pack_kernel(int frag0, int KS, int MT, int korder, const float* __restrict__ W,
                                  const float* __restrict__ b, int nin, int nout, const float* __restrict__ W2,
                                  const float* __restrict__ b2, int nout2, int split_row, float bias_add,
                                  int bias_off, const float* __restrict__ scale, int64_t img_floats,
                                  float* __restrict__ packed) {
    _Float16* ph = reinterpret_cast<_Float16*>(packed) + (int64_t)frag0 * 512;
    const int64_t halves = (int64_t)MT * KS * 2 * 512;
    const float sc = scale[0];
    for (int64_t h = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; h < halves + 16 * MT;
         h += (int64_t)gridDim.x * blockDim.x) {
        if (h >= halves) {  // bias rows in the same (possibly remapped) order
            const int row = (int)(h - halves);
            float v = 0.0f;
            if (W2) {
                if (row < split_row) v = row < nout ? b[row] : 0.0f;
                else v = row - split_row < nout2 ? b2[row - split_row] : 0.0f;
            } else if (row < nout) {
                v = b[row] + bias_add;
            }
            packed[img_floats + bias_off + row] = v;
            continue;
        }
        const int j = (int)(h & 7), lane = (int)((h >> 3) & 63), part = (int)((h >> 9) & 1);
        const int64_t pair = h >> 10;  // m * KS + s
        const int s = (int)(pair % KS), m = (int)(pair / KS);
        const int g = lane >> 4, i = lane & 15;
        const int in = korder ? 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3) : 32 * s + 8 * g + j;
        const int out = 16 * m + i;
        float w = 0.0f;
        if (in < nin) {
            if (W2) {
                if (out < split_row) w = out < nout ? W[(int64_t)in * nout + out] : 0.0f;
                else w = out - split_row < nout2 ? W2[(int64_t)in * nout2 + (out - split_row)] : 0.0f;
            } else if (out < nout) {
                w = W[(int64_t)in * nout + out];
            }
        }
        w *= sc;  // exact: a power of two
        const _Float16 hi = (_Float16)w;
        ph[h] = part == 0 ? hi : (_Float16)(w - (float)hi);
    }
}

// ---- device side ---------------------------------------------------------------------------------------
// B operands of every k-step of one activation tensor, for this wave's two voxel tiles, as packed f16 pairs:
// dword d of fragment [k-step][voxel tile] holds k-slots 2 d, 2 d + 1.  (Kept as opaque 32-bit values: left as
// f16x8 vectors filled element by element, the compiler carries every half in a register of its own until the
// MFMA -- 256 registers per panel.)
template <int KS>
struct Panel {
    uint32_t hi[KS][2][4], lo[KS][2][4];
};
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
template <int KS>
__device__ __forceinline__ f16x8 frag_hi(const Panel<KS>& P, int s, int vt) {
    return __builtin_bit_cast(f16x8, u32x4v{P.hi[s][vt][0], P.hi[s][vt][1], P.hi[s][vt][2], P.hi[s][vt][3]});
}
template <int KS>
__device__ __forceinline__ f16x8 frag_lo(const Panel<KS>& P, int s, int vt) {
    return __builtin_bit_cast(f16x8, u32x4v{P.lo[s][vt][0], P.lo[s][vt][1], P.lo[s][vt][2], P.lo[s][vt][3]});
}
__device__ __forceinline__ uint32_t pack2(_Float16 a, _Float16 b) {
    uint32_t d = __builtin_bit_cast(uint32_t, qb::f16x2{a, b});
    asm volatile("" : "+v"(d));  // one packed register from here on
    return d;
}

// The weight-fragment stream of this wave.
struct Stream {
    const char* img;    // image + this wave's quarter of a stage (wave-uniform: SGPRs)
    uint32_t lane16;    // 16 lane: the only per-lane part of an LDS-direct load's address
    uint32_t src_off;   // byte offset of the next stage to fetch (wave-uniform)
    uint4* ring;        // LDS ring + this wave's quarter of a slot
    uint32_t ring_lds;  // LDS byte address of the ring + 16 lane
    uint32_t cur;       // LDS byte address of the stage being read (+ 16 lane)
    int issue_slot;     // ring slot the next issue fills
    int read_slot;      // ring slot of the stage being read
    uint32_t pass_bytes;
};

// A wave fetches fragments 4 w .. 4 w + 3 of a stage: scalar base + one lane-offset register, one M0, four
// LDS-direct loads told apart by their immediate offset (which the hardware adds to the global and to the LDS
// address alike).
template <int K>
__device__ __forceinline__ void glds16_off(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, K * 1024, 0);
}
__device__ __forceinline__ void stream_issue(Stream& S) {
    uint4* dst = S.ring + S.issue_slot * (kStageFrags * 64);
    const char* src = S.img + S.src_off + S.lane16;
    glds16_off<0>(src, dst);
    glds16_off<1>(src, dst);
    glds16_off<2>(src, dst);
    glds16_off<3>(src, dst);
    uint32_t nxt = S.src_off + kStageFrags * 1024;
    nxt = nxt == S.pass_bytes ? 0u : nxt;  // the stream wraps from pass to pass
    // opaque: the offsets of a pass repeat from pass to pass, and left visible the compiler computes all of
    // them ahead of the pass loop and spills them
    asm volatile("" : "+s"(nxt));
    S.src_off = nxt;
    S.issue_slot = (S.issue_slot + 1) & (kRing - 1);
}
// Mid-stage handshake: my quarter of the NEXT stage has landed (all but the youngest stage's four loads
// are done), everyone has left the previous stage, whose slot the new issue overwrites.
__device__ __forceinline__ void stream_sync(Stream& S) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kAhead - 2)) : "memory");
#ifndef QB_FUSED_X_NOBARRIER  // timing experiments only (scripts/dev/fused_variants.sh): results are invalid
    __builtin_amdgcn_s_barrier();
#endif
    stream_issue(S);
}
__device__ __forceinline__ void stream_next_stage(Stream& S) {
    S.read_slot = (S.read_slot + 1) & (kRing - 1);
    S.cur = S.ring_lds + (uint32_t)S.read_slot * (kStageFrags * 1024);
}

struct Frag {   // one (hi, lo) fragment pair of the weight stream
    u32x4 hi, lo;
};
struct Acc {    // a finished (or running) output tile: one accumulator per voxel tile and the tile's bias rows
    f32x4 o[2];
    u32x4 bias;
};
template <int V>
struct IC {
    static constexpr int value = V;
};
// all LDS reads but the two youngest (the pair requested last) have returned
#ifdef QB_FUSED_X_NOLDSWAIT  // timing experiments only: results are invalid
#define QB_FUSED_WAIT_BUT2 ""
#else
#define QB_FUSED_WAIT_BUT2 "s_waitcnt lgkmcnt(2)"
#endif
__device__ __forceinline__ void lds_wait_but2(u32x4& a, u32x4& b) {
    asm volatile(QB_FUSED_WAIT_BUT2 : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lds_wait_but2(u32x4& a, u32x4& b, u32x4& c) {
    asm volatile(QB_FUSED_WAIT_BUT2 : "+v"(a), "+v"(b), "+v"(c));
}

// Request the fragment pair that starts at fragment F of the pass; F >= FP: a pair of the NEXT pass (same
// weights: the stream wraps).  By the last quarter of a stage the next one has landed (mid-stage handshake),
// so the read-ahead never stalls at a tile, stage or pass boundary.
template <int F, int FP>
__device__ __forceinline__ void frag_fetch(Stream& S, Frag& w) {
    constexpr int FO = (F % FP) % kStageFrags;
    if constexpr (FO == 0) stream_next_stage(S);
    w.hi = lds_read16<(FO + 0) * 1024>(S.cur);
    w.lo = lds_read16<(FO + 1) * 1024>(S.cur);
}

// k-steps S_ .. KSOP-1 of one 16-row output tile, both voxel tiles.  wa holds the pair of step S_ (returned),
// wb the pair of step S_+1 (requested); the pair of step S_+2 is requested before the six MFMAs of step S_,
// and after them the wait leaves only that youngest pair outstanding.  The previous tile's epilogue (prev:
// eight pieces, one per output value of a lane) is spread over the k-steps so that its vector instructions
// issue between this tile's MFMAs, two or three per MFMA.
// oa / ob: the two pairs that follow the tile.  (Recursive template: every LDS offset is an immediate.)
template <int KSOP, int F0, int FP, int BOFF, int S_, class Prev>
__device__ __forceinline__ void tile_steps(Stream& S, uint32_t bias_lds, Frag wa, Frag wb, Frag& oa, Frag& ob,
                                           const Panel<KSOP>& in, Acc& acc, const Prev& prev) {
    if constexpr (S_ < KSOP) {
        if constexpr ((F0 + 2 * S_) % kStageFrags == kStageFrags / 2) stream_sync(S);
        if constexpr (S_ == 0) acc.bias = lds_read16<BOFF * 4>(bias_lds);
        Frag n;
        frag_fetch<F0 + 2 * S_ + 4, FP>(S, n);
        // the reads are in flight BEFORE this step's MFMAs and waited for AFTER them: nothing crosses either fence
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 h = __builtin_bit_cast(f16x8, wa.hi), l = __builtin_bit_cast(f16x8, wa.lo);
        const f16x8 h0 = frag_hi(in, S_, 0), h1 = frag_hi(in, S_, 1), l0 = frag_lo(in, S_, 0), l1 = frag_lo(in, S_, 1);
        acc.o[0] = QB_MFMA_F16(h, h0, acc.o[0]);
        acc.o[1] = QB_MFMA_F16(h, h1, acc.o[1]);
        acc.o[0] = QB_MFMA_F16(h, l0, acc.o[0]);
        acc.o[1] = QB_MFMA_F16(h, l1, acc.o[1]);
        acc.o[0] = QB_MFMA_F16(l, h0, acc.o[0]);
        acc.o[1] = QB_MFMA_F16(l, h1, acc.o[1]);
#ifndef QB_FUSED_X_NOEPI
        if constexpr ((0 * KSOP) / 8 == S_) prev(IC<0>{});
        if constexpr ((1 * KSOP) / 8 == S_) prev(IC<1>{});
        if constexpr ((2 * KSOP) / 8 == S_) prev(IC<2>{});
        if constexpr ((3 * KSOP) / 8 == S_) prev(IC<3>{});
        if constexpr ((4 * KSOP) / 8 == S_) prev(IC<4>{});
        if constexpr ((5 * KSOP) / 8 == S_) prev(IC<5>{});
        if constexpr ((6 * KSOP) / 8 == S_) prev(IC<6>{});
        if constexpr ((7 * KSOP) / 8 == S_) prev(IC<7>{});
#endif
        // issue order inside the step: one MFMA, then a share of the epilogue's vector instructions (which
        // issue while the matrix pipe works on it), six times over
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, QB_FUSED_VALU_PER_MFMA, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S_ == 0) lds_wait_but2(wb.hi, wb.lo, acc.bias);
        else lds_wait_but2(wb.hi, wb.lo);
        tile_steps<KSOP, F0, FP, BOFF, S_ + 1>(S, bias_lds, wb, n, oa, ob, in, acc, prev);
    } else {
        oa = wa;
        ob = wb;
    }
}

// One output tile into acc.  F0: fragment index within the pass of the tile's first fragment (a tile never
// straddles a stage); BOFF: float offset of the tile's bias rows in the LDS bias image.  wa / wb: in, the
// tile's first two pairs; out, the two pairs that follow it.
template <int KSOP, int F0, int FP, int BOFF, class Prev>
__device__ __forceinline__ void tile_mma(Stream& S, uint32_t bias_lds, Frag& wa, Frag& wb, const Panel<KSOP>& in,
                                         Acc& acc, const Prev& prev) {
    static_assert(F0 % kStageFrags + 2 * KSOP <= kStageFrags, "a tile's fragments stay inside one stage");
    static_assert(BOFF * 4 + 64 < 65536, "bias offset is a ds_read immediate");
    __builtin_amdgcn_sched_barrier(0);  // tiles are scheduled one at a time: three panels leave no slack
    acc.o[0] = acc.o[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    tile_steps<KSOP, F0, FP, BOFF, 0>(S, bias_lds, wa, wb, wa, wb, in, acc, prev);
}

// (W in + b)[16 M + 4 g + r] of voxel tile vt: piece E = 4 vt + r of a finished tile.  inv_scale undoes the
// dense op's power-of-two weight scale.
template <int E>
__device__ __forceinline__ float acc_elem(const Acc& a, float inv_scale) {
    constexpr int vt = E / 4, r = E % 4;
    const float4 b4 = __builtin_bit_cast(float4, a.bias);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
    return fmaf(a.o[vt][r], inv_scale, bb[r]);
}
// x = hi + lo, hi = f16(x), lo = f16(x - hi): the packed hi pair is converted once and its halves are read
// back for the residual (x - hi is exact in float32)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack2((_Float16)a, (_Float16)b);
    const qb::f16x2 h = __builtin_bit_cast(qb::f16x2, hi);
    lo = pack2((_Float16)__builtin_fmaf((float)h[0], -1.0f, a), (_Float16)__builtin_fmaf((float)h[1], -1.0f, b));
}
// k-slots 4 (M & 1) + 2 d, + 1 of k-step M / 2 hold units 16 M + 4 g + 2 d, + 1 (C = 2 vt + d)
template <int M, int C, int KS>
__device__ __forceinline__ void put_pair(Panel<KS>& P, float a, float b) {
    constexpr int vt = C / 2, d = C % 2;
    split_pair(a, b, P.hi[M / 2][vt][2 * (M & 1) + d], P.lo[M / 2][vt][2 * (M & 1) + d]);
}
// value E = 4 vt + r of tile M, read back from a panel
template <int M, int E, int KS>
__device__ __forceinline__ float get_elem(const Panel<KS>& P) {
    constexpr int vt = E / 4, d = (E % 4) / 2, e = E % 2;
    // laundered: otherwise the float32 values of the hi halves computed when the panel was written (split_pair)
    // are kept alive from that op to this one -- a second, spilled copy of the panel
    uint32_t dh = P.hi[M / 2][vt][2 * (M & 1) + d], dl = P.lo[M / 2][vt][2 * (M & 1) + d];
    asm volatile("" : "+v"(dh), "+v"(dl));
    const qb::f16x2 h = __builtin_bit_cast(qb::f16x2, dh);
    const qb::f16x2 l = __builtin_bit_cast(qb::f16x2, dl);
    return __builtin_fmaf((float)l[e], 1.0f, (float)h[e]);
}

// relu on a split panel, in place (the Activation in front of the first 3x3x1 convolution, model.py:151):
// the sign of hi + 2^-11 lo is the sign bit of hi (lo is below half an ulp of hi; x tiny negative gives hi = -0)
typedef short s16x2 __attribute__((ext_vector_type(2)));
template <int KS>
__device__ __forceinline__ void relu_panel(Panel<KS>& P) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int vt = 0; vt < 2; ++vt)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const s16x2 neg = __builtin_bit_cast(s16x2, P.hi[s][vt][d]) >> (short)15;  // 0xffff where negative
                const uint32_t keep = ~__builtin_bit_cast(uint32_t, neg);
                P.hi[s][vt][d] &= keep;
                P.lo[s][vt][d] &= keep;
            }
}

enum { EPI_RELU = 0, EPI_LINEAR = 1, EPI_GATE = 2 };

// Piece E (= 4 vt + r) of tile M's epilogue; the even piece of a pair parks its value in `carry`, the odd one
// splits and stores the pair.  EPI_GATE: out = skip (1 - g) + r g with g = sigmoid(W r + b) (model.py:164-170).
template <int EPI, int M, int E, int KS>
__device__ __forceinline__ void epi_elem(const Acc& a, float inv_scale, Panel<KS>& out, const Panel<KS>& skip,
                                         const Panel<KS>& rr, float& carry) {
    float y = acc_elem<E>(a, inv_scale);
    if constexpr (EPI == EPI_RELU) y = fmaxf(y, 0.0f);
    if constexpr (EPI == EPI_GATE) {
        const float sk = get_elem<M, E>(skip), r = get_elem<M, E>(rr);
        const float gate = qb::sigmoidf_(y);  // model.py:169
        y = fmaf(gate, r - sk, sk);           // skip (1 - g) + r g, model.py:170
    }
    if constexpr (E % 2 == 0) carry = y;
    else put_pair<M, E / 2>(out, carry, y);
}

// the dense op's 2^-e (LDS aux image, float index IDX): wave-uniform
template <int IDX>
__device__ __forceinline__ float op_inv_scale(uint32_t aux_lds) {
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(aux_lds), "n"(IDX * 4));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(r));
}

// A dense op: tiles M .. MT-1 of out = epi(W in + b).  Tile M-1's epilogue runs inside tile M's MFMA stream
// (two accumulator sets in rotation); the last tile's runs after the op.
template <int KS, int KSIN, int MT, int F0, int FP, int BOFF, int EPI, int M = 0>
__device__ __forceinline__ void dense_op(Stream& S, uint32_t bias_lds, float inv_scale, Frag& wa, Frag& wb,
                                         const Panel<KSIN>& in, Panel<KS>& out, const Panel<KS>& skip,
                                         const Panel<KS>& rr, Acc (&acc)[2]) {
    if constexpr (M < MT) {
        float carry = 0.0f;
        auto prev = [&](auto c) {
            if constexpr (M > 0)
                epi_elem<EPI, (M > 0 ? M - 1 : 0), decltype(c)::value>(acc[(M + 1) & 1], inv_scale, out, skip, rr, carry);
        };
        tile_mma<KSIN, F0 + M * 2 * KSIN, FP, BOFF + 16 * M>(S, bias_lds, wa, wb, in, acc[M & 1], prev);
        dense_op<KS, KSIN, MT, F0, FP, BOFF, EPI, M + 1>(S, bias_lds, inv_scale, wa, wb, in, out, skip, rr, acc);
    } else {
        __builtin_amdgcn_sched_barrier(0);
        float carry = 0.0f;
        epi_elem<EPI, MT - 1, 0>(acc[(MT - 1) & 1], inv_scale, out, skip, rr, carry);
        epi_elem<EPI, MT - 1, 1>(acc[(MT - 1) & 1], inv_scale, out, skip, rr, carry);
        epi_elem<EPI, MT - 1, 2>(acc[(MT - 1) & 1], inv_scale, out, skip, rr, carry);
        epi_elem<EPI, MT - 1, 3>(acc[(MT - 1) & 1], inv_scale, out, skip, rr, carry);
        epi_elem<EPI, MT - 1, 4>(acc[(MT - 1) & 1], inv_scale, out, skip, rr, carry);
        epi_elem<EPI, MT - 1, 5>(acc[(MT - 1) & 1], inv_scale, out, skip, rr, carry);
        epi_elem<EPI, MT - 1, 6>(acc[(MT - 1) & 1], inv_scale, out, skip, rr, carry);
        epi_elem<EPI, MT - 1, 7>(acc[(MT - 1) & 1], inv_scale, out, skip, rr, carry);
    }
}

struct FusedArgs {
    const float* x;     // [N][T]
    const uint4* img;   // fused weight image
    const float* bias;  // bias image (floats)
    float* q;           // [N][5]
    float* ls;          // [N][T] log sigma
    int64_t N;
    int T, se_idx, multi_norm;
};

// Piece E (= 4 vt + r) of head tile M: tiles 0 .. TT-1 are log-sigma rows 16 M + 4 g + r, tile TT holds the
// five q rows.  Even pieces park their value, odd ones store the pair.
template <int TT, int M, int E>
__device__ __forceinline__ void head_elem(const Acc& acc, float inv_scale, const FusedArgs& a, const int64_t (&v)[2],
                                          int g, float& carry) {
    constexpr int vt = E / 4, d = (E % 4) / 2;
    const float y = acc_elem<E>(acc, inv_scale);
    if constexpr (E % 2 == 0) {
        carry = y;
        return;
    }
    const float y0 = carry, y1 = y;
    if (v[vt] >= a.N) return;
    if constexpr (M < TT) {
        const int row = 16 * M + 4 * g + 2 * d;
        float* dst = a.ls + v[vt] * a.T + row;
        if ((a.T & 1) == 0 && row + 1 < a.T) {
            *reinterpret_cast<float2*>(dst) = make_float2(y0, y1);
        } else {
            if (row < a.T) dst[0] = y0;
            if (row + 1 < a.T) dst[1] = y1;
        }
    } else {
        const int row = 4 * g + 2 * d;
        if (row < 5) a.q[v[vt] * 5 + row] = y0;
        if (row + 1 < 5) a.q[v[vt] * 5 + row + 1] = y1;
    }
}
template <int KS, int TT, int F0, int FP, int BOFF, int M = 0>
__device__ __forceinline__ void head_op(Stream& S, uint32_t bias_lds, float inv_scale, Frag& wa, Frag& wb,
                                        const Panel<KS>& b, const FusedArgs& a, const int64_t (&v)[2], int g,
                                        Acc (&acc)[2]) {
    if constexpr (M <= TT) {
        float carry = 0.0f;
        auto prev = [&](auto c) {
            if constexpr (M > 0)
                head_elem<TT, (M > 0 ? M - 1 : 0), decltype(c)::value>(acc[(M + 1) & 1], inv_scale, a, v, g, carry);
        };
        tile_mma<KS, F0 + M * 2 * KS, FP, BOFF + 16 * M>(S, bias_lds, wa, wb, b, acc[M & 1], prev);
        head_op<KS, TT, F0, FP, BOFF, M + 1>(S, bias_lds, inv_scale, wa, wb, b, a, v, g, acc);
    } else {
        __builtin_amdgcn_sched_barrier(0);
        float carry = 0.0f;
        head_elem<TT, TT, 0>(acc[TT & 1], inv_scale, a, v, g, carry);
        head_elem<TT, TT, 1>(acc[TT & 1], inv_scale, a, v, g, carry);
        head_elem<TT, TT, 2>(acc[TT & 1], inv_scale, a, v, g, carry);
        head_elem<TT, TT, 3>(acc[TT & 1], inv_scale, a, v, g, carry);
        head_elem<TT, TT, 4>(acc[TT & 1], inv_scale, a, v, g, carry);
        head_elem<TT, TT, 5>(acc[TT & 1], inv_scale, a, v, g, carry);
        head_elem<TT, TT, 6>(acc[TT & 1], inv_scale, a, v, g, carry);
        head_elem<TT, TT, 7>(acc[TT & 1], inv_scale, a, v, g, carry);
    }
}

// Gated residual blocks LB .. L-1 (model.py:147-172): b comes in P0 and leaves in P1; the panels rotate by
// name from block to block, nothing is copied.  SC0: float index of the first block op's 2^-e in the aux image.
template <int KS, int MT, int L, int TT, int FB0, int FP, int SC0, int LB>
__device__ __forceinline__ void blocks_and_head(Stream& S, uint32_t bias_lds, uint32_t aux_lds, Frag& wa, Frag& wb,
                                                Panel<KS>& P0, Panel<KS>& P1, Panel<KS>& P2, const FusedArgs& a,
                                                const int64_t (&v)[2], int g, Acc (&acc)[2]) {
    constexpr int FOP = MT * KS * 2;  // fragments per dense op
    constexpr int U = 16 * MT;
    if constexpr (LB < L) {
        constexpr int F = FB0 + LB * 4 * FOP, B = U + LB * 4 * U, SC = SC0 + 8 * LB;
        dense_op<KS, KS, MT, F, FP, B, EPI_RELU>(S, bias_lds, op_inv_scale<SC>(aux_lds), wa, wb, P0, P2, P2, P2,
                                                 acc);                                             // skip, :148
        if constexpr (LB > 0) relu_panel(P0);  // block 0's input is a relu output already            :151
        dense_op<KS, KS, MT, F + FOP, FP, B + U, EPI_RELU>(S, bias_lds, op_inv_scale<SC + 2>(aux_lds), wa, wb, P0, P1,
                                                           P1, P1, acc);                            // t, :152-155
        dense_op<KS, KS, MT, F + 2 * FOP, FP, B + 2 * U, EPI_LINEAR>(S, bias_lds, op_inv_scale<SC + 4>(aux_lds), wa, wb,
                                                                     P1, P0, P0, P0, acc);          // r, :156
        dense_op<KS, KS, MT, F + 3 * FOP, FP, B + 3 * U, EPI_GATE>(S, bias_lds, op_inv_scale<SC + 6>(aux_lds), wa, wb,
                                                                   P0, P1, P2, P0, acc);            // :164-170
        blocks_and_head<KS, MT, L, TT, FB0, FP, SC0, LB + 1>(S, bias_lds, aux_lds, wa, wb, P1, P0, P2, a, v, g, acc);
    } else {
        head_op<KS, TT, FB0 + L * 4 * FOP, FP, U + L * 4 * U>(S, bias_lds, op_inv_scale<SC0 + 8 * L>(aux_lds), wa, wb,
                                                              P0, a, v, g, acc);
    }
}

template <int TT, int L>
__global__ __launch_bounds__(kFB) void wide_fused_kernel(FusedArgs a) {
    constexpr int U = 256, KS = U / 32, MT = U / 16, KS1 = (TT + 1) / 2;
    constexpr FusedLayout fl = make_fused_layout(16 * TT, U, L);  // T only pads inside its 16-row tile
    constexpr int FP = fl.frags_pass;
    static_assert(FP % kStageFrags == 0, "a pass is a whole number of stages");
    extern __shared__ __align__(16) uint4 smem[];
    uint4* ring = smem;                                                      // [kRing][16][64]
    float* lbias = reinterpret_cast<float*>(smem + kRing * kStageFrags * 64);  // [bias_total]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    for (int k = threadIdx.x; k < fl.aux_floats; k += kFB) lbias[k] = a.bias[k];  // biases, then per-op scales

    Stream S;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    S.img = reinterpret_cast<const char*>(a.img) + wave_u * 4096;
    S.lane16 = 16u * lane;
    S.src_off = 0;
    S.ring = ring + wave_u * 4 * 64;
    S.ring_lds = lds_addr(ring) + 16u * lane;
    S.cur = S.ring_lds;
    S.issue_slot = 0;
    S.read_slot = 0;
    S.pass_bytes = (uint32_t)fl.stages_pass * kStageFrags * 1024;
    const uint32_t bias_lds = lds_addr(lbias) + 16u * g;
    const uint32_t aux_lds = lds_addr(lbias);
    constexpr int SC0 = fl.scale_off + 1;  // float index of the first layer's 2^-e; op k's is SC0 + 2 k
#pragma unroll
    for (int k = 0; k < kAhead; ++k) stream_issue(S);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kAhead - 1)) : "memory");  // stage 0 has landed (my quarter)
    __syncthreads();                                                          // ... everyone's; the biases too
    Frag wa, wb;  // the next two fragment pairs of the stream
    wa.hi = lds_read16<0>(S.cur);
    wa.lo = lds_read16<1024>(S.cur);
    wb.hi = lds_read16<2048>(S.cur);
    wb.lo = lds_read16<3072>(S.cur);
    lds_wait(wa.hi, wa.lo, wb.hi, wb.lo);

    const int64_t nblk = (a.N + kVoxPerPass - 1) / kVoxPerPass;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        int64_t v[2];
        Panel<KS1> X;
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            v[vt] = blk * kVoxPerPass + wave * 32 + vt * 16 + i;
            const int64_t vc = v[vt] < a.N ? v[vt] : a.N - 1;  // clamp: every lane takes part in the MFMAs
            const float* xv = a.x + vc * a.T;
            float den;                                          // normalise_data, model.py:97-113
            if (a.multi_norm)
                den = (qb::clampf_(xv[a.se_idx - 1], 1e-2f, 1e8f) + qb::clampf_(xv[a.se_idx], 1e-2f, 1e8f) +
                       qb::clampf_(xv[a.se_idx + 1], 1e-2f, 1e8f)) / 3.0f;
            else
                den = qb::clampf_(xv[a.se_idx], 1e-2f, 1e8f);
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                float f[8];
                const int t0 = 32 * s + 8 * g;
                if ((a.T & 3) == 0) {
                    const float4 lo4 = t0 + 3 < a.T ? *reinterpret_cast<const float4*>(xv + t0) : make_float4(0, 0, 0, 0);
                    const float4 hi4 =
                        t0 + 7 < a.T ? *reinterpret_cast<const float4*>(xv + t0 + 4) : make_float4(0, 0, 0, 0);
                    f[0] = lo4.x; f[1] = lo4.y; f[2] = lo4.z; f[3] = lo4.w;
                    f[4] = hi4.x; f[5] = hi4.y; f[6] = hi4.z; f[7] = hi4.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = t0 + j < a.T ? xv[t0 + j] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    f[j] = t0 + j < a.T ? logf(qb::clampf_(f[j], 1e-2f, 1e8f) / den) : 0.0f;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    split_pair(f[2 * d], f[2 * d + 1], X.hi[s][vt][d], X.lo[s][vt][d]);
                }
            }
        }
        Panel<KS> P0, P1, P2;
        Acc acc[2];
        dense_op<KS, KS1, MT, 0, FP, 0, EPI_RELU>(S, bias_lds, op_inv_scale<SC0>(aux_lds), wa, wb, X, P0, P0, P0,
                                                  acc);  // first layer, model.py:181
        blocks_and_head<KS, MT, L, TT, MT * KS1 * 2, FP, SC0 + 2, 0>(S, bias_lds, aux_lds, wa, wb, P0, P1, P2, a, v, g,
                                                                     acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-direct load may land after the block has gone
}

}  // namespace

namespace qb {
bool wide_fused_supported(const qbold_encoder_shape* s) { return fused_supported(s); }

int wide_fused_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* packed, const float* x,
                   float* out_q, float* out_log_sigma, int64_t N, hipStream_t s) {
    const FusedLayout fl = make_fused_layout(shape->T, shape->U, shape->L);
    FusedArgs a{};
    a.x = x;
    a.img = reinterpret_cast<const uint4*>(packed);
    a.bias = packed + fl.img_floats;
    a.q = out_q;
    a.ls = out_log_sigma;
    a.N = N;
    a.T = shape->T;
    a.se_idx = ctx->dev.se_idx;
    a.multi_norm = ctx->dev.multi_norm;
    const size_t smem = sizeof(uint4) * kRing * kStageFrags * 64 + sizeof(float) * fl.aux_floats;
    const int64_t nblk = (N + kVoxPerPass - 1) / kVoxPerPass;
    const int grid = (int)(nblk < ctx->num_cus ? nblk : ctx->num_cus);
#define QB_LAUNCH_FUSED(TT, LL)                                                                              \
    do {                                                                                                     \
        auto k = wide_fused_kernel<TT, LL>;                                                                  \
        QB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)smem));                                                              \
        hipLaunchKernelGGL(k, dim3(grid), dim3(kFB), smem, s, a);                                            \
    } while (0)
#ifndef QB_FUSED_DEV
    if (fl.TT == 1 && shape->L == 1) QB_LAUNCH_FUSED(1, 1);
    else if (fl.TT == 1 && shape->L == 2) QB_LAUNCH_FUSED(1, 2);
    else if (fl.TT == 4 && shape->L == 1) QB_LAUNCH_FUSED(4, 1);
    else
#endif
    if (fl.TT == 4 && shape->L == 2) QB_LAUNCH_FUSED(4, 2);
    else {
        qb::set_error("wide_fused_fwd: shape not instantiated");
        return QBOLD_ERR_UNSUPPORTED;
    }
#undef QB_LAUNCH_FUSED
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}
}  // namespace qb

extern "C" int64_t qbold_encoder_fused_packed_floats(const qbold_encoder_shape* s) {
    if (!fused_supported(s)) return QBOLD_ERR_UNSUPPORTED;
    return make_fused_layout(s->T, s->U, s->L).total_floats;
}

extern "C" int qbold_encoder_fused_pack(const qbold_ctx* ctx, const qbold_encoder_shape* shape,
                                        const float* weights, float* packed, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (!fused_supported(shape)) {
        qb::set_error("qbold_encoder_fused_pack: the one-launch wide encoder is built for U = 256, L = 1 or 2, "
                      "T <= 16 or 49 <= T <= 64, channel-wise gating, QBOLD_ENC_F32");
        return QBOLD_ERR_UNSUPPORTED;
    }
    QB_REQUIRE(shape->T == ctx->dev.T, "qbold_encoder_fused_pack: encoder shape T differs from the context's tau grid");
    QB_REQUIRE(weights && packed && reinterpret_cast<uintptr_t>(packed) % 16 == 0,
               "qbold_encoder_fused_pack: null or misaligned buffer");
    const int T = shape->T, U = shape->U, L = shape->L;
    const FusedLayout fl = make_fused_layout(T, U, L);
    const qb::CanonLayout c = qb::make_canon(T, U, L, shape->channelwise_gating, shape->spatial_taps);
    hipStream_t s = (hipStream_t)stream;
    int frag = 0, op = 0;
    auto pack = [&](int KS, int MT, int korder, const float* W, const float* b, int nin, int nout, const float* W2,
                    const float* b2, int nout2, int split_row, float add, int bias_off) {
        float* scale = packed + fl.img_floats + fl.scale_off + 2 * op;
        hipLaunchKernelGGL(fused_scale_kernel, dim3(1), dim3(256), 0, s, W, (int64_t)nin * nout, W2,
                           W2 ? (int64_t)nin * nout2 : 0, scale);
        const int64_t n = (int64_t)MT * KS * 2 * 512 + 16 * MT;
        hipLaunchKernelGGL(fused_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, frag, KS, MT, korder,
                           W, b, nin, nout, W2, b2, nout2, split_row, add, bias_off, scale, fl.img_floats, packed);
        frag += MT * KS * 2;
        ++op;
    };
    pack(fl.KS1, fl.MT, 0, weights + c.W0, weights + c.b0, T, U, nullptr, nullptr, 0, 0, 0.0f, fl.bias_first);
    const int ctr = c.taps == 9 ? 4 * U * U : 0;  // voxel batches see the centre tap of the 3x3x1 kernels
    for (int l = 0; l < L; ++l) {
        const float* wb = weights + c.blk0 + (int64_t)l * c.blk_stride;
        const int b0 = fl.bias_blk0 + l * 4 * U;
        pack(fl.KS, fl.MT, 1, wb + c.Wc, wb + c.bc, U, U, nullptr, nullptr, 0, 0, 0.0f, b0);
        pack(fl.KS, fl.MT, 1, wb + c.Wr1 + ctr, wb + c.br1, U, U, nullptr, nullptr, 0, 0, 0.0f, b0 + U);
        pack(fl.KS, fl.MT, 1, wb + c.Wr2 + ctr, wb + c.br2, U, U, nullptr, nullptr, 0, 0, 0.0f, b0 + 2 * U);
        pack(fl.KS, fl.MT, 1, wb + c.Wg, wb + c.bg, U, U, nullptr, nullptr, 0, 0, shape->gate_offset, b0 + 3 * U);
    }
    pack(fl.KS, fl.HT, 1, weights + c.Ws, weights + c.bs, U, T, weights + c.Wf, weights + c.bf, 5, 16 * fl.TT, 0.0f,
         fl.bias_head);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_encoder_fused_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* packed,
                                       const float* x, float* out_q, float* out_log_sigma, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (!fused_supported(shape)) {
        qb::set_error("qbold_encoder_fused_fwd: shape outside the one-launch wide encoder (see qbold_encoder_fused_pack)");
        return QBOLD_ERR_UNSUPPORTED;
    }
    QB_REQUIRE(shape->T == ctx->dev.T, "qbold_encoder_fused_fwd: encoder shape T differs from the context's tau grid");
    QB_REQUIRE(N >= 0, "qbold_encoder_fused_fwd: negative N");
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(packed && x && out_q && out_log_sigma, "qbold_encoder_fused_fwd: null buffer");
    QB_REQUIRE(reinterpret_cast<uintptr_t>(packed) % 16 == 0 && (shape->T % 4 != 0 || reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                   (shape->T % 4 != 0 || reinterpret_cast<uintptr_t>(out_log_sigma) % 16 == 0),
               "qbold_encoder_fused_fwd: packed image, x and out_log_sigma must be 16-byte aligned");
    return qb::wide_fused_fwd(ctx, shape, packed, x, out_q, out_log_sigma, N, (hipStream_t)stream);
}
