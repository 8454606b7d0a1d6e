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
// register file.  A wave owns 32 voxels = the N side of v_mfma_f32_32x32x16_f16; every dense layer is computed
// transposed (weights = A operand, activations = B operand) so that the 32x32 accumulator layout (voxel on
// lane & 31; register r of lane half h = lane >> 5 holds unit 8 (r >> 2) + 4 h + (r & 3) of the tile) IS the next
// layer's B operand (k-slot j of lane half h) if the weight image is stored in that k order: registers
// 8 t .. 8 t + 7 of output tile M are the eight k-slots of k-step 2 M + t.  A 256-unit activation panel of 32
// voxels is kept as ready-made B operands -- split f16 halves hi / lo of every value, 16 k-steps x (4 + 4)
// VGPRs = 128 VGPRs -- and a gated block needs three panels (b / relu b, skip, t -> r), 384 VGPRs, plus two
// output tiles' accumulators (16 each).  The loop nest is output tile outermost: the B operands of all k-steps
// are register-resident, so a tile's accumulator lives for 48 MFMAs only and its result goes straight into
// the next panel (bias, activation, split).
//
// Why 32x32x16.  With one wave per SIMD the epilogue's vector instructions have to issue in the shadow of this
// wave's own MFMAs.  A 16x16x32 MFMA (16 cycles) leaves 8 cycles of vector issue -- two instructions, and a
// dependent pair already overruns it; measured, that form of this kernel ran at 2.2x its matrix-pipe time with
// the vector pipe and the matrix pipe taking turns.  A 32x32x16 MFMA does the same work per cycle in 32-cycle
// pieces and leaves 24 cycles: the ~3.4 vector instructions per MFMA of this kernel fit.
//
// Arithmetic: x = hi + lo with hi = f16(x), lo = f16(x - hi) UNSCALED, so that all three products (hi.hi, hi.lo,
// lo.hi) add into ONE accumulator and the epilogue has nothing to combine; weights are pre-multiplied per dense
// op by a power of two that puts max |w| in [2^12, 2^13) (their lo halves stay normal f16 numbers down to
// |w| = 2^-15 max |w|), undone for free in the epilogue's y = acc 2^-e + bias.  An activation's lo half is a
// normal number for |x| >= 2^-3 and carries an absolute error <= 2^-25 below that; |x| beyond 65504 overflows hi.
//
// Weights.  All dense ops of one pass over 128 voxels form one flat stream of 1 KiB MFMA fragments in
// consumption order ([op][out tile][k-step][hi, lo]; 2.2 MB for config 3, L2-resident).  Stages of 16
// fragments travel L2 -> LDS by LDS-direct loads (global_load_lds_dwordx4, each wave issues a quarter) into a
// ring of eight 16 KiB slots, four stages ahead; the handshake sits in the MIDDLE of a stage -- counted
// s_waitcnt vmcnt, raw s_barrier, issue of the stage that reuses the slot everyone has just left -- so the
// next stage is known to have landed before the current one ends and the fragment reads run two pairs ahead of
// the MFMAs without a bubble at tile, stage or pass boundaries.  The stream wraps from one pass to the next
// (same weights), so the ring never drains.  Biases and per-op scales live in LDS for the whole launch.
//
// Per pass and wave: 3,312 MFMAs (32 cycles each) = 106 k cycles of matrix pipe for 32 voxels; at N = 1 M that
// is 32 passes per CU = 1.41 ms at 2.4 GHz -- the floor this kernel is measured against.
#include "canon_layout.h"
#include "encoder_core.h"
#include "qbold_ctx.h"
#include "wide_common.h"

namespace {

using namespace qbw;

#ifndef QB_FUSED_VALU_PER_MFMA
#define QB_FUSED_VALU_PER_MFMA 5
#endif
constexpr int kFB = 256;          // threads per block: one wave per SIMD
constexpr int kRing = 8;          // LDS ring slots of 16 fragments (128 KiB)
constexpr int kAhead = 4;         // stages requested beyond the one being read (16 loads in flight per wave: the
                                  // 6-bit vmcnt also has to hold a pass's signal loads and head stores)
constexpr int kStageFrags = 16;   // 1 KiB fragments per stage
constexpr int kVoxPerPass = 128;  // 4 waves x 32 voxels
#ifndef QB_FUSED_PAIRS_AHEAD
#define QB_FUSED_PAIRS_AHEAD 1
#endif
constexpr int kPairsAhead = QB_FUSED_PAIRS_AHEAD;  // fragment pairs requested ahead of the MFMAs (1 or 2; two cost
                                                   // eight more VGPRs, which three panels do not leave)

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define QB_MFMA32_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// ---- fused image layout (host + device) ------------------------------------------------------------
struct FusedLayout {
    int T, U, L;
    int KS1, KS, MT, LT, HT;  // first-layer k-steps (16 deep), body k-steps, body tiles (32 rows), log-sigma tiles, head tiles
    int frags_first, frags_op, frags_head, frags_pass, stages_pass;
    int bias_first, bias_blk0, bias_head, bias_total;  // float offsets inside the bias image
    int scale_off, n_ops, aux_floats;                   // per-op (2^e, 2^-e) pairs behind the biases
    int64_t img_floats, total_floats;                   // weight fragments, + biases and scales
};
__host__ __device__ constexpr inline FusedLayout make_fused_layout(int T, int U, int L) {
    FusedLayout f{};
    f.T = T; f.U = U; f.L = L;
    f.KS1 = (T + 15) / 16;
    f.KS = U / 16;
    f.MT = U / 32;
    f.LT = (T + 31) / 32;
    f.HT = f.LT + 1;
    f.frags_first = f.MT * f.KS1 * 2;
    f.frags_op = f.MT * f.KS * 2;
    f.frags_head = f.HT * f.KS * 2;
    f.frags_pass = f.frags_first + 4 * L * f.frags_op + f.frags_head;
    f.stages_pass = (f.frags_pass + kStageFrags - 1) / kStageFrags;
    f.bias_first = 0;
    f.bias_blk0 = U;
    f.bias_head = U + 4 * L * U;
    f.bias_total = f.bias_head + 32 * f.HT;
    f.scale_off = (f.bias_total + 3) & ~3;
    f.n_ops = 2 + 4 * L;  // first layer, four per block, heads
    f.aux_floats = (f.scale_off + 2 * f.n_ops + 3) & ~3;
    f.img_floats = (int64_t)f.stages_pass * kStageFrags * 256;
    f.total_floats = f.img_floats + f.aux_floats;
    return f;
}
inline bool fused_supported(const qbold_encoder_shape* s) {
    // the unrolled kernels below are instantiated for these shapes (T <= 16 or 49 .. 64 taus: one or four
    // first-layer k-steps); anything else keeps the layer-wise path
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

// One dense op into the fused image: fragment (tile m, k-step s, part) holds, for lane (h = lane >> 5,
// i = lane & 31) and k-slot j, W[in][out = 32 m + i] with in = 16 s + 8 h + j (korder 0: the first layer, rows of
// x) or in = 16 s + 8 (j >> 2) + 4 h + (j & 3) (korder 1: the accumulator order, see the header).  Head
// (W2 != null): output row < split_row -> log-sigma row of W (Ws), else q row (row - split_row) of W2 (Wf).
// Weights are stored times scale[0] as hi = f16(w'), lo = f16(w' - hi); biases unscaled, but for the gate ops:
// theirs are stored as -log2(e) (b + gate_offset), the form the sigmoid's v_exp_f32 takes (epi_elem).
__global__ void fused_pack_kernel(int frag0, int KS, int MT, int korder, const float* __restrict__ W,
                                  const float* __restrict__ b, int nin, int nout, const float* __restrict__ W2,
                                  const float* __restrict__ b2, int nout2, int split_row, float bias_add, float bias_mul,
                                  int bias_off, const float* __restrict__ scale, int64_t img_floats,
                                  float* __restrict__ packed) {
    _Float16* ph = reinterpret_cast<_Float16*>(packed) + (int64_t)frag0 * 512;
    const int64_t halves = (int64_t)MT * KS * 2 * 512;
    const float sc = scale[0];
    for (int64_t h = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; h < halves + 32 * MT;
         h += (int64_t)gridDim.x * blockDim.x) {
        if (h >= halves) {  // bias rows in the same (possibly remapped) order
            const int row = (int)(h - halves);
            float v = 0.0f;
            if (W2) {
                if (row < split_row) v = row < nout ? b[row] : 0.0f;
                else v = row - split_row < nout2 ? b2[row - split_row] : 0.0f;
            } else if (row < nout) {
                v = (b[row] + bias_add) * bias_mul;
            }
            packed[img_floats + bias_off + row] = v;
            continue;
        }
        const int j = (int)(h & 7), lane = (int)((h >> 3) & 63), part = (int)((h >> 9) & 1);
        const int64_t pair = h >> 10;  // m * KS + s
        const int s = (int)(pair % KS), m = (int)(pair / KS);
        const int hh = lane >> 5, i = lane & 31;
        const int in = korder ? 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3) : 16 * s + 8 * hh + j;
        const int out = 32 * m + i;
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
// B operands of every k-step of one activation tensor, for this wave's 32 voxels, as packed f16 pairs:
// dword d of k-step s holds k-slots 2 d, 2 d + 1.  (Kept as opaque 32-bit values: left as f16x8 vectors filled
// element by element, the compiler carries every half in a register of its own until the MFMA.)
template <int KS>
struct Panel {
    uint32_t hi[KS][4], lo[KS][4];
};
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
template <int KS>
__device__ __forceinline__ f16x8 frag_hi(const Panel<KS>& P, int s) {
    return __builtin_bit_cast(f16x8, u32x4v{P.hi[s][0], P.hi[s][1], P.hi[s][2], P.hi[s][3]});
}
template <int KS>
__device__ __forceinline__ f16x8 frag_lo(const Panel<KS>& P, int s) {
    return __builtin_bit_cast(f16x8, u32x4v{P.lo[s][0], P.lo[s][1], P.lo[s][2], P.lo[s][3]});
}
// x = hi + lo, hi = f16(x), lo = f16(x - hi), for a pair of values: one packed conversion for the hi halves,
// and each lo half straight from a mixed-precision FMA (x - hi is exact in float32; the FMA rounds it to f16)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo, float& amax) {
    amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b)));  // v_max3_f32: operand range guard (encoder_core.h)
    asm volatile("" : "+v"(amax));  // here and now: left free, the compiler parks every value in scratch and
                                    // evaluates the whole max tree where amax is read, at the head
    uint32_t h = __builtin_bit_cast(uint32_t, qb::f16x2{(_Float16)a, (_Float16)b}), l;
    asm volatile("" : "+v"(h));  // one packed register from here on
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
    hi = h;
    lo = l;
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
// Mid-stage handshake: my quarter of the NEXT stage has landed (all but the youngest kAhead - 2 stages' loads
// are done), everyone has left the previous stage, whose slot the new issue overwrites.
// EXTRA: vector-memory instructions other than the stream's own that are certain to have been issued after the
// awaited stage's loads (sync_extras below) -- vmcnt counts loads and stores alike and retires them in order,
// so without the allowance a sync soon after the signal loads of the next pass (HBM latency) or the head's
// stores would wait for those as well.
template <int EXTRA>
__device__ __forceinline__ void stream_sync(Stream& S) {
    static_assert(4 * (kAhead - 2) + EXTRA < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kAhead - 2) + EXTRA) : "memory");
    __builtin_amdgcn_s_barrier();
    stream_issue(S);
}
__device__ __forceinline__ void stream_next_stage(Stream& S) {
    S.read_slot = (S.read_slot + 1) & (kRing - 1);
    S.cur = S.ring_lds + (uint32_t)S.read_slot * (kStageFrags * 1024);
}

// The sync in the middle of stage n needs stage n + 1, which the sync of stage n - 3 requested (kAhead = 4):
// whatever else was issued since then may still be in flight.  Counted for the V4 kernels (four first-layer
// k-steps, T % 4 == 0: two log-sigma tiles, one q tile; NS stages per pass, the head takes the last six), whose
// head issues, in program order and fenced by the syncs' memory clobbers,
//   11 signal loads | tile 0: sync NS-6, sync NS-5 | tile 1: store, sync NS-4, 2 stores, sync NS-3, store |
//   tile 2: store, sync NS-2, 2 stores, sync NS-1, store | 4 q stores | next pass
// (the 16-byte log-sigma stores of tile M leave in steps 3, 7, 11, 15 of tile M + 1, the syncs sit in front of
// steps 4 and 12).  The next pass starts by waiting for its signals -- at the loop head the compiler makes that
// a full drain -- so its first syncs have nothing of the head's behind them and take no allowance.
// scripts/check_vmcnt_ring.py replays the kernel's ISA against this table (tests/test_host.py: every allowance
// is tight, one more at any site is caught).
__host__ __device__ constexpr inline int sync_extras(bool v4, int n, int NS) {
    static_assert(kAhead == 4, "the table counts three syncs back");
#if defined(QB_X_NO_LS_STORE) || defined(QB_X_GLOBAL_STORE)
    return 0;  // timing / debugging variants issue other stores
#endif
    if (!v4) return 0;
    if (n == NS - 6 || n == NS - 5) return 11;
    if (n == NS - 4) return 12;
    if (n == NS - 3) return 3;
    if (n == NS - 2) return 5;
    if (n == NS - 1) return 6;
    return 0;
}

struct Frag {   // one (hi, lo) fragment pair of the weight stream
    u32x4 hi, lo;
};
template <int V>
struct IC {
    static constexpr int value = V;
};
__device__ __forceinline__ void lds_wait_all(u32x4& a, u32x4& b, u32x4& c) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void lds_wait_all(u32x4& a, u32x4& b, uint32_t& c) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
}
// every LDS read but the two youngest (the pair requested last) has returned
__device__ __forceinline__ void lds_wait_but2(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lds_wait_but2(u32x4& a, u32x4& b, u32x4& c) {
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void lds_wait_but2(u32x4& a, u32x4& b, uint32_t& c) {
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b), "+v"(c));
}
// biases requested one k-step ahead travel as a pair of plain variables: one value (sixteen-step tiles) or four
// (four-step tiles)
// ... together with the lane's operand range guard (largest activation split so far, encoder_core.h)
#define QB_BIAS_REFS uint32_t &bias, u32x4 &bias4, float &amax
#define QB_BIAS_ARGS bias, bias4, amax
template <int OFF>
__device__ __forceinline__ uint32_t lds_read4(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read_b32 immediate offset is 16 bits");
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF>
__device__ __forceinline__ float lds_read4_now(uint32_t addr) {
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "n"(OFF));
    return __builtin_bit_cast(float, r);
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

// unit 32 M + 8 (E >> 2) + 4 h + (E & 3) of the tile: float offset of its bias relative to the lane half's base
__host__ __device__ constexpr inline int bias_row(int E) { return 8 * (E >> 2) + (E & 3); }

// k-steps S_ .. KSOP-1 of one 32-row output tile.  wa holds the pair of step S_ (returned), wb the pair of step
// S_+1 (requested); the pair of step S_+2 is requested before the three MFMAs of step S_, and after them the
// wait leaves only that youngest pair outstanding.  The PREVIOUS tile's epilogue (prev: sixteen pieces, one
// per output value of a lane) is spread over the k-steps so that its vector instructions issue between this
// tile's MFMAs.  The biases of the pieces of step S_+1 are requested in step S_ (one ds_read_b32 with sixteen
// k-steps, one ds_read_b128 with four) -- PBOFF: bias offset of the previous tile, BOFF of this one, whose
// first pieces are requested in the last step -- and travel in `bias`.  oa / ob: the two pairs that follow the
// tile.  (Recursive template: every LDS offset is an instruction immediate.)
template <int KSOP, int F0, int FP, bool V4, int BOFF, int PBOFF, bool HAVE_PREV, int S_, class Prev>
__device__ __forceinline__ void tile_steps(Stream& S, uint32_t bias_lds, Frag wa, Frag wb, Frag& oa, Frag& ob,
                                           const Panel<KSOP>& in, f32x16& acc, QB_BIAS_REFS, const Prev& prev) {
    constexpr int PPS = 16 / KSOP;  // epilogue pieces per k-step
    constexpr bool AHEAD = PPS <= 4;
    if constexpr (S_ < KSOP) {
        if constexpr ((F0 + 2 * S_) % kStageFrags == kStageFrags / 2)
            stream_sync<sync_extras(V4, (F0 + 2 * S_) / kStageFrags, FP / kStageFrags)>(S);
        // NB a register an asm LDS read is still writing must reach its wait untouched: whole variables only
        // (a read into one component of a vector makes the compiler copy the pending register into the tuple)
        uint32_t nbias = bias;
        u32x4 nbias4 = bias4;
        if constexpr (AHEAD) {
            constexpr bool last = S_ + 1 == KSOP;
            constexpr int E0 = last ? 0 : PPS * (S_ + 1);           // first piece of the step the bias is for
            constexpr int off = ((last ? BOFF : PBOFF) + bias_row(E0)) * 4;
            if constexpr (last || HAVE_PREV) {
                if constexpr (PPS == 1) nbias = lds_read4<off>(bias_lds);
                else nbias4 = lds_read16<off>(bias_lds);  // pieces E0 .. E0 + 3 are four consecutive rows
            }
        }
        Frag n;
        frag_fetch<F0 + 2 * S_ + 2 * kPairsAhead, FP>(S, n);
        // the reads are in flight BEFORE this step's MFMAs and waited for AFTER them: nothing crosses either fence
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 h = __builtin_bit_cast(f16x8, wa.hi), l = __builtin_bit_cast(f16x8, wa.lo);
        const f16x8 bh = frag_hi(in, S_), bl = frag_lo(in, S_);
        acc = QB_MFMA32_F16(h, bh, acc);
        acc = QB_MFMA32_F16(h, bl, acc);
        acc = QB_MFMA32_F16(l, bh, acc);
        if constexpr (HAVE_PREV) {
            if constexpr (PPS == 1) {
                prev(IC<S_>{}, __builtin_bit_cast(float, bias));
            } else if constexpr (PPS == 4) {
                const float4 b4 = __builtin_bit_cast(float4, bias4);
                prev(IC<4 * S_ + 0>{}, b4.x);
                prev(IC<4 * S_ + 1>{}, b4.y);
                prev(IC<4 * S_ + 2>{}, b4.z);
                prev(IC<4 * S_ + 3>{}, b4.w);
            } else {  // a single k-step (T <= 16): biases read where they are used
                prev(IC<0>{}, 0.0f); prev(IC<1>{}, 0.0f); prev(IC<2>{}, 0.0f); prev(IC<3>{}, 0.0f);
                prev(IC<4>{}, 0.0f); prev(IC<5>{}, 0.0f); prev(IC<6>{}, 0.0f); prev(IC<7>{}, 0.0f);
                prev(IC<8>{}, 0.0f); prev(IC<9>{}, 0.0f); prev(IC<10>{}, 0.0f); prev(IC<11>{}, 0.0f);
                prev(IC<12>{}, 0.0f); prev(IC<13>{}, 0.0f); prev(IC<14>{}, 0.0f); prev(IC<15>{}, 0.0f);
            }
        }
        // issue order inside the step: one MFMA, then a share of the epilogue's vector instructions (which
        // issue while the matrix pipe works on it), three times over
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, QB_FUSED_VALU_PER_MFMA, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kPairsAhead == 2) {
            if constexpr (PPS == 1) lds_wait_but2(wb.hi, wb.lo, nbias);
            else if constexpr (PPS == 4) lds_wait_but2(wb.hi, wb.lo, nbias4);
            else lds_wait_but2(wb.hi, wb.lo);
        } else {  // one pair ahead: everything requested has to be back
            if constexpr (PPS == 1) lds_wait_all(n.hi, n.lo, nbias);
            else if constexpr (PPS == 4) lds_wait_all(n.hi, n.lo, nbias4);
            else lds_wait(n.hi, n.lo);
        }
        bias = nbias;
        bias4 = nbias4;
        if constexpr (kPairsAhead == 2)
            tile_steps<KSOP, F0, FP, V4, BOFF, PBOFF, HAVE_PREV, S_ + 1>(S, bias_lds, wb, n, oa, ob, in, acc, QB_BIAS_ARGS, prev);
        else
            tile_steps<KSOP, F0, FP, V4, BOFF, PBOFF, HAVE_PREV, S_ + 1>(S, bias_lds, n, n, oa, ob, in, acc, QB_BIAS_ARGS, prev);
    } else {
        oa = wa;
        ob = wb;
    }
}

// One output tile into acc.  F0: fragment index within the pass of the tile's first fragment; BOFF / PBOFF:
// float offsets of this / the previous tile's bias rows in the LDS bias image.  wa / wb: in, the tile's first
// two pairs; out, the two pairs that follow it.
template <int KSOP, int F0, int FP, bool V4, int BOFF, int PBOFF, bool HAVE_PREV, class Prev>
__device__ __forceinline__ void tile_mma(Stream& S, uint32_t bias_lds, Frag& wa, Frag& wb, const Panel<KSOP>& in,
                                         f32x16& acc, QB_BIAS_REFS, const Prev& prev) {
    static_assert((BOFF + 32) * 4 < 65536, "bias offset is a ds_read immediate");
    __builtin_amdgcn_sched_barrier(0);  // tiles are scheduled one at a time: three panels leave no slack
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    tile_steps<KSOP, F0, FP, V4, BOFF, PBOFF, HAVE_PREV, 0>(S, bias_lds, wa, wb, wa, wb, in, acc, QB_BIAS_ARGS, prev);
}

// value E (register E of the accumulator = unit 32 M + 8 (E >> 2) + 4 h + (E & 3)) read back from a panel
template <int M, int E, int KS>
__device__ __forceinline__ float get_elem(const Panel<KS>& P) {
    // hi + lo in one v_fma_mix_f32 reading the f16 halves in place (the compiler's own version: two conversions
    // and an add; it also kept the float32 values of the hi halves alive from the op that wrote the panel to this
    // one -- a second, spilled copy of the panel -- which the opaque asm rules out)
    const uint32_t dh = P.hi[2 * M + E / 8][(E % 8) / 2], dl = P.lo[2 * M + E / 8][(E % 8) / 2];
    float r;
    if constexpr (E % 2 == 0) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(dl), "v"(dh));
    else asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(dl), "v"(dh));
    return r;
}

// relu on a split panel, in place (the Activation in front of the first 3x3x1 convolution, model.py:151):
// the sign of hi + lo is the sign bit of hi (|lo| is at most half an ulp of hi; x tiny negative gives hi = -0)
typedef short s16x2 __attribute__((ext_vector_type(2)));
template <int KS>
__device__ __forceinline__ void relu_panel(Panel<KS>& P) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const s16x2 neg = __builtin_bit_cast(s16x2, P.hi[s][d]) >> (short)15;  // 0xffff where negative
            const uint32_t keep = ~__builtin_bit_cast(uint32_t, neg);
            P.hi[s][d] &= keep;
            P.lo[s][d] &= keep;
        }
}

enum { EPI_RELU = 0, EPI_LINEAR = 1, EPI_GATE = 2 };

// Piece E of tile M's epilogue: y = acc[E] 2^-e + bias; the even piece of a pair parks its value in `carry`, the
// odd one splits and stores the pair.  EPI_GATE: out = skip (1 - g) + r g, g = sigmoid(W r + b) (model.py:164-170).
template <int EPI, int M, int E, int KS>
__device__ __forceinline__ void epi_elem(const f32x16& acc, float inv_scale, float bias, Panel<KS>& out,
                                         const Panel<KS>& skip, const Panel<KS>& rr, float& carry, float& amax) {
    float y = fmaf(acc[E], inv_scale, bias);
    if constexpr (EPI == EPI_RELU) y = fmaxf(y, 0.0f);
    if constexpr (EPI == EPI_GATE) {  // here inv_scale and bias carry the factor -log2(e): y = -log2(e) (W r + b)
        const float sk = get_elem<M, E>(skip), r = get_elem<M, E>(rr);
        const float gate = qb::rcpf_(1.0f + qb::exp2f_(y));  // sigmoid, model.py:169
        y = fmaf(gate, r - sk, sk);                           // skip (1 - g) + r g, model.py:170
    }
    if constexpr (E % 2 == 0) carry = y;
    else split_pair(carry, y, out.hi[2 * M + E / 8][(E % 8) / 2], out.lo[2 * M + E / 8][(E % 8) / 2], amax);
}

// the dense op's 2^-e (LDS aux image, float index IDX): wave-uniform
template <int IDX>
__device__ __forceinline__ float op_inv_scale(uint32_t aux_lds) {
    const float r = lds_read4_now<IDX * 4>(aux_lds);
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, r)));
}

// the sixteen biases of a lane for one tile (rows 8 q + 4 h + 0..3, q = 0..3), read and waited for at once
template <int BOFF>
__device__ __forceinline__ void last_tile_bias(uint32_t bias_lds, float (&b)[16]) {
    u32x4 r0 = lds_read16<(BOFF + 0) * 4>(bias_lds), r1 = lds_read16<(BOFF + 8) * 4>(bias_lds);
    u32x4 r2 = lds_read16<(BOFF + 16) * 4>(bias_lds), r3 = lds_read16<(BOFF + 24) * 4>(bias_lds);
    lds_wait(r0, r1, r2, r3);
    // whole-vector casts: __builtin_bit_cast(float, r[k]) of a vector element yields element 0 for every k here
    const float4 f[4] = {__builtin_bit_cast(float4, r0), __builtin_bit_cast(float4, r1), __builtin_bit_cast(float4, r2),
                         __builtin_bit_cast(float4, r3)};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        b[4 * q + 0] = f[q].x;
        b[4 * q + 1] = f[q].y;
        b[4 * q + 2] = f[q].z;
        b[4 * q + 3] = f[q].w;
    }
}

// A dense op: tiles M .. MT-1 of out = epi(W in + b).  Tile M-1's epilogue runs inside tile M's MFMA stream
// (two accumulators in rotation); the last tile's runs after the op.
template <int KS, int KSIN, int MT, int F0, int FP, bool V4, int BOFF, int EPI, int M = 0>
__device__ __forceinline__ void dense_op(Stream& S, uint32_t bias_lds, float inv_scale, Frag& wa, Frag& wb,
                                         const Panel<KSIN>& in, Panel<KS>& out, const Panel<KS>& skip,
                                         const Panel<KS>& rr, f32x16 (&acc)[2], QB_BIAS_REFS) {
    constexpr bool AHEAD = KSIN >= 4;  // as tile_steps: biases requested a k-step ahead
    if constexpr (M < MT) {
        constexpr int PM = M > 0 ? M - 1 : 0;
        float carry = 0.0f;
        auto prev = [&](auto c, float b) {
            constexpr int E = decltype(c)::value;
#ifndef QB_X_SYNC_BIAS
            if constexpr (!AHEAD)
#endif
                b = lds_read4_now<(BOFF + 32 * PM + bias_row(E)) * 4>(bias_lds);
            epi_elem<EPI, PM, E>(acc[(M + 1) & 1], inv_scale, b, out, skip, rr, carry, amax);
        };
        tile_mma<KSIN, F0 + M * 2 * KSIN, FP, V4, BOFF + 32 * M, BOFF + 32 * PM, (M > 0)>(S, bias_lds, wa, wb, in,
                                                                                       acc[M & 1], QB_BIAS_ARGS, prev);
        dense_op<KS, KSIN, MT, F0, FP, V4, BOFF, EPI, M + 1>(S, bias_lds, inv_scale, wa, wb, in, out, skip, rr, acc, QB_BIAS_ARGS);
    } else {
        __builtin_amdgcn_sched_barrier(0);
        float carry = 0.0f;
        float bl[16];
        last_tile_bias<BOFF + 32 * (MT - 1)>(bias_lds, bl);
        auto last = [&](auto c) {
            constexpr int E = decltype(c)::value;
            epi_elem<EPI, MT - 1, E>(acc[(MT - 1) & 1], inv_scale, bl[E], out, skip, rr, carry, amax);
        };
        last(IC<0>{}); last(IC<1>{}); last(IC<2>{}); last(IC<3>{}); last(IC<4>{}); last(IC<5>{}); last(IC<6>{});
        last(IC<7>{}); last(IC<8>{}); last(IC<9>{}); last(IC<10>{}); last(IC<11>{}); last(IC<12>{}); last(IC<13>{});
        last(IC<14>{}); last(IC<15>{});
    }
}

struct FusedArgs {
    const float* x;     // [N][T]
    const uint4* img;   // fused weight image
    const float* bias;  // bias image, then per-op scales (floats)
    float* q;           // [N][5]
    float* ls;          // [N][T] log sigma
    int64_t N;
    int T, se_idx, multi_norm;
    unsigned long long* stamps;  // diagnostic builds only (QB_FUSED_STAMP): s_memtime at op boundaries
    int* stamp_idx;
};
#ifdef QB_FUSED_STAMP
#define QB_STAMP(a, k)                                                                              \
    do {                                                                                            \
        if ((a).stamps && blockIdx.x == 0 && threadIdx.x == 0) {                                    \
            (a).stamps[(*(a).stamp_idx)++] = __builtin_amdgcn_s_memtime();                                 \
        }                                                                                           \
    } while (0)
#else
#define QB_STAMP(a, k) do {} while (0)
#endif

// Raw signals of this lane's voxel for the first layer's B operand: k-step s, lane half h holds taus
// 16 s + 8 h .. + 7; plus the images the normaliser needs (model.py:102-106).
template <int KS1>
struct XRaw {
    float f[KS1][8];
    float d[3];
};
template <int KS1, bool V4>
__device__ __forceinline__ void load_x(const FusedArgs& a, int64_t v, int h, XRaw<KS1>& xr) {
    // opaque: keeps the address arithmetic here, at the loads -- hoisted out of the pass loop its pieces sit in
    // registers through the whole pass, where three panels leave none to spare
    asm volatile("" : "+v"(v), "+v"(h));
    const int64_t vc = v < a.N ? v : a.N - 1;  // clamp: every lane takes part in the MFMAs (a.N >= 1)
    const float* xv = a.x + vc * a.T;
    const int se = a.se_idx;
    xr.d[1] = xv[se];
    xr.d[0] = xv[se > 0 ? se - 1 : 0];
    xr.d[2] = xv[se + 1 < a.T ? se + 1 : se];
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
        const int t0 = 16 * s + 8 * h;
        if (V4 || ((a.T & 3) == 0 && a.T >= 8)) {  // wave-uniform; V4 (four k-steps, T % 4 == 0): known here
            // each 16-byte half clamped into the row on its own: a half that starts inside the row is fetched from its
            // own address (T % 4 == 0: it then lies wholly inside), one beyond it fetches the row's last four taus and
            // is masked in convert_x -- no window ever has to be picked apart
            const int tl = t0 <= a.T - 4 ? t0 : a.T - 4, th = t0 + 4 <= a.T - 4 ? t0 + 4 : a.T - 4;
#ifdef QB_FUSED_X_NT
            const float4 lo4 = __builtin_nontemporal_load(reinterpret_cast<const float4*>(xv + tl));
            const float4 hi4 = __builtin_nontemporal_load(reinterpret_cast<const float4*>(xv + th));
#else
            const float4 lo4 = *reinterpret_cast<const float4*>(xv + tl);
            const float4 hi4 = *reinterpret_cast<const float4*>(xv + th);
#endif
            xr.f[s][0] = lo4.x; xr.f[s][1] = lo4.y; xr.f[s][2] = lo4.z; xr.f[s][3] = lo4.w;
            xr.f[s][4] = hi4.x; xr.f[s][5] = hi4.y; xr.f[s][6] = hi4.z; xr.f[s][7] = hi4.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) xr.f[s][j] = xv[t0 + j < a.T ? t0 + j : a.T - 1];
        }
    }
}
// normalise_data (model.py:97-113) and the split into B operands
template <int KS1>
__device__ __forceinline__ void convert_x(const FusedArgs& a, int h, const XRaw<KS1>& xr, Panel<KS1>& X, float& amax) {
    float den = qb::clampf_(xr.d[1], 1e-2f, 1e8f);
    if (a.multi_norm) den = (qb::clampf_(xr.d[0], 1e-2f, 1e8f) + den + qb::clampf_(xr.d[2], 1e-2f, 1e8f)) / 3.0f;
    const float inv_den = 1.0f / den;
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
        const int t0 = 16 * s + 8 * h;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)  // v_log_f32 as the LDS-resident kernels (encoder_core.h normalise); taus beyond the
                                     // row (whatever load_x fetched for them) are zero
            f[j] = t0 + j < a.T ? QB_LN2 * qb::log2f_(qb::clampf_(xr.f[s][j], 1e-2f, 1e8f) * inv_den) : 0.0f;
#pragma unroll
        for (int d = 0; d < 4; ++d) split_pair(f[2 * d], f[2 * d + 1], X.hi[s][d], X.lo[s][d], amax);
    }
}

// Where a pass's head outputs go: raw buffer resources over this pass's voxels only, so that the hardware drops
// the stores of lanes beyond the batch (their offsets lie outside the resource) -- no branch in the epilogue
// pieces, which therefore stay inside the MFMA stream's scheduling regions.
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct HeadOut {
    __amdgpu_buffer_rsrc_t ls, q;
    uint32_t ls_off, q_off;  // byte offsets of this lane's voxel inside the pass
    int T;
    bool overflow;           // an activation of this voxel left the f16 operand range: its outputs become NaN
#ifdef QB_X_GLOBAL_STORE
    float* pls; float* pq; int64_t vv, N;
#endif
};
#define QB_OOB 0x7ffffff0u  // beyond any resource: a dropped store
// Piece E of head tile M: tiles 0 .. LT-1 are log-sigma rows 32 M + 8 (E >> 2) + 4 h + (E & 3), tile LT holds the
// five q rows.
#ifndef QB_FUSED_LS_AUX
#define QB_FUSED_LS_AUX 0  // cache policy of the log-sigma stores.  nt (2) was worth 8 % while the ring's syncs still
                           // waited for these stores (sync_extras); since then it gains nothing here and costs the
                           // ELBO kernel, which reads log sigma next, its hits in the memory-side cache
#endif
// V4: T is a multiple of four, so the four registers E & ~3 .. | 3 of a lane (taus 8 (E >> 2) + 4 h + 0..3 of its
// voxel) are one aligned 16-byte piece of the voxel's row: one store instead of four, whole 32-byte sectors per
// lane pair.  hc parks the first three values of a group.
template <int LT, int M, int E, bool V4>
__device__ __forceinline__ void head_elem(const f32x16& acc, float inv_scale, float bias, const HeadOut& o, int h,
                                          float (&hc)[3]) {
    const float y = o.overflow ? __builtin_nanf("") : fmaf(acc[E], inv_scale, bias);
    const int row_in_tile = 8 * (E >> 2) + 4 * h + (E & 3);
#ifdef QB_X_GLOBAL_STORE
    if (o.vv < o.N) {
        if constexpr (M < LT) {
            if (32 * M + row_in_tile < o.T) o.pls[o.vv * o.T + 32 * M + row_in_tile] = y;
        } else if (row_in_tile < 5) o.pq[o.vv * 5 + row_in_tile] = y;
    }
    return;
#endif
    if constexpr (M < LT) {
#ifdef QB_X_NO_LS_STORE
        return;
#endif
        if constexpr (V4) {
            if constexpr ((E & 3) != 3) hc[E & 3] = y;
            else {
                const int row = 32 * M + row_in_tile - 3;
                const uint32_t off = row < o.T ? o.ls_off + 4u * row : QB_OOB;
                const u32x4 d = {__builtin_bit_cast(uint32_t, hc[0]), __builtin_bit_cast(uint32_t, hc[1]),
                                 __builtin_bit_cast(uint32_t, hc[2]), __builtin_bit_cast(uint32_t, y)};
                __builtin_amdgcn_raw_buffer_store_b128(d, o.ls, off, 0, QB_FUSED_LS_AUX);
            }
        } else {
            const int row = 32 * M + row_in_tile;
            const uint32_t off = row < o.T ? o.ls_off + 4u * row : QB_OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, y), o.ls, off, 0, QB_FUSED_LS_AUX);
        }
    } else if constexpr (E < 4) {  // rows 0-3 (h = 0) and 4 (h = 1, E = 0); registers 4 .. 15 hold padding rows
        const uint32_t off = row_in_tile < 5 ? o.q_off + 4u * row_in_tile : QB_OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, y), o.q, off, 0, 0);
    }
}
template <int KS, int LT, int F0, int FP, int BOFF, bool V4, int M = 0>
__device__ __forceinline__ void head_op(Stream& S, uint32_t bias_lds, float inv_scale, Frag& wa, Frag& wb,
                                        const Panel<KS>& b, const HeadOut& o, int h, f32x16 (&acc)[2], QB_BIAS_REFS) {
    if constexpr (M <= LT) {
        constexpr int PM = M > 0 ? M - 1 : 0;
        float hc[3] = {0.0f, 0.0f, 0.0f};
        auto prev = [&](auto c, float bb) {
#ifdef QB_X_SYNC_BIAS
            bb = lds_read4_now<(BOFF + 32 * PM + bias_row(decltype(c)::value)) * 4>(bias_lds);
#endif
            head_elem<LT, PM, decltype(c)::value, V4>(acc[(M + 1) & 1], inv_scale, bb, o, h, hc);
        };
        tile_mma<KS, F0 + M * 2 * KS, FP, V4, BOFF + 32 * M, BOFF + 32 * PM, (M > 0)>(S, bias_lds, wa, wb, b, acc[M & 1],
                                                                                   QB_BIAS_ARGS, prev);
        head_op<KS, LT, F0, FP, BOFF, V4, M + 1>(S, bias_lds, inv_scale, wa, wb, b, o, h, acc, QB_BIAS_ARGS);
    } else {
        __builtin_amdgcn_sched_barrier(0);
        float bl[16];
        float hc[3] = {0.0f, 0.0f, 0.0f};
        last_tile_bias<BOFF + 32 * LT>(bias_lds, bl);
        auto last = [&](auto c) {
            constexpr int E = decltype(c)::value;
            head_elem<LT, LT, E, V4>(acc[LT & 1], inv_scale, bl[E], o, h, hc);
        };
        last(IC<0>{}); last(IC<1>{}); last(IC<2>{}); last(IC<3>{}); last(IC<4>{}); last(IC<5>{}); last(IC<6>{});
        last(IC<7>{}); last(IC<8>{}); last(IC<9>{}); last(IC<10>{}); last(IC<11>{}); last(IC<12>{}); last(IC<13>{});
        last(IC<14>{}); last(IC<15>{});
    }
}

// Gated residual blocks LB .. L-1 (model.py:147-172): b comes in P0 and leaves in P1; the panels rotate by
// name from block to block, nothing is copied.  SC0: float index of the first block op's 2^-e in the aux image.
template <int KS, int KS1, int MT, int L, int LT, int FB0, int FP, int SC0, bool V4, int LB>
__device__ __forceinline__ void blocks_and_head(Stream& S, uint32_t bias_lds, uint32_t aux_lds, Frag& wa, Frag& wb,
                                                Panel<KS>& P0, Panel<KS>& P1, Panel<KS>& P2, const FusedArgs& a,
                                                int64_t v, int64_t v_next, int h, f32x16 (&acc)[2], QB_BIAS_REFS,
                                                XRaw<KS1>& xr) {
    constexpr int FOP = MT * KS * 2;  // fragments per dense op
    constexpr int U = 32 * MT;
    if constexpr (LB < L) {
        constexpr int F = FB0 + LB * 4 * FOP, B = U + LB * 4 * U, SC = SC0 + 8 * LB;
        dense_op<KS, KS, MT, F, FP, V4, B, EPI_RELU>(S, bias_lds, op_inv_scale<SC>(aux_lds), wa, wb, P0, P2, P2, P2, acc,
                                                 QB_BIAS_ARGS);                                            // skip, :148
        QB_STAMP(a, 0);
        if constexpr (LB > 0) relu_panel(P0);  // block 0's input is a relu output already            :151
        dense_op<KS, KS, MT, F + FOP, FP, V4, B + U, EPI_RELU>(S, bias_lds, op_inv_scale<SC + 2>(aux_lds), wa, wb, P0, P1,
                                                           P1, P1, acc, QB_BIAS_ARGS);                      // t, :152-155
        QB_STAMP(a, 1);
        dense_op<KS, KS, MT, F + 2 * FOP, FP, V4, B + 2 * U, EPI_LINEAR>(S, bias_lds, op_inv_scale<SC + 4>(aux_lds), wa, wb,
                                                                     P1, P0, P0, P0, acc, QB_BIAS_ARGS);    // r, :156
        QB_STAMP(a, 2);
        dense_op<KS, KS, MT, F + 3 * FOP, FP, V4, B + 3 * U, EPI_GATE>(S, bias_lds, -QB_LOG2E * op_inv_scale<SC + 6>(aux_lds), wa, wb,
                                                                   P0, P1, P2, P0, acc, QB_BIAS_ARGS);      // :164-170
        QB_STAMP(a, 3);
        blocks_and_head<KS, KS1, MT, L, LT, FB0, FP, SC0, V4, LB + 1>(S, bias_lds, aux_lds, wa, wb, P1, P0, P2, a, v, v_next,
                                                                  h, acc, QB_BIAS_ARGS, xr);
    } else {
        load_x<KS1, V4>(a, v_next, h, xr);  // the next pass's signals (clamped beyond the batch: then unused)
        HeadOut o;
        {
            const int64_t v0 = __builtin_amdgcn_readfirstlane((int)(v >> 7)) * (int64_t)kVoxPerPass;  // pass's first voxel
            const int64_t left = a.N - v0;
            const uint32_t nv = (uint32_t)(left < kVoxPerPass ? left : kVoxPerPass);
            o.ls = __builtin_amdgcn_make_buffer_rsrc(a.ls + v0 * a.T, 0, nv * (uint32_t)a.T * 4u, 0x00020000);
            o.q = __builtin_amdgcn_make_buffer_rsrc(a.q + v0 * 5, 0, nv * 20u, 0x00020000);
            const uint32_t local = (uint32_t)(v - v0);
            o.ls_off = local * (uint32_t)a.T * 4u;
            o.q_off = local * 20u;
            o.T = a.T;
            // (the head's own input was split by the last gate op: amax is complete here)
            o.overflow = qb::split_overflowed(fmaxf(amax, __shfl_xor(amax, 32, 64)));
#ifdef QB_X_GLOBAL_STORE
            o.pls = a.ls; o.pq = a.q; o.vv = v; o.N = a.N;
#endif
        }
        head_op<KS, LT, FB0 + L * 4 * FOP, FP, U + L * 4 * U, V4>(S, bias_lds, op_inv_scale<SC0 + 8 * L>(aux_lds), wa, wb,
                                                              P0, o, h, acc, QB_BIAS_ARGS);
        QB_STAMP(a, 4);
    }
}

// TT: first-layer k-steps (ceil(T / 16): 1 or 4); V4: T % 4 == 0, log sigma leaves in 16-byte stores
template <int TT, int L, bool V4>
__global__ __launch_bounds__(kFB) void wide_fused_kernel(FusedArgs a) {
#ifdef QB_FUSED_STAMP
    int stamp_counter = 0;
    a.stamp_idx = &stamp_counter;
#endif
    constexpr int U = 256, KS = U / 16, MT = U / 32, KS1 = TT, LT = (TT + 1) / 2;
    constexpr FusedLayout fl = make_fused_layout(16 * TT, U, L);  // T only pads inside its tile
    constexpr int FP = fl.frags_pass;
    static_assert(FP % kStageFrags == 0, "a pass is a whole number of stages");
    static_assert(fl.KS1 == KS1 && fl.LT == LT, "layout and kernel agree on the tile counts");
    extern __shared__ __align__(16) uint4 smem[];
    uint4* ring = smem;                                                      // [kRing][16][64]
    float* lbias = reinterpret_cast<float*>(smem + kRing * kStageFrags * 64);  // biases, then per-op scales
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, i = lane & 31;
    for (int k = threadIdx.x; k < fl.aux_floats; k += kFB) lbias[k] = a.bias[k];

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
    const uint32_t bias_lds = lds_addr(lbias) + 16u * h;  // rows 4 h .. 4 h + 3 of every group of eight
    const uint32_t aux_lds = lds_addr(lbias);
    constexpr int SC0 = fl.scale_off + 1;  // float index of the first layer's 2^-e; op k's is SC0 + 2 k
#pragma unroll
    for (int k = 0; k < kAhead; ++k) stream_issue(S);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kAhead - 1)) : "memory");  // stage 0 has landed (my quarter)
    __syncthreads();                                                          // ... everyone's; the biases too
    Frag wa, wb;  // the next fragment pair(s) of the stream
    wa.hi = lds_read16<0>(S.cur);
    wa.lo = lds_read16<1024>(S.cur);
    if constexpr (kPairsAhead == 2) {
        wb.hi = lds_read16<2048>(S.cur);
        wb.lo = lds_read16<3072>(S.cur);
        lds_wait(wa.hi, wa.lo, wb.hi, wb.lo);
    } else {
        lds_wait(wa.hi, wa.lo);
        wb = wa;
    }
    uint32_t bias = 0u;                    // bias of the next step's epilogue piece (requested one step ahead)
    u32x4 bias4 = u32x4{0u, 0u, 0u, 0u};   // ... of the next step's four pieces (the first layer)

    // The signals of a pass are requested one pass ahead, at the start of the previous pass's head op (one panel
    // live, registers to spare, ~2 us of matrix work to cover the HBM latency): branch-free loads with clamped
    // addresses, so that the compiler can count them (a load behind a branch made it wait for vmcnt(0), i.e.
    // for every LDS-direct load in flight as well -- ten times per pass).
    const int64_t nblk = (a.N + kVoxPerPass - 1) / kVoxPerPass;
    XRaw<KS1> xr;
    load_x<KS1, V4>(a, (int64_t)blockIdx.x * kVoxPerPass + wave * 32 + i, h, xr);
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t v = blk * kVoxPerPass + wave * 32 + i;
        Panel<KS1> X;
        QB_STAMP(a, 5);
        float amax = 0.0f;   // this pass's voxel: largest activation split (both lane halves are merged at the head)
        convert_x<KS1>(a, h, xr, X, amax);
        QB_STAMP(a, 6);
        Panel<KS> P0, P1, P2;
        f32x16 acc[2];
        dense_op<KS, KS1, MT, 0, FP, V4, 0, EPI_RELU>(S, bias_lds, op_inv_scale<SC0>(aux_lds), wa, wb, X, P0, P0, P0, acc,
                                                  QB_BIAS_ARGS);  // first layer, model.py:181
        bias4 = u32x4{0u, 0u, 0u, 0u};  // dead until the next pass's first layer
        QB_STAMP(a, 7);
        blocks_and_head<KS, KS1, MT, L, LT, MT * KS1 * 2, FP, SC0 + 2, V4, 0>(S, bias_lds, aux_lds, wa, wb, P0, P1, P2, a, v,
                                                                          v + (int64_t)gridDim.x * kVoxPerPass, h, acc,
                                                                          QB_BIAS_ARGS, xr);
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
    a.stamps = nullptr;
    a.stamp_idx = nullptr;
#ifdef QB_FUSED_STAMP
    if (const char* p = getenv("QBOLD_FUSED_STAMPS")) a.stamps = reinterpret_cast<unsigned long long*>(strtoull(p, nullptr, 0));
#endif
    const size_t smem = sizeof(uint4) * kRing * kStageFrags * 64 + sizeof(float) * fl.aux_floats;
    const int64_t nblk = (N + kVoxPerPass - 1) / kVoxPerPass;
    const int grid = (int)(nblk < ctx->num_cus ? nblk : ctx->num_cus);
#define QB_LAUNCH_FUSED(TT, LL, V4)                                                                          \
    do {                                                                                                     \
        auto k = wide_fused_kernel<TT, LL, V4>;                                                              \
        QB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)smem));                                                              \
        hipLaunchKernelGGL(k, dim3(grid), dim3(kFB), smem, s, a);                                            \
    } while (0)
    const bool v4 = shape->T % 4 == 0 && reinterpret_cast<uintptr_t>(out_log_sigma) % 16 == 0;
#ifndef QB_FUSED_DEV
    if (fl.KS1 == 1 && shape->L == 1) QB_LAUNCH_FUSED(1, 1, false);  // one log-sigma tile: nothing to gain
    else if (fl.KS1 == 1 && shape->L == 2) QB_LAUNCH_FUSED(1, 2, false);
    else if (fl.KS1 == 4 && shape->L == 1 && v4) QB_LAUNCH_FUSED(4, 1, true);
    else if (fl.KS1 == 4 && shape->L == 1) QB_LAUNCH_FUSED(4, 1, false);
    else if (fl.KS1 == 4 && shape->L == 2 && !v4) QB_LAUNCH_FUSED(4, 2, false);
    else
#endif
    if (fl.KS1 == 4 && shape->L == 2 && v4) QB_LAUNCH_FUSED(4, 2, true);
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
                    const float* b2, int nout2, int split_row, float add, int bias_off, float bias_mul = 1.0f) {
        float* scale = packed + fl.img_floats + fl.scale_off + 2 * op;
        hipLaunchKernelGGL(fused_scale_kernel, dim3(1), dim3(256), 0, s, W, (int64_t)nin * nout, W2,
                           W2 ? (int64_t)nin * nout2 : 0, scale);
        const int64_t n = (int64_t)MT * KS * 2 * 512 + 32 * MT;
        hipLaunchKernelGGL(fused_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, frag, KS, MT, korder,
                           W, b, nin, nout, W2, b2, nout2, split_row, add, bias_mul, bias_off, scale, fl.img_floats, packed);
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
        pack(fl.KS, fl.MT, 1, wb + c.Wg, wb + c.bg, U, U, nullptr, nullptr, 0, 0, shape->gate_offset, b0 + 3 * U,
             -QB_LOG2E);
    }
    pack(fl.KS, fl.HT, 1, weights + c.Ws, weights + c.bs, U, T, weights + c.Wf, weights + c.bf, 5, 32 * fl.LT, 0.0f,
         fl.bias_head);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_encoder_fused_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* packed,
                                       const float* x, float* out_q, float* out_log_sigma, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_RELU_ONLY(shape, "qbold_encoder_fused_fwd");
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
