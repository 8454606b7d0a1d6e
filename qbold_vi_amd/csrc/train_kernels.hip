// train_kernels.hip -- encoder training: forward with saved activations, backward (data and
// weight gradients), the pre-training loss gradient and the AdamW update.
//
// Reference: the Keras fit loops of train.py (:285-376 fine-tuning, :379-427 pre-training) with
// tfa.optimizers.AdamW; TensorFlow autodiff through create_encoder (model.py:122-223) for voxel
// batches (3x3x1 convolutions = their centre tap).
//
// Unlike the inference kernels these are plain row-major [N][64] float32 GEMMs on the exact-f32
// matrix instruction (v_mfma_f32_16x16x4_f32), one layer per launch, activations saved to HBM:
// training steps are 190-256 k voxels in the reference (train.py:68,103), ~1 GB of activations,
// and HBM-bound at a few ms per step -- off the voxel-ELBO headline path, correctness first.
#include <cmath>

#include "canon_layout.h"
#include "encoder_core.h"
#include "qbold_ctx.h"

namespace qb {
int check_encoder_shape(const qbold_ctx* ctx, const qbold_encoder_shape* s);
}  // namespace qb

namespace {

using qb::f32x4;
#define QB_MFMA16F(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Cross-workgroup sums that have no scratch to leave partials in (one double per quantity in the caller's buffer) are
// added as 64-bit FIXED-POINT integers: integer addition is associative, so the result does not depend on the order the
// workgroups arrive in -- a floating-point atomicAdd would make these the only sums of the library that are not
// reproducible bit for bit.  fixed_to_double_kernel turns the integers into doubles in place.
__device__ __forceinline__ void atomic_add_fixed(double* acc, double v, double scale) {
    atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)llrint(v * scale));
}
__global__ void fixed_to_double_kernel(double* p, int n, double inv_scale) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) p[k] = (double)(*reinterpret_cast<const long long*>(p + k)) * inv_scale;
}
constexpr double kTvFixed = 4294967296.0;      // 2^32: a TV sum is < 2^25 (N < 2^23 voxels, four unit differences each)
constexpr double kStatsFixed = 16777216.0;     // 2^24: |sum log v|, sum 1 / v < 2^35

constexpr int kLd = 64;        // row stride of the activation / delta tensors for U <= 64
constexpr int kMaxU = 256;     // forward-only layer-wise path (BASELINE config 3)
constexpr int kWs = 65;        // LDS row stride of a staged weight matrix

// bit 0: relu on the output; bit 1: relu on the INPUT rows as they are loaded (the reference's Activation
// layer in front of a convolution, model.py:151 -- no materialised relu(b) tensor)
// bits 2 / 3: the same with Keras' 'gelu' (exact: 0.5 x (1 + erf(x / sqrt 2)); activation_type of the reference's
// EncoderTrainer, model.py:60, 115-120) -- forward only, on the general kernels (xw_kernel, conv9_kernel)
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU_IN = 2, ACT_GELU = 4, ACT_GELU_IN = 8 };
__device__ __forceinline__ float gelu_(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// activation of an output value / of an input value as it is loaded
__device__ __forceinline__ float act_out(float y, int act) {
    if (act & ACT_RELU) return fmaxf(y, 0.0f);
    if (act & ACT_GELU) return gelu_(y);
    return y;
}
__device__ __forceinline__ float act_in(float a, int act) {
    if (act & ACT_RELU_IN) return fmaxf(a, 0.0f);
    if (act & ACT_GELU_IN) return gelu_(a);
    return a;
}

// Row gather of one 3x3x1 tap on a [B][X][Y][Z] crop batch: row v reads its (dx, dy) neighbour, or
// zeros outside the crop ('same' padding).  Z == 0 disables the gather.
struct Gather {
    int X, Y, Z, dx, dy;
    double iX, iY, iZ;  // reciprocals: the flat index is split by multiplies, not 64-bit divisions
};
inline Gather make_gather(int X, int Y, int Z, int dx, int dy) {
    return Gather{X, Y, Z, dx, dy, X > 0 ? 1.0 / (double)X : 0.0, Y > 0 ? 1.0 / (double)Y : 0.0,
                  Z > 0 ? 1.0 / (double)Z : 0.0};
}
// n = q d + r, 0 <= r < d, for 0 <= n < 2^51: the double product is within one of the quotient
__device__ __forceinline__ void divmod(int64_t n, int d, double inv, int64_t& q, int& r) {
    q = (int64_t)((double)n * inv);
    int64_t rr = n - q * d;
    if (rr < 0) { rr += d; --q; }
    else if (rr >= d) { rr -= d; ++q; }
    r = (int)rr;
}
__device__ __forceinline__ int64_t gather_row(const Gather& gt, int64_t v) {
    if (gt.Z == 0) return v;
    // v = ((b X + x) Y + y) Z + z
    int64_t q2, q3, q4;
    int z, y, x;
    divmod(v, gt.Z, gt.iZ, q2, z);
    divmod(q2, gt.Y, gt.iY, q3, y);
    divmod(q3, gt.X, gt.iX, q4, x);
    (void)z;
    (void)q4;
    const int xx = x + gt.dx, yy = y + gt.dy;
    if (xx < 0 || xx >= gt.X || yy < 0 || yy >= gt.Y) return -1;
    return v + ((int64_t)gt.dx * gt.Y + gt.dy) * gt.Z;
}

// Row tensors read through a buffer resource of N x 256 bytes: an offset at or beyond kOutside is outside any of
// them (N < 2^23) and the hardware returns zeros for it -- padding without a select behind the load.
constexpr uint32_t kOutside = 0x80000000u;
// one v_max_f32 (fmaxf would quiet a signalling NaN first: two instructions per value)
__device__ __forceinline__ float max_1op(float x, float floor) {
    float y;
    asm("v_max_f32 %0, %1, %2" : "=v"(y) : "s"(floor), "v"(x));
    return y;
}
__device__ __forceinline__ float relu_1op(float x) {
    float y;
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
    return y;
}

// Staging a weight block into LDS: a thread's loads first, all in flight together, then its LDS stores.  Written as
// one loop -- `if (inside) v = W[..]; Wl[..] = v;` -- the compiler keeps each load inside its branch and waits for it
// before the store: 16 dependent L2 round trips per thread, 10 us at the head of EVERY launch of these kernels (a
// third of a 1 x 1 layer's 30 us at the crop batch's 190 k voxels).  load(e) must be branch-free and read a valid
// address whatever e (clamp, do not skip); store(e, v) decides what element e becomes.
template <int THREADS, class Load, class Store>
__device__ __forceinline__ void stage_in_flight(int total, Load load, Store store) {
    for (int base = 0; base < total; base += 16 * THREADS) {
        float wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = base + u * THREADS + (int)threadIdx.x;
            wv[u] = load(e < total ? e : total - 1);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = base + u * THREADS + (int)threadIdx.x;
            if (e < total) store(e, wv[u]);
        }
    }
}
__device__ __forceinline__ int clamp_hi(int v, int n) { return v < n ? v : n - 1; }   // n >= 1

// Y[N][ldy] (cols < ndim) = act(X[N][ldx] (cols < kdim) . W + b), W given as Wl[k][j]:
//   trans = 0: Wl[k][j] = W[k * ldw + j]   (forward, W canonical [in][out])
//   trans = 1: Wl[k][j] = W[j * ldw + k]   (backward-data: dX = dY . W^T)
// accum: Y += ...;  mask: result *= (mask[v][j] > 0)  (relu backward)
// blockIdx.y selects a 64-column slab of the output; the block stages that slab of W for the whole
// K range (kpad rows, kpad = kdim rounded up to 4) in dynamic LDS.  kdim, ndim <= 256.
__global__ __launch_bounds__(256) void xw_kernel(const float* __restrict__ X, int ldx, int kdim,
                                                 const float* __restrict__ W, int ldw, int trans,
                                                 const float* __restrict__ b, float* __restrict__ Y,
                                                 int ldy, int ndim, int act, int accum,
                                                 const float* __restrict__ mask, int ldm, int64_t N,
                                                 Gather gt) {
    extern __shared__ float Wl[];
    const int n0 = blockIdx.y * 64;
    // rows of X that are 16-byte aligned are read as float4: in MFMA k-step (q, c) lane group g then
    // supplies k = 16 q + 4 g + c (any bijection of k over (step, group) sums the same products)
    const bool vec = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    // 16-byte aligned output rows are written as float4: output tile m of lane i then stands for column
    // 4 i + m instead of 16 m + i (the weight slab is staged with its columns permuted to match), so a lane
    // holds four consecutive columns of each of its rows and a 16-lane group writes a whole 256-byte row
    const bool vout = (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0 &&
                      (!mask || ((ldm & 3) == 0 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0));
    const int kpad = vec ? (kdim + 15) & ~15 : (kdim + 3) & ~3;
    stage_in_flight<256>(
        kpad * 64,
        [&](int e) {
            const int k = clamp_hi(e >> 6, kdim), j = clamp_hi(n0 + (e & 63), ndim);
            return trans ? W[j * ldw + k] : W[k * ldw + j];
        },
        [&](int e, float v) {
            const int k = e >> 6, jj = e & 63;
            Wl[k * kWs + (vout ? 16 * (jj & 3) + (jj >> 2) : jj)] = k < kdim && n0 + jj < ndim ? v : 0.0f;
        });
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int64_t ntile = (N + 15) / 16;
    const int ksteps = kpad >> 2;
    // kdim <= 64 with aligned rows: the tile's four float4 per lane are loaded one tile ahead, unconditionally
    // (a padded tap reads row 0; columns beyond kdim stay inside the row stride), and zeroed where they are
    // used -- the next tile's rows are in flight during this tile's 64 MFMAs
    const bool pre = vec && kdim <= 64 && ldx >= 64;
    const int nq = (kdim + 15) >> 4;
    float4 nxt[4];
    int64_t nva = 0;
    auto fetch = [&](int64_t tile) {
        const int64_t t = tile < ntile ? tile : ntile - 1;
        const int64_t v = t * 16 + i < N ? t * 16 + i : N - 1;
        nva = gather_row(gt, v);
        const float* xr = X + (nva < 0 ? 0 : nva) * ldx + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nq) nxt[q] = *reinterpret_cast<const float4*>(xr + 16 * q);
    };
    if (pre && (int64_t)blockIdx.x * 4 + wave < ntile) fetch((int64_t)blockIdx.x * 4 + wave);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += (int64_t)gridDim.x * 4) {
        const int64_t v0 = tile * 16;
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (pre) {
            float4 cur[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
            const bool rowok = nva >= 0;
            fetch(tile + (int64_t)gridDim.x * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q >= nq) break;
                const int k0 = 16 * q + 4 * g;
                float ac[4] = {rowok && k0 + 0 < kdim ? cur[q].x : 0.0f, rowok && k0 + 1 < kdim ? cur[q].y : 0.0f,
                               rowok && k0 + 2 < kdim ? cur[q].z : 0.0f, rowok && k0 + 3 < kdim ? cur[q].w : 0.0f};
                if (act & (ACT_RELU_IN | ACT_GELU_IN)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ac[c] = act_in(ac[c], act);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* wr = Wl + (k0 + c) * kWs + i;
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = QB_MFMA16F(ac[c], wr[16 * m], acc[m]);
                }
            }
        }
        const int64_t va = pre ? 0 : gather_row(gt, v0 + i < N ? v0 + i : N - 1);
        const float* xr = X + (va < 0 ? 0 : va) * ldx;
        if (pre) {
        } else if (vec) {
            for (int q = 0; 16 * q < kdim; ++q) {
                const int k0 = 16 * q + 4 * g;
                float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (va >= 0 && k0 < kdim) a = *reinterpret_cast<const float4*>(xr + k0);
                float ac[4] = {k0 + 0 < kdim ? a.x : 0.0f, k0 + 1 < kdim ? a.y : 0.0f,
                               k0 + 2 < kdim ? a.z : 0.0f, k0 + 3 < kdim ? a.w : 0.0f};
                if (act & (ACT_RELU_IN | ACT_GELU_IN)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ac[c] = act_in(ac[c], act);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* wr = Wl + (k0 + c) * kWs + i;
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = QB_MFMA16F(ac[c], wr[16 * m], acc[m]);
                }
            }
        } else {
            for (int s = 0; s < ksteps; ++s) {
                const int k = 4 * s + g;
                float a = (k < kdim && va >= 0) ? xr[k] : 0.0f;
                a = act_in(a, act);
                const float* wr = Wl + k * kWs + i;
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = QB_MFMA16F(a, wr[16 * m], acc[m]);
            }
        }
        if (vout) {
            const int j = n0 + 4 * i;
            if (j < ndim) {
                float bj[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) bj[m] = (b && j + m < ndim) ? b[j + m] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t v = v0 + 4 * g + r;
                    if (v >= N) continue;
                    float y[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) y[m] = acc[m][r] + bj[m];
                    float4* yp = reinterpret_cast<float4*>(Y + v * ldy + j);
                    if (j + 3 < ndim) {
                        if (accum) {
                            const float4 o = *yp;
                            y[0] += o.x; y[1] += o.y; y[2] += o.z; y[3] += o.w;
                        }
                        if (act & (ACT_RELU | ACT_GELU)) {
#pragma unroll
                            for (int m = 0; m < 4; ++m) y[m] = act_out(y[m], act);
                        }
                        if (mask) {
                            const float4 mk = *reinterpret_cast<const float4*>(mask + v * ldm + j);
                            y[0] = mk.x > 0.0f ? y[0] : 0.0f; y[1] = mk.y > 0.0f ? y[1] : 0.0f;
                            y[2] = mk.z > 0.0f ? y[2] : 0.0f; y[3] = mk.w > 0.0f ? y[3] : 0.0f;
                        }
                        *yp = make_float4(y[0], y[1], y[2], y[3]);
                    } else {  // ragged last columns
                        for (int m = 0; m < 4 && j + m < ndim; ++m) {
                            float t = y[m];
                            if (accum) t += Y[v * ldy + j + m];
                            t = act_out(t, act);
                            if (mask) t = mask[v * ldm + j + m] > 0.0f ? t : 0.0f;
                            Y[v * ldy + j + m] = t;
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int j = n0 + 16 * m + i;
                if (j >= ndim) continue;
                const float bj = b ? b[j] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t v = v0 + 4 * g + r;
                    if (v >= N) continue;
                    float y = acc[m][r] + bj;
                    if (accum) y += Y[v * ldy + j];
                    y = act_out(y, act);
                    if (mask) y = mask[v * ldm + j] > 0.0f ? y : 0.0f;
                    Y[v * ldy + j] = y;
                }
            }
        }
    }
}

// The common shape of xw_kernel as a leaner kernel: kdim, ndim <= 64, 16-byte aligned rows on both sides, no
// gather.  Same staging and k / column permutations; ACCUM / MASK are compile-time, and the rows of Y and of
// the mask that the epilogue needs are requested BEFORE the tile's 64 MFMAs (read right before their use
// they cost four dependent memory latencies per tile -- half of the backward launches accumulate or mask).
template <bool ACCUM, bool MASK>
__global__ __launch_bounds__(256) void xw64_kernel(const float* __restrict__ X, int ldx, int kdim,
                                                   const float* __restrict__ W, int ldw, int trans,
                                                   const float* __restrict__ b, float* __restrict__ Y, int ldy,
                                                   int ndim, int act, const float* __restrict__ mask, int ldm,
                                                   int64_t N) {
    extern __shared__ float Wl[];
    const int kpad = (kdim + 15) & ~15;
    stage_in_flight<256>(
        kpad * 64,
        [&](int e) {
            const int k = clamp_hi(e >> 6, kdim), j = clamp_hi(e & 63, ndim);
            return trans ? W[j * ldw + k] : W[k * ldw + j];
        },
        [&](int e, float v) {
            const int k = e >> 6, j = e & 63;
            Wl[k * kWs + 16 * (j & 3) + (j >> 2)] = k < kdim && j < ndim ? v : 0.0f;   // tile m of lane i = column 4 i + m
        });
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int64_t ntile = (N + 15) / 16;
    const int nq = (kdim + 15) >> 4;
    const int j = 4 * i;
    float bj[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) bj[m] = (b && j + m < ndim) ? b[j + m] : 0.0f;
    float4 nxt[4];
    auto fetch = [&](int64_t tile) {
        const int64_t t = tile < ntile ? tile : ntile - 1;
        const int64_t v = t * 16 + i < N ? t * 16 + i : N - 1;
        const float* xr = X + v * ldx + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nq) nxt[q] = *reinterpret_cast<const float4*>(xr + 16 * q);
    };
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile < ntile) fetch(tile);
    for (; tile < ntile; tile += stride) {
        const int64_t v0 = tile * 16;
        float4 cur[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        fetch(tile + stride);
        float4 old[4], mk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t v = v0 + 4 * g + r < N ? v0 + 4 * g + r : N - 1;
            if (ACCUM) old[r] = *reinterpret_cast<const float4*>(Y + v * ldy + j);
            if (MASK) mk[r] = *reinterpret_cast<const float4*>(mask + v * ldm + j);
        }
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= nq) break;
            const int k0 = 16 * q + 4 * g;
            float ac[4] = {k0 + 0 < kdim ? cur[q].x : 0.0f, k0 + 1 < kdim ? cur[q].y : 0.0f,
                           k0 + 2 < kdim ? cur[q].z : 0.0f, k0 + 3 < kdim ? cur[q].w : 0.0f};
            if (act & ACT_RELU_IN) {
#pragma unroll
                for (int c = 0; c < 4; ++c) ac[c] = fmaxf(ac[c], 0.0f);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float* wr = Wl + (k0 + c) * kWs + i;
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = QB_MFMA16F(ac[c], wr[16 * m], acc[m]);
            }
        }
        if (j >= ndim) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t v = v0 + 4 * g + r;
            if (v >= N) continue;
            float y[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) y[m] = acc[m][r] + bj[m];
            if (ACCUM) {
                y[0] += old[r].x; y[1] += old[r].y; y[2] += old[r].z; y[3] += old[r].w;
            }
            if (act & ACT_RELU) {
#pragma unroll
                for (int m = 0; m < 4; ++m) y[m] = fmaxf(y[m], 0.0f);
            }
            if (MASK) {
                y[0] = mk[r].x > 0.0f ? y[0] : 0.0f; y[1] = mk[r].y > 0.0f ? y[1] : 0.0f;
                y[2] = mk[r].z > 0.0f ? y[2] : 0.0f; y[3] = mk[r].w > 0.0f ? y[3] : 0.0f;
            }
            if (j + 3 < ndim) {
                *reinterpret_cast<float4*>(Y + v * ldy + j) = make_float4(y[0], y[1], y[2], y[3]);
            } else {  // ragged last columns
                for (int m = 0; m < 4 && j + m < ndim; ++m) Y[v * ldy + j + m] = y[m];
            }
        }
    }
}

// Both heads from one pass over the last activation rows, written straight into the caller's compact
// [N][5] and [N][T] buffers: columns 0-4 of the stacked product are the q head, 5 .. 5 + T - 1 the log-sigma head.
__global__ __launch_bounds__(256) void xw64_heads_kernel(const float* __restrict__ X, int ld, int kdim,
                                                         const float* __restrict__ Wf, const float* __restrict__ bf,
                                                         const float* __restrict__ Ws, const float* __restrict__ bs,
                                                         int T, float* __restrict__ out_q, float* __restrict__ out_ls,
                                                         int64_t N) {
    extern __shared__ float Wl[];
    const int ndim = 5 + T;
    const bool compact = ndim <= 16;
    const int kpad = (kdim + 15) & ~15;
    stage_in_flight<256>(
        kpad * 64,
        [&](int e) {
            const int k = clamp_hi(e >> 6, kdim), j = clamp_hi(e & 63, ndim);
            const float* src = j < 5 ? Wf + k * 5 + j : Ws + k * T + (j - 5);
            return *src;
        },
        [&](int e, float v) {
            const int k = e >> 6, j = e & 63;
            // tile m of lane i = column 4 i + m; with at most 16 columns (T <= 11) they all sit in tile 0, lane i = column i:
            // one MFMA per k step instead of four, and every lane has a column to store
            Wl[k * kWs + (compact ? j : 16 * (j & 3) + (j >> 2))] = k < kdim && j < ndim ? v : 0.0f;
        });
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int64_t ntile = (N + 15) / 16;
    const int nq = (kdim + 15) >> 4;
    const int j = 4 * i;
    float bj[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) bj[m] = j + m < 5 ? bf[j + m] : (j + m < ndim ? bs[j + m - 5] : 0.0f);
    const float bi = i < 5 ? bf[i] : (i < ndim ? bs[i - 5] : 0.0f);   // compact: this lane's one column
    float4 nxt[4];
    auto fetch = [&](int64_t tile) {
        const int64_t t = tile < ntile ? tile : ntile - 1;
        const int64_t v = t * 16 + i < N ? t * 16 + i : N - 1;
        const float* xr = X + v * ld + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nq) nxt[q] = *reinterpret_cast<const float4*>(xr + 16 * q);
    };
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile < ntile) fetch(tile);
    for (; tile < ntile; tile += stride) {
        const int64_t v0 = tile * 16;
        float4 cur[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        fetch(tile + stride);
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= nq) break;
            const int k0 = 16 * q + 4 * g;
            const float ac[4] = {k0 + 0 < kdim ? cur[q].x : 0.0f, k0 + 1 < kdim ? cur[q].y : 0.0f,
                                 k0 + 2 < kdim ? cur[q].z : 0.0f, k0 + 3 < kdim ? cur[q].w : 0.0f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float* wr = Wl + (k0 + c) * kWs + i;
                if (compact) {
                    acc[0] = QB_MFMA16F(ac[c], wr[0], acc[0]);
                    continue;
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = QB_MFMA16F(ac[c], wr[16 * m], acc[m]);
            }
        }
        if (compact) {
            if (i >= ndim) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t v = v0 + 4 * g + r;
                if (v >= N) continue;
                const float y = acc[0][r] + bi;
                if (i < 5) out_q[v * 5 + i] = y;
                else out_ls[v * T + (i - 5)] = y;
            }
            continue;
        }
        if (j >= ndim) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t v = v0 + 4 * g + r;
            if (v >= N) continue;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int col = j + m;
                const float y = acc[m][r] + bj[m];
                if (col < 5) out_q[v * 5 + col] = y;
                else if (col < ndim) out_ls[v * T + (col - 5)] = y;
            }
        }
    }
}

// Forward of a residual block's two input branches in one pass over the input rows (voxel batches):
//   Y1 = relu(X W1 + b1)              skip = relu(b Wc + bc)                       model.py:148
//   Y2 = relu(relu(X) W2 + b2)        t = relu(relu(b) Wr1 + br1)                  model.py:151-155
// Both weight slabs stay in LDS; the two chains run one after the other on the same row registers and
// accumulator set.
__global__ __launch_bounds__(256) void xw64_fork_kernel(const float* __restrict__ X, int ld, int kdim, int ndim,
                                                        const float* __restrict__ W1, const float* __restrict__ b1,
                                                        float* __restrict__ Y1, const float* __restrict__ W2,
                                                        const float* __restrict__ b2, float* __restrict__ Y2,
                                                        int ldw, int64_t N) {
    extern __shared__ float Wl[];  // [2][kpad][kWs]
    const int kpad = (kdim + 15) & ~15;
    float* Wl2 = Wl + kpad * kWs;
    stage_in_flight<256>(
        2 * kpad * 64,
        [&](int e) {
            const int which = e >= kpad * 64, ee = which ? e - kpad * 64 : e;
            const int k = clamp_hi(ee >> 6, kdim), j = clamp_hi(ee & 63, ndim);
            return (which ? W2 : W1)[k * ldw + j];
        },
        [&](int e, float v) {
            const int which = e >= kpad * 64, ee = which ? e - kpad * 64 : e;
            const int k = ee >> 6, j = ee & 63;
            (which ? Wl2 : Wl)[k * kWs + 16 * (j & 3) + (j >> 2)] = k < kdim && j < ndim ? v : 0.0f;
        });
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int64_t ntile = (N + 15) / 16;
    const int nq = (kdim + 15) >> 4;
    const int j = 4 * i;
    float bj1[4], bj2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        bj1[m] = (b1 && j + m < ndim) ? b1[j + m] : 0.0f;
        bj2[m] = (b2 && j + m < ndim) ? b2[j + m] : 0.0f;
    }
    float4 nxt[4];
    auto fetch = [&](int64_t tile) {
        const int64_t t = tile < ntile ? tile : ntile - 1;
        const int64_t v = t * 16 + i < N ? t * 16 + i : N - 1;
        const float* xr = X + v * ld + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nq) nxt[q] = *reinterpret_cast<const float4*>(xr + 16 * q);
    };
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile < ntile) fetch(tile);
    for (; tile < ntile; tile += stride) {
        const int64_t v0 = tile * 16;
        float4 cur[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        fetch(tile + stride);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const float* Ws = pass ? Wl2 : Wl;
            f32x4 acc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q >= nq) break;
                const int k0 = 16 * q + 4 * g;
                float ac[4] = {k0 + 0 < kdim ? cur[q].x : 0.0f, k0 + 1 < kdim ? cur[q].y : 0.0f,
                               k0 + 2 < kdim ? cur[q].z : 0.0f, k0 + 3 < kdim ? cur[q].w : 0.0f};
                if (pass) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ac[c] = fmaxf(ac[c], 0.0f);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* wr = Ws + (k0 + c) * kWs + i;
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = QB_MFMA16F(ac[c], wr[16 * m], acc[m]);
                }
            }
            if (j >= ndim) continue;
            float* Y = pass ? Y2 : Y1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t v = v0 + 4 * g + r;
                if (v >= N) continue;
                float y[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) y[m] = fmaxf(acc[m][r] + (pass ? bj2[m] : bj1[m]), 0.0f);
                if (j + 3 < ndim) {
                    *reinterpret_cast<float4*>(Y + v * ld + j) = make_float4(y[0], y[1], y[2], y[3]);
                } else {
                    for (int m = 0; m < 4 && j + m < ndim; ++m) Y[v * ld + j + m] = y[m];
                }
            }
        }
    }
}

// Backward-data of a residual block's two input branches in one pass (voxel batches):
//   Y = (X1 W1^T) * (M > 0) + X2 W2^T        d b_in = (dE Wr1^T) * (b_in > 0) + dC Wc^T
// instead of a masked GEMM into Y followed by an accumulating GEMM that reads Y back.  One accumulator set:
// the first chain is masked in place before the second chain adds to it.  Both weight slabs stay in LDS.
__global__ __launch_bounds__(256) void xw64_dual_kernel(const float* __restrict__ X1, const float* __restrict__ W1,
                                                        const float* __restrict__ M, const float* __restrict__ X2,
                                                        const float* __restrict__ W2, int ld, int kdim, int ndim,
                                                        int ldw, float* __restrict__ Y, int64_t N) {
    extern __shared__ float Wl[];  // [2][kpad][kWs]
    const int kpad = (kdim + 15) & ~15;
    float* Wl2 = Wl + kpad * kWs;
    stage_in_flight<256>(
        2 * kpad * 64,
        [&](int e) {
            const int which = e >= kpad * 64, ee = which ? e - kpad * 64 : e;
            const int k = clamp_hi(ee >> 6, kdim), j = clamp_hi(ee & 63, ndim);
            return (which ? W2 : W1)[j * ldw + k];   // transposed: dX = dY W^T
        },
        [&](int e, float v) {
            const int which = e >= kpad * 64, ee = which ? e - kpad * 64 : e;
            const int k = ee >> 6, j = ee & 63;
            (which ? Wl2 : Wl)[k * kWs + 16 * (j & 3) + (j >> 2)] = k < kdim && j < ndim ? v : 0.0f;
        });
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int64_t ntile = (N + 15) / 16;
    const int nq = (kdim + 15) >> 4;
    const int j = 4 * i;
    float4 n1[4], n2[4];
    auto fetch = [&](int64_t tile) {
        const int64_t t = tile < ntile ? tile : ntile - 1;
        const int64_t v = t * 16 + i < N ? t * 16 + i : N - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nq) {
                n1[q] = *reinterpret_cast<const float4*>(X1 + v * ld + 4 * g + 16 * q);
                n2[q] = *reinterpret_cast<const float4*>(X2 + v * ld + 4 * g + 16 * q);
            }
    };
    auto chain = [&](const float4 (&cur)[4], const float* Ws, f32x4 (&acc)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= nq) break;
            const int k0 = 16 * q + 4 * g;
            const float ac[4] = {k0 + 0 < kdim ? cur[q].x : 0.0f, k0 + 1 < kdim ? cur[q].y : 0.0f,
                                 k0 + 2 < kdim ? cur[q].z : 0.0f, k0 + 3 < kdim ? cur[q].w : 0.0f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float* wr = Ws + (k0 + c) * kWs + i;
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = QB_MFMA16F(ac[c], wr[16 * m], acc[m]);
            }
        }
    };
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile < ntile) fetch(tile);
    for (; tile < ntile; tile += stride) {
        const int64_t v0 = tile * 16;
        float4 c1[4], c2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c1[q] = n1[q];
            c2[q] = n2[q];
        }
        fetch(tile + stride);
        float4 mk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t v = v0 + 4 * g + r < N ? v0 + 4 * g + r : N - 1;
            mk[r] = *reinterpret_cast<const float4*>(M + v * ld + j);
        }
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        chain(c1, Wl, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // through the relu of the first branch, before the second one is added
            acc[0][r] = mk[r].x > 0.0f ? acc[0][r] : 0.0f;
            acc[1][r] = mk[r].y > 0.0f ? acc[1][r] : 0.0f;
            acc[2][r] = mk[r].z > 0.0f ? acc[2][r] : 0.0f;
            acc[3][r] = mk[r].w > 0.0f ? acc[3][r] : 0.0f;
        }
        chain(c2, Wl2, acc);
        if (j >= ndim) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t v = v0 + 4 * g + r;
            if (v >= N) continue;
            if (j + 3 < ndim) {
                *reinterpret_cast<float4*>(Y + v * ld + j) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
            } else {
                for (int m = 0; m < 4 && j + m < ndim; ++m) Y[v * ld + j + m] = acc[m][r];
            }
        }
    }
}

// The gating layer of a residual block with its blend as the epilogue (channel-wise gating, model.py:164-170):
//   gl = r Wg + bg            (kept: the backward needs the logits)
//   bout = skip (1 - g) + r g,   g = sigmoid(gl + gate_offset)
// instead of a second pass over gl, skip and r.  The rows of skip and r that the epilogue blends are requested
// before the tile's MFMAs (r is the GEMM input itself: a cache hit in another lane layout).
__global__ __launch_bounds__(256) void xw64_gate_kernel(const float* __restrict__ X, int ldx, int kdim,
                                                        const float* __restrict__ W, int ldw,
                                                        const float* __restrict__ b, float* __restrict__ GL,
                                                        const float* __restrict__ skip, float* __restrict__ bout,
                                                        int ldy, int ndim, float offset, int64_t N) {
    extern __shared__ float Wl[];
    const int kpad = (kdim + 15) & ~15;
    stage_in_flight<256>(
        kpad * 64,
        [&](int e) {
            const int k = clamp_hi(e >> 6, kdim), j = clamp_hi(e & 63, ndim);
            return W[k * ldw + j];
        },
        [&](int e, float v) {
            const int k = e >> 6, j = e & 63;
            Wl[k * kWs + 16 * (j & 3) + (j >> 2)] = k < kdim && j < ndim ? v : 0.0f;   // tile m of lane i = column 4 i + m
        });
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int64_t ntile = (N + 15) / 16;
    const int nq = (kdim + 15) >> 4;
    const int j = 4 * i;
    float bj[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) bj[m] = (b && j + m < ndim) ? b[j + m] : 0.0f;
    float4 nxt[4];
    auto fetch = [&](int64_t tile) {
        const int64_t t = tile < ntile ? tile : ntile - 1;
        const int64_t v = t * 16 + i < N ? t * 16 + i : N - 1;
        const float* xr = X + v * ldx + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nq) nxt[q] = *reinterpret_cast<const float4*>(xr + 16 * q);
    };
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile < ntile) fetch(tile);
    for (; tile < ntile; tile += stride) {
        const int64_t v0 = tile * 16;
        float4 cur[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        fetch(tile + stride);
        float4 sk[4], rr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t v = v0 + 4 * g + r < N ? v0 + 4 * g + r : N - 1;
            sk[r] = *reinterpret_cast<const float4*>(skip + v * ldy + j);
            rr[r] = *reinterpret_cast<const float4*>(X + v * ldx + j);
        }
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= nq) break;
            const int k0 = 16 * q + 4 * g;
            const float ac[4] = {k0 + 0 < kdim ? cur[q].x : 0.0f, k0 + 1 < kdim ? cur[q].y : 0.0f,
                                 k0 + 2 < kdim ? cur[q].z : 0.0f, k0 + 3 < kdim ? cur[q].w : 0.0f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float* wr = Wl + (k0 + c) * kWs + i;
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = QB_MFMA16F(ac[c], wr[16 * m], acc[m]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t v = v0 + 4 * g + r;
            if (v >= N) continue;
            const float s4[4] = {sk[r].x, sk[r].y, sk[r].z, sk[r].w}, r4[4] = {rr[r].x, rr[r].y, rr[r].z, rr[r].w};
            float gl[4], o[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                gl[m] = acc[m][r] + bj[m];
                const float gate = 1.0f / (1.0f + expf(-(gl[m] + offset)));
                o[m] = j + m < ndim ? s4[m] * (1.0f - gate) + r4[m] * gate : 0.0f;   // zero beyond U, as gate_fwd_kernel
            }
            *reinterpret_cast<float4*>(bout + v * ldy + j) = make_float4(o[0], o[1], o[2], o[3]);
            if (j + 3 < ndim) {
                *reinterpret_cast<float4*>(GL + v * ldy + j) = make_float4(gl[0], gl[1], gl[2], gl[3]);
            } else {
                for (int m = 0; m < 4 && j + m < ndim; ++m) GL[v * ldy + j + m] = gl[m];
            }
        }
    }
}

// The whole 3x3x1 'same' convolution of one layer in ONE launch (U <= 64):
//   Y[v] = act(sum_tap X[nbr(v, tap)] K[tap] + b)            (flip = 0, model.py:152-157)
//   Y[v] = (sum_tap X[nbr(v, -tap)] K[tap]^T) * (mask[v] > 0)  (flip = 1: adjoint wrt the input)
// All nine tap kernels (<= 64 x 64 floats each) are staged in LDS once per workgroup (one 1024-thread
// workgroup per CU); a wave keeps its 16-voxel tile's accumulators in registers across the taps --
// instead of nine launches that each read and rewrite the output tensor.  Activation rows are read 16 bytes
// per lane: in MFMA k-step (q, c) lane group g supplies k = 16 q + 4 g + c (any bijection of k over
// (step, group) sums the same products), so one float4 load feeds four k-steps.
__global__ __launch_bounds__(1024) void conv9_kernel(const float* __restrict__ X, int ldx, int U,
                                                     const float* __restrict__ K9, int flip,
                                                     const float* __restrict__ b, float* __restrict__ Y,
                                                     int ldy, int act, const float* __restrict__ mask,
                                                     int ldm, int64_t N, Gather g0) {
    extern __shared__ float Wl[];  // [9][64][kWs]: all nine taps stay resident (146 KiB), one block per CU
    const bool vout = (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0 &&
                      (!mask || ((ldm & 3) == 0 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0));
    stage_in_flight<1024>(
        9 * 64 * 64,
        [&](int e) {
            const int tap = e >> 12, k = clamp_hi((e >> 6) & 63, U), j = clamp_hi(e & 63, U);
            const float* W = K9 + (int64_t)tap * U * U;
            return flip ? W[j * U + k] : W[k * U + j];
        },
        [&](int e, float v) {
            const int tap = e >> 12, k = (e >> 6) & 63, j = e & 63;
            Wl[(tap * 64 + k) * kWs + (vout ? 16 * (j & 3) + (j >> 2) : j)] = k < U && j < U ? v : 0.0f;  // columns permuted as in xw_kernel
        });
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int64_t ntile = (N + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * 16 + wave; tile < ntile; tile += (int64_t)gridDim.x * 16) {
        const int64_t v0 = tile * 16;
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int tap = 0; tap < 9; ++tap) {
            const int dx = tap / 3 - 1, dy = tap % 3 - 1;
            Gather gt = g0;
            gt.dx = flip ? -dx : dx;
            gt.dy = flip ? -dy : dy;
            const int64_t va = gather_row(gt, v0 + i < N ? v0 + i : N - 1);
            if (__builtin_amdgcn_ballot_w64(va >= 0) == 0) continue;  // the whole tile reads padding
            const float* xr = X + (va < 0 ? 0 : va) * ldx + 4 * g;
            const float* Wt = Wl + tap * 64 * kWs;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (16 * q >= U) break;
                float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (va >= 0) a = *reinterpret_cast<const float4*>(xr + 16 * q);
                const int k0 = 16 * q + 4 * g;
                float ac[4] = {k0 + 0 < U ? a.x : 0.0f, k0 + 1 < U ? a.y : 0.0f,
                               k0 + 2 < U ? a.z : 0.0f, k0 + 3 < U ? a.w : 0.0f};
                if (act & (ACT_RELU_IN | ACT_GELU_IN)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ac[c] = act_in(ac[c], act);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* wr = Wt + (k0 + c) * kWs + i;
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = QB_MFMA16F(ac[c], wr[16 * m], acc[m]);
                }
            }
        }
        if (vout) {  // lane i holds columns 4 i .. 4 i + 3 of its four rows: one float4 store per row
            const int j = 4 * i;
            if (j < U) {
                float bj[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) bj[m] = (b && j + m < U) ? b[j + m] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t v = v0 + 4 * g + r;
                    if (v >= N) continue;
                    float y[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        y[m] = acc[m][r] + bj[m];
                        y[m] = act_out(y[m], act);
                    }
                    if (j + 3 < U) {
                        if (mask) {
                            const float4 mk = *reinterpret_cast<const float4*>(mask + v * ldm + j);
                            y[0] = mk.x > 0.0f ? y[0] : 0.0f; y[1] = mk.y > 0.0f ? y[1] : 0.0f;
                            y[2] = mk.z > 0.0f ? y[2] : 0.0f; y[3] = mk.w > 0.0f ? y[3] : 0.0f;
                        }
                        *reinterpret_cast<float4*>(Y + v * ldy + j) = make_float4(y[0], y[1], y[2], y[3]);
                    } else {
                        for (int m = 0; m < 4 && j + m < U; ++m)
                            Y[v * ldy + j + m] = (mask && !(mask[v * ldm + j + m] > 0.0f)) ? 0.0f : y[m];
                    }
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int j = 16 * m + i;
                if (j >= U) continue;
                const float bj = b ? b[j] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t v = v0 + 4 * g + r;
                    if (v >= N) continue;
                    float y = acc[m][r] + bj;
                    y = act_out(y, act);
                    if (mask) y = mask[v * ldm + j] > 0.0f ? y : 0.0f;
                    Y[v * ldy + j] = y;
                }
            }
        }
    }
}

// conv9_kernel on the f16 matrix pipe: Y^T[unit][voxel] with every operand split in two halves (x = hi +
// 2^-11 lo; three v_mfma_f32_16x16x32_f16 per tile, tap and k-step, float32 accumulation -- encoder_core.h,
// per-product error <= ~7e-7).  The exact-f32 form spends 9 x 64 x 32 = 18.4 k vector-pipe cycles per tile on
// MFMAs; this one 9 x 24 x 16 = 3.5 k matrix-pipe cycles plus the operand splits.  The nine tap kernels are
// converted once per workgroup into the weight image of encoder_core.h (9 x 16 KiB of LDS); a lane owns one
// voxel: its crop coordinates are split once per tile, every tap reads the neighbour's row (or zeros).
// SCALED (the backward-data launches): the rows are deltas that carry the loss's 1 / sum(mask) -- 1e-6 and below, where
// an f16 high half is subnormal or zero.  They are multiplied by 2^floor(log2 sum(mask)) on arrival (in place of the
// forward's relu-on-arrival: the same instruction count) and the outputs by its inverse: exact both ways, and the
// split sees the per-voxel gradients at their own magnitude, whatever the batch size.
template <bool SCALED>
__global__ __launch_bounds__(1024) void conv9h_kernel(const float* __restrict__ X, int ldx, int U,
                                                      const float* __restrict__ K9, int flip,
                                                      const float* __restrict__ b, float* __restrict__ Y,
                                                      int ldy, int act, const float* __restrict__ mask,
                                                      int ldm, int64_t N, Gather g0,
                                                      const double* __restrict__ sums) {
    extern __shared__ __align__(16) float img[];  // [9][s 2][m 4][hi, lo][lane 64][8 halves], then bias[64]
    float* bias = img + 9 * 4096;
    // the nine tap kernels into the weight image: thread (a, b) of a 16 x 64 patch reads W[tap][a][b] (rows of U
    // floats, coalesced; 36 independent loads per thread, all in flight at once -- walking the IMAGE in order
    // instead made 72 dependent scattered reads per thread, half of the launch) and scatters its two halves
#ifndef QB_CONV9H_ABL   // timing experiments only (scripts/dev/variant.sh): 1 no weight image, 2 no MFMAs, 4 one tap's loads, 8 no split
#define QB_CONV9H_ABL 0
#endif
    if (!(QB_CONV9H_ABL & 1)) {
        const int tb = threadIdx.x & 63, ta = threadIdx.x >> 6;
        const float bv = b && threadIdx.x < 64 && (int)threadIdx.x < U ? b[threadIdx.x] : 0.0f;   // in flight with the weights
        float wv[9][4];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
                const int ra = 16 * a4 + ta;
                wv[tap][a4] = ra < U && tb < U ? K9[(int64_t)tap * U * U + ra * U + tb] : 0.0f;
            }
        _Float16* ih = reinterpret_cast<_Float16*>(img);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
                const int ra = 16 * a4 + ta;
                const int k = flip ? tb : ra, j = flip ? ra : tb;   // the image holds Wl[k][j] = flip ? W[j][k] : W[k][j]
                const int e = tap * 8192 + (k >> 5) * 4096 + (j >> 4) * 1024 + (((k >> 2) & 3) * 16 + (j & 15)) * 8 +
                              ((k >> 4) & 1) * 4 + (k & 3);
                const float w = wv[tap][a4];
                const _Float16 hi = (_Float16)w;
                ih[e] = hi;
                ih[e + 512] = (_Float16)((w - (float)hi) * QB_LO_SCALE);
            }
        if (threadIdx.x < 64) bias[threadIdx.x] = bv;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int64_t ntile = (N + 15) / 16;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (uint32_t)(N * 256), 0x00020000);
    uint32_t col[4];   // byte offset of this lane's quarter q of a row (units 16 q + 4 g ..), outside for padding units
#pragma unroll
    for (int q = 0; q < 4; ++q) col[q] = 16 * q + 4 * g < U ? 4u * (16 * q + 4 * g) : kOutside;
    const float in_floor = act & ACT_RELU_IN ? 0.0f : -INFINITY;   // relu on the rows as they arrive, or nothing
    float s_in = 1.0f, s_out = 1.0f;
    if constexpr (SCALED) {
        const float sm = (float)sums[2];
        if (sm >= 1.0f && sm < 1e30f) {
            int e;
            (void)frexpf(sm, &e);          // sm = m 2^e, m in [0.5, 1)
            s_in = ldexpf(1.0f, e - 1);
            s_out = ldexpf(1.0f, 1 - e);
        }
    }
    const float lo_unscale = QB_LO_UNSCALE * s_out;
    for (int64_t tile = (int64_t)blockIdx.x * 16 + wave; tile < ntile; tile += (int64_t)gridDim.x * 16) {
        const int64_t v = tile * 16 + i;
        const bool ok = v < N;
        const int64_t vc = ok ? v : N - 1;
        int64_t q2, q3, q4;
        int z, y, x;
        divmod(vc, g0.Z, g0.iZ, q2, z);
        divmod(q2, g0.Y, g0.iY, q3, y);
        divmod(q3, g0.X, g0.iX, q4, x);
        (void)z;
        (void)q4;
        f32x4 out[4], cross[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            out[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            cross[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        // Rows come through a buffer resource of N x 256 bytes: a padded tap, a voxel beyond the batch or a
        // padding column reads at an offset outside it and the hardware returns zeros -- no selects behind the
        // loads.  The nine taps are unrolled with the next tap's four row quarters in flight during this tap's
        // 24 MFMAs (two register sets in rotation: a copy out of an in-flight load would wait for it).
        const uint32_t rowoff = ok ? (uint32_t)v * 256u : kOutside;
        float4 rows[2][4];
        bool some[2];
        auto load_tap = [&](int tap, float4 (&w)[4], bool& any) {
            const int dx = flip ? 1 - tap / 3 : tap / 3 - 1, dy = flip ? 1 - tap % 3 : tap % 3 - 1;
            const int xx = x + dx, yy = y + dy;
            const bool in = ok && xx >= 0 && xx < g0.X && yy >= 0 && yy < g0.Y;
            any = __builtin_amdgcn_ballot_w64(in) != 0;
            const uint32_t o = in ? rowoff + (uint32_t)((dx * g0.Y + dy) * g0.Z * 256) : kOutside;
#pragma unroll
            for (int q = 0; q < 4; ++q)   // o (a multiple of 256) | col[q] (< 256): their sum, or outside if either is
                w[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, o | col[q], 0, 0));
        };
        load_tap(0, rows[0], some[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < 8 && !(QB_CONV9H_ABL & 4)) load_tap(tap + 1, rows[(tap + 1) & 1], some[(tap + 1) & 1]);
            if ((QB_CONV9H_ABL & 4) && tap < 8) { some[(tap + 1) & 1] = some[0];
#pragma unroll
                for (int q = 0; q < 4; ++q) rows[(tap + 1) & 1][q] = rows[tap & 1][q]; }
            if (!some[tap & 1]) continue;  // the whole tile reads padding
            const float4(&rw)[4] = rows[tap & 1];
            qb::f16x8 bhi[2], blo[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float x8[8] = {rw[2 * st].x, rw[2 * st].y, rw[2 * st].z, rw[2 * st].w,
                               rw[2 * st + 1].x, rw[2 * st + 1].y, rw[2 * st + 1].z, rw[2 * st + 1].w};
#pragma unroll
                for (int j8 = 0; j8 < 8; ++j8) x8[j8] = SCALED ? x8[j8] * s_in : max_1op(x8[j8], in_floor);
                if (QB_CONV9H_ABL & 8) {
                    bhi[st] = __builtin_bit_cast(qb::f16x8, rw[2 * st]);
                    blo[st] = __builtin_bit_cast(qb::f16x8, rw[2 * st + 1]);
                } else
                qb::split8<false>(x8, bhi[st], blo[st]);
            }
            const float* A = img + tap * 4096;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const qb::f16x8 whi = qb::lds_frag(A, (st * 4 + m) * 2 + 0, lane);
                    const qb::f16x8 wlo = qb::lds_frag(A, (st * 4 + m) * 2 + 1, lane);
                    if (QB_CONV9H_ABL & 2) {
                        out[m][0] += (float)whi[0] * (float)bhi[st][0] + (float)wlo[1] * (float)blo[st][1];
                        continue;
                    }
                    out[m] = QB_MFMA_F16(whi, bhi[st], out[m]);
                    cross[m] = QB_MFMA_F16(whi, blo[st], cross[m]);
                    cross[m] = QB_MFMA_F16(wlo, bhi[st], cross[m]);
                }
            }
        }
        if (!ok) continue;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int j = 16 * m + 4 * g;
            if (j >= U) continue;
            const f32x4 bj = qb::load4(bias + j);
            float yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                yv[r] = SCALED ? fmaf(cross[m][r], lo_unscale, out[m][r] * s_out) + bj[r]
                               : fmaf(cross[m][r], QB_LO_UNSCALE, out[m][r]) + bj[r];
                if (act & ACT_RELU) yv[r] = fmaxf(yv[r], 0.0f);
            }
            if (j + 3 < U) {
                if (mask) {
                    const float4 mk = *reinterpret_cast<const float4*>(mask + v * ldm + j);
                    yv[0] = mk.x > 0.0f ? yv[0] : 0.0f; yv[1] = mk.y > 0.0f ? yv[1] : 0.0f;
                    yv[2] = mk.z > 0.0f ? yv[2] : 0.0f; yv[3] = mk.w > 0.0f ? yv[3] : 0.0f;
                }
                *reinterpret_cast<float4*>(Y + v * ldy + j) = make_float4(yv[0], yv[1], yv[2], yv[3]);
            } else {
                for (int r = 0; r < 4 && j + r < U; ++r)
                    Y[v * ldy + j + r] = (mask && !(mask[v * ldm + j + r] > 0.0f)) ? 0.0f : yv[r];
            }
        }
    }
}

// partial[blk][64*64 + 64]: dW[i][j] = sum_v X[v][i] D[v][j] over this block's voxels, then db[j]
// gridDim.y == 9: blockIdx.y is the tap of a 3x3x1 kernel (gt carries the crop geometry only) and the
// partials of tap t start at partial + t * gridDim.x * (64*64 + 64).
// 1024 threads = 16 waves, each accumulating the whole 64 x 64 product over its share of the 4-voxel steps
// (64 accumulator registers per lane): 16 MFMAs per pair of 16-byte loads keep the per-step index / mask
// arithmetic, which f32 MFMAs do not overlap, at a quarter of the MFMA time (a 32 x 32 quadrant per wave
// spent more on it than on its 4 MFMAs).  One block per CU; the in-block reduction uses eight 16 KiB tiles
// of dynamic LDS: waves 0-7 store their tiles, waves 8-15 add theirs on top (one writer per address), and all
// threads add the eight tiles in a fixed order -- three barriers, bitwise reproducible, no float atomics.
constexpr int kXtdTiles = 8;
constexpr size_t kXtdSmem = sizeof(float) * kXtdTiles * (64 * 64 + 64);
template <bool VEC>
__global__ __launch_bounds__(1024) void xtd_kernel(const float* __restrict__ X, int ldx, int kdim,
                                                   const float* __restrict__ D, int ldd, int ndim,
                                                   float* __restrict__ partial, int64_t N, Gather gt,
                                                   int relu_x, const float* __restrict__ Dref) {
    // Dref (optional, laid out like D): D is taken as D * (Dref > 0) -- the step through a relu folded into
    // the operand load instead of a masked copy of the delta tensor
    extern __shared__ float red8[];  // [kXtdTiles][64 * 64 + 64]
    if (gridDim.y == 9) {
        gt.dx = (int)blockIdx.y / 3 - 1;
        gt.dy = (int)blockIdx.y % 3 - 1;
        partial += (int64_t)blockIdx.y * gridDim.x * (64 * 64 + 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float dbsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // 16-byte aligned rows are read as float4: tile m of lane i then stands for column 4 i + m instead of
    // 16 m + i (a permutation of the output rows / columns, undone where the tiles are reduced)
    constexpr bool vec = VEC;  // the launcher checks alignment
    const int64_t nstep = (N + 3) / 4;  // 4 voxels per MFMA k-step
    const int64_t stride = (int64_t)gridDim.x * 16;
    // unconditional loads, zeroing deferred to the point of use (a select right behind a load makes the
    // compiler wait for it and the prefetch would buy nothing); padded / invalid rows read row 0 or v itself,
    // columns beyond kdim / ndim stay inside the row stride
    struct Raw {
        float x[4], d[4], r[4];
        unsigned in;  // bit 0: X row valid, bit 1: D row valid
    };
    auto load = [&](int64_t st, Raw& w) {
        const int64_t v0 = st * 4 + g;
        const bool ok = st < nstep && v0 < N;
        const int64_t v = ok ? v0 : 0;
        const int64_t vx = ok ? gather_row(gt, v) : -1;
        w.in = (vx >= 0 ? 1u : 0u) | (ok ? 2u : 0u);
        const float* xr = X + (vx >= 0 ? vx : v) * ldx;
        const float* dr = D + v * ldd;
        if (vec) {
            const float4 a = *reinterpret_cast<const float4*>(xr + 4 * i);
            const float4 b = *reinterpret_cast<const float4*>(dr + 4 * i);
            w.x[0] = a.x; w.x[1] = a.y; w.x[2] = a.z; w.x[3] = a.w;
            w.d[0] = b.x; w.d[1] = b.y; w.d[2] = b.z; w.d[3] = b.w;
            if (Dref) {
                const float4 c = *reinterpret_cast<const float4*>(Dref + v * ldd + 4 * i);
                w.r[0] = c.x; w.r[1] = c.y; w.r[2] = c.z; w.r[3] = c.w;
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                w.x[m] = xr[16 * m + i];
                w.d[m] = dr[16 * m + i];
                if (Dref) w.r[m] = Dref[v * ldd + 16 * m + i];
            }
        }
    };
    bool kx[4], nd[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int col = vec ? 4 * i + m : 16 * m + i;
        kx[m] = col < kdim;
        nd[m] = col < ndim;
    }
    // Two buffers in rotation, the loop unrolled by two so that no buffer is ever copied: a register move
    // out of an in-flight load makes the wave wait for it, which (with `cur = nxt` at the end of the trip)
    // delayed the next load until the previous one had landed.  Steps past the end read row 0 and
    // contribute zeros.
    auto compute = [&](const Raw& w) {
        const bool okx = (w.in & 1u) != 0, okd = (w.in & 2u) != 0;
        float xa[4], dd[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            xa[m] = okx && kx[m] ? w.x[m] : 0.0f;
            if (relu_x) xa[m] = fmaxf(xa[m], 0.0f);
            dd[m] = okd && nd[m] ? w.d[m] : 0.0f;
            if (Dref) dd[m] = w.r[m] > 0.0f ? dd[m] : 0.0f;
            dbsum[m] += dd[m];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][c] = QB_MFMA16F(xa[a], dd[c], acc[a][c]);
    };
    Raw b0, b1;
    int64_t st = (int64_t)blockIdx.x * 16 + wave;
    load(st, b0);
    for (; st < nstep; st += 2 * stride) {
        load(st + stride, b1);
        compute(b0);
        load(st + 2 * stride, b0);
        compute(b1);
    }
    // MFMA output: acc[a][c][r] of lane (i, g) is row 4 g + r, column i of tile (a, c)
    float sm[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        sm[m] = dbsum[m];
        sm[m] += __shfl_xor(sm[m], 16, 64);
        sm[m] += __shfl_xor(sm[m], 32, 64);
    }
    float* red = red8 + (wave & (kXtdTiles - 1)) * (64 * 64 + 64);
    for (int ph = 0; ph < 2; ++ph) {
        if ((wave >> 3) == ph) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = vec ? 4 * (4 * g + r) + a : 16 * a + 4 * g + r;
                        const int col = vec ? 4 * i + c : 16 * c + i;
                        float* p = red + row * 64 + col;
                        *p = ph == 0 ? acc[a][c][r] : *p + acc[a][c][r];
                    }
            if (g == 0) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float* p = red + 64 * 64 + (vec ? 4 * i + m : 16 * m + i);
                    *p = ph == 0 ? sm[m] : *p + sm[m];
                }
            }
        }
        __syncthreads();
    }
    float* out = partial + (int64_t)blockIdx.x * (64 * 64 + 64);
    for (int e = threadIdx.x; e < 64 * 64 + 64; e += 1024) {
        float t[kXtdTiles];
#pragma unroll
        for (int b = 0; b < kXtdTiles; ++b) t[b] = red8[b * (64 * 64 + 64) + e];
        out[e] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
}

// The eight waves of an xtd9 workgroup (2 row groups x 4 output quadrants) put their nine tap gradients into the
// slabs: per tap, row group 0 writes its quadrants, row group 1 adds (fixed order), all threads store the slab.
__device__ __forceinline__ void xtd9_store(float* __restrict__ red, float* __restrict__ partial,
                                           const f32x4 (&acc)[9][2][2], const float (&dbsum)[2], int lane, int wave) {
    const int g = lane >> 4, i = lane & 15;
    const int rg = wave >> 2, qa = (wave >> 1) & 1, qc = wave & 1;
    const int cd = 32 * qc + 2 * i;
    float sm[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        sm[m] = dbsum[m];
        sm[m] += __shfl_xor(sm[m], 16, 64);
        sm[m] += __shfl_xor(sm[m], 32, 64);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            if (rg == ph) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* p = red + (32 * qa + 2 * (4 * g + r) + a) * 64 + 32 * qc + 2 * i + c;
                            *p = ph == 0 ? acc[t][a][c][r] : *p + acc[t][a][c][r];
                        }
                if (qa == 0 && g == 0) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        float* p = red + 64 * 64 + cd + m;
                        const float val = t == 0 ? sm[m] : 0.0f;
                        *p = ph == 0 ? val : *p + val;
                    }
                }
            }
            __syncthreads();
        }
        float* out = partial + ((int64_t)t * gridDim.x + blockIdx.x) * (64 * 64 + 64);
        for (int e = threadIdx.x; e < 64 * 64 + 64; e += 512) out[e] = red[e];
        __syncthreads();
    }
}


// All nine tap gradients of a 3x3x1 kernel in one pass over the rows: dK[tap] = X[nbr(., tap)]^T D.
// A wave splits a 4-voxel step's flat index into crop coordinates once, reads its D half-row once and the
// nine neighbour half-rows of X, and issues 36 MFMAs -- the per-step address arithmetic, which bounds the
// one-tap-per-block form (about 100 VALU instructions against 4 MFMAs), is amortised over the taps.
// 512 threads = 8 waves = 2 row groups x 4 output quadrants (144 accumulator registers per lane); partials
// as xtd_kernel with gridDim.y == 9: tap t at partial + t * gridDim.x * (64*64 + 64), db in tap 0's slab.
__global__ __launch_bounds__(512) void xtd9_kernel(const float* __restrict__ X, int ldx, int kdim,
                                                   const float* __restrict__ D, int ldd, int ndim,
                                                   float* __restrict__ partial, int64_t N, Gather gt,
                                                   int relu_x) {
    __shared__ float red[64 * 64 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int rg = wave >> 2, qa = (wave >> 1) & 1, qc = wave & 1;
    f32x4 acc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[t][a][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float dbsum[2] = {0.0f, 0.0f};
    const int cx = 32 * qa + 2 * i, cd = 32 * qc + 2 * i;
    const bool vec = ((ldx | ldd) & 1) == 0 &&
                     ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(D)) & 7) == 0;
    const int64_t nstep = (N + 3) / 4;
    const int64_t stride = (int64_t)gridDim.x * 2;
    // Loads are unconditional (a padded tap reads the voxel's own row, columns beyond kdim / ndim stay inside
    // the row stride) and their zeroing is deferred to the point of use: a select right behind a load makes
    // the compiler wait for it, which would serialise the ten loads of a step.
    struct Raw {
        float2 x[9], d;
        unsigned in;  // bit t: tap t inside the crop; bit 9: row valid
    };
    auto load = [&](int64_t st, Raw& w) {
        const int64_t v0 = st * 4 + g;
        const bool ok = st < nstep && v0 < N;
        const int64_t v = ok ? v0 : 0;
        int64_t q2, q3, q4;
        int z, y, x;
        divmod(v, gt.Z, gt.iZ, q2, z);
        divmod(q2, gt.Y, gt.iY, q3, y);
        divmod(q3, gt.X, gt.iX, q4, x);
        (void)z;
        (void)q4;
        w.in = ok ? 512u : 0u;
        const float* dr = D + v * ldd + cd;
        if (vec) w.d = *reinterpret_cast<const float2*>(dr);
        else w.d = make_float2(dr[0], dr[1]);
        // one base pointer per voxel and nine uniform row offsets; the neighbour is inside the crop iff its
        // x-neighbour and its y-neighbour are (three + three comparisons instead of four per tap)
        const float* x0 = X + v * ldx + cx;
        const bool okx[3] = {ok && x > 0, ok, ok && x + 1 < gt.X};
        const bool oky[3] = {y > 0, true, y + 1 < gt.Y};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const bool in = okx[t / 3] && oky[t % 3];
            w.in |= in ? 1u << t : 0u;
            const int64_t off = ((int64_t)(t / 3 - 1) * gt.Y + (t % 3 - 1)) * gt.Z * ldx;   // wave-uniform
            const float* xr = x0 + (in ? off : 0);
            if (vec) w.x[t] = *reinterpret_cast<const float2*>(xr);
            else w.x[t] = make_float2(xr[0], xr[1]);
        }
    };
    const bool kx0 = cx < kdim, kx1 = cx + 1 < kdim, nd0 = cd < ndim, nd1 = cd + 1 < ndim;
    auto compute = [&](const Raw& w) {
        const bool ok = (w.in & 512u) != 0;
        const float dd[2] = {ok && nd0 ? w.d.x : 0.0f, ok && nd1 ? w.d.y : 0.0f};
        dbsum[0] += dd[0];
        dbsum[1] += dd[1];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const bool in = (w.in >> t) & 1u;
            float xa[2] = {in && kx0 ? w.x[t].x : 0.0f, in && kx1 ? w.x[t].y : 0.0f};
            if (relu_x) {
                xa[0] = fmaxf(xa[0], 0.0f);
                xa[1] = fmaxf(xa[1], 0.0f);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[t][a][c] = QB_MFMA16F(xa[a], dd[c], acc[t][a][c]);
        }
    };
    // two buffers in rotation, unrolled by two: no register copies out of in-flight loads (see xtd_kernel)
    Raw b0, b1;
    int64_t st = (int64_t)blockIdx.x * 2 + rg;
    load(st, b0);
    for (; st < nstep; st += 2 * stride) {
        load(st + stride, b1);   // the next step's ten half-rows are in flight during these 36 MFMAs
        compute(b0);
        load(st + 2 * stride, b0);
        compute(b1);
    }
    xtd9_store(red, partial, acc, dbsum, lane, wave);
}

// ---- weight gradients on the bf16 matrix pipe: a float32 as three bfloat16 pieces ------------------------------
// x = p0 + p1 + p2 EXACTLY: p0 is x with its low 16 bits cleared (8 significant bits), p1 the same of what is
// left, p2 the rest (24 bits in all; the two subtractions are exact).  bfloat16 keeps float32's exponent, so
// deltas that carry the loss's 1 / sum(mask) (1e-6 and below) need no scaling -- the reason this product cannot
// use the forward's two-half f16 split, whose per-voxel scale does not factor out of a sum over voxels.  Of the
// nine piece products the six of order <= 2 are kept:  x d ~ x0 d0 + x1 d0 + x0 d1 + x1 d1 + x0 d2 + x2 d0
// (dropped: x1 d2 + x2 d1 + x2 d2 <= 2^-23 |x d|), accumulated in float32 by v_mfma_f32_16x16x32_bf16.  A wave's
// MFMA takes K = 32: a lane group holds FOUR voxels x TWO pieces in its eight k slots, so three MFMAs
//   [x0 | x0] . [d1 | d0],   [x1 | x1] . [d1 | d0],   [x2 | x0] . [d0 | d2]
// add all six products of 16 voxels -- 48 matrix-pipe cycles where v_mfma_f32_16x16x4_f32 spends 128 on the
// vector pipe.
typedef uint32_t u32x4b __attribute__((ext_vector_type(4)));
struct Bf3 {
    uint32_t p0[2], p1[2], p2[2];   // piece p of k slots (0, 1) and (2, 3), two bfloat16 per register
};
__device__ __forceinline__ uint32_t bf_pack(float lo, float hi) {   // the upper halves of two float32
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ void bf3_split(const float (&x)[4], Bf3& o) {
    float r[4], c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = x[j] - __uint_as_float(__float_as_uint(x[j]) & 0xffff0000u);
        c[j] = r[j] - __uint_as_float(__float_as_uint(r[j]) & 0xffff0000u);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        o.p0[h] = bf_pack(x[2 * h], x[2 * h + 1]);
        o.p1[h] = bf_pack(r[2 * h], r[2 * h + 1]);
        o.p2[h] = bf_pack(c[2 * h], c[2 * h + 1]);
    }
}
// acc += X^T D over the 16 voxels whose pieces the wave holds (x: the row operand's tile, d: the column operand's).
// The repeated pieces sit on the row operand: the column operand's two forms are windows of six registers.
struct Bf3Cols {
    uint32_t q[6];   // d1 | d0 | d2
};
__device__ __forceinline__ void bf3_cols(const float (&d)[4], Bf3Cols& o) {
    Bf3 t;
    bf3_split(d, t);
    o.q[0] = t.p1[0]; o.q[1] = t.p1[1]; o.q[2] = t.p0[0]; o.q[3] = t.p0[1]; o.q[4] = t.p2[0]; o.q[5] = t.p2[1];
}
__device__ __forceinline__ f32x4 bf3_mfma(const Bf3& x, const Bf3Cols& d, f32x4 acc) {
    const qb::bf16x8 a00 = __builtin_bit_cast(qb::bf16x8, u32x4b{x.p0[0], x.p0[1], x.p0[0], x.p0[1]});
    const qb::bf16x8 a11 = __builtin_bit_cast(qb::bf16x8, u32x4b{x.p1[0], x.p1[1], x.p1[0], x.p1[1]});
    const qb::bf16x8 a20 = __builtin_bit_cast(qb::bf16x8, u32x4b{x.p2[0], x.p2[1], x.p0[0], x.p0[1]});
    const qb::bf16x8 b10 = __builtin_bit_cast(qb::bf16x8, u32x4b{d.q[0], d.q[1], d.q[2], d.q[3]});
    const qb::bf16x8 b02 = __builtin_bit_cast(qb::bf16x8, u32x4b{d.q[2], d.q[3], d.q[4], d.q[5]});
    acc = QB_MFMA_BF16(a00, b10, acc);    // x0 d1 + x0 d0
    acc = QB_MFMA_BF16(a11, b10, acc);    // x1 d1 + x1 d0
    return QB_MFMA_BF16(a20, b02, acc);   // x2 d0 + x0 d2
}

// ---- the same products on the f16 matrix pipe: a float32 as TWO f16 halves, deltas under a running scale --------
// x = hi + lo, hi = f16(x), lo = f16(x - hi) UNSCALED (22 significant bits while lo is normal, i.e. |x| >= 2^-3 of
// the f16 unit; below that an absolute 2^-25).  All four half products are kept and land in ONE accumulator: with
// K = 32 = 16 voxels x two halves,  [xh | xh] . [dh | dl]  and  [xl | xl] . [dh | dl]  are two MFMAs where the
// three-piece bf16 form takes three, and a split costs 1.5 vector instructions per value (a packed conversion and
// one mixed-precision FMA) where the bf16 pieces cost 5.5 -- on kernels whose time is those instructions.
// What made this form impossible in round 3 -- deltas carry the loss's 1 / sum(mask), far under f16's range, and a
// per-voxel scale does not factor out of a sum over voxels -- is met by a scale per WAVE that only ever falls: the
// deltas are multiplied by s (a power of two) before the split and the wave's accumulators hold s x the sums.  A step
// whose largest |delta| x s reaches 2^14 (one v_max3 per pair, one compare, one scalar branch) takes the slow path
// once: the wave's maximum sets s so that it lands in [2^11, 2^12) and the accumulators are rescaled (exact).  s
// starts at 2^60, so the first step with a delta above 1e-14 sets it; the expected number of later events is the
// number of record highs of a sequence, ~ ln(steps).  Smaller deltas that follow keep 2^-25 x 2^-12 of the
// largest seen as their absolute error -- nothing in a sum the large ones dominate.  Activations are taken as they
// are: the forward's own limit (|x| < 65504, include/qbold_hip.h) applies, beyond it the gradient is NaN, not a clamp.
struct H2 {
    uint32_t h[2], l[2];   // halves of k slots (0, 1) and (2, 3), two f16 per register
};
__device__ __forceinline__ void h2_split(const float (&x)[4], H2& o) {
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
        const float a = x[2 * p2], b = x[2 * p2 + 1];
        const uint32_t h = __builtin_bit_cast(uint32_t, qb::f16x2{(_Float16)a, (_Float16)b});
        uint32_t l;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
        o.h[p2] = h;
        o.l[p2] = l;
    }
}
// The repeated halves sit on the row operand (short-lived: one tap's or one tile's), the column operand -- kept for
// a whole step -- is four registers: dh | dl.
struct H2Rows {
    uint32_t q[8];   // xh | xh, xl | xl
};
__device__ __forceinline__ void h2_rows(const float (&x)[4], H2Rows& o) {
    H2 t;
    h2_split(x, t);
    o.q[0] = t.h[0]; o.q[1] = t.h[1]; o.q[2] = t.h[0]; o.q[3] = t.h[1];
    o.q[4] = t.l[0]; o.q[5] = t.l[1]; o.q[6] = t.l[0]; o.q[7] = t.l[1];
}
__device__ __forceinline__ void h2_cols(const float (&d)[4], float scale, H2& o) {
    const float ds[4] = {d[0] * scale, d[1] * scale, d[2] * scale, d[3] * scale};
    h2_split(ds, o);
}
__device__ __forceinline__ f32x4 h2_mfma(const H2Rows& x, const H2& d, f32x4 acc) {
    const qb::f16x8 ah = __builtin_bit_cast(qb::f16x8, u32x4b{x.q[0], x.q[1], x.q[2], x.q[3]});
    const qb::f16x8 al = __builtin_bit_cast(qb::f16x8, u32x4b{x.q[4], x.q[5], x.q[6], x.q[7]});
    const qb::f16x8 b = __builtin_bit_cast(qb::f16x8, u32x4b{d.h[0], d.h[1], d.l[0], d.l[1]});
    acc = QB_MFMA_F16(ah, b, acc);     // xh dh + xh dl
    return QB_MFMA_F16(al, b, acc);    // xl dh + xl dl
}
// the operand forms of the weight-gradient kernels: H16 = two f16 halves under the wave's delta scale, else three
// bfloat16 pieces (QBOLD_KSEL_DW_BF16_PIECES)
template <bool H16> struct DwOps;
template <> struct DwOps<false> {
    typedef Bf3 Rows;
    typedef Bf3Cols Cols;
    static __device__ __forceinline__ void rows(const float (&x)[4], Rows& o) { bf3_split(x, o); }
    static __device__ __forceinline__ void cols(const float (&d)[4], float, Cols& o) { bf3_cols(d, o); }
    static __device__ __forceinline__ f32x4 mfma(const Rows& x, const Cols& d, f32x4 acc) { return bf3_mfma(x, d, acc); }
};
template <> struct DwOps<true> {
    typedef H2Rows Rows;
    typedef H2 Cols;
    static __device__ __forceinline__ void rows(const float (&x)[4], Rows& o) { h2_rows(x, o); }
    static __device__ __forceinline__ void cols(const float (&d)[4], float s, Cols& o) { h2_cols(d, s, o); }
    static __device__ __forceinline__ f32x4 mfma(const Rows& x, const Cols& d, f32x4 acc) { return h2_mfma(x, d, acc); }
};
constexpr float kDeltaScale0 = 1.152921504606846976e18f;   // 2^60
constexpr float kDeltaTrip = 16384.0f;                      // 2^14: f16 ends at 65504
// the slow path of the wave's delta scale: m = this lane's largest |delta| of the step; returns the factor the
// accumulators take (new scale / old scale) and sets the new scale
__device__ __forceinline__ float delta_rescale(float m, float& scale) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (!(m < 1e30f)) return 1.0f;   // inf / NaN deltas: they propagate as they are
    int e;
    (void)frexpf(m, &e);             // m = f 2^e, f in [0.5, 1)
    const float ns = ldexpf(1.0f, 12 - e);   // m ns in [2^11, 2^12)
    const float f = ns / scale;
    scale = ns;
    return f;
}

// xtd9_kernel on the bf16 matrix pipe, 16 voxels per step, for layers of U <= 64, U % 4 == 0 units.  A lane group takes FOUR
// CONSECUTIVE voxels: with Z % 4 == 0 they are one z run of one (x, y) column, so a step splits one flat index
// (32-bit multiply-high divisions), tests one neighbourhood and reads every tap's four rows off one offset
// (256-byte strides).  Rows come through a buffer resource of N x 256 bytes: a padded tap or a step beyond the
// batch reads at an offset outside it and the hardware returns zeros -- no selects behind the loads.
// 512 threads = 8 waves = 2 row groups x 4 ROW tiles of the gradient: a wave owns 16 input units (lane i: unit
// 16 qa + i, one float per tap row) against all 64 output units (lane i: units 4 i .. 4 i + 3, one float4 of
// the delta row; column tile c = unit 4 i + c), so every tap row is split into its pieces once per workgroup and
// the delta row once per wave and step: 52 values per step where 2 x 2 quadrants split 80.  144 accumulator
// registers per lane.  The taps are walked with the rows of the next two taps in flight.  Same slabs, same
// fixed-order reduction as xtd9_kernel.
struct Xtd9Pos {
    uint32_t row;   // byte offset of the four voxels' first row, or kOutside
    unsigned in;    // bit t: tap t inside the crop
};
// n = q d + r for n < 2^32 / d:  m = floor(2^32 / d) + 1 overestimates the quotient by at most one
// (d = 1: m does not fit 32 bits and magic32 returns 0, for which the quotient is n itself)
__device__ __forceinline__ uint32_t magic32(int d) { return d > 1 ? (uint32_t)(0x100000000ull / (uint32_t)d) + 1u : 0u; }
__device__ __forceinline__ void divmod32(uint32_t n, uint32_t d, uint32_t m, uint32_t& q, uint32_t& r) {
    q = m ? __umulhi(n, m) : n;
    uint32_t p = q * d;
    if (p > n) { --q; p -= d; }
    r = n - p;
}
template <bool RELU_X, bool H16>
__global__ __launch_bounds__(512) void xtd9b_kernel(const float* __restrict__ X, const float* __restrict__ D,
                                                    float* __restrict__ partial, int64_t N, Gather gt, int U) {
    typedef DwOps<H16> Ops;
    float dscale = H16 ? kDeltaScale0 : 1.0f;
    __shared__ __align__(16) float red[64 * 64 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int rg = wave >> 2, qa = wave & 3;
    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float dbsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // byte offsets inside a row; units beyond U (a multiple of 4) are padding nobody wrote: read outside instead
    const bool xcol = 16 * qa + i < U;
    const uint32_t cx = 4u * (16 * qa + i), cd = 4 * i < U ? 16u * i : kOutside;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (uint32_t)(N * 256), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(D), 0, (uint32_t)(N * 256), 0x00020000);
    const uint32_t nvox = (uint32_t)N;
    const uint32_t nstep = (nvox + 15u) / 16u;
    const uint32_t stride = gridDim.x * 2u;
    const uint32_t mZ = magic32(gt.Z), mY = magic32(gt.Y), mX = magic32(gt.X);
    auto locate = [&](uint32_t st, Xtd9Pos& p) {
        const uint32_t v0 = st * 16u + 4u * g;
        const bool ok = st < nstep && v0 < nvox;   // N is a multiple of Z, Z of 4: four voxels or none
        uint32_t q2, q3, q4, z, y, x;
        divmod32(ok ? v0 : 0u, (uint32_t)gt.Z, mZ, q2, z);
        divmod32(q2, (uint32_t)gt.Y, mY, q3, y);
        divmod32(q3, (uint32_t)gt.X, mX, q4, x);
        (void)z;
        (void)q4;
        const bool okx[3] = {ok && x > 0, ok, ok && (int)x + 1 < gt.X};
        const bool oky[3] = {y > 0, true, (int)y + 1 < gt.Y};
        p.in = 0u;
#pragma unroll
        for (int t = 0; t < 9; ++t) p.in |= okx[t / 3] && oky[t % 3] ? 1u << t : 0u;
        p.row = ok ? v0 * 256u : kOutside;
    };
    auto load_tap = [&](const Xtd9Pos& p, int t, float (&w)[4]) {
        const int32_t off = ((t / 3 - 1) * gt.Y + (t % 3 - 1)) * gt.Z * 256;   // wave-uniform
        const uint32_t o = ((p.in >> t) & 1u) && xcol ? p.row + (uint32_t)off + cx : kOutside;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, o + 256u * j, 0, 0));
    };
    auto load_d = [&](const Xtd9Pos& p, f32x4 (&w)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)   // row (a multiple of 1024) | cd (< 256): their sum, or outside if either is
            w[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, (p.row | cd) + 256u * j, 0, 0));
    };
    // one 16-voxel step.  The tap rows travel through a ring of three register sets, two taps ahead of their use (at
    // the end of a step: the next step's first two taps); nine taps: tap t always sits in set t % 3.
    constexpr int kXtdAhead = 2;
    float xb[3][4];
    f32x4 dw[4];
    auto step = [&](const Xtd9Pos& cur, const Xtd9Pos& nxt) {
        typename Ops::Cols dp[4];
        if constexpr (H16) {
            float m = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) m = fmaxf(m, fabsf(dw[j][c]));
            if (__builtin_amdgcn_ballot_w64(m * dscale >= kDeltaTrip) != 0) {
                const float f = delta_rescale(m, dscale);
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[t][c] *= f;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float dd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dd[j] = dw[j][c];
                dbsum[c] += dd[j];
            }
            Ops::cols(dd, dscale, dp[c]);
        }
        load_d(nxt, dw);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + kXtdAhead < 9) load_tap(cur, t + kXtdAhead, xb[(t + kXtdAhead) % 3]);
            else load_tap(nxt, t + kXtdAhead - 9, xb[(t + kXtdAhead) % 3]);
            float xa[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xa[j] = RELU_X ? relu_1op(xb[t % 3][j]) : xb[t % 3][j];
            typename Ops::Rows xp;
            Ops::rows(xa, xp);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][c] = Ops::mfma(xp, dp[c], acc[t][c]);
        }
    };
    Xtd9Pos p0, p1;
    uint32_t st = blockIdx.x * 2u + rg;
    locate(st, p0);
    load_d(p0, dw);
#pragma unroll
    for (int t = 0; t < kXtdAhead; ++t) load_tap(p0, t, xb[t]);
    for (; st < nstep; st += 2 * stride) {   // two steps per trip: the positions alternate without a copy
        locate(st + stride, p1);
        step(p0, p1);
        locate(st + 2 * stride, p0);
        step(p1, p0);
    }
    if constexpr (H16) {   // the wave's accumulators hold dscale x the sums
        const float inv = 1.0f / dscale;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][c] *= inv;
    }
    // acc[t][c][r] of lane (g, i): input unit 16 qa + 4 g + r, output unit 4 i + c
    float sm[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        sm[c] = dbsum[c];
        sm[c] += __shfl_xor(sm[c], 16, 64);
        sm[c] += __shfl_xor(sm[c], 32, 64);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            if (rg == ph) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float4* p = reinterpret_cast<float4*>(red + (16 * qa + 4 * g + r) * 64 + 4 * i);
                    float4 o = make_float4(acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]);
                    if (ph == 1) {
                        const float4 q = *p;
                        o = make_float4(q.x + o.x, q.y + o.y, q.z + o.z, q.w + o.w);
                    }
                    *p = o;
                }
                if (qa == 0 && g == 0) {
                    float4* p = reinterpret_cast<float4*>(red + 64 * 64 + 4 * i);
                    float4 o = t == 0 ? make_float4(sm[0], sm[1], sm[2], sm[3]) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (ph == 1) {
                        const float4 q = *p;
                        o = make_float4(q.x + o.x, q.y + o.y, q.z + o.z, q.w + o.w);
                    }
                    *p = o;
                }
            }
            __syncthreads();
        }
        float* out = partial + ((int64_t)t * gridDim.x + blockIdx.x) * (64 * 64 + 64);
        for (int e = threadIdx.x; e < 64 * 64 + 64; e += 512) out[e] = red[e];
        __syncthreads();
    }
}

// xtd_kernel on the bf16 matrix pipe (three-piece operands), 16 voxels per step: [N][64] row tensors without a
// gather, kdim and ndim multiples of 4.  512 threads = 8 waves, each accumulating the whole 64 x 64 product over
// its share of the steps; a lane group reads four consecutive rows of X and of D as float4 (tile m of lane i =
// column 4 i + m, as xtd_kernel's aligned form), through buffer resources of N x 256 bytes: rows beyond the batch
// and padding columns read as zeros without a select.  Eight 16-byte loads per lane are in flight per step where
// the f32 form has two per 4-voxel step -- at crop-batch sizes (190 k rows) that latency, not the MFMAs, was the
// kernel's time.  Slabs and their fixed-order sum as xtd_kernel (one 16 KiB tile per wave, added in wave order).
template <bool RELU_X, bool DREF, bool H16>   // DREF: D is taken as D * (Dref > 0), the step through a relu folded into the load
__global__ __launch_bounds__(512) void xtdb_kernel(const float* __restrict__ X, int kdim, const float* __restrict__ D,
                                                   int ndim, float* __restrict__ partial, int64_t N,
                                                   const float* __restrict__ Dref) {
    typedef DwOps<H16> Ops;
    float dscale = H16 ? kDeltaScale0 : 1.0f;
    extern __shared__ float red8[];  // [kXtdTiles][64 * 64 + 64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float dbsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (uint32_t)(N * 256), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(D), 0, (uint32_t)(N * 256), 0x00020000);
    const uint32_t colx = 4 * i < kdim ? 16u * i : kOutside, cold = 4 * i < ndim ? 16u * i : kOutside;
    const uint32_t nstep = ((uint32_t)N + 15u) / 16u;
    const uint32_t stride = gridDim.x * 8u;
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DREF ? Dref : D), 0, (uint32_t)(N * 256), 0x00020000);
    struct Raw {
        f32x4 x[4], d[4], r[DREF ? 4 : 1];
    };
    auto load = [&](uint32_t st, Raw& w) {   // a step beyond the batch reads beyond the buffers (N < 2^23: no wrap)
        const uint32_t row = (st * 16u + 4u * g) * 256u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w.x[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, row + colx + 256u * j, 0, 0));
            w.d[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, row + cold + 256u * j, 0, 0));
            if (DREF) w.r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, row + cold + 256u * j, 0, 0));
        }
    };
    auto compute = [&](const Raw& w) {
        typename Ops::Rows xp[4];
        typename Ops::Cols dp[4];
        if constexpr (H16) {   // (a delta the Dref mask removes counts: it can only make the scale more careful)
            float mx = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < 4; ++m) mx = fmaxf(mx, fabsf(w.d[j][m]));
            if (__builtin_amdgcn_ballot_w64(mx * dscale >= kDeltaTrip) != 0) {
                const float f = delta_rescale(mx, dscale);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[a][c] *= f;
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float xa[4], dd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xa[j] = RELU_X ? relu_1op(w.x[j][m]) : w.x[j][m];
                dd[j] = DREF ? (w.r[j][m] > 0.0f ? w.d[j][m] : 0.0f) : w.d[j][m];
                dbsum[m] += dd[j];
            }
            Ops::rows(xa, xp[m]);
            Ops::cols(dd, dscale, dp[m]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][c] = Ops::mfma(xp[a], dp[c], acc[a][c]);
    };
    Raw b0, b1;   // two sets in rotation, the loop unrolled by two (see xtd_kernel)
    uint32_t st = blockIdx.x * 8u + wave;
    load(st, b0);
    for (; st < nstep; st += 2 * stride) {
        load(st + stride, b1);
        compute(b0);
        load(st + 2 * stride, b0);
        compute(b1);
    }
    if constexpr (H16) {   // the wave's accumulators hold dscale x the sums
        const float inv = 1.0f / dscale;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][c] *= inv;
    }
    // acc[a][c][r] of lane (g, i): row 4 (4 g + r) + a, column 4 i + c
    float* red = red8 + wave * (64 * 64 + 64);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4*>(red + (4 * (4 * g + r) + a) * 64 + 4 * i) =
                make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
    float sm[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        sm[m] = dbsum[m];
        sm[m] += __shfl_xor(sm[m], 16, 64);
        sm[m] += __shfl_xor(sm[m], 32, 64);
    }
    if (g == 0) *reinterpret_cast<float4*>(red + 64 * 64 + 4 * i) = make_float4(sm[0], sm[1], sm[2], sm[3]);
    __syncthreads();
    float* out = partial + (int64_t)blockIdx.x * (64 * 64 + 64);
    for (int e = threadIdx.x; e < 64 * 64 + 64; e += 512) {
        float t[kXtdTiles];
#pragma unroll
        for (int b = 0; b < kXtdTiles; ++b) t[b] = red8[b * (64 * 64 + 64) + e];
        out[e] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
}

// dW[i * ldw + j] (+)= sum_blk partial[blk][i][j]; db[j] (+)= sum_blk partial[blk][4096 + j]
// gridDim.y == 9: one 3x3x1 kernel -- tap t reads its own partials and writes dW + t * kdim * ndim; the
// bias gradient comes from tap 0 only.
__global__ __launch_bounds__(1024) void slab_reduce_kernel(const float* __restrict__ partial, int nblk,
                                                            float* __restrict__ dW, int ldw, int kdim, int ndim,
                                                            float* __restrict__ db, int accum, int j0) {
    // j0: the slab's columns j0 .. j0 + ndim - 1 go to dW[:, 0 .. ndim - 1] (two heads sharing one slab)
    // 64 elements per block x 16 interleaved parts of the slab range, added in a fixed order
    constexpr int kParts = 16;
    __shared__ double part[kParts][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), p = threadIdx.x >> 6;
    if (gridDim.y == 9) {
        partial += (int64_t)blockIdx.y * nblk * (64 * 64 + 64);
        dW += (int64_t)blockIdx.y * kdim * ndim;
        if (blockIdx.y != 0) db = nullptr;
    }
    double a = 0.0;
    if (e < 64 * 64 + 64)
        for (int bk = p; bk < nblk; bk += kParts) a += (double)partial[(int64_t)bk * (64 * 64 + 64) + e];
    part[p][threadIdx.x & 63] = a;
    __syncthreads();
    if (p != 0 || e >= 64 * 64 + 64) return;
    a = 0.0;
#pragma unroll
    for (int q = 0; q < kParts; ++q) a += part[q][threadIdx.x];
    if (e < 64 * 64) {
        const int i = e >> 6, j = (e & 63) - j0;
        if (i < kdim && j >= 0 && j < ndim) dW[i * ldw + j] = (accum ? dW[i * ldw + j] : 0.0f) + (float)a;
    } else {
        const int j = e - 64 * 64 - j0;
        if (db && j >= 0 && j < ndim) db[j] = (accum ? db[j] : 0.0f) + (float)a;
    }
}

// The same sums for a QUEUE of slabs in one launch (blockIdx.y = job): a crop step's thirteen weight-gradient kernels
// each left a reduction of their own behind (65 blocks, ~8 us, mostly launch and latency); queued, they run as a few
// launches of hundreds of blocks.  Same order of additions per element as slab_reduce_kernel: bitwise the same sums.
struct SlabJob {
    const float* partial;
    float* dW;
    float* db;
    int nblk, ldw, kdim, ndim, j0;
};
constexpr int kMaxSlabJobs = 40;
struct SlabJobs {
    SlabJob job[kMaxSlabJobs];
};
// A thread takes FOUR consecutive elements (one 16-byte load per slab) of one of the sixteen parts, four slabs'
// loads in flight per trip; per element the additions are slab_reduce_kernel's, in its order.
__global__ __launch_bounds__(512) void slab_reduce_jobs_kernel(SlabJobs jobs) {
    // 32 float4 columns x 16 parts per workgroup: 33 workgroups per job, so that a queue of nine jobs (one nine-tap
    // gradient) is 297 workgroups -- with 64 columns per workgroup it was 153 on 256 CUs
    constexpr int kParts = 16, kCols = 32, kSlab = 64 * 64 + 64;
    __shared__ double part[kParts][kCols][4];
    const SlabJob jb = jobs.job[blockIdx.y];
    const int t = threadIdx.x & (kCols - 1), p = threadIdx.x / kCols;
    const int e4 = blockIdx.x * kCols + t;       // float4 index within a slab
    const bool in = e4 * 4 < kSlab;              // kSlab is a multiple of 4
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    if (in) {
        const float4* src = reinterpret_cast<const float4*>(jb.partial) + e4;
        for (int bk = p; bk < jb.nblk; bk += 4 * kParts) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = bk + u * kParts;
                v[u] = src[(int64_t)(b < jb.nblk ? b : bk) * (kSlab / 4)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (bk + u * kParts < jb.nblk) {
                    a[0] += (double)v[u].x; a[1] += (double)v[u].y; a[2] += (double)v[u].z; a[3] += (double)v[u].w;
                }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) part[p][t][c] = a[c];
    __syncthreads();
    // thread (t, c = p) of the first four part rows finishes element 4 e4 + c
    if (p >= 4 || !in) return;
    double sum = 0.0;
#pragma unroll
    for (int q = 0; q < kParts; ++q) sum += part[q][t][p];
    const int e = 4 * e4 + p;
    if (e < 64 * 64) {
        const int i = e >> 6, j = (e & 63) - jb.j0;
        if (i < jb.kdim && j >= 0 && j < jb.ndim) jb.dW[i * jb.ldw + j] = (float)sum;
    } else {
        const int j = e - 64 * 64 - jb.j0;
        if (jb.db && j >= 0 && j < jb.ndim) jb.db[j] = (float)sum;
    }
}

// Stacked head weights for the backward-data GEMM of both heads in one pass: Wst[k][j] (k < 5 + T rows of
// U columns) = Wf[j][k] for k < 5, Ws[j][k - 5] beyond -- d last = [g_q | g_ls] Wst.
__global__ void stack_heads_kernel(const float* __restrict__ Wf, const float* __restrict__ Ws, int T, int U,
                                   float* __restrict__ Wst) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (5 + T) * U) return;
    const int k = e / U, j = e % U;
    Wst[e] = k < 5 ? Wf[j * 5 + k] : Ws[j * T + (k - 5)];
}

// Both heads' backward in one pass over the rows, for 5 + T <= 16 head outputs and U % 4 == 0 units:
//   d last = [g_q | g_ls] / sum(mask) . [Wf^T; Ws^T]      and      d[Wf | Ws] = last^T [g_q | g_ls] / sum(mask),
// which the layer-wise form runs as head_delta_kernel + stack_heads_kernel + a GEMM + a weight-gradient pass: the
// 16-column delta tensor written and read twice, `last` read once more.  Here a wave takes 16 voxels per tile and reads
// the two head gradients straight from g_q / g_ls (twice, in the two operand layouts: 64 bytes per voxel, cache
// hits), `last` once, and writes d last: 0.6 GB per million voxels instead of 1.2.  Exact float32 products
// (v_mfma_f32_16x16x4_f32: K = 16 head outputs one way, 16 voxels the other -- 32 MFMAs per tile, far below the
// memory time).  The weight gradients leave as one xtd-format slab per workgroup ([unit][64] + 64 bias sums, of
// which columns 0 .. 15 are written), summed by slab_reduce_kernel in a fixed order like every other slab.
__global__ __launch_bounds__(512) void heads_bwd_kernel(const float* __restrict__ g_q, const float* __restrict__ g_ls,
                                                         int T, const double* __restrict__ sums,
                                                         const float* __restrict__ last, const float* __restrict__ Wf,
                                                         const float* __restrict__ Ws, int U, float* __restrict__ d_last,
                                                         float* __restrict__ partial, int64_t N) {
    constexpr int kTile = 64 * 16 + 16;   // one wave's [unit][16] gradients + 16 bias sums
    __shared__ float red[8][kTile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const float scale = sums ? (float)(1.0 / sums[2]) : 1.0f;   // 1 / sum(mask)
    const int H = 5 + T;
    // stacked head weights as the row operand of d last^T = Wst^T dA^T: unit 16 t + i, head output 4 s + g
    float wst[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int u = 16 * t + i, k = 4 * s4 + g;
            wst[t][s4] = u < U ? (k < 5 ? Wf[u * 5 + k] : (k < H ? Ws[u * T + k - 5] : 0.0f)) : 0.0f;
        }
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(last), 0, (uint32_t)(N * 256), 0x00020000);
    const uint32_t coll = 4 * i < U ? 16u * i : kOutside;
    // head output k of voxel v: g_q[5 v + k] or g_ls[T v + k - 5], zero beyond 5 + T and beyond the batch -- one load
    // from each array through a buffer resource, the one that does not apply at an offset outside it (returns 0):
    // no branch around a load, all of a tile's loads in flight together
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g_q), 0, (uint32_t)(N * 20), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g_ls), 0, (uint32_t)(N * 4 * T), 0x00020000);
    auto col_q = [&](int k) { return k < 5 ? 4u * k : kOutside; };
    auto col_s = [&](int k) { return k >= 5 && k < H ? 4u * (k - 5) : kOutside; };
    uint32_t q1[4], s1[4];   // columns 4 s + g (the data product's operand)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        q1[s4] = col_q(4 * s4 + g);
        s1[s4] = col_s(4 * s4 + g);
    }
    const uint32_t q2 = col_q(i), s2 = col_s(i);   // column i (the gradient product's operand)
    const uint32_t rowq = 20u, rows = 4u * T;
    auto delta = [&](uint32_t v, uint32_t cq, uint32_t cs) {   // v < 2^23 + 2^16: no 32-bit wrap, rows beyond N are outside
        const float a = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rq, v * rowq + cq, 0, 0));
        const float b = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, v * rows + cs, 0, 0));
        return a + b;
    };
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float dbsum = 0.0f;
    struct Raw {
        float d1[4], d2[4];   // the delta as column operand of the data product (voxel i, output 4 s + g) and of the
        f32x4 x[4];           // gradient product (voxel 4 s + g, output i); last rows 4 s + g, units 4 i ..
    };
    auto load = [&](int64_t tile, Raw& w) {
        const uint32_t v0 = (uint32_t)tile * 16u;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            w.d1[s4] = delta(v0 + i, q1[s4], s1[s4]);
            w.d2[s4] = delta(v0 + 4 * s4 + g, q2, s2);
            // (rows beyond the batch are beyond the buffer)
            w.x[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, (v0 + 4 * s4 + g) * 256u + coll, 0, 0));
        }
    };
    const int64_t ntile = (N + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * 8;
    auto compute = [&](int64_t tile, const Raw& w) {
        // d last^T [unit][voxel]
        f32x4 o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            o[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) o[t] = QB_MFMA16F(wst[t][s4], w.d1[s4] * scale, o[t]);
        }
        const int64_t v = tile * 16 + i;
        if (v < N) {
#pragma unroll
            for (int t = 0; t < 4; ++t)   // o[t][r]: unit 16 t + 4 g + r of voxel i
                *reinterpret_cast<float4*>(d_last + v * 64 + 16 * t + 4 * g) = make_float4(o[t][0], o[t][1], o[t][2], o[t][3]);
        }
        // d [Wf | Ws][unit][output] += last^T delta over the tile's voxels
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float d2 = w.d2[s4] * scale;
            dbsum += d2;
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = QB_MFMA16F(w.x[s4][m], d2, acc[m]);
        }
    };
    Raw b0, b1;   // two sets in rotation, the loop unrolled by two (see xtd_kernel)
    int64_t tile = (int64_t)blockIdx.x * 8 + wave;
    load(tile < ntile ? tile : ntile, b0);
    for (; tile < ntile; tile += 2 * stride) {
        load(tile + stride, b1);
        compute(tile, b0);
        load(tile + 2 * stride, b0);
        if (tile + stride < ntile) compute(tile + stride, b1);
    }
    // acc[m][r] of lane (g, i): unit 4 (4 g + r) + m, head output i.  One 4 KiB tile per wave, added in wave order.
    dbsum += __shfl_xor(dbsum, 16, 64);
    dbsum += __shfl_xor(dbsum, 32, 64);
    float* tl = red[wave];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) tl[(4 * (4 * g + r) + m) * 16 + i] = acc[m][r];
    if (g == 0) tl[64 * 16 + i] = dbsum;
    __syncthreads();
    float* out = partial + (int64_t)blockIdx.x * (64 * 64 + 64);
    for (int e = threadIdx.x; e < kTile; e += 512) {
        float t8[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) t8[b] = red[b][e];
        const float sum = ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
        out[e < 64 * 16 ? (e >> 4) * 64 + (e & 15) : 64 * 64 + (e - 64 * 16)] = sum;
    }
}

// normalise_data into a [N][64] slot (model.py:97-113)
__global__ void normalise64_kernel(QbDev c, const float* __restrict__ x, float* __restrict__ out,
                                   int ld, int64_t N) {
    // one thread per (voxel, tau): coalesced reads and 4 wc-byte row heads written; columns T .. wc - 1 are
    // zeroed, the rest of a row is never used (its readers take kdim = T columns and zero the others by select)
    const int T = c.T, se = c.se_idx;
    const int wc = ((T + 3) & ~3) < ld ? ((T + 3) & ~3) : ld;
    const int64_t total = N * wc;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(e % wc);
        const int64_t v = e / wc;
        const float* xv = x + v * T;
        float o = 0.0f;
        if (t < T) {
            float den;
            if (c.multi_norm)
                den = (qb::clampf_(xv[se - 1], 1e-2f, 1e8f) + qb::clampf_(xv[se], 1e-2f, 1e8f) +
                       qb::clampf_(xv[se + 1], 1e-2f, 1e8f)) / 3.0f;
            else
                den = qb::clampf_(xv[se], 1e-2f, 1e8f);
            o = logf(qb::clampf_(xv[t], 1e-2f, 1e8f) / den);
        }
        out[v * ld + t] = o;
    }
}

// gate: gl <- g = sigmoid(gl + offset);  bout = skip (1 - g) + r g          (model.py:167-170)
__global__ void gate_fwd_kernel(float* __restrict__ gl, const float* __restrict__ skip,
                                const float* __restrict__ r, float* __restrict__ bout, float offset,
                                int U, int G, int ld, int64_t N) {
    const int64_t total = N * ld;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % ld);
        const int64_t v = e / ld;
        float o = 0.0f;
        if (j < U) {
            const float gate = 1.0f / (1.0f + expf(-(gl[v * ld + (G == 1 ? 0 : j)] + offset)));
            o = skip[e] * (1.0f - gate) + r[e] * gate;
        }
        bout[e] = o;
    }
}
// gate backward: d_b in/out -> d_skip_pre (masked by skip > 0; relu_mask = 0: d skip itself, the gelu path applies
// its own derivative), d_r, d_gl (channelwise or summed)
__global__ void gate_bwd_kernel(const float* __restrict__ d_b, const float* __restrict__ gl,
                                const float* __restrict__ skip, const float* __restrict__ r,
                                float* __restrict__ d_skip_pre, float* __restrict__ d_r,
                                float* __restrict__ d_gl, float offset, int U, int G, int ld, int64_t N,
                                int relu_mask = 1) {
    for (int64_t v = blockIdx.x * (int64_t)(blockDim.x / 64) + (threadIdx.x >> 6); v < N;
         v += (int64_t)gridDim.x * (blockDim.x / 64)) {
        float shared_sum = 0.0f;
        for (int j = threadIdx.x & 63; j < ld; j += 64) {
            const int64_t e = v * ld + j;
            float dgl = 0.0f, ds = 0.0f, dr = 0.0f;
            if (j < U) {
                const float gate = 1.0f / (1.0f + expf(-(gl[v * ld + (G == 1 ? 0 : j)] + offset)));
                const float db = d_b[e];
                ds = (!relu_mask || skip[e] > 0.0f) ? db * (1.0f - gate) : 0.0f;
                dr = db * gate;
                dgl = db * (r[e] - skip[e]) * gate * (1.0f - gate);
            }
            d_skip_pre[e] = ds;
            d_r[e] = dr;
            if (G == 1) shared_sum += dgl;  // shared gate: sum the contributions of all units
            else d_gl[e] = dgl;
        }
        if (G == 1) {
            shared_sum = qb::wave_sum(shared_sum);
            for (int j = threadIdx.x & 63; j < ld; j += 64) d_gl[v * ld + j] = j == 0 ? shared_sum : 0.0f;
        }
    }
}


// gate_bwd_kernel and the gating conv's backward-data product in one launch, for channel-wise gates on [N][64] rows
// (U = G <= 64, U % 4 == 0): the crop backward's first two kernels read d b, gl, skip, r and wrote d s', d r, d gl,
// then re-read d gl and d r to add d gl Wg^T -- ten tensor passes, here seven.  A wave takes 16 voxels.  In the
// product's INPUT layout (lane (i, g): voxel i, units 16 q + 4 g ..) it forms d gl and d s' from the four rows and
// stores them; d gl feeds the 64 exact-f32 MFMAs of xw64_kernel as it is formed.  d r = d b g + d gl Wg^T leaves in the
// product's OUTPUT layout (lane (i, g): voxels 4 g + r, units 4 i ..), where d b and gl are read a second time (the
// rows the tile just brought in: cache hits) and the gate evaluated again -- requested before the MFMAs.
__global__ __launch_bounds__(256) void gate_bwd_wg_kernel(const float* __restrict__ d_b, const float* __restrict__ gl,
                                                          const float* __restrict__ skip, const float* __restrict__ r,
                                                          const float* __restrict__ Wg, float* __restrict__ d_skip_pre,
                                                          float* __restrict__ d_r, float* __restrict__ d_gl,
                                                          float offset, int U, int64_t N) {
    extern __shared__ float Wl[];
    constexpr int ld = kLd;
    const int kpad = (U + 15) & ~15;
    stage_in_flight<256>(
        kpad * 64,
        [&](int e) {   // Wl[k][j] = Wg[j][k]: d r = d gl Wg^T
            const int k = clamp_hi(e >> 6, U), j = clamp_hi(e & 63, U);
            return Wg[j * U + k];
        },
        [&](int e, float v) {
            const int k = e >> 6, j = e & 63;
            Wl[k * kWs + 16 * (j & 3) + (j >> 2)] = k < U && j < U ? v : 0.0f;   // tile m of lane i = column 4 i + m
        });
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int64_t ntile = (N + 15) / 16;
    const int nq = (U + 15) >> 4;
    const int j = 4 * i;
    struct Rows {
        float4 db[4], gl[4], sk[4], r[4];
    };
    auto fetch = [&](int64_t tile, Rows& w) {
        const int64_t t = tile < ntile ? tile : ntile - 1;
        const int64_t v = t * 16 + i < N ? t * 16 + i : N - 1;
        const int64_t o = v * ld + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nq) {
                w.db[q] = *reinterpret_cast<const float4*>(d_b + o + 16 * q);
                w.gl[q] = *reinterpret_cast<const float4*>(gl + o + 16 * q);
                w.sk[q] = *reinterpret_cast<const float4*>(skip + o + 16 * q);
                w.r[q] = *reinterpret_cast<const float4*>(r + o + 16 * q);
            }
    };
    // (no software prefetch of the next tile's rows: three waves per SIMD at 148 registers beat two at 196 with it,
    // 1.449 against 1.455 ms per crop step)
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += stride) {
        const int64_t v0 = tile * 16;
        Rows cur;
        fetch(tile, cur);
        float4 odb[4], ogl[4];   // the output layout's rows of d b and gl
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t v = v0 + 4 * g + rr < N ? v0 + 4 * g + rr : N - 1;
            const int jj = j < U ? j : 0;
            odb[rr] = *reinterpret_cast<const float4*>(d_b + v * ld + jj);
            ogl[rr] = *reinterpret_cast<const float4*>(gl + v * ld + jj);
        }
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const bool row_ok = v0 + i < N;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= nq) break;
            const int k0 = 16 * q + 4 * g;
            const bool in = k0 < U;   // U % 4 == 0: a float4 of units is inside or outside
            const float db4[4] = {cur.db[q].x, cur.db[q].y, cur.db[q].z, cur.db[q].w};
            const float gl4[4] = {cur.gl[q].x, cur.gl[q].y, cur.gl[q].z, cur.gl[q].w};
            const float sk4[4] = {cur.sk[q].x, cur.sk[q].y, cur.sk[q].z, cur.sk[q].w};
            const float r4[4] = {cur.r[q].x, cur.r[q].y, cur.r[q].z, cur.r[q].w};
            float dgl[4], ds[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float gate = qb::sigmoidf_(gl4[c] + offset);
                ds[c] = in && sk4[c] > 0.0f ? db4[c] * (1.0f - gate) : 0.0f;
                dgl[c] = in ? db4[c] * (r4[c] - sk4[c]) * gate * (1.0f - gate) : 0.0f;
            }
            if (row_ok) {
                const int64_t o = (v0 + i) * ld + k0;
                *reinterpret_cast<float4*>(d_gl + o) = make_float4(dgl[0], dgl[1], dgl[2], dgl[3]);
                *reinterpret_cast<float4*>(d_skip_pre + o) = make_float4(ds[0], ds[1], ds[2], ds[3]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float* wr = Wl + (k0 + c) * kWs + i;
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = QB_MFMA16F(dgl[c], wr[16 * m], acc[m]);
            }
        }
        // padding units of the rows (U .. 63): zeros, as gate_bwd_kernel leaves them
        if (row_ok)
            for (int q = nq; q < 4; ++q) {
                const int64_t o = (v0 + i) * ld + 16 * q + 4 * g;
                *reinterpret_cast<float4*>(d_gl + o) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                *reinterpret_cast<float4*>(d_skip_pre + o) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t v = v0 + 4 * g + rr;
            if (v >= N) continue;
            float y[4];
            const float db4[4] = {odb[rr].x, odb[rr].y, odb[rr].z, odb[rr].w};
            const float gl4[4] = {ogl[rr].x, ogl[rr].y, ogl[rr].z, ogl[rr].w};
#pragma unroll
            for (int m = 0; m < 4; ++m) y[m] = j < U ? fmaf(db4[m], qb::sigmoidf_(gl4[m] + offset), acc[m][rr]) : 0.0f;
            *reinterpret_cast<float4*>(d_r + v * ld + j) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
}

// ---- data side of one gated block's backward in ONE launch ------------------------------------------------
// For voxel batches of the LDS-resident shapes.  Per 16-voxel wave tile (the lane layout of encoder_core.h: a
// lane holds four consecutive units of its voxel per 16-row tile, so [N][64] rows load and store as float4):
// read b_in and d b_out, RECOMPUTE the block (skip, t, r, gate: four dense64 on the block's forward image --
// bit-identical to the forward that ran, so the relu masks agree), form the element-wise deltas, run the four
// backward-data products on the TRANSPOSED image and write the four deltas the weight-gradient kernels need
// plus d b_in:
//   d gl  = d b (r - skip) g (1 - g)                         -> xtd(r, d gl)       = dWg
//   d r   = d b g + d gl Wg^T                                -> xtd(t, d r)        = dWr2
//   d t'  = (d r Wr2^T) (t > 0)                              -> xtd(relu b_in, d t') = dWr1
//   d s'  = d b (1 - g) (skip > 0)                           -> xtd(b_in, d s')    = dWc
//   d b_in = (d t' Wr1^T) (b_in > 0) + d s' Wc^T
// 2 tensor reads + 5 writes instead of the 17 passes of gate_bwd_kernel + three xw64 launches.  Deltas carry
// the 1 / sum(mask) of the loss (1e-6 and below): each product's input is scaled per voxel by a power of two
// into [2^13, 2^14) before the f16 split and the output scaled back, exact in both directions.
// eight waves per workgroup (one per CU: the two images take 133 KB of LDS): 168 VGPRs, no scratch.  Measured per
// 1 M-voxel step: 1024 threads (128 VGPRs, 116 B of scratch per lane) 3.47 ms, 768 3.24, 512 3.15, 256 3.39.
#ifndef QB_BLK_THREADS
#define QB_BLK_THREADS 512
#endif
constexpr int kBlkThreads = QB_BLK_THREADS;
constexpr uint32_t kDropped = 0x80000000u;   // beyond any row buffer (N < 2^23): the hardware drops the store

// (columns U .. 63 of a row tensor are padding nobody is obliged to write: read as zero)
__device__ __forceinline__ void load_rows(const float* __restrict__ slot, uint32_t voff, int g, int U, f32x4 (&a)[4]) {
    const char* sb = reinterpret_cast<const char*>(slot);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float4 q = *reinterpret_cast<const float4*>(sb + voff + 64 * m);
        a[m] = f32x4{q.x, q.y, q.z, q.w};
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[m][k] = 16 * m + 4 * g + k < U ? a[m][k] : 0.0f;
}
typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_rows(__amdgpu_buffer_rsrc_t rs, uint32_t soff, const f32x4 (&a)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, a[m]), rs, soff + 64u * m, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_buffer(float* p, int64_t N) {
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, (uint32_t)(N * 256), 0x00020000);
}
// out = W in (no bias: the transposed image is packed with zero biases) with a per-voxel power-of-two scale;
// `in` is scaled in place (every caller is done with it)
__device__ __forceinline__ void dense64_scaled(const float* __restrict__ A, const float* __restrict__ bias,
                                               f32x4 (&in)[4], f32x4 (&out)[4], int lane) {
    float mx = 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) mx = fmaxf(mx, fabsf(in[m][k]));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));   // the voxel's largest delta (its units sit in four lanes)
    int se = 267 - ((__float_as_int(mx) >> 23) & 0xff);   // 2^(13 - exponent) as a biased exponent
    se = se < 1 ? 1 : (se > 250 ? 250 : se);
    const float sc = __int_as_float(se << 23), inv = __int_as_float((254 - se) << 23);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) in[m][k] *= sc;
    qb::dense64<false>(A, bias, in, out, lane);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) out[m][k] *= inv;
}

__global__ __launch_bounds__(kBlkThreads) void block_bwd_kernel(
    const float* __restrict__ img_f, const float* __restrict__ img_b, const float* __restrict__ b_in,
    const float* d_b, float* d_gl, float* d_r, float* d_t, float* d_s, float* d_bin, int U, int64_t N) {
    extern __shared__ __align__(16) float lds_img[];
    float* F = lds_img;
    float* B = lds_img + qb::BLK_FLOATS;
    qb::copy_to_lds<kBlkThreads>(F, img_f, qb::BLK_FLOATS / 4);
    qb::copy_to_lds<kBlkThreads>(B, img_b, qb::BLK_FLOATS / 4);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    constexpr int NW = kBlkThreads / 64;
    const __amdgpu_buffer_rsrc_t o_gl = row_buffer(d_gl, N), o_r = row_buffer(d_r, N), o_t = row_buffer(d_t, N),
                                 o_s = row_buffer(d_s, N), o_b = row_buffer(d_bin, N);
    const int64_t ntile = (N + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntile; tile += (int64_t)gridDim.x * NW) {
        const int64_t v = tile * 16 + i;
        const bool live = v < N;
        const uint32_t voff = (uint32_t)(live ? v : N - 1) * 256u + 16u * (uint32_t)g;
        const uint32_t soff = live ? voff : kDropped;
        f32x4 b[4], skip[4], t[4], r[4];
        load_rows(b_in, voff, g, U, b);
        uint32_t m_bin = 0u, m_t = 0u;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) m_bin |= b[m][k] > 0.0f ? 1u << (4 * m + k) : 0u;
        // the block again, as encoder_core.h block_stream2 computed it
        qb::dense64<false>(F + qb::BLK_WC_A, F + qb::BLK_WC_B, b, skip, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            skip[m] = qb::relu4(skip[m]);
            b[m] = qb::relu4(b[m]);
        }
        qb::dense64<false>(F + qb::BLK_R1_A, F + qb::BLK_R1_B, b, t, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            t[m] = qb::relu4(t[m]);
#pragma unroll
            for (int k = 0; k < 4; ++k) m_t |= t[m][k] > 0.0f ? 1u << (4 * m + k) : 0u;
        }
        qb::dense64<false>(F + qb::BLK_R2_A, F + qb::BLK_R2_B, t, r, lane);
        qb::dense64<false>(F + qb::BLK_G_A, F + qb::BLK_G_B, r, t, lane);   // gate logits + gate_offset
        load_rows(d_b, voff, g, U, b);
        // element-wise deltas: r <- d gl, skip <- d s', b <- d b g
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gate = qb::sigmoidf_(t[m][k]);
                const float db = b[m][k], sk = skip[m][k];
                r[m][k] = db * (r[m][k] - sk) * gate * (1.0f - gate);
                skip[m][k] = sk > 0.0f ? db * (1.0f - gate) : 0.0f;
                b[m][k] = db * gate;
            }
        }
        store_rows(o_gl, soff, r);
        store_rows(o_s, soff, skip);
        dense64_scaled(B + qb::BLK_G_A, B + qb::BLK_G_B, r, t, lane);        // d gl Wg^T
#pragma unroll
        for (int m = 0; m < 4; ++m) b[m] += t[m];                            // d r
        store_rows(o_r, soff, b);
        dense64_scaled(B + qb::BLK_R2_A, B + qb::BLK_R2_B, b, t, lane);      // d r Wr2^T
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[m][k] = (m_t >> (4 * m + k)) & 1u ? t[m][k] : 0.0f;
        store_rows(o_t, soff, t);
        dense64_scaled(B + qb::BLK_R1_A, B + qb::BLK_R1_B, t, r, lane);      // d t' Wr1^T
        dense64_scaled(B + qb::BLK_WC_A, B + qb::BLK_WC_B, skip, b, lane);   // d s' Wc^T
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) b[m][k] += (m_bin >> (4 * m + k)) & 1u ? r[m][k] : 0.0f;
        store_rows(o_b, soff, b);
    }
}


// slab_reduce_kernel for many slabs (one per wave of block_bwd_dw_kernel: 1,024): 64 interleaved parts per
// workgroup; parts added in a fixed order.
struct SlabTargets {   // where the sum of matrix q's slabs goes (blockIdx.y = q)
    float* dW[4];
    float* db[4];
    int ldw[4], kdim[4], ndim[4];
};
__global__ __launch_bounds__(1024) void slab_reduce_many_kernel(const float* __restrict__ partial, int nblk,
                                                                 int64_t per_matrix, SlabTargets tg) {
    // 64 elements x 64 interleaved parts per workgroup: a thread takes four consecutive elements (16-byte loads, a
    // part's sixteen lanes read 256 contiguous bytes of a slab), four slabs' loads in flight per trip; per element the
    // additions and their order are the ones of the 16-element form this replaces (parts of stride 64, then the parts)
    constexpr int kParts = 64, kLanes = 16, kSlab = 64 * 64 + 64;
    __shared__ double part[kParts][kLanes][4];
    const int q = blockIdx.y;
    partial += (int64_t)q * per_matrix;
    float* __restrict__ dW = tg.dW[q];
    float* __restrict__ db = tg.db[q];
    const int ldw = tg.ldw[q], kdim = tg.kdim[q], ndim = tg.ndim[q];
    const int le = threadIdx.x & (kLanes - 1), p = threadIdx.x / kLanes;
    const int e4 = blockIdx.x * kLanes + le;     // float4 index within a slab (kSlab / 4 = 1040 = 65 x 16)
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const float4* src = reinterpret_cast<const float4*>(partial) + e4;
    for (int bk = p; bk < nblk; bk += 4 * kParts) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = bk + u * kParts;
            v[u] = src[(int64_t)(b < nblk ? b : bk) * (kSlab / 4)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (bk + u * kParts < nblk) {
                a[0] += (double)v[u].x; a[1] += (double)v[u].y; a[2] += (double)v[u].z; a[3] += (double)v[u].w;
            }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) part[p][le][c] = a[c];
    __syncthreads();
    if (threadIdx.x >= 4 * kLanes) return;   // thread (le, c) finishes element 4 e4 + c
    const int c = threadIdx.x / kLanes;
    double sum = 0.0;
#pragma unroll 8
    for (int qq = 0; qq < kParts; ++qq) sum += part[qq][le][c];
    const int e = 4 * e4 + c;
    if (e < 64 * 64) {
        const int i = e >> 6, j = e & 63;
        if (i < kdim && j < ndim) dW[i * ldw + j] = (float)sum;
    } else {
        const int j = e - 64 * 64;
        if (db && j < ndim) db[j] = (float)sum;
    }
}

// ---- the same with the block's four weight gradients accumulated in the kernel ----------------------------
// block_bwd_kernel still hands its four deltas to four xtd_kernel launches, which re-read each delta and its
// activation from HBM (0.5 ms per block on 1 M voxels).  Here a wave keeps the four 64 x 64 float32 gradient
// accumulators (256 registers) for the whole launch -- ONE wave per SIMD, so that it owns the 512-entry register
// file -- and adds each 16-voxel tile's contribution  dW[in][out] += sum_v X[v][in] D[v][out]  on
// v_mfma_f32_16x16x4_f32 (exact float32 products, k = voxel).  Both operands of that MFMA want the UNIT on the
// lane, while everything else in the kernel has the VOXEL on the lane: each tensor takes one trip through a
// 5 KB LDS tile per wave ([voxel][80] floats: the row stride keeps the 16 lanes of the two k groups of a
// half-wave on different banks) -- 4 ds_write_b128, 16 ds_read_b32.  The only tensor written is d b_in: 2 reads +
// 1 write instead of 2 + 5 and eight more for the xtd launches.  Gradients leave as one slab per wave and matrix
// ([in][out] + 64 bias sums, the format slab_reduce_kernel adds up in a fixed order: bitwise reproducible).
constexpr int kDwThreads = 256;
constexpr int kTrStride = 80;

struct TileFrags {   // a [16 voxel][64 unit] tile as MFMA operand: f[tile][ks] = T[voxel 4 ks + lane / 16][unit 16 tile + lane % 16]
    float f[4][4];
};
__device__ __forceinline__ void to_frags(float* __restrict__ buf, int lane, const f32x4 (&a)[4], TileFrags& o) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int m = 0; m < 4; ++m)
        *reinterpret_cast<float4*>(buf + i * kTrStride + 16 * m + 4 * g) = make_float4(a[m][0], a[m][1], a[m][2], a[m][3]);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) o.f[t][ks] = buf[(4 * ks + g) * kTrStride + 16 * t + i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is free for the next tensor
    __builtin_amdgcn_wave_barrier();
}
// acc[mt][nt] += X^T D over the tile's 16 voxels; bias[nt] += this lane's share of the column sums of D
template <bool RELU_X>
__device__ __forceinline__ void dw_accumulate(const TileFrags& X, const TileFrags& D, f32x4 (&acc)[4][4], float (&bias)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float a = RELU_X ? relu_1op(X.f[mt][ks]) : X.f[mt][ks];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = QB_MFMA16F(a, D.f[nt][ks], acc[mt][nt]);
        }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bias[nt] += D.f[nt][ks];
}
// this wave's slab of one matrix: [in][out] then the 64 bias sums
__device__ __forceinline__ void write_slab(float* __restrict__ slab, int lane, const f32x4 (&acc)[4][4], float (&bias)[4]) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(16 * mt + 4 * g + r) * 64 + 16 * nt + j] = acc[mt][nt][r];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        float b = bias[nt];
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (g == 0) slab[64 * 64 + 16 * nt + j] = b;
    }
}

__global__ __launch_bounds__(kDwThreads) void block_bwd_dw_kernel(
    const float* __restrict__ img_f, const float* __restrict__ img_b, const float* __restrict__ b_in,
    const float* d_b, float* d_bin, float* __restrict__ slabs, int nwaves_total, int U, int64_t N) {
    extern __shared__ __align__(16) float lds_img[];
    float* F = lds_img;
    float* B = lds_img + qb::BLK_FLOATS;
    qb::copy_to_lds<kDwThreads>(F, img_f, qb::BLK_FLOATS / 4);
    qb::copy_to_lds<kDwThreads>(B, img_b, qb::BLK_FLOATS / 4);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    constexpr int NW = kDwThreads / 64;
    float* tr = lds_img + 2 * qb::BLK_FLOATS + wave * (16 * kTrStride);
    const __amdgpu_buffer_rsrc_t o_b = row_buffer(d_bin, N);
    f32x4 acc_g[4][4], acc_r2[4][4], acc_r1[4][4], acc_c[4][4];
    float bs_g[4] = {0, 0, 0, 0}, bs_r2[4] = {0, 0, 0, 0}, bs_r1[4] = {0, 0, 0, 0}, bs_c[4] = {0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc_g[a][c] = f32x4{0, 0, 0, 0};
            acc_r2[a][c] = f32x4{0, 0, 0, 0};
            acc_r1[a][c] = f32x4{0, 0, 0, 0};
            acc_c[a][c] = f32x4{0, 0, 0, 0};
        }
    const int64_t ntile = (N + 15) / 16;
    // (Requesting the next tile's block input two products ahead was measured and lost, 3.18 against 3.11 ms per
    // step: with every register taken it adds spills, and the loads' latency is not what bounds this kernel.  So
    // was moving the four weight-gradient products to the bf16 matrix pipe with xtd9b_kernel's three-piece operands,
    // 3.03 against 2.69 ms per step: 192 MFMAs of 16 cycles replace 256 of 32, but the pieces cost 770 more vector
    // instructions per tile and 250 bytes of scratch per lane, on a wave that has nothing to overlap them with.
    // Round 4, the two-half f16 form of xtdb_kernel (128 f16 MFMAs for the 256 f32 ones, fragments as {h, l} pairs in
    // the same 16 registers, one running delta scale per matrix): 2.58 against 2.30 ms per step WITHOUT the rescaling
    // path (192 bytes of scratch), 4.50 with it (1,064 bytes: the slow path's reads of 64 accumulation registers sit
    // inside the tile loop).  Third form of the same answer: this kernel's time is its register file.)
    for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntile; tile += (int64_t)gridDim.x * NW) {
        const int64_t v = tile * 16 + i;
        const bool live = v < N;
        const uint32_t voff = (uint32_t)(live ? v : N - 1) * 256u + 16u * (uint32_t)g;
        const uint32_t soff = live ? voff : kDropped;
        f32x4 b[4], skip[4], t[4], r[4];
        load_rows(b_in, voff, g, U, b);
        if (!live) {   // a lane beyond the batch repeats the last voxel for the MFMAs' sake: it must not count twice
#pragma unroll
            for (int m = 0; m < 4; ++m) b[m] = f32x4{0, 0, 0, 0};
        }
        TileFrags fb, fx, fd;
        to_frags(tr, lane, b, fb);                                           // b_in, unit on the lane
        uint32_t m_bin = 0u, m_t = 0u;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) m_bin |= b[m][k] > 0.0f ? 1u << (4 * m + k) : 0u;
        qb::dense64<false>(F + qb::BLK_WC_A, F + qb::BLK_WC_B, b, skip, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            skip[m] = qb::relu4(skip[m]);
#pragma unroll
            for (int k = 0; k < 4; ++k) b[m][k] = relu_1op(b[m][k]);   // b came from memory: fmaxf would quiet it first
        }
        qb::dense64<false>(F + qb::BLK_R1_A, F + qb::BLK_R1_B, b, t, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            t[m] = qb::relu4(t[m]);
#pragma unroll
            for (int k = 0; k < 4; ++k) m_t |= t[m][k] > 0.0f ? 1u << (4 * m + k) : 0u;
        }
        qb::dense64<false>(F + qb::BLK_R2_A, F + qb::BLK_R2_B, t, r, lane);
        TileFrags ft;
        to_frags(tr, lane, t, ft);                                           // t
        to_frags(tr, lane, r, fx);                                           // r
        qb::dense64<false>(F + qb::BLK_G_A, F + qb::BLK_G_B, r, t, lane);   // gate logits + gate_offset
        load_rows(d_b, voff, g, U, b);
        if (!live) {
#pragma unroll
            for (int m = 0; m < 4; ++m) b[m] = f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gate = qb::sigmoidf_(t[m][k]);
                const float db = b[m][k], sk = skip[m][k];
                r[m][k] = db * (r[m][k] - sk) * gate * (1.0f - gate);
                skip[m][k] = sk > 0.0f ? db * (1.0f - gate) : 0.0f;
                b[m][k] = db * gate;
            }
        }
        to_frags(tr, lane, r, fd);                                           // d gl
        dw_accumulate<false>(fx, fd, acc_g, bs_g);                           // dWg += r^T d gl
        to_frags(tr, lane, skip, fd);                                        // d s'
        dw_accumulate<false>(fb, fd, acc_c, bs_c);                           // dWc += b_in^T d s'
        dense64_scaled(B + qb::BLK_G_A, B + qb::BLK_G_B, r, t, lane);        // d gl Wg^T
#pragma unroll
        for (int m = 0; m < 4; ++m) b[m] += t[m];                            // d r
        to_frags(tr, lane, b, fd);
        dw_accumulate<false>(ft, fd, acc_r2, bs_r2);                         // dWr2 += t^T d r
        dense64_scaled(B + qb::BLK_R2_A, B + qb::BLK_R2_B, b, t, lane);      // d r Wr2^T
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[m][k] = (m_t >> (4 * m + k)) & 1u ? t[m][k] : 0.0f;
        to_frags(tr, lane, t, fd);
        dw_accumulate<true>(fb, fd, acc_r1, bs_r1);                          // dWr1 += relu(b_in)^T d t'
        dense64_scaled(B + qb::BLK_R1_A, B + qb::BLK_R1_B, t, r, lane);      // d t' Wr1^T
        dense64_scaled(B + qb::BLK_WC_A, B + qb::BLK_WC_B, skip, b, lane);   // d s' Wc^T
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) b[m][k] += (m_bin >> (4 * m + k)) & 1u ? r[m][k] : 0.0f;
        store_rows(o_b, soff, b);
    }
    // slabs: matrix q of wave w at slabs[(q * nwaves_total + w) * (64 * 64 + 64)];  q: 0 Wg, 1 Wr2, 2 Wr1, 3 Wc
    const int64_t w = (int64_t)blockIdx.x * NW + wave;
    constexpr int64_t kSlab = 64 * 64 + 64;
    write_slab(slabs + (0 * (int64_t)nwaves_total + w) * kSlab, lane, acc_g, bs_g);
    write_slab(slabs + (1 * (int64_t)nwaves_total + w) * kSlab, lane, acc_r2, bs_r2);
    write_slab(slabs + (2 * (int64_t)nwaves_total + w) * kSlab, lane, acc_r1, bs_r1);
    write_slab(slabs + (3 * (int64_t)nwaves_total + w) * kSlab, lane, acc_c, bs_c);
}

// the block matrices of the canonical blob transposed in place of themselves (centre tap of 3x3x1 kernels),
// everything else zero: packed by qbold_encoder_pack this is the image whose dense64 multiplies by W^T
__global__ void blob_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, qb::CanonLayout c) {
    const int U = c.U, ctr = c.taps == 9 ? 4 * U * U : 0;
    const int per = 4 * U * U;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < c.L * per; e += gridDim.x * blockDim.x) {
        const int l = e / per, q = e % per, which = q / (U * U), ij = q % (U * U), ii = ij / U, jj = ij % U;
        const int off = c.blk0 + l * c.blk_stride +
                        (which == 0 ? c.Wc : which == 1 ? c.Wr1 + ctr : which == 2 ? c.Wr2 + ctr : c.Wg);
        wt[off + jj * U + ii] = w[off + ii * U + jj];
    }
}

// out = in * (ref > 0): the relu adjoint
__global__ void mask_mul_kernel(const float* __restrict__ in, const float* __restrict__ ref,
                                float* __restrict__ out, int64_t n) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x)
        out[e] = ref[e] > 0.0f ? in[e] : 0.0f;
}

// synthetic_data_loss (model.py:449-514) and its gradient: loss_v = -log p(y_true; q)
// Optional inverse-gamma prior on the two marginal variances (model.py:492-507, use_mvg branch):
// loss -= log IG(exp(s_o)^2; a, b) + log IG(exp(s_d)^2 + q[4]^2; a, b) -- the RAW fifth parameter, as
// the reference writes it (:499).  ig_a = 0 switches it off; ig_c0 = lgamma(a) - a log b.
// out = gelu(in);  out = d * gelu'(z), gelu'(z) = Phi(z) + z phi(z)  -- the gelu backward works on pre-activations it
// recomputes (train_bwd_gelu below)
__global__ void gelu_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        out[e] = gelu_(in[e]);
}
// d[row][col] *= gelu'(z[row][col]) in place, columns < U only: the padding columns of Z hold stale workspace data
// (possibly NaN / Inf), which must not reach the delta tensors.
__global__ void gelu_bwd_mul_kernel(float* d, const float* __restrict__ z, int64_t rows, int ld, int U) {
    const int64_t n = rows * (int64_t)U;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = (i / U) * ld + (i % U);
        const float v = z[e];
        const float dg = 0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * expf(-0.5f * v * v);
        d[e] *= dg;
    }
}

__global__ void nlogp_bwd_kernel(const float* __restrict__ y_true, int ldy, const float* __restrict__ q,
                                 float* __restrict__ g_q, int ldg, float* __restrict__ loss, float scale,
                                 float ig_a, float ig_b, float ig_c0, int64_t N) {
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < N;
         v += (int64_t)gridDim.x * blockDim.x) {
        float p[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) p[k] = q[v * 5 + k];
        const qb::LogitMvn m = qb::make_mvn(p);
        const qb::LogitObs o = qb::make_obs(y_true[v * ldy], y_true[v * ldy + 1]);
        const float r0 = o.l0 - m.mu_o, r1 = o.l1 - m.mu_d;
        const float w0 = r0 * m.i_so, w1 = r1 * m.i_sd + r0 * m.i_bl;
        float lv = 1.8378770664093453f + (m.s_o + m.s_d) + 0.5f * (w0 * w0 + w1 * w1) + o.jac;
        const float th1 = (m.s_o + 1.0f) * (1.0f / 3.0f), th3 = (m.s_d + 1.0f) * (1.0f / 3.0f);
        const float th4 = m.c * 7.38905609893065f;
        float g1 = (1.0f - w0 * w0 - w1 * r0 * m.i_bl) * 3.0f * (1.0f - th1 * th1);
        float g3 = (1.0f - w1 * w1) * 3.0f * (1.0f - th3 * th3);
        float g4 = (w1 * -r0 * m.i_so * m.i_sd) * 0.1353352832366127f * (1.0f - th4 * th4);
        if (ig_a > 0.0f) {
            const float xo = 1.0f / (m.i_so * m.i_so);               // exp(s_o)^2
            const float ed = 1.0f / (m.i_sd * m.i_sd);               // exp(s_d)^2
            const float xd = ed + p[4] * p[4];
            // -log IG(x) = lgamma(a) - a log b + (a + 1) log x + b / x
            lv += 2.0f * ig_c0 + (ig_a + 1.0f) * (logf(xo) + logf(xd)) + ig_b / xo + ig_b / xd;
            const float do_ = (ig_a + 1.0f) / xo - ig_b / (xo * xo);  // d / d x_o
            const float dd = (ig_a + 1.0f) / xd - ig_b / (xd * xd);   // d / d x_d
            g1 += do_ * 2.0f * xo * 3.0f * (1.0f - th1 * th1);
            g3 += dd * 2.0f * ed * 3.0f * (1.0f - th3 * th3);
            g4 += dd * 2.0f * p[4];
        }
        if (loss) loss[v] = lv;
        float* g = g_q + v * ldg;
        g[0] = scale * -(w0 * m.i_so + w1 * m.i_bl);
        g[1] = scale * g1;
        g[2] = scale * -(w1 * m.i_sd);
        g[3] = scale * g3;
        g[4] = scale * g4;
    }
}

// The learned inverse-gamma hyper-prior of the pre-training loss (infer_inv_gamma with the diagonal family,
// model.py:201-205, 454-455, 493-507): per voxel
//   term = - log IG(v_o; a_o, b_o) - log IG(v_d; a_d, b_d),   v = exp(2 s),  s = transform_std(raw)
//        = sum_dim [ lgamma(a) - a log b + (a + 1) log v + b / v ]
// ADDED to loss_v; scale * d term / d q ADDED to g_q (columns 1 and 3); and the four sums the gradient with
// respect to (a_o, b_o, a_d, b_d) needs -- sum log v_o, sum 1 / v_o, sum log v_d, sum 1 / v_d -- into stats.
__global__ __launch_bounds__(256) void hyper_prior_kernel(const float* __restrict__ q, float a_o, float b_o, float c_o,
                                                          float a_d, float b_d, float c_d, float scale,
                                                          float* __restrict__ g_q, float* __restrict__ loss,
                                                          double* __restrict__ stats, int64_t N) {
    __shared__ double red[4][4];
    double s_lo = 0.0, s_io = 0.0, s_ld = 0.0, s_id = 0.0;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < N;
         v += (int64_t)gridDim.x * blockDim.x) {
        const float so = qb::transform_std(q[v * 5 + 1]), sd = qb::transform_std(q[v * 5 + 3]);
        const float lvo = 2.0f * so, lvd = 2.0f * sd;              // log v
        const float ivo = __expf(-lvo), ivd = __expf(-lvd);        // 1 / v
        if (loss) loss[v] += (c_o + (a_o + 1.0f) * lvo + b_o * ivo) + (c_d + (a_d + 1.0f) * lvd + b_d * ivd);
        if (g_q) {
            const float to = (so + 1.0f) * (1.0f / 3.0f), td = (sd + 1.0f) * (1.0f / 3.0f);   // tanh(raw)
            g_q[v * 5 + 1] += scale * (2.0f * (a_o + 1.0f) - 2.0f * b_o * ivo) * 3.0f * (1.0f - to * to);
            g_q[v * 5 + 3] += scale * (2.0f * (a_d + 1.0f) - 2.0f * b_d * ivd) * 3.0f * (1.0f - td * td);
        }
        s_lo += lvo; s_io += ivo; s_ld += lvd; s_id += ivd;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double vals[4] = {s_lo, s_io, s_ld, s_id};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double x = vals[k];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        if (lane == 0) red[wave][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < 4 && stats)
        atomic_add_fixed(stats + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x],
                         kStatsFixed);
}

// copy g_q [N][5] and g_ls [N][T] into one [N][ld] delta tensor (cols 0-4, 5..5+T-1), scaled.  Only the
// first 5 + T columns (rounded up to 4) of a row are written: the GEMMs that read this tensor take ndim /
// kdim = 5 or T columns of it and zero everything beyond by select, so the rest of the row is never used.
__global__ void head_delta_kernel(const float* __restrict__ g_q, const float* __restrict__ g_ls, int T,
                                  const double* __restrict__ sums, float* __restrict__ d, int ld, int64_t N) {
    const float scale = sums ? (float)(1.0 / sums[2]) : 1.0f;  // 1 / sum(mask)
    const int wcols = ((5 + T + 3) & ~3) < ld ? ((5 + T + 3) & ~3) : ld;
    const int64_t total = N * wcols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % wcols);
        const int64_t v = e / wcols;
        float o = 0.0f;
        if (j < 5) o = g_q[v * 5 + j] * scale;
        else if (g_ls && j < 5 + T) o = g_ls[v * T + j - 5] * scale;
        d[v * ld + j] = o;
    }
}

// tfa.optimizers.AdamW step (decoupled weight decay; Keras Adam with epsilon 1e-7):
//   var -= wd * var;  m,v update;  var -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
__global__ void adamw_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, float lr_t, float b1, float b2, float eps,
                             float wd) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        const float ge = g[e];
        const float me = b1 * m[e] + (1.0f - b1) * ge;
        const float ve = b2 * v[e] + (1.0f - b2) * ge * ge;
        m[e] = me;
        v[e] = ve;
        float we = w[e];
        we -= wd * we;
        we -= lr_t * me / (sqrtf(ve) + eps);
        w[e] = we;
    }
}

struct Launcher {
    const qbold_ctx* ctx;
    hipStream_t s;
    int64_t N;
    int ld;  // row stride of the activation tensors (64, or U rounded up to 64 beyond that)
    Gather gather = make_gather(0, 0, 0, 0, 0);
    // Deferred slab sums (slab_reduce_jobs_kernel): while `bump` is set, every weight-gradient kernel writes its slabs
    // to a region of its own carved from [bump, bump_end) and queues its reduction instead of launching it; the queue
    // is flushed when the region or the job table is full, and at the end of the backward.
    mutable float* bump = nullptr;
    mutable float* bump0 = nullptr;
    mutable float* bump_end = nullptr;
    mutable SlabJobs jobs{};
    mutable int njobs = 0;
    void defer_slabs(float* region, float* region_end) const { bump = bump0 = region; bump_end = region_end; njobs = 0; }
    void flush_slabs() const {
        if (njobs > 0)
            hipLaunchKernelGGL(slab_reduce_jobs_kernel, dim3(((64 * 64 + 64) / 4 + 31) / 32, njobs), dim3(512), 0, s, jobs);
        njobs = 0;
        bump = bump0;
    }
    // a slab region for `count` groups of nblk slabs, or null when slabs are not deferred
    float* slab_region(int nblk, int count, int jobs_needed) const {
        if (!bump0) return nullptr;
        const int64_t need = (int64_t)count * nblk * (64 * 64 + 64);
        if (need > bump_end - bump0) return nullptr;          // larger than the whole region: the undeferred path
        if (need > bump_end - bump || njobs + jobs_needed > kMaxSlabJobs) flush_slabs();
        float* r = bump;
        bump += need;
        return r;
    }
    void queue_slab(const float* partial, int nblk, float* dW, int ldw, int kdim, int ndim, float* db, int j0) const {
        jobs.job[njobs++] = SlabJob{partial, dW, db, nblk, ldw, kdim, ndim, j0};
    }
    int grid() const {
        int64_t nb = (N + 63) / 64;
        int64_t cap = (int64_t)ctx->num_cus * 4;  // 256-thread blocks per CU; measured per 1 M-voxel step with the
                                                  // prefetching xw_kernel: 2 -> 8.4, 3 -> 7.6, 4 -> 7.3, 6 -> 7.7, 8 -> 7.4, 16 -> 7.5 ms
        return (int)(nb < cap ? (nb > 0 ? nb : 1) : cap);
    }
    // Y (row stride ldy) = act(X W + b); ndim output columns in slabs of 64
    int xw_ld(const float* X, int ldx, int kdim, const float* W, int ldw, int trans, const float* b,
              float* Y, int ldy, int ndim, int act, int accum, const float* mask) const {
        const size_t smem = sizeof(float) * (size_t)((kdim + 15) & ~15) * kWs;  // room for the float4 k order
        if (smem > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xw_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return qb::hip_fail(e, "hipFuncSetAttribute(xw_kernel)");
        }
        const bool aligned = (reinterpret_cast<uintptr_t>(X) & 15) == 0 && (ldx & 3) == 0 && ldx >= 64 &&
                             (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && (ldy & 3) == 0 && ldy >= 64 &&
                             (!mask || ((reinterpret_cast<uintptr_t>(mask) & 15) == 0 && (ld & 3) == 0 && ld >= 64));
        if (aligned && kdim <= 64 && ndim <= 64 && gather.Z == 0 && !(ctx->kernel_sel & 512) &&
            !(act & (ACT_GELU | ACT_GELU_IN))) {   // gelu: the general kernel
            auto kern = accum ? (mask ? xw64_kernel<true, true> : xw64_kernel<true, false>)
                              : (mask ? xw64_kernel<false, true> : xw64_kernel<false, false>);
            hipLaunchKernelGGL(kern, dim3(grid()), dim3(256), smem, s, X, ldx, kdim, W, ldw, trans, b, Y, ldy, ndim,
                               act, mask, ld, N);
            return QBOLD_OK;
        }
        hipLaunchKernelGGL(xw_kernel, dim3(grid(), (ndim + 63) / 64), dim3(256), smem, s, X, ldx, kdim, W,
                           ldw, trans, b, Y, ldy, ndim, act, accum, mask, ld, N, gather);
        return QBOLD_OK;
    }
    // 3x3x1 'same' convolution as nine gathered GEMMs: Y = act(sum_taps X[nbr] K[tap] + b).
    // flip = 1 is the adjoint wrt the input (taps mirrored, kernels transposed).
    // delta_sums (backward-data launches): the device double[3] whose [2] is the sum(mask) the deltas were divided by
    void conv3x3(const float* X, const float* K9, int U, const float* b, float* Y, int act, int flip,
                 const float* mask, const qbold_geometry& gm, const double* delta_sums = nullptr) {
        if (U <= 64 && ld == kLd && !(ctx->kernel_sel & 256)) {  // one launch, accumulators in registers
            const int64_t nb = (N + 255) / 256;
            const int64_t cap = (int64_t)ctx->num_cus;
            const size_t smem = sizeof(float) * 9 * 64 * kWs;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv9_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            const bool aligned = ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) |
                                   reinterpret_cast<uintptr_t>(mask)) & 15) == 0;
            if (aligned && U % 4 == 0 && N < (1 << 23) && !(ctx->kernel_sel & 65536) && !(act & (ACT_GELU | ACT_GELU_IN))) {   // split-f16 matrix pipe (bit 65536 / gelu / odd widths: the exact-f32 form)
                const size_t smh = sizeof(float) * (9 * 4096 + 64);
                const bool scaled = delta_sums && !(act & ACT_RELU_IN);
                auto kern = scaled ? conv9h_kernel<true> : conv9h_kernel<false>;
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smh);
                hipLaunchKernelGGL(kern, dim3((unsigned)(nb < cap ? (nb > 0 ? nb : 1) : cap)), dim3(1024), smh, s,
                                   X, ld, U, K9, flip, b, Y, ld, act, mask, ld, N, make_gather(gm.X, gm.Y, gm.Z, 0, 0),
                                   delta_sums);
                gather = make_gather(0, 0, 0, 0, 0);
                return;
            }
            hipLaunchKernelGGL(conv9_kernel, dim3((unsigned)(nb < cap ? (nb > 0 ? nb : 1) : cap)), dim3(1024), smem, s,
                               X, ld, U, K9, flip, b, Y, ld, act, mask, ld, N, make_gather(gm.X, gm.Y, gm.Z, 0, 0));
            gather = make_gather(0, 0, 0, 0, 0);
            return;
        }
        for (int tap = 0; tap < 9; ++tap) {
            const int dx = tap / 3 - 1, dy = tap % 3 - 1;
            gather = make_gather(gm.X, gm.Y, gm.Z, flip ? -dx : dx, flip ? -dy : dy);
            (void)xw_ld(X, ld, U, K9 + (int64_t)tap * U * U, U, flip, tap == 0 ? b : nullptr, Y, ld, U,
                        (tap == 8 ? (act & (ACT_RELU | ACT_GELU)) : ACT_NONE) | (act & (ACT_RELU_IN | ACT_GELU_IN)), tap != 0,
                        tap == 8 ? mask : nullptr);
        }
        gather = make_gather(0, 0, 0, 0, 0);
    }
    void xw(const float* X, int ldx, int kdim, const float* W, int ldw, int trans, const float* b,
            float* Y, int ndim, int act, int accum, const float* mask) const {
        (void)xw_ld(X, ldx, kdim, W, ldw, trans, b, Y, ld, ndim, act, accum, mask);
    }
    // dW (+)= X^T D, db (+)= sum D: 64 x 64 slabs of the (kdim x ndim) product, one launch pair per slab
    // kzero > kdim: columns kdim .. kzero - 1 of X are known to hold zeros (a reader may take them along)
    void xtd(const float* X, int kdim, const float* D, int ndim, float* partial, int nblk, float* dW,
             int ldw, float* db, int accum, int relu_x = 0, const float* dref = nullptr, int kzero = 0) const {
        for (int a = 0; a < kdim; a += 64)
            for (int c = 0; c < ndim; c += 64) {
                const int ka = kdim - a < 64 ? kdim - a : 64, nc = ndim - c < 64 ? ndim - c : 64;
                const float* Xa = X + a;
                const float* Dc = D + c;
                const float* Rc = dref ? dref + c : nullptr;
                const bool vec = (ld & 3) == 0 &&
                                 ((reinterpret_cast<uintptr_t>(Xa) | reinterpret_cast<uintptr_t>(Dc) |
                                   reinterpret_cast<uintptr_t>(Rc)) & 15) == 0;
                const int kread = kzero > kdim && kdim <= 64 ? (kzero < 64 ? kzero : 64) : ka;
                float* own = accum ? nullptr : slab_region(nblk, 1, 1);
                if (!own && njobs > 0) flush_slabs();   // the shared region below may hold queued slabs
                float* part = own ? own : partial;
                if (pieces_ok(Xa, kread, Dc, nc, Rc)) {
                    launch_xtdb(Xa, kread, Dc, nc, part, nblk, relu_x, Rc);
                } else {
                    auto kern = vec ? xtd_kernel<true> : xtd_kernel<false>;
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kXtdSmem);
                    hipLaunchKernelGGL(kern, dim3(nblk), dim3(1024), kXtdSmem, s, Xa, ld, ka, Dc, ld, nc, part, N,
                                       gather, relu_x, Rc);
                }
                if (own) {
                    queue_slab(part, nblk, dW + (int64_t)a * ldw + c, ldw, ka, nc, a == 0 && db ? db + c : nullptr, 0);
                    continue;
                }
                hipLaunchKernelGGL(slab_reduce_kernel, dim3((64 * 64 + 64) / 64), dim3(1024), 0, s, part,
                                   nblk, dW + (int64_t)a * ldw + c, ldw, ka, nc, a == 0 && db ? db + c : nullptr, accum, 0);
            }
    }
    // one 64 x 64 slab (kdim, ndim <= 64) without its reduction, and the reduction of columns j0 .. of a slab
    // the bf16 matrix pipe's form (xtdb_kernel): aligned [N][64] rows, no gather, widths that are multiples of 4
    // (bit 524288 of the kernel selection: the exact-f32 kernel)
    bool pieces_ok(const float* X, int kdim, const float* D, int ndim, const float* dref) const {
        return ld == kLd && gather.Z == 0 && kdim % 4 == 0 && ndim % 4 == 0 && kdim <= 64 && ndim <= 64 &&
               N < (1 << 23) && !(ctx->kernel_sel & 524288) &&
               ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(D) | reinterpret_cast<uintptr_t>(dref)) & 15) == 0;
    }
    void launch_xtdb(const float* X, int kdim, const float* D, int ndim, float* partial, int nblk, int relu_x,
                     const float* dref) const {
        const bool h16 = !(ctx->kernel_sel & QBOLD_KSEL_DW_BF16_PIECES);
        auto kern = h16 ? (dref ? (relu_x ? xtdb_kernel<true, true, true> : xtdb_kernel<false, true, true>)
                                : (relu_x ? xtdb_kernel<true, false, true> : xtdb_kernel<false, false, true>))
                        : (dref ? (relu_x ? xtdb_kernel<true, true, false> : xtdb_kernel<false, true, false>)
                                : (relu_x ? xtdb_kernel<true, false, false> : xtdb_kernel<false, false, false>));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)kXtdSmem);
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), kXtdSmem, s, X, kdim, D, ndim, partial, N, dref);
    }
    void xtd_only(const float* X, int kdim, const float* D, int ndim, float* partial, int nblk) const {
        if (pieces_ok(X, kdim, D, ndim, nullptr)) {
            launch_xtdb(X, kdim, D, ndim, partial, nblk, 0, nullptr);
            return;
        }
        const bool vec = (ld & 3) == 0 &&
                         ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(D)) & 15) == 0;
        auto kern = vec ? xtd_kernel<true> : xtd_kernel<false>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)kXtdSmem);
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(1024), kXtdSmem, s, X, ld, kdim, D, ld, ndim, partial, N, gather, 0,
                           static_cast<const float*>(nullptr));
    }
    void reduce_cols(const float* partial, int nblk, float* dW, int ldw, int kdim, int ndim, float* db, int j0) const {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((64 * 64 + 64) / 64), dim3(1024), 0, s, partial, nblk, dW, ldw,
                           kdim, ndim, db, 0, j0);
    }
    // all nine taps of a 3x3x1 kernel gradient in two launches: dK[tap] = X[nbr(., tap)]^T D, db = sum D
    void xtd9(const float* X, int U, const float* D, float* partial, int nblk, float* dK9, float* db,
              const qbold_geometry& gm, int relu_x = 0) const {
        // bf16 matrix pipe (three-piece operands): crops whose lane groups' four voxels are one z run
        // (bit 524288 of the kernel selection: the exact-f32 kernel)
        const bool pieces = U <= 64 && U % 4 == 0 && gm.Z % 4 == 0 && N % 4 == 0 && N >= 4 && N < (1 << 23) &&
                            gm.X <= 512 && gm.Y <= 512 && gm.Z <= 512 && !(ctx->kernel_sel & 524288) &&
                            ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(D)) & 7) == 0;
        float* own = slab_region(nblk, 9, 9);
        if (!own && njobs > 0) flush_slabs();
        if (own) partial = own;
        if (pieces) {
            const bool h16 = !(ctx->kernel_sel & QBOLD_KSEL_DW_BF16_PIECES);
            auto kern = h16 ? (relu_x ? xtd9b_kernel<true, true> : xtd9b_kernel<false, true>)
                            : (relu_x ? xtd9b_kernel<true, false> : xtd9b_kernel<false, false>);
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), 0, s, X, D, partial, N, make_gather(gm.X, gm.Y, gm.Z, 0, 0), U);
        }
        else
            hipLaunchKernelGGL(xtd9_kernel, dim3(nblk), dim3(512), 0, s, X, kLd, U, D, kLd, U, partial, N,
                               make_gather(gm.X, gm.Y, gm.Z, 0, 0), relu_x);
        if (own) {
            for (int tap = 0; tap < 9; ++tap)
                queue_slab(partial + (int64_t)tap * nblk * (64 * 64 + 64), nblk, dK9 + (int64_t)tap * U * U, U, U, U,
                           tap == 0 ? db : nullptr, 0);
            return;
        }
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((64 * 64 + 64) / 64, 9), dim3(1024), 0, s, partial,
                           nblk, dK9, U, U, U, db, 0, 0);
    }
    int ew() const {
        int64_t nb = (N * ld + 255) / 256;
        int64_t cap = (int64_t)ctx->num_cus * 8;
        return (int)(nb < cap ? (nb > 0 ? nb : 1) : cap);
    }
};

int train_ld(int U) { return U <= 64 ? kLd : ((U + 63) / 64) * 64; }

// the layer-wise path accepts wider / longer encoders than the LDS-resident fused kernels
int check_layerwise_shape(const qbold_ctx* ctx, const qbold_encoder_shape* s) {
    if (!s) { qb::set_error("encoder shape is null"); return QBOLD_ERR_INVALID; }
    if (s->T != ctx->dev.T) { qb::set_error("encoder shape T differs from the context's tau grid"); return QBOLD_ERR_INVALID; }
    if (s->U < 1 || s->U > kMaxU || s->L < 1 || s->L > 8) {
        qb::set_error("layer-wise encoder path: need 1 <= U <= 256, 1 <= L <= 8");
        return QBOLD_ERR_UNSUPPORTED;
    }
    if (s->activation != QBOLD_ACT_RELU && s->activation != QBOLD_ACT_GELU) {
        qb::set_error("encoder shape: activation must be QBOLD_ACT_RELU or QBOLD_ACT_GELU");
        return QBOLD_ERR_INVALID;
    }
    return QBOLD_OK;
}

constexpr int kSlabBlocks = 512;  // xtd partial slabs per launch: two 1024-thread blocks per CU

}  // namespace

// ---------------------------------------------------------------------------------------------
// activation slots (each [N][64] floats):
//   0: n   1: h   then per block l (base 2 + 5 l): skip, t, r, gl (gate logits), bout
//   stream 1 uses: 0: n, 1: h, 2 + l: a_l
// ---------------------------------------------------------------------------------------------
// ---- add_normalizer (model.py:131-140): Dropout, GroupNormalization(groups = 1, axis = -1), then the Activation --------
// One "normalizer" sits in front of each of a residual path's two activations (model.py:150-151, 154-155):
//   u = Dropout(x)         training only: u = x keep / (1 - rate), keep ~ Bernoulli(1 - rate) per element
//   v = gamma (u - mean_g) rstd_g + beta    tfa GroupNormalization, groups = 1: mean and biased variance over ALL
//                          positions and channels of one batch element g (a voxel of an (N,1,1,1,C) batch, a whole crop
//                          of a [B][X][Y][Z][C] batch), rstd = 1 / sqrt(var + 1e-3)
//   a = act(v)
// The library's own dropout stream (the reference's is TensorFlow's): element (row, column c) of normalizer `layer` is
// dropped iff half-word c & 7 of Philox4x32-7(ctr = (row_lo, row_hi, (c >> 3) | layer << 16, 5), key = seed) is below
// rate 2^16 -- regenerated from the same keys by the backward.  Off the hot path: general element-wise kernels.
struct NormSpec {
    int layer_norm;        // GroupNormalization on
    int act;               // qbold_activation
    uint32_t drop_thresh;  // 0: no dropout
    float keep_scale;      // 1 / (1 - rate)
    uint64_t seed;
    uint32_t layer;
    int64_t rows_per_group;
};
constexpr float kLnEps = 1e-3f;   // tfa GroupNormalization's default epsilon

__device__ __forceinline__ float drop_factor(const NormSpec& sp, int64_t row, int col) {
    if (sp.drop_thresh == 0u) return 1.0f;
    const uint4 o = qb::philox4x32_7(make_uint4((uint32_t)row, (uint32_t)((uint64_t)row >> 32),
                                                (uint32_t)(col >> 3) | (sp.layer << 16), 5u),
                                     make_uint2((uint32_t)sp.seed, (uint32_t)(sp.seed >> 32)));
    const uint32_t w = ((col >> 1) & 3) == 0 ? o.x : ((col >> 1) & 3) == 1 ? o.y : ((col >> 1) & 3) == 2 ? o.z : o.w;
    const uint32_t hw = (col & 1) ? (w >> 16) : (w & 0xffffu);
    return hw < sp.drop_thresh ? 0.0f : sp.keep_scale;
}
__device__ __forceinline__ float act_of(int act, float v) {
    return act == QBOLD_ACT_GELU ? 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)) : fmaxf(v, 0.0f);
}
__device__ __forceinline__ float dact_of(int act, float v) {
    if (act != QBOLD_ACT_GELU) return v > 0.0f ? 1.0f : 0.0f;
    return 0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * expf(-0.5f * v * v);
}

// mean and rstd of u = Dropout(x) per group: one 256-thread block per group, two passes (mean, then the variance about it)
__global__ __launch_bounds__(256) void norm_stats_kernel(const float* __restrict__ x, int ld, int U, NormSpec sp,
                                                         float2* __restrict__ stats) {
    __shared__ double red[256];
    const int64_t g = blockIdx.x, r0 = g * sp.rows_per_group;
    const int64_t n = sp.rows_per_group * (int64_t)U;
    double mean = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
        double acc = 0.0;
        for (int64_t i = threadIdx.x; i < n; i += 256) {
            const int64_t row = r0 + i / U;
            const int col = (int)(i % U);
            const double u = (double)(x[row * ld + col] * drop_factor(sp, row, col));
            acc += pass == 0 ? u : (u - mean) * (u - mean);
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        const double tot = red[0] / (double)n;
        __syncthreads();
        if (pass == 0) mean = tot;
        else if (threadIdx.x == 0) stats[g] = make_float2((float)mean, 1.0f / sqrtf((float)tot + kLnEps));
    }
}
// the same for groups of one row (voxel batches): a wave per row
__global__ __launch_bounds__(256) void norm_stats_rows_kernel(const float* __restrict__ x, int ld, int U, NormSpec sp,
                                                              float2* __restrict__ stats, int64_t N) {
    const int lane = threadIdx.x & 63;
    for (int64_t row = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6); row < N; row += (int64_t)gridDim.x * 4) {
        float u[4] = {0.0f, 0.0f, 0.0f, 0.0f}, acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col = lane + 64 * k;
            if (col < U) u[k] = x[row * ld + col] * drop_factor(sp, row, col);
            acc += u[k];
        }
        const float mean = qb::wave_sum(acc) / (float)U;
        acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (lane + 64 * k < U) acc += (u[k] - mean) * (u[k] - mean);
        const float var = qb::wave_sum(acc) / (float)U;
        if (lane == 0) stats[row] = make_float2(mean, 1.0f / sqrtf(var + kLnEps));
    }
}
// a = act(LN(Dropout(x)));  gb: gamma [U], beta [U] (layer_norm) or null
__global__ void norm_act_fwd_kernel(const float* __restrict__ x, int ld, int U, NormSpec sp,
                                    const float2* __restrict__ stats, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float* __restrict__ out, int64_t N) {
    const int64_t n = N * (int64_t)U;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / U;
        const int col = (int)(i % U);
        float v = x[row * ld + col] * drop_factor(sp, row, col);
        if (sp.layer_norm) {
            const float2 st = stats[row / sp.rows_per_group];
            v = fmaf((v - st.x) * st.y, gamma[col], beta[col]);
        }
        out[row * ld + col] = act_of(sp.act, v);
    }
}
// Backward, step 1 (layer_norm): per group S1 = sum dxh, S2 = sum dxh xh with dxh = d_a act'(v) gamma; one block per group
__global__ __launch_bounds__(256) void norm_bwd_group_sums_kernel(const float* __restrict__ x, const float* __restrict__ d,
                                                                  int ld, int U, NormSpec sp,
                                                                  const float2* __restrict__ stats,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  float2* __restrict__ gsum) {
    __shared__ double red[2][256];
    const int64_t g = blockIdx.x, r0 = g * sp.rows_per_group;
    const int64_t n = sp.rows_per_group * (int64_t)U;
    const float2 st = stats[g];
    double a1 = 0.0, a2 = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const int64_t row = r0 + i / U;
        const int col = (int)(i % U);
        const float xh = (x[row * ld + col] * drop_factor(sp, row, col) - st.x) * st.y;
        const float v = fmaf(xh, gamma[col], beta[col]);
        const float dxh = d[row * ld + col] * dact_of(sp.act, v) * gamma[col];
        a1 += (double)dxh;
        a2 += (double)(dxh * xh);
    }
    red[0][threadIdx.x] = a1;
    red[1][threadIdx.x] = a2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) gsum[g] = make_float2((float)(red[0][0] / (double)n), (float)(red[1][0] / (double)n));
}
// the same for groups of one row (voxel batches): a wave per row
__global__ __launch_bounds__(256) void norm_bwd_row_sums_kernel(const float* __restrict__ x, const float* __restrict__ d,
                                                                int ld, int U, NormSpec sp, const float2* __restrict__ stats,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float2* __restrict__ gsum, int64_t N) {
    const int lane = threadIdx.x & 63;
    for (int64_t row = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6); row < N; row += (int64_t)gridDim.x * 4) {
        const float2 st = stats[row];
        float a1 = 0.0f, a2 = 0.0f;
        for (int col = lane; col < U; col += 64) {
            const float xh = (x[row * ld + col] * drop_factor(sp, row, col) - st.x) * st.y;
            const float dxh = d[row * ld + col] * dact_of(sp.act, fmaf(xh, gamma[col], beta[col])) * gamma[col];
            a1 += dxh;
            a2 = fmaf(dxh, xh, a2);
        }
        a1 = qb::wave_sum(a1);
        a2 = qb::wave_sum(a2);
        if (lane == 0) gsum[row] = make_float2(a1 / (float)U, a2 / (float)U);
    }
}
// Backward, step 2 (layer_norm): d gamma[c] = sum_rows d_v xh, d beta[c] = sum_rows d_v, d_v = d_a act'(v).  A block owns
// a contiguous row range, a thread a column; partial [block][2 U] doubles, added in block order by the kernel below
// (bitwise reproducible like every other weight gradient).
constexpr int kNormBlocks = 256;
__global__ __launch_bounds__(256) void norm_bwd_param_partial_kernel(const float* __restrict__ x, const float* __restrict__ d,
                                                                     int ld, int U, NormSpec sp,
                                                                     const float2* __restrict__ stats,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta,
                                                                     double* __restrict__ partial, int64_t N) {
    const int col = threadIdx.x;
    const int64_t per = (N + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = r0 + per < N ? r0 + per : N;
    double ag = 0.0, ab = 0.0;
    if (col < U)
        for (int64_t row = r0; row < r1; ++row) {
            const float2 st = stats[row / sp.rows_per_group];
            const float xh = (x[row * ld + col] * drop_factor(sp, row, col) - st.x) * st.y;
            const float dv = d[row * ld + col] * dact_of(sp.act, fmaf(xh, gamma[col], beta[col]));
            ag += (double)(dv * xh);
            ab += (double)dv;
        }
    if (col < U) {
        partial[(int64_t)blockIdx.x * 2 * U + col] = ag;
        partial[(int64_t)blockIdx.x * 2 * U + U + col] = ab;
    }
}
__global__ void norm_bwd_param_reduce_kernel(const double* __restrict__ partial, int nblk, int U,
                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * U) return;
    double a = 0.0;
    for (int b = 0; b < nblk; ++b) a += partial[(int64_t)b * 2 * U + c];
    (c < U ? dgamma[c] : dbeta[c - U]) = (float)a;
}
// Backward, step 3: d_x in place of d_a.  layer_norm: d_u = rstd (dxh - S1 - xh S2); then through the dropout
__global__ void norm_act_bwd_kernel(const float* __restrict__ x, float* __restrict__ d, int ld, int U, NormSpec sp,
                                    const float2* __restrict__ stats, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float2* __restrict__ gsum, int64_t N) {
    const int64_t n = N * (int64_t)U;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / U;
        const int col = (int)(i % U);
        const float keep = drop_factor(sp, row, col);
        const float u = x[row * ld + col] * keep;
        float du;
        if (sp.layer_norm) {
            const int64_t g = row / sp.rows_per_group;
            const float2 st = stats[g], gs = gsum[g];
            const float xh = (u - st.x) * st.y;
            const float dxh = d[row * ld + col] * dact_of(sp.act, fmaf(xh, gamma[col], beta[col])) * gamma[col];
            du = st.y * (dxh - gs.x - xh * gs.y);
        } else {
            du = d[row * ld + col] * dact_of(sp.act, u);
        }
        d[row * ld + col] = du * keep;
    }
}
// d *= act'(z) over the live columns (the skip path's activation)
__global__ void dact_mul_kernel(float* d, const float* __restrict__ z, int act, int64_t rows, int ld, int U) {
    const int64_t n = rows * (int64_t)U;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = (i / U) * ld + (i % U);
        d[e] *= dact_of(act, z[e]);
    }
}

static bool norm_mode(const qbold_encoder_shape* s) {
    return s->layer_norm || (s->dropout_rate > 0.0f && s->dropout_seed != 0);
}
static NormSpec norm_spec(const qbold_encoder_shape* s, int layer, const qbold_geometry* gm) {
    NormSpec sp;
    sp.layer_norm = s->layer_norm ? 1 : 0;
    sp.act = s->activation;
    const bool drop = s->dropout_rate > 0.0f && s->dropout_seed != 0;
    const float rate = s->dropout_rate < 0.999f ? s->dropout_rate : 0.999f;
    sp.drop_thresh = drop ? (uint32_t)lrintf(rate * 65536.0f) : 0u;
    sp.keep_scale = drop ? 1.0f / (1.0f - (float)sp.drop_thresh / 65536.0f) : 1.0f;
    sp.seed = s->dropout_seed;
    sp.layer = (uint32_t)layer;
    sp.rows_per_group = gm ? (int64_t)gm->X * gm->Y * gm->Z : 1;
    return sp;
}
// scratch behind the slabs: group statistics and backward sums (float2 per group each), parameter partials
static int64_t norm_scratch_floats(const qbold_encoder_shape* s, int64_t N) {
    return s && (s->layer_norm || s->dropout_rate > 0.0f) ? 4 * N + 2 * (int64_t)kNormBlocks * 2 * 256 + 16 : 0;
}

extern "C" int64_t qbold_train_workspace_floats(const qbold_encoder_shape* shape, int64_t N) {
    if (!shape || N < 0) return QBOLD_ERR_INVALID;
    const int64_t slots = 2 + 5 * (int64_t)shape->L + 5;  // activations + 5 delta scratch tensors
    return slots * N * train_ld(shape->U) + (int64_t)9 * kSlabBlocks * (64 * 64 + 64)   // 9: taps of a 3x3x1 kernel
           + norm_scratch_floats(shape, N);
}

static int train_fwd_impl(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* w,
                          const float* x, int stream_sel, float* ws, float* out_q, float* out_log_sigma,
                          int64_t N, void* stream, const qbold_geometry* gm) {
    QB_NEED_DEVICE(ctx);
    int rc = check_layerwise_shape(ctx, shape);
    if (rc) return rc;
    QB_REQUIRE(N > 0 && w && x && ws && out_q, "qbold_encoder_train_fwd: bad argument");
    if (gm) {
        QB_REQUIRE(shape->spatial_taps == 9, "spatial encoder needs 9-tap (3x3x1) residual kernels");
        QB_REQUIRE(shape->U <= 64, "spatial encoder path is built for U <= 64");
        QB_REQUIRE(stream_sel == 2, "only stream 2 has spatial convolutions");
    }
    QB_REQUIRE(stream_sel == 1 || stream_sel == 2, "qbold_encoder_train_fwd: stream must be 1 or 2");
    const qb::CanonLayout c = qb::make_canon(shape->T, shape->U, shape->L, shape->channelwise_gating, shape->spatial_taps, shape->layer_norm);
    const int T = c.T, U = c.U, L = c.L, G = c.G;
    const int ld = train_ld(U);
    Launcher k{ctx, (hipStream_t)stream, N, ld};
    auto slot = [&](int i) { return ws + (int64_t)i * N * ld; };
    // slot 0 holds the normalised input with its own row stride (T may exceed ld only if T > 64: not allowed)
    hipLaunchKernelGGL(normalise64_kernel, dim3(k.ew()), dim3(256), 0, k.s, ctx->dev, x, slot(0), ld, N);
    // activation_type (model.py:60, 115-120): relu, or Keras' exact gelu on the general kernels (forward only)
    const bool gelu = shape->activation == QBOLD_ACT_GELU;
    const int A_OUT = gelu ? ACT_GELU : ACT_RELU, A_IN = gelu ? ACT_GELU_IN : ACT_RELU_IN;
    k.xw(slot(0), ld, T, w + c.W0, U, 0, w + c.b0, slot(1), U, A_OUT, 0, nullptr);
    const float* cur = slot(1);
    float* head = slot(2 + 5 * L);  // scratch slot for the head output
    if (stream_sel == 1) {
        for (int l = 0; l < L; ++l) {
            const float* wb = w + c.blk0 + l * c.blk_stride;
            k.xw(cur, ld, U, wb + c.Wc, U, 0, wb + c.bc, slot(2 + l), U, A_OUT, 0, nullptr);
            cur = slot(2 + l);
        }
    } else {
        for (int l = 0; l < L; ++l) {
            const float* wb = w + c.blk0 + l * c.blk_stride;
            float* skip = slot(2 + 5 * l), *t = slot(3 + 5 * l), *r = slot(4 + 5 * l);
            float* gl = slot(5 + 5 * l), *bout = slot(6 + 5 * l);
            if (norm_mode(shape)) {
                // model.py:147-157 with add_normalizer live: skip as ever; the residual path goes
                //   a1 = act(LN1(D1(b)))  ->  p = conv1(a1)  [kept in the `t` slot, pre-normalizer]  ->
                //   a2 = act(LN2(D2(p)))  ->  r = conv2(a2)
                // a1 / a2 live in the backward's scratch slots (free during the forward; the backward recomputes them)
                const int ctr = (!gm && c.taps == 9) ? 4 * U * U : 0;
                float* a1 = slot(2 + 5 * L), *a2 = slot(2 + 5 * L + 1);
                float2* stats = reinterpret_cast<float2*>(ws + ((int64_t)(2 + 5 * L + 5) * N * ld +
                                                                (int64_t)9 * kSlabBlocks * (64 * 64 + 64)));
                const float* lnp = c.ln ? w + c.ln + (int64_t)l * 4 * U : nullptr;
                auto normalizer = [&](const float* xin, int which, float* out) {
                    const NormSpec sp = norm_spec(shape, 2 * l + which, gm);
                    if (sp.layer_norm) {
                        if (sp.rows_per_group == 1)
                            hipLaunchKernelGGL(norm_stats_rows_kernel, dim3((unsigned)((N + 3) / 4 < 4096 ? (N + 3) / 4 : 4096)),
                                               dim3(256), 0, k.s, xin, ld, U, sp, stats, N);
                        else
                            hipLaunchKernelGGL(norm_stats_kernel, dim3((unsigned)(N / sp.rows_per_group)), dim3(256), 0, k.s,
                                               xin, ld, U, sp, stats);
                    }
                    hipLaunchKernelGGL(norm_act_fwd_kernel, dim3(k.ew()), dim3(256), 0, k.s, xin, ld, U, sp, stats,
                                       lnp ? lnp + 2 * which * U : nullptr, lnp ? lnp + (2 * which + 1) * U : nullptr, out, N);
                };
                k.xw(cur, ld, U, wb + c.Wc, U, 0, wb + c.bc, skip, U, A_OUT, 0, nullptr);
                normalizer(cur, 0, a1);
                if (gm) k.conv3x3(a1, wb + c.Wr1, U, wb + c.br1, t, ACT_NONE, 0, nullptr, *gm);
                else k.xw(a1, ld, U, wb + c.Wr1 + ctr, U, 0, wb + c.br1, t, U, ACT_NONE, 0, nullptr);
                normalizer(t, 1, a2);
                if (gm) k.conv3x3(a2, wb + c.Wr2, U, wb + c.br2, r, ACT_NONE, 0, nullptr, *gm);
                else k.xw(a2, ld, U, wb + c.Wr2 + ctr, U, 0, wb + c.br2, r, U, ACT_NONE, 0, nullptr);
                k.xw(r, ld, U, wb + c.Wg, G, 0, wb + c.bg, gl, G, ACT_NONE, 0, nullptr);
                hipLaunchKernelGGL(gate_fwd_kernel, dim3(k.ew()), dim3(256), 0, k.s, gl, skip, r, bout,
                                   shape->gate_offset, U, G, ld, N);
                cur = bout;
                continue;
            }
            const bool fork = !gm && U <= 64 && ld == kLd && !(ctx->kernel_sel & 8192) && !gelu;
            if (!fork) k.xw(cur, ld, U, wb + c.Wc, U, 0, wb + c.bc, skip, U, A_OUT, 0, nullptr);
            // relu(b) feeds the first residual conv (model.py:151): applied to the rows as they are loaded
            if (gm) {  // 3x3x1 'same' convolutions, model.py:152-157
                k.conv3x3(cur, wb + c.Wr1, U, wb + c.br1, t, A_OUT | A_IN, 0, nullptr, *gm);
                k.conv3x3(t, wb + c.Wr2, U, wb + c.br2, r, ACT_NONE, 0, nullptr, *gm);
            } else {   // voxel batch: centre tap only
                const int ctr = c.taps == 9 ? 4 * U * U : 0;
                if (fork)   // skip and t from one pass over the block's input rows
                    hipLaunchKernelGGL(xw64_fork_kernel, dim3(k.grid()), dim3(256), sizeof(float) * 2 * 64 * kWs, k.s,
                                       cur, ld, U, U, wb + c.Wc, wb + c.bc, skip, wb + c.Wr1 + ctr, wb + c.br1, t, U, N);
                else
                    k.xw(cur, ld, U, wb + c.Wr1 + ctr, U, 0, wb + c.br1, t, U, A_OUT | A_IN, 0, nullptr);
                k.xw(t, ld, U, wb + c.Wr2 + ctr, U, 0, wb + c.br2, r, U, ACT_NONE, 0, nullptr);
            }
            const bool fuse_gate = G == U && U <= 64 && ld == kLd && !(ctx->kernel_sel & 2048) &&
                                   ((reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(gl) |
                                     reinterpret_cast<uintptr_t>(skip) | reinterpret_cast<uintptr_t>(bout)) & 15) == 0;
            if (fuse_gate) {   // gating GEMM with the blend as its epilogue
                hipLaunchKernelGGL(xw64_gate_kernel, dim3(k.grid()), dim3(256), sizeof(float) * 64 * kWs, k.s, r, ld,
                                   U, wb + c.Wg, G, wb + c.bg, gl, skip, bout, ld, G, shape->gate_offset, N);
            } else {
                k.xw(r, ld, U, wb + c.Wg, G, 0, wb + c.bg, gl, G, ACT_NONE, 0, nullptr);
                hipLaunchKernelGGL(gate_fwd_kernel, dim3(k.ew()), dim3(256), 0, k.s, gl, skip, r, bout,
                                   shape->gate_offset, U, G, ld, N);
            }
            cur = bout;
        }
    }
    // heads straight into the caller's [N][5] / [N][T] buffers
    (void)head;
    if (stream_sel == 2 && out_log_sigma && U <= 64 && ld == kLd && 5 + T <= 64 &&
        (reinterpret_cast<uintptr_t>(cur) & 15) == 0 && !(ctx->kernel_sel & 32768)) {
        hipLaunchKernelGGL(xw64_heads_kernel, dim3(k.grid()), dim3(256), sizeof(float) * 64 * kWs, k.s, cur, ld, U,
                           w + c.Wf, w + c.bf, w + c.Ws, w + c.bs, T, out_q, out_log_sigma, N);
    } else {
        rc = k.xw_ld(cur, ld, U, w + c.Wf, 5, 0, w + c.bf, out_q, 5, 5, ACT_NONE, 0, nullptr);
        if (rc) return rc;
        if (stream_sel == 2 && out_log_sigma) {
            rc = k.xw_ld(cur, ld, U, w + c.Ws, T, 0, w + c.bs, out_log_sigma, T, T, ACT_NONE, 0, nullptr);
            if (rc) return rc;
        }
    }
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_encoder_train_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape,
                                       const float* w, const float* x, int stream_sel, float* ws,
                                       float* out_q, float* out_log_sigma, int64_t N, void* stream) {
    return train_fwd_impl(ctx, shape, w, x, stream_sel, ws, out_q, out_log_sigma, N, stream, nullptr);
}

extern "C" int qbold_encoder_spatial_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape,
                                         const float* w, const float* x, const qbold_geometry* geom,
                                         float* ws, float* out_q, float* out_log_sigma, void* stream) {
    QB_REQUIRE(geom && geom->B > 0 && geom->X > 0 && geom->Y > 0 && geom->Z > 0,
               "qbold_encoder_spatial_fwd: bad geometry");
    const int64_t N = (int64_t)geom->B * geom->X * geom->Y * geom->Z;
    return train_fwd_impl(ctx, shape, w, x, 2, ws, out_q, out_log_sigma, N, stream, geom);
}

// the condition under which a stream-2 voxel-batch backward runs block_bwd_kernel (one definition: the forward
// leaves out what that kernel recomputes only if this says so)
static bool block_bwd_applies(const qbold_ctx* ctx, const qbold_encoder_shape* s, int64_t N) {
    return ctx && s && s->activation == QBOLD_ACT_RELU && !s->layer_norm && !(s->dropout_rate > 0.0f) && s->U >= 1 && s->U <= 64 &&
           s->channelwise_gating && s->L >= 1 && s->L <= 2 && s->T <= 27 &&
           s->T == ctx->dev.T && s->precision == QBOLD_ENC_F32 && N > 0 && N < ((int64_t)1 << 23) &&
           !(ctx->kernel_sel & 131072);
}
// ... and the one under which the block kernel also accumulates the weight gradients (block_bwd_dw_kernel)
static int dw_grid(const qbold_ctx* ctx, int64_t N) {
    const int64_t ntile = (N + 15) / 16, nb = (ntile + kDwThreads / 64 - 1) / (kDwThreads / 64);
    return (int)(nb < ctx->num_cus ? nb : ctx->num_cus);
}
static bool block_bwd_dw_applies(const qbold_ctx* ctx, const qbold_encoder_shape* s, int64_t N) {
    return block_bwd_applies(ctx, s, N) && !(ctx->kernel_sel & 262144) &&
           4 * (int64_t)dw_grid(ctx, N) * (kDwThreads / 64) <= 8 * (int64_t)kSlabBlocks;
}
extern "C" int qbold_encoder_train_bwd_recomputes(const qbold_ctx* ctx, const qbold_encoder_shape* shape, int64_t N) {
    return block_bwd_dw_applies(ctx, shape, N) ? 2 : (block_bwd_applies(ctx, shape, N) ? 1 : 0);
}

// The backward of the layer-wise forward with activation_type = 'gelu' (model.py:60, 115-120, 151, 155).  gelu is not
// invertible, so its derivative needs the PRE-activation, which the forward (shared with relu: post-activation tensors
// in the workspace slots) does not keep: each pre-activation is recomputed by the GEMM / convolution that produced
// it, without bias-free shortcuts, right where its derivative is needed, and the delta is multiplied by gelu'(z).  All
// on the general kernels (xw_kernel, conv9_kernel, xtd_kernel, xtd9_kernel); scratch: the head-delta slot once the
// heads are done, and the block's `skip` slot once gate_bwd_kernel has read it.
static int train_bwd_gelu(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* w, int stream_sel,
                          float* ws, const float* g_q, const float* g_ls, const double* sums, float* grad, int64_t N,
                          void* stream, const qbold_geometry* gm) {
    const qb::CanonLayout c = qb::make_canon(shape->T, shape->U, shape->L, shape->channelwise_gating, shape->spatial_taps, shape->layer_norm);
    const int T = c.T, U = c.U, L = c.L, G = c.G;
    const int ld = train_ld(shape->U);
    Launcher k{ctx, (hipStream_t)stream, N, ld};
    auto slot = [&](int i) { return ws + (int64_t)i * N * ld; };
    const int base = 2 + 5 * L;
    float* dA = slot(base), *dB = slot(base + 1), *dC = slot(base + 2), *dD = slot(base + 3), *dE = slot(base + 4);
    float* partial = ws + (int64_t)(base + 5) * N * ld;
    QB_HIP(hipMemsetAsync(grad, 0, sizeof(float) * c.total, k.s));
    hipLaunchKernelGGL(head_delta_kernel, dim3(k.ew()), dim3(256), 0, k.s, g_q, stream_sel == 2 ? g_ls : nullptr, T,
                       sums, dA, ld, N);
    const float* last = stream_sel == 1 ? slot(2 + L - 1) : slot(6 + 5 * (L - 1));
    int64_t nb = N / (16 * 16 * 4);
    const int64_t cap = ctx->num_cus < kSlabBlocks ? ctx->num_cus : kSlabBlocks;
    const int slabs = (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
    const int slabs9 = (int)((N + 511) / 512 < ctx->num_cus ? ((N + 511) / 512 > 0 ? (N + 511) / 512 : 1)
                                                           : (ctx->num_cus < kSlabBlocks ? ctx->num_cus : kSlabBlocks));
    const int64_t ne = N * (int64_t)ld;
    auto gelu_of = [&](const float* in, float* out) {
        hipLaunchKernelGGL(gelu_kernel, dim3(k.ew()), dim3(256), 0, k.s, in, out, ne);
    };
    auto times_dgelu = [&](float* d, const float* z) {   // d *= gelu'(z)
        hipLaunchKernelGGL(gelu_bwd_mul_kernel, dim3(k.ew()), dim3(256), 0, k.s, d, z, N, ld, U);
    };
    // heads: dWf, dbf (and dWs, dbs); dB = g_q Wf^T (+ g_ls Ws^T)
    k.xtd(last, U, dA, 5, partial, slabs, grad + c.Wf, 5, grad + c.bf, 0);
    k.xw(dA, ld, 5, w + c.Wf, 5, 1, nullptr, dB, U, ACT_NONE, 0, nullptr);
    if (stream_sel == 2 && g_ls) {
        k.xtd(last, U, dA + 5, T, partial, slabs, grad + c.Ws, T, grad + c.bs, 0);
        k.xw(dA + 5, ld, T, w + c.Ws, T, 1, nullptr, dB, U, ACT_NONE, 1, nullptr);
    }
    float* Z = dA;   // the head delta has been consumed: scratch for recomputed pre-activations
    if (stream_sel == 1) {
        for (int l = L - 1; l >= 0; --l) {
            const float* wb = w + c.blk0 + l * c.blk_stride;
            float* gb = grad + c.blk0 + l * c.blk_stride;
            const float* a_in = l == 0 ? slot(1) : slot(2 + l - 1);
            k.xw(a_in, ld, U, wb + c.Wc, U, 0, wb + c.bc, Z, U, ACT_NONE, 0, nullptr);   // z_l, model.py:145
            times_dgelu(dB, Z);
            k.xtd(a_in, U, dB, U, partial, slabs, gb + c.Wc, U, gb + c.bc, 0);
            k.xw(dB, ld, U, wb + c.Wc, U, 1, nullptr, dC, U, ACT_NONE, 0, nullptr);
            float* tmp = dB; dB = dC; dC = tmp;
        }
    } else {
        for (int l = L - 1; l >= 0; --l) {
            const float* wb = w + c.blk0 + l * c.blk_stride;
            float* gb = grad + c.blk0 + l * c.blk_stride;
            float* skip = slot(2 + 5 * l);
            const float* t = slot(3 + 5 * l), *r = slot(4 + 5 * l), *gl = slot(5 + 5 * l);
            const float* b_in = l == 0 ? slot(1) : slot(6 + 5 * (l - 1));
            const int ctr = (!gm && c.taps == 9) ? 4 * U * U : 0;
            // dB = d b_out  ->  dC = d skip (post-activation), dD = d r, dE = d gate logits
            hipLaunchKernelGGL(gate_bwd_kernel, dim3(k.grid()), dim3(256), 0, k.s, dB, gl, skip, r, dC, dD, dE,
                               shape->gate_offset, U, G, ld, N, 0);
            k.xtd(r, U, dE, G, partial, slabs, gb + c.Wg, G, gb + c.bg, 0);
            k.xw(dE, ld, G, wb + c.Wg, G, 1, nullptr, dD, U, ACT_NONE, 1, nullptr);       // d r += d gl Wg^T
            float* Gb = skip;                                                             // gelu(b_in), model.py:151
            gelu_of(b_in, Gb);
            if (gm) {
                k.xtd9(t, U, dD, partial, slabs9, gb + c.Wr2, gb + c.br2, *gm);
                k.conv3x3(dD, wb + c.Wr2, U, nullptr, dE, ACT_NONE, 1, nullptr, *gm, sums);     // d t (post-activation)
                k.conv3x3(Gb, wb + c.Wr1, U, wb + c.br1, Z, ACT_NONE, 0, nullptr, *gm);   // z_t, model.py:152
            } else {
                k.xtd(t, U, dD, U, partial, slabs, gb + c.Wr2 + ctr, U, gb + c.br2, 0);
                k.xw(dD, ld, U, wb + c.Wr2 + ctr, U, 1, nullptr, dE, U, ACT_NONE, 0, nullptr);
                k.xw(Gb, ld, U, wb + c.Wr1 + ctr, U, 0, wb + c.br1, Z, U, ACT_NONE, 0, nullptr);
            }
            times_dgelu(dE, Z);                                                           // d z_t
            if (gm) {
                k.xtd9(Gb, U, dE, partial, slabs9, gb + c.Wr1, gb + c.br1, *gm, 0);
                k.conv3x3(dE, wb + c.Wr1, U, nullptr, dB, ACT_NONE, 1, nullptr, *gm, sums);     // d gelu(b_in)
            } else {
                k.xtd(Gb, U, dE, U, partial, slabs, gb + c.Wr1 + ctr, U, gb + c.br1, 0, 0);
                k.xw(dE, ld, U, wb + c.Wr1 + ctr, U, 1, nullptr, dB, U, ACT_NONE, 0, nullptr);
            }
            times_dgelu(dB, b_in);                                                        // through Activation(b_in)
            k.xw(b_in, ld, U, wb + c.Wc, U, 0, wb + c.bc, Z, U, ACT_NONE, 0, nullptr);    // z_skip, model.py:148
            times_dgelu(dC, Z);
            k.xtd(b_in, U, dC, U, partial, slabs, gb + c.Wc, U, gb + c.bc, 0);
            k.xw(dC, ld, U, wb + c.Wc, U, 1, nullptr, dB, U, ACT_NONE, 1, nullptr);       // d b_in += d z_skip Wc^T
        }
    }
    // first layer: z_0 = n W0 + b0; dW0 = n^T (dB gelu'(z_0))
    k.xw(slot(0), ld, T, w + c.W0, U, 0, w + c.b0, Z, U, ACT_NONE, 0, nullptr);
    times_dgelu(dB, Z);
    k.xtd(slot(0), T, dB, U, partial, slabs, grad + c.W0, U, grad + c.b0, 0);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

// The backward of stream 2 with add_normalizer live (use_layer_norm and / or dropout in a training step), relu or gelu:
// the structure of train_bwd_gelu -- every pre-activation where its derivative is needed -- with each residual
// activation's normalizer differentiated in front of it (norm_act_bwd_kernel) and the GroupNormalization parameters'
// gradients beside the kernels' (behind the heads in the canonical blob).
static int train_bwd_norm(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* w, float* ws,
                          const float* g_q, const float* g_ls, const double* sums, float* grad, int64_t N, void* stream,
                          const qbold_geometry* gm) {
    const qb::CanonLayout c = qb::make_canon(shape->T, shape->U, shape->L, shape->channelwise_gating, shape->spatial_taps, shape->layer_norm);
    const int T = c.T, U = c.U, L = c.L, G = c.G;
    const int ld = train_ld(shape->U), act = shape->activation;
    const int A_OUT = act == QBOLD_ACT_GELU ? ACT_GELU : ACT_RELU;
    (void)A_OUT;
    Launcher k{ctx, (hipStream_t)stream, N, ld};
    auto slot = [&](int i) { return ws + (int64_t)i * N * ld; };
    const int base = 2 + 5 * L;
    float* dA = slot(base), *dB = slot(base + 1), *dC = slot(base + 2), *dD = slot(base + 3), *dE = slot(base + 4);
    float* partial = ws + (int64_t)(base + 5) * N * ld;
    float* nscratch = partial + (int64_t)9 * kSlabBlocks * (64 * 64 + 64);
    float2* stats = reinterpret_cast<float2*>(nscratch);
    float2* gsum = reinterpret_cast<float2*>(nscratch + 2 * N);
    double* ppart = reinterpret_cast<double*>(nscratch + 4 * N);
    QB_HIP(hipMemsetAsync(grad, 0, sizeof(float) * c.total, k.s));
    hipLaunchKernelGGL(head_delta_kernel, dim3(k.ew()), dim3(256), 0, k.s, g_q, g_ls, T, sums, dA, ld, N);
    const float* last = slot(6 + 5 * (L - 1));
    int64_t nb = N / (16 * 16 * 4);
    const int64_t cap = ctx->num_cus < kSlabBlocks ? ctx->num_cus : kSlabBlocks;
    const int slabs = (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
    const int slabs9 = (int)((N + 511) / 512 < ctx->num_cus ? ((N + 511) / 512 > 0 ? (N + 511) / 512 : 1)
                                                           : (ctx->num_cus < kSlabBlocks ? ctx->num_cus : kSlabBlocks));
    auto times_dact = [&](float* d, const float* z) {
        hipLaunchKernelGGL(dact_mul_kernel, dim3(k.ew()), dim3(256), 0, k.s, d, z, act, N, ld, U);
    };
    k.xtd(last, U, dA, 5, partial, slabs, grad + c.Wf, 5, grad + c.bf, 0);
    k.xw(dA, ld, 5, w + c.Wf, 5, 1, nullptr, dB, U, ACT_NONE, 0, nullptr);
    if (g_ls) {
        k.xtd(last, U, dA + 5, T, partial, slabs, grad + c.Ws, T, grad + c.bs, 0);
        k.xw(dA + 5, ld, T, w + c.Ws, T, 1, nullptr, dB, U, ACT_NONE, 1, nullptr);
    }
    float* Z = dA;   // the head delta has been consumed: scratch
    for (int l = L - 1; l >= 0; --l) {
        const float* wb = w + c.blk0 + l * c.blk_stride;
        float* gb = grad + c.blk0 + l * c.blk_stride;
        float* skip = slot(2 + 5 * l);
        const float* p = slot(3 + 5 * l), *r = slot(4 + 5 * l), *gl = slot(5 + 5 * l);
        const float* b_in = l == 0 ? slot(1) : slot(6 + 5 * (l - 1));
        const int ctr = (!gm && c.taps == 9) ? 4 * U * U : 0;
        const float* lnp = c.ln ? w + c.ln + (int64_t)l * 4 * U : nullptr;
        float* lng = c.ln ? grad + c.ln + (int64_t)l * 4 * U : nullptr;
        // recompute a = act(LN(D(x))) of normalizer `which` into `out`
        auto normalizer = [&](const float* xin, int which, float* out) {
            const NormSpec sp = norm_spec(shape, 2 * l + which, gm);
            if (sp.layer_norm) {
                if (sp.rows_per_group == 1)
                    hipLaunchKernelGGL(norm_stats_rows_kernel, dim3((unsigned)((N + 3) / 4 < 4096 ? (N + 3) / 4 : 4096)), dim3(256),
                                       0, k.s, xin, ld, U, sp, stats, N);
                else
                    hipLaunchKernelGGL(norm_stats_kernel, dim3((unsigned)(N / sp.rows_per_group)), dim3(256), 0, k.s, xin, ld,
                                       U, sp, stats);
            }
            hipLaunchKernelGGL(norm_act_fwd_kernel, dim3(k.ew()), dim3(256), 0, k.s, xin, ld, U, sp, stats,
                               lnp ? lnp + 2 * which * U : nullptr, lnp ? lnp + (2 * which + 1) * U : nullptr, out, N);
        };
        // d (in: d a, out: d x) through normalizer `which` applied to xin; `stats` must hold xin's statistics
        auto normalizer_bwd = [&](const float* xin, int which, float* d) {
            const NormSpec sp = norm_spec(shape, 2 * l + which, gm);
            const float* ga = lnp ? lnp + 2 * which * U : nullptr, *be = lnp ? lnp + (2 * which + 1) * U : nullptr;
            if (sp.layer_norm) {
                const int64_t groups = N / sp.rows_per_group;
                if (sp.rows_per_group == 1)
                    hipLaunchKernelGGL(norm_bwd_row_sums_kernel, dim3((unsigned)((N + 3) / 4 < 4096 ? (N + 3) / 4 : 4096)), dim3(256),
                                       0, k.s, xin, d, ld, U, sp, stats, ga, be, gsum, N);
                else
                    hipLaunchKernelGGL(norm_bwd_group_sums_kernel, dim3((unsigned)groups), dim3(256), 0, k.s, xin, d, ld, U, sp,
                                       stats, ga, be, gsum);
                const int nblk = (int)(N < kNormBlocks ? N : kNormBlocks);
                hipLaunchKernelGGL(norm_bwd_param_partial_kernel, dim3(nblk), dim3(256), 0, k.s, xin, d, ld, U, sp, stats, ga,
                                   be, ppart, N);
                hipLaunchKernelGGL(norm_bwd_param_reduce_kernel, dim3((2 * U + 255) / 256), dim3(256), 0, k.s, ppart, nblk, U,
                                   lng + 2 * which * U, lng + (2 * which + 1) * U);
            }
            hipLaunchKernelGGL(norm_act_bwd_kernel, dim3(k.ew()), dim3(256), 0, k.s, xin, d, ld, U, sp, stats, ga, be, gsum, N);
        };
        // dB = d b_out  ->  dC = d skip (post-activation), dD = d r, dE = d gate logits
        hipLaunchKernelGGL(gate_bwd_kernel, dim3(k.grid()), dim3(256), 0, k.s, dB, gl, skip, r, dC, dD, dE,
                           shape->gate_offset, U, G, ld, N, 0);
        k.xtd(r, U, dE, G, partial, slabs, gb + c.Wg, G, gb + c.bg, 0);
        k.xw(dE, ld, G, wb + c.Wg, G, 1, nullptr, dD, U, ACT_NONE, 1, nullptr);       // d r += d gl Wg^T
        // second convolution: input a2 = act(LN2(D2(p)))
        normalizer(p, 1, Z);                                                          // a2 (and p's statistics)
        if (gm) {
            k.xtd9(Z, U, dD, partial, slabs9, gb + c.Wr2, gb + c.br2, *gm);
            k.conv3x3(dD, wb + c.Wr2, U, nullptr, dE, ACT_NONE, 1, nullptr, *gm, sums);     // d a2
        } else {
            k.xtd(Z, U, dD, U, partial, slabs, gb + c.Wr2 + ctr, U, gb + c.br2, 0);
            k.xw(dD, ld, U, wb + c.Wr2 + ctr, U, 1, nullptr, dE, U, ACT_NONE, 0, nullptr);
        }
        normalizer_bwd(p, 1, dE);                                                     // d p
        // first convolution: input a1 = act(LN1(D1(b_in)))
        float* A1 = skip;                                                             // the skip slot is free from here
        normalizer(b_in, 0, A1);
        if (gm) {
            k.xtd9(A1, U, dE, partial, slabs9, gb + c.Wr1, gb + c.br1, *gm, 0);
            k.conv3x3(dE, wb + c.Wr1, U, nullptr, dB, ACT_NONE, 1, nullptr, *gm, sums);     // d a1
        } else {
            k.xtd(A1, U, dE, U, partial, slabs, gb + c.Wr1 + ctr, U, gb + c.br1, 0, 0);
            k.xw(dE, ld, U, wb + c.Wr1 + ctr, U, 1, nullptr, dB, U, ACT_NONE, 0, nullptr);
        }
        normalizer_bwd(b_in, 0, dB);                                                  // d b_in, residual path
        // skip path: z = b_in Wc + bc, d z = d skip act'(z)
        k.xw(b_in, ld, U, wb + c.Wc, U, 0, wb + c.bc, Z, U, ACT_NONE, 0, nullptr);
        times_dact(dC, Z);
        k.xtd(b_in, U, dC, U, partial, slabs, gb + c.Wc, U, gb + c.bc, 0);
        k.xw(dC, ld, U, wb + c.Wc, U, 1, nullptr, dB, U, ACT_NONE, 1, nullptr);       // d b_in += d z Wc^T
    }
    // first layer: z_0 = n W0 + b0; dW0 = n^T (dB act'(z_0))
    k.xw(slot(0), ld, T, w + c.W0, U, 0, w + c.b0, Z, U, ACT_NONE, 0, nullptr);
    times_dact(dB, Z);
    k.xtd(slot(0), T, dB, U, partial, slabs, grad + c.W0, U, grad + c.b0, 0);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

// g_head_q [N][5], g_head_ls [N][T] (stream 2 only; may be NULL), sums: device double[3] whose
// third entry is sum(mask) (NULL = gradients already normalised).  grad: canonical layout, overwritten.
static int train_bwd_impl(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* w,
                          int stream_sel, float* ws, const float* g_q, const float* g_ls,
                          const double* sums, float* grad, int64_t N, void* stream,
                          const qbold_geometry* gm) {
    QB_NEED_DEVICE(ctx);

    int rc = check_layerwise_shape(ctx, shape);
    if (rc) return rc;
    if (gm && shape->U > 64) {
        qb::set_error("qbold_encoder_spatial_bwd: the 3x3x1 path is built for U <= 64");
        return QBOLD_ERR_UNSUPPORTED;
    }
    QB_REQUIRE(N > 0 && w && ws && g_q && grad, "qbold_encoder_train_bwd: bad argument");
    QB_REQUIRE(stream_sel == 1 || stream_sel == 2, "qbold_encoder_train_bwd: stream must be 1 or 2");
    const qb::CanonLayout c = qb::make_canon(shape->T, shape->U, shape->L, shape->channelwise_gating, shape->spatial_taps, shape->layer_norm);
    const int T = c.T, U = c.U, L = c.L, G = c.G;
    const int ld = train_ld(shape->U);
    QB_REQUIRE(5 + shape->T <= ld, "qbold_encoder_train_bwd: the head delta (5 + T columns) exceeds the row stride");
    if (stream_sel == 2 && norm_mode(shape))
        return train_bwd_norm(ctx, shape, w, ws, g_q, g_ls, sums, grad, N, stream, gm);
    if (shape->activation == QBOLD_ACT_GELU)
        return train_bwd_gelu(ctx, shape, w, stream_sel, ws, g_q, g_ls, sums, grad, N, stream, gm);
    Launcher k{ctx, (hipStream_t)stream, N, ld};
    auto slot = [&](int i) { return ws + (int64_t)i * N * ld; };
    const int base = 2 + 5 * L;
    float* dA = slot(base), *dB = slot(base + 1), *dC = slot(base + 2), *dD = slot(base + 3),
          *dE = slot(base + 4);
    float* partial = ws + (int64_t)(base + 5) * N * ld;
    QB_HIP(hipMemsetAsync(grad, 0, sizeof(float) * c.total, k.s));
    const float* last = stream_sel == 1 ? slot(2 + L - 1) : slot(6 + 5 * (L - 1));
    // dWf, dbf (and dWs, dbs), d_last = g_q Wf^T (+ g_ls Ws^T)
    // xtd slabs per launch: whole rounds of two 1024-thread blocks per CU, about 1024 voxels per block
    // and tap (fewer, longer blocks leave a ragged last round; more pay the 16 KiB epilogue too often)
    // xtd slabs per launch: one 1024-thread block per CU, at least 16 four-voxel steps per wave
    auto slab_count = [&](int taps) {
        (void)taps;
#ifndef QB_SLAB_VOX
#define QB_SLAB_VOX 512
#endif
        int64_t nb = N / QB_SLAB_VOX;   // (a workgroup per CU from 131 k voxels on: at 1,024 voxels per slab a crop batch of
                                        // 190 k filled 185 of the 256 CUs)
        const int64_t cap = ctx->num_cus < kSlabBlocks ? ctx->num_cus : kSlabBlocks;
        return (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
    };
    const int slabs = slab_count(1);
    // the nine-tap kernel runs one 512-thread block per CU (144 accumulator registers per lane)
    const int slabs9 = (int)((N + 511) / 512 < ctx->num_cus ? ((N + 511) / 512 > 0 ? (N + 511) / 512 : 1)
                                                           : (ctx->num_cus < kSlabBlocks ? ctx->num_cus : kSlabBlocks));
    if (gm && !(ctx->kernel_sel & QBOLD_KSEL_SLAB_SUMS_SEPARATE))   // crop batches: the step's slab sums queued (QBOLD_KSEL bit QBOLD_KSEL_SLAB_SUMS_SEPARATE: one by one)
        k.defer_slabs(partial, partial + (int64_t)9 * kSlabBlocks * (64 * 64 + 64));
    const bool heads_one_pass = stream_sel == 2 && g_ls && U <= 64 && U % 4 == 0 && ld == kLd && 5 + T <= 16 &&
                                N < (1 << 23) && !(ctx->kernel_sel & (16384 | 1048576)) &&
                                ((reinterpret_cast<uintptr_t>(last) | reinterpret_cast<uintptr_t>(dB)) & 15) == 0;
    if (!heads_one_pass)   // head delta [N][64]: cols 0-4 = g_q, 5.. = g_ls, scaled by 1 / sum(mask)
        hipLaunchKernelGGL(head_delta_kernel, dim3(k.ew()), dim3(256), 0, k.s, g_q,
                           stream_sel == 2 ? g_ls : nullptr, T, sums, dA, ld, N);
    if (heads_one_pass) {
        // both heads' data and weight gradients in one pass over the rows (heads_bwd_kernel): no delta tensor
        float* hp = k.slab_region(slabs, 1, 2);
        hipLaunchKernelGGL(heads_bwd_kernel, dim3(slabs), dim3(512), 0, k.s, g_q, g_ls, T, sums, last, w + c.Wf, w + c.Ws,
                           U, dB, hp ? hp : partial, N);
        if (hp) {
            k.queue_slab(hp, slabs, grad + c.Wf, 5, U, 5, grad + c.bf, 0);
            k.queue_slab(hp, slabs, grad + c.Ws, T, U, T, grad + c.bs, 5);
        } else {
            k.reduce_cols(partial, slabs, grad + c.Wf, 5, U, 5, grad + c.bf, 0);
            k.reduce_cols(partial, slabs, grad + c.Ws, T, U, T, grad + c.bs, 5);
        }
    } else if (stream_sel == 2 && g_ls && U <= 64 && ld == kLd && 5 + T <= 64 && !(ctx->kernel_sel & 16384)) {
        // both heads at once: one weight-gradient pass over the 5 + T delta columns (slab columns 0-4 -> Wf,
        // 5 .. -> Ws) and one backward-data GEMM with the stacked weights [Wf^T; Ws^T] -- instead of two
        // passes each, one of them over the unaligned column block dA + 5
        float* Wst = partial + (int64_t)8 * kSlabBlocks * (64 * 64 + 64);   // scratch beyond the slabs in use
        k.xtd_only(last, U, dA, 5 + T, partial, slabs);
        k.reduce_cols(partial, slabs, grad + c.Wf, 5, U, 5, grad + c.bf, 0);
        k.reduce_cols(partial, slabs, grad + c.Ws, T, U, T, grad + c.bs, 5);
        hipLaunchKernelGGL(stack_heads_kernel, dim3(((5 + T) * U + 255) / 256), dim3(256), 0, k.s, w + c.Wf, w + c.Ws,
                           T, U, Wst);
        k.xw(dA, ld, 5 + T, Wst, U, 0, nullptr, dB, U, ACT_NONE, 0, nullptr);
    } else {
        k.xtd(last, U, dA, 5, partial, slabs, grad + c.Wf, 5, grad + c.bf, 0);
        k.xw(dA, ld, 5, w + c.Wf, 5, 1, nullptr, dB, U, ACT_NONE, 0, nullptr);
        if (stream_sel == 2 && g_ls) {
            k.xtd(last, U, dA + 5, T, partial, slabs, grad + c.Ws, T, grad + c.bs, 0);
            k.xw(dA + 5, ld, T, w + c.Ws, T, 1, nullptr, dB, U, ACT_NONE, 1, nullptr);
        }
    }
    // dB = gradient wrt the last activation tensor
    // Voxel batches of the LDS-resident shapes: per block one launch for the data side (block_bwd_kernel).  Its
    // two weight images (the block as the forward packs it, and its transpose) are packed here, into the ninth
    // slab region of the workspace, which voxel batches do not use.
    const bool blk_fused = stream_sel == 2 && !gm && block_bwd_applies(ctx, shape, N);
    const qb::EncLayout el = qb::make_enc_layout(T, 64, L);
    float* img_f = partial + (int64_t)8 * kSlabBlocks * (64 * 64 + 64) + 64 * 64;
    float* img_b = img_f + el.total;
    if (blk_fused) {
        float* wt = img_b + el.total;
        // (images and blob zeroed first: the pack kernel writes the slots it owns, the workspace is uninitialised)
        QB_HIP(hipMemsetAsync(img_f, 0, sizeof(float) * (2 * (size_t)el.total + c.total), k.s));
        hipLaunchKernelGGL(blob_transpose_kernel, dim3(64), dim3(256), 0, k.s, w, wt, c);
        qbold_encoder_shape plain = *shape;
        plain.gate_offset = 0.0f;   // the transposed image carries no biases
        rc = qbold_encoder_pack(ctx, shape, w, img_f, k.s);
        if (rc) return rc;
        rc = qbold_encoder_pack(ctx, &plain, wt, img_b, k.s);
        if (rc) return rc;
    }
    if (stream_sel == 1) {
        for (int l = L - 1; l >= 0; --l) {
            const float* wb = w + c.blk0 + l * c.blk_stride;
            float* gb = grad + c.blk0 + l * c.blk_stride;
            const float* a_out = slot(2 + l);
            const float* a_in = l == 0 ? slot(1) : slot(2 + l - 1);
            // through the relu: dC = dB * (a_out > 0)
            hipLaunchKernelGGL(mask_mul_kernel, dim3(k.ew()), dim3(256), 0, k.s, dB, a_out, dC, N * ld);
            k.xtd(a_in, U, dC, U, partial, slabs, gb + c.Wc, U, gb + c.bc, 0);
            k.xw(dC, ld, U, wb + c.Wc, U, 1, nullptr, dB, U, ACT_NONE, 0, nullptr);
        }
    } else {
        for (int l = L - 1; l >= 0; --l) {
            const float* wb = w + c.blk0 + l * c.blk_stride;
            float* gb = grad + c.blk0 + l * c.blk_stride;
            const float* skip = slot(2 + 5 * l), *t = slot(3 + 5 * l), *r = slot(4 + 5 * l);
            const float* gl = slot(5 + 5 * l);
            const float* b_in = l == 0 ? slot(1) : slot(6 + 5 * (l - 1));
            bool fused_in = false;
            if (blk_fused && block_bwd_dw_applies(ctx, shape, N)) {
                // data side AND the four weight gradients in one launch (block_bwd_dw_kernel), then four slab sums
                const int grid = dw_grid(ctx, N);
                const int nw = grid * (kDwThreads / 64);
                {
                    const size_t smem = sizeof(float) * (2 * qb::BLK_FLOATS + (kDwThreads / 64) * 16 * kTrStride);
                    QB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(block_bwd_dw_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                    hipLaunchKernelGGL(block_bwd_dw_kernel, dim3(grid), dim3(kDwThreads), smem, k.s,
                                       img_f + el.blk0 + l * el.blk_stride, img_b + el.blk0 + l * el.blk_stride, b_in,
                                       dB, dB, partial, nw, U, N);
                    const int ctr = c.taps == 9 ? 4 * U * U : 0;
                    const int64_t per = (int64_t)nw * (64 * 64 + 64);
                    // the four matrices' slabs in one launch (grid y = matrix): 1,040 workgroups instead of four rounds
                    // of 260 on 256 CUs
                    const dim3 rg((64 * 64 + 64) / 64, 4);
                    SlabTargets tg;
                    tg.dW[0] = gb + c.Wg;        tg.db[0] = gb + c.bg;  tg.ldw[0] = G; tg.kdim[0] = U; tg.ndim[0] = G;
                    tg.dW[1] = gb + c.Wr2 + ctr; tg.db[1] = gb + c.br2; tg.ldw[1] = U; tg.kdim[1] = U; tg.ndim[1] = U;
                    tg.dW[2] = gb + c.Wr1 + ctr; tg.db[2] = gb + c.br1; tg.ldw[2] = U; tg.kdim[2] = U; tg.ndim[2] = U;
                    tg.dW[3] = gb + c.Wc;        tg.db[3] = gb + c.bc;  tg.ldw[3] = U; tg.kdim[3] = U; tg.ndim[3] = U;
                    hipLaunchKernelGGL(slab_reduce_many_kernel, rg, dim3(1024), 0, k.s, partial, nw, per, tg);
                    continue;
                }
            }
            if (blk_fused) {   // the block's data side in one launch (block_bwd_kernel), then its four weight gradients
                const size_t smem = sizeof(float) * 2 * qb::BLK_FLOATS;
                QB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(block_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                const int64_t per_blk = (kBlkThreads / 64) * 16, nb = (N + per_blk - 1) / per_blk;
                hipLaunchKernelGGL(block_bwd_kernel, dim3((unsigned)(nb < ctx->num_cus ? nb : ctx->num_cus)),
                                   dim3(kBlkThreads), smem, k.s, img_f + el.blk0 + l * el.blk_stride,
                                   img_b + el.blk0 + l * el.blk_stride, b_in, dB, dE, dD, dA, dC, dB, U, N);
                const int ctr = c.taps == 9 ? 4 * U * U : 0;
                k.xtd(r, U, dE, G, partial, slabs, gb + c.Wg, G, gb + c.bg, 0);
                k.xtd(t, U, dD, U, partial, slabs, gb + c.Wr2 + ctr, U, gb + c.br2, 0);
                k.xtd(b_in, U, dA, U, partial, slabs, gb + c.Wr1 + ctr, U, gb + c.br1, 0, 1);
                k.xtd(b_in, U, dC, U, partial, slabs, gb + c.Wc, U, gb + c.bc, 0);
                continue;
            }
            // dB = d b_out.  dC = d skip_pre, dD = d r, dE = d gate logits
            const bool gate_fused = G == U && U <= 64 && U % 4 == 0 && ld == kLd && !(ctx->kernel_sel & 2048) &&
                                    ((reinterpret_cast<uintptr_t>(dB) | reinterpret_cast<uintptr_t>(gl) |
                                      reinterpret_cast<uintptr_t>(skip) | reinterpret_cast<uintptr_t>(r) |
                                      reinterpret_cast<uintptr_t>(dC) | reinterpret_cast<uintptr_t>(dD) |
                                      reinterpret_cast<uintptr_t>(dE)) & 15) == 0;
            if (gate_fused) {   // gate backward and d r += dE Wg^T in one launch (QBOLD_KSEL_SEPARATE_GATE: two)
                const size_t smem = sizeof(float) * (size_t)((U + 15) & ~15) * kWs;
                hipLaunchKernelGGL(gate_bwd_wg_kernel, dim3(k.grid()), dim3(256), smem, k.s, dB, gl, skip, r, wb + c.Wg, dC,
                                   dD, dE, shape->gate_offset, U, N);
            } else {
                hipLaunchKernelGGL(gate_bwd_kernel, dim3(k.grid()), dim3(256), 0, k.s, dB, gl, skip, r, dC, dD,
                                   dE, shape->gate_offset, U, G, ld, N);
            }
            // gating conv: dWg = r^T dE; d r += dE Wg^T
            k.xtd(r, U, dE, G, partial, slabs, gb + c.Wg, G, gb + c.bg, 0);
            if (!gate_fused) k.xw(dE, ld, G, wb + c.Wg, G, 1, nullptr, dD, U, ACT_NONE, 1, nullptr);
            if (gm) {
                // second residual conv (3x3x1): dK2[tap] = t[nbr]^T dD; d t_pre = conv^T(dD) * (t > 0) -> dE
                k.xtd9(t, U, dD, partial, slabs9, gb + c.Wr2, gb + c.br2, *gm);
                k.conv3x3(dD, wb + c.Wr2, U, nullptr, dE, ACT_NONE, 1, t, *gm, sums);
                // first residual conv: input relu(b_in)
                k.xtd9(b_in, U, dE, partial, slabs9, gb + c.Wr1, gb + c.br1, *gm, 1);
                k.conv3x3(dE, wb + c.Wr1, U, nullptr, dB, ACT_NONE, 1, b_in, *gm, sums);
            } else {
                const int ctr = c.taps == 9 ? 4 * U * U : 0;
                // second residual conv: dWr2 = t^T dD; d t_pre = (dD Wr2^T) * (t > 0)  -> dE
                k.xtd(t, U, dD, U, partial, slabs, gb + c.Wr2 + ctr, U, gb + c.br2, 0);
                k.xw(dD, ld, U, wb + c.Wr2 + ctr, U, 1, nullptr, dE, U, ACT_NONE, 0, t);
                // first residual conv: input relu(b_in): dWr1 = relu(b_in)^T dE; d b_in = (dE Wr1^T) * (b_in > 0)
                k.xtd(b_in, U, dE, U, partial, slabs, gb + c.Wr1 + ctr, U, gb + c.br1, 0, 1);
                fused_in = U <= 64 && ld == kLd && !(ctx->kernel_sel & 4096);
                if (!fused_in) k.xw(dE, ld, U, wb + c.Wr1 + ctr, U, 1, nullptr, dB, U, ACT_NONE, 0, b_in);
            }
            // skip conv: dWc = b_in^T dC; d b_in += dC Wc^T
            k.xtd(b_in, U, dC, U, partial, slabs, gb + c.Wc, U, gb + c.bc, 0);
            if (fused_in) {   // d b_in = (dE Wr1^T) * (b_in > 0) + dC Wc^T in one pass
                const int ctr = c.taps == 9 ? 4 * U * U : 0;
                const size_t smem = sizeof(float) * 2 * 64 * kWs;
                hipLaunchKernelGGL(xw64_dual_kernel, dim3(k.grid()), dim3(256), smem, k.s, dE, wb + c.Wr1 + ctr, b_in,
                                   dC, wb + c.Wc, ld, U, U, U, dB, N);
            } else {
                k.xw(dC, ld, U, wb + c.Wc, U, 1, nullptr, dB, U, ACT_NONE, 1, nullptr);
            }
        }
    }
    // first layer: delta_pre = dB * (h > 0); dW0 = n^T delta_pre.  Columns T .. of n are zero up to the next multiple
    // of 4 (normalise64_kernel, encoder_train_fwd_kernel): the reader may take whole float4s.
    // (Having block 0's one-launch backward apply the mask it holds anyway was measured and lost: 16 selects more and
    // the 512-register kernel spills 336 bytes per lane instead of 44, 1.00 ms against 0.78.)
    k.xtd(slot(0), T, dB, U, partial, slabs, grad + c.W0, U, grad + c.b0, 0, 0, slot(1), (T + 3) & ~3);
    k.flush_slabs();
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_encoder_train_bwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape,
                                       const float* w, int stream_sel, float* ws, const float* g_q,
                                       const float* g_ls, const double* sums, float* grad, int64_t N,
                                       void* stream) {
    return train_bwd_impl(ctx, shape, w, stream_sel, ws, g_q, g_ls, sums, grad, N, stream, nullptr);
}

extern "C" int qbold_encoder_spatial_bwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape,
                                         const float* w, const qbold_geometry* geom, float* ws,
                                         const float* g_q, const float* g_ls, const double* sums,
                                         float* grad, void* stream) {
    QB_REQUIRE(geom && geom->B > 0 && geom->X > 0 && geom->Y > 0 && geom->Z > 0,
               "qbold_encoder_spatial_bwd: bad geometry");
    QB_REQUIRE(shape && shape->spatial_taps == 9, "qbold_encoder_spatial_bwd: needs 9-tap kernels");
    const int64_t N = (int64_t)geom->B * geom->X * geom->Y * geom->Z;
    return train_bwd_impl(ctx, shape, w, 2, ws, g_q, g_ls, sums, grad, N, stream, geom);
}

namespace {
// smoothness_loss (model.py:726-754): p = forward_transform(mean) / range = sigmoid(mu) + min/range;
// sum over x- and y-neighbour pairs with both masks > 0 of |p_v - p_w| (OEF and DBV channels).
// Each voxel owns the pairs towards +x and +y for the sum, and gathers the sign of all four
// differences it takes part in for the gradient (no atomics on g_q).
__global__ void smoothness_kernel(const float* __restrict__ q, const float* __restrict__ mask,
                                  qbold_geometry gm, float weight, float* __restrict__ g_q,
                                  double* __restrict__ tv_sum, int64_t N) {
    __shared__ double red[4];
    float acc = 0.0f;
    const int64_t yz = (int64_t)gm.Y * gm.Z;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < N;
         v += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)((v / yz) % gm.X), y = (int)((v / gm.Z) % gm.Y);
        const bool mv = mask[v] > 0.0f;
        const float so = 1.0f / (1.0f + expf(-q[5 * v])), sd = 1.0f / (1.0f + expf(-q[5 * v + 2]));
        float go = 0.0f, gd = 0.0f;
        const int64_t off[4] = {yz, (int64_t)gm.Z, -yz, -(int64_t)gm.Z};
        const bool inside[4] = {x + 1 < gm.X, y + 1 < gm.Y, x > 0, y > 0};
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (!inside[n]) continue;
            const int64_t w = v + off[n];
            if (!(mv && mask[w] > 0.0f)) continue;
            const float wo = 1.0f / (1.0f + expf(-q[5 * w])), wd = 1.0f / (1.0f + expf(-q[5 * w + 2]));
            const float dO = so - wo, dD = sd - wd;
            if (n < 2) acc += fabsf(dO) + fabsf(dD);
            go += (dO > 0.0f) - (dO < 0.0f);
            gd += (dD > 0.0f) - (dD < 0.0f);
        }
        if (g_q) {
            g_q[5 * v] += weight * go * so * (1.0f - so);
            g_q[5 * v + 2] += weight * gd * sd * (1.0f - sd);
        }
    }
    acc = qb::wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (double)acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += red[w];
        atomic_add_fixed(tv_sum, a, kTvFixed);
    }
}
}  // namespace

extern "C" int qbold_smoothness(const qbold_ctx* ctx, const float* q, const float* mask,
                                const qbold_geometry* geom, float weight, float* g_q, double* tv_sum,
                                void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_REQUIRE(q && mask && geom && tv_sum, "qbold_smoothness: null argument");
    QB_REQUIRE(geom->B > 0 && geom->X > 0 && geom->Y > 0 && geom->Z > 0, "qbold_smoothness: bad geometry");
    const int64_t N = (int64_t)geom->B * geom->X * geom->Y * geom->Z;
    hipStream_t s = (hipStream_t)stream;
    QB_HIP(hipMemsetAsync(tv_sum, 0, sizeof(double), s));
    int64_t nb = (N + 255) / 256;
    int64_t cap = (int64_t)ctx->num_cus * 8;
    hipLaunchKernelGGL(smoothness_kernel, dim3((int)(nb < cap ? nb : cap)), dim3(256), 0, s, q, mask, *geom,
                       weight, g_q, tv_sum, N);
    hipLaunchKernelGGL(fixed_to_double_kernel, dim3(1), dim3(64), 0, s, tv_sum, 1, 1.0 / kTvFixed);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

// synthetic_data_loss (model.py:449-514, use_mvg, no r2p term) and its gradient with respect to q:
// loss_v [N] (may be NULL), g_q [N][5] = scale * d loss_v / d q.
extern "C" int qbold_synth_loss_bwd(const qbold_ctx* ctx, const float* y_true, int ld_y, const float* q,
                                    float* g_q, float* loss_v, float scale, double inv_gamma_alpha,
                                    double inv_gamma_beta, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_REQUIRE(N > 0 && y_true && q && g_q && ld_y >= 2, "qbold_synth_loss_bwd: bad argument");
    QB_REQUIRE(inv_gamma_alpha >= 0.0 && inv_gamma_beta >= 0.0, "qbold_synth_loss_bwd: negative inverse-gamma parameter");
    const bool ig = inv_gamma_alpha * inv_gamma_beta > 0.0;  // model.py:492
    const float ig_a = ig ? (float)inv_gamma_alpha : 0.0f, ig_b = ig ? (float)inv_gamma_beta : 0.0f;
    const float ig_c0 = ig ? (float)(lgamma(inv_gamma_alpha) - inv_gamma_alpha * log(inv_gamma_beta)) : 0.0f;
    int64_t nb = (N + 255) / 256;
    int64_t cap = (int64_t)ctx->num_cus * 8;
    hipLaunchKernelGGL(nlogp_bwd_kernel, dim3((int)(nb < cap ? nb : cap)), dim3(256), 0,
                       (hipStream_t)stream, y_true, ld_y, q, g_q, 5, loss_v, scale, ig_a, ig_b, ig_c0, N);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_hyper_prior_bwd(const qbold_ctx* ctx, const float* q, const double* ig_host, float scale,
                                     float* g_q, float* loss_v, double* stats, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_REQUIRE(N > 0 && q && ig_host && (g_q || loss_v || stats), "qbold_hyper_prior_bwd: bad argument");
    for (int k = 0; k < 4; ++k)
        QB_REQUIRE(ig_host[k] > 0.0 && std::isfinite(ig_host[k]), "qbold_hyper_prior_bwd: inverse-gamma parameters must be positive");
    hipStream_t s = (hipStream_t)stream;
    if (stats) QB_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(double), s));
    const double ao = ig_host[0], bo = ig_host[1], ad = ig_host[2], bd = ig_host[3];
    int64_t nb = (N + 255) / 256;
    int64_t cap = (int64_t)ctx->num_cus * 8;
    hipLaunchKernelGGL(hyper_prior_kernel, dim3((int)(nb < cap ? nb : cap)), dim3(256), 0, s, q, (float)ao, (float)bo,
                       (float)(lgamma(ao) - ao * log(bo)), (float)ad, (float)bd, (float)(lgamma(ad) - ad * log(bd)),
                       scale, g_q, loss_v, stats, N);
    if (stats) hipLaunchKernelGGL(fixed_to_double_kernel, dim3(1), dim3(64), 0, s, stats, 4, 1.0 / kStatsFixed);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

// One tfa.optimizers.AdamW step on a flat parameter blob (train.py:308-310, 382-385):
// decoupled weight decay `wd` (already scheduled), Keras-Adam bias correction at step t >= 1.
extern "C" int qbold_adamw_step(const qbold_ctx* ctx, float* params, const float* grads, float* m,
                                float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                                double weight_decay, int64_t t, void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_REQUIRE(n > 0 && params && grads && m && v && t >= 1, "qbold_adamw_step: bad argument");
    const double lr_t = lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t));
    int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)(nb < 1024 ? nb : 1024)), dim3(256), 0, (hipStream_t)stream,
                       params, grads, m, v, n, (float)lr_t, (float)beta1, (float)beta2, (float)eps,
                       (float)weight_decay);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}
