// qbold_ctx.h -- host-side context of libqbold_hip.so (private; the public face is
// include/qbold_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/qbold_hip.h"
#include "qbold_dev.h"

struct qbold_ctx {
    int device = 0;
    qbold_consts consts{};
    qbold_loss_cfg loss{};
    QbDev dev{};                    // folded constants, passed by value to kernels
    float4* d_tab = nullptr;        // device copy of the F(x) cubic table
    std::vector<float> h_tab;       // host copy (4 floats per segment)
    float4* d_gtab = nullptr;       // per-tau OEF-indexed table of the sampling fast path (qbold_dev.h, GtLds)
    std::vector<float> h_gtab;      // host copy: [j - 1][segment][4]
    bool gtab_ok = false;           // built: T = 11, spin echo at index 2 with tau = 0 there, taus mirrored about it
    bool grid_mirrors = false;      // tau = 0 at the spin-echo image and the taus mirror about it (ctx.hip)
    int num_cus = 256;
    float dF_node0_ref = 0.0f;      // TF-gradient slope of Simpson node 0 (see qbold_ctx_set_grad_node0)
    int kernel_sel = 0;             // qbold_ctx_set_kernel_selection: which of several EQUIVALENT kernels runs
};

namespace qb {
void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);
}  // namespace qb

#define QB_HIP(call)                                              \
    do {                                                          \
        hipError_t _e = (call);                                   \
        if (_e != hipSuccess) return qb::hip_fail(_e, #call);     \
    } while (0)

#define QB_NEED_DEVICE(ctx)                                                   \
    do {                                                                      \
        if (!(ctx) || (ctx)->device < 0) {                                    \
            qb::set_error("context has no device (host-only or null)");       \
            return QBOLD_ERR_NO_DEVICE;                                       \
        }                                                                     \
    } while (0)

// entry points whose kernels implement relu only (gelu: the layer-wise forward, include/qbold_hip.h)
#define QB_RELU_ONLY(shape, what)                                                                      \
    do {                                                                                               \
        if ((shape) && ((shape)->activation != QBOLD_ACT_RELU || (shape)->layer_norm)) {               \
            qb::set_error(what ": activation 'gelu' and use_layer_norm run through the layer-wise "    \
                               "entry points (qbold_encoder_train_fwd / _bwd, qbold_encoder_spatial_fwd / _bwd)");   \
            return QBOLD_ERR_UNSUPPORTED;                                                              \
        }                                                                                              \
    } while (0)

#define QB_REQUIRE(cond, msg)              \
    do {                                   \
        if (!(cond)) {                     \
            qb::set_error(msg);            \
            return QBOLD_ERR_INVALID;      \
        }                                  \
    } while (0)
