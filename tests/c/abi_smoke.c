/* Plain-C consumer of include/qbold_hip.h: proves the boundary is a C ABI (no C++ or torch types).
 * Builds a host-only context (device = -1), reads the tau grid and the F(x) table, and checks
 * that compute entry points refuse to run without a device.  Compiled and run by
 * tests/test_host.py with gcc. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "qbold_hip.h"

int main(void) {
    qbold_consts c;
    memset(&c, 0, sizeof c);
    c.gamma = 2.67513e8; c.b0 = 3.0; c.dchi = 2.64e-7; c.te = 0.074; c.r2t = 11.5;
    c.tr = 3.0; c.ti = 1.21; c.t1b = 1.58; c.hct = 0.34;
    c.tau_start = -0.016; c.tau_end = 0.065; c.tau_step = 0.008;
    c.full_model = 1; c.include_blood = 1;
    qbold_loss_cfg l;
    memset(&l, 0, sizeof l);
    qbold_ctx* ctx = NULL;
    if (qbold_abi_version() != QBOLD_ABI_VERSION) return 1;
    if (qbold_ctx_create(&c, &l, -1, &ctx) != QBOLD_OK || !ctx) { printf("%s\n", qbold_last_error()); return 2; }
    if (qbold_ctx_num_taus(ctx) != 11 || qbold_ctx_se_idx(ctx) != 2) return 3;
    float taus[QBOLD_MAX_T];
    if (qbold_ctx_taus(ctx, taus) != QBOLD_OK || taus[2] != 0.0f || fabsf(taus[10] - 0.064f) > 1e-7f) return 4;
    float x[3] = {0.0f, 1.0f, 4.0f}, F[3], dF[3];
    if (qbold_ctx_table_eval(ctx, x, F, dF, 3) != QBOLD_OK) return 5;
    /* float32-semantics values of the Simpson-129 integral (0.2886, 3.0380) */
    if (F[0] != 0.0f || fabsf(F[1] - 0.28860f) > 2e-4f || fabsf(F[2] - 3.0380f) > 2e-3f) return 6;
    if (qbold_signal_fwd(ctx, NULL, NULL, 4, NULL) != QBOLD_ERR_NO_DEVICE) return 7;
    qbold_encoder_shape s = {11, 60, 2, 1, -3.0f, 1};
    if (qbold_encoder_num_params(&s) != 30976) return 8;
    qbold_ctx_destroy(ctx);
    printf("abi ok: F(1)=%.6f F(4)=%.6f dF(4)=%.6f\n", F[1], F[2], dF[2]);
    return 0;
}
