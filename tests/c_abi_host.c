/* A plain-C host for the C ABI (include/ursa_hip.h): no Python, no torch. Allocates with the HIP
 * runtime, runs the fused update (Philox noise, momentum, weight decay, fused grad zeroing) and the
 * BMA reduction through libursa_hip.so and checks them against the CPU oracle (liboracle.so).
 * Built and run by tests/test_c_abi_host.py on the GPU box:
 *   gcc -D__HIP_PLATFORM_AMD__ tests/c_abi_host.c -Iinclude -I/opt/rocm/include -Lursabench_amd/csrc -lursa_hip
 *       -Loracle -loracle -L/opt/rocm/lib -lamdhip64 -lm
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ursa_hip.h"

int oracle_sgmcmc_step_f32(float*, float*, float*, const float*, float*, int64_t, float, float, float, float, float,
                           uint64_t, uint64_t, uint32_t);
int oracle_bma_accumulate_f32(const float*, float*, float*, float*, const float*, int32_t, int64_t, int32_t, float,
                              float, uint32_t);

int oracle_bn_relu_fwd_f32(const float*, float*, const float*, const float*, float*, float*, float*, float*, int64_t,
                           int64_t, int64_t, float, float, int);
int oracle_bn_relu_bwd_f32(const float*, const float*, float*, const float*, const float*, const float*, const float*,
                           float*, float*, int64_t, int64_t, int64_t, int);
int oracle_bn_relu_bwd_gated_f32(const float*, const float*, float*, const float*, const float*, const float*, const float*,
                                 float*, float*, int64_t, int64_t, int64_t, int, const int32_t*, const uint8_t*, int64_t);

int oracle_conv_wgrad_f32(const float*, const float*, float*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t);
int oracle_conv1x1_f32(const float*, const float*, float*, int64_t, int64_t, int64_t, int64_t, int);
int oracle_conv3x3_f32(const float*, const float*, float*, int64_t, int64_t, int64_t, int64_t, int64_t, int, int64_t);
int oracle_preact_fwd_f32(const float*, const float*, const float*, float*, float*, const float*, const float*, float*, float*, float*, double*,
                          int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, float, float, int);
int oracle_preact_bwd_f32(const float*, const float*, const float*, const float*, float*, double*, int64_t, int64_t, int64_t, int64_t, int64_t,
                          int64_t);
int oracle_bn_bwd_dx_f32(const float*, const float*, const float*, float*, const float*, const float*, const double*, float*, float*, int64_t,
                         int64_t, int64_t);
int oracle_fc_ce_f32(const float*, const float*, const float*, const int64_t*, float*, float*, float*, float*, float*, int64_t, int64_t, int64_t,
                     int64_t);

#define CHECK(x) do { int rc_ = (x); if (rc_) { printf("FAIL %s -> %d (%s)\n", #x, rc_, ursa_strerror(rc_)); return 1; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) / 8388608.0f) - 1.0f; }

int main(void)
{
    if (ursa_abi_version() != URSA_ABI_VERSION) { printf("FAIL abi version\n"); return 1; }
    const int64_t n = 61706 + 3;                 /* not a multiple of 4: exercises the scalar tail */
    const size_t bytes = (size_t)n * sizeof(float);
    float *th = malloc(bytes), *gr = malloc(bytes), *mo = malloc(bytes), *back = malloc(bytes);
    unsigned seed = 12345u;
    for (int64_t i = 0; i < n; ++i) { th[i] = frand(&seed); gr[i] = 3.0f * frand(&seed); mo[i] = 0.1f * frand(&seed); }
    float *dth, *dgr, *dmo;
    CHECK(hipMalloc((void**)&dth, bytes)); CHECK(hipMalloc((void**)&dgr, bytes)); CHECK(hipMalloc((void**)&dmo, bytes));
    CHECK(hipMemcpy(dth, th, bytes, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dgr, gr, bytes, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dmo, mo, bytes, hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const uint32_t flags = URSA_STEP_NOISE | URSA_STEP_WD | URSA_STEP_ZERO_GRAD;
    for (int k = 0; k < 3; ++k) {                /* three steps: device vs oracle trajectories must stay bit-equal */
        CHECK(ursa_sgmcmc_step_f32(dth, dgr, dmo, NULL, NULL, n, 0.05f, 0.9f, 4.0f / 50000.0f, 0.1f, 50000.0f, 77u,
                                   (uint64_t)k, flags, st));
        oracle_sgmcmc_step_f32(th, gr, mo, NULL, NULL, n, 0.05f, 0.9f, 4.0f / 50000.0f, 0.1f, 50000.0f, 77u, (uint64_t)k,
                               flags);
        for (int64_t i = 0; i < n; ++i) gr[i] = frand(&seed);
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(dgr, gr, bytes, hipMemcpyHostToDevice));
    }
    CHECK(hipMemcpy(back, dth, bytes, hipMemcpyDeviceToHost));
    if (memcmp(back, th, bytes)) { printf("FAIL theta differs from the oracle\n"); return 1; }
    CHECK(hipMemcpy(back, dmo, bytes, hipMemcpyDeviceToHost));
    if (memcmp(back, mo, bytes)) { printf("FAIL momentum differs from the oracle\n"); return 1; }

    /* BMA reduction: S=5 members, B=777 rows, C=10 classes */
    const int S = 5, C = 10; const int64_t B = 777;
    const size_t zb = (size_t)S * B * C * sizeof(float), pb = (size_t)B * C * sizeof(float), eb = (size_t)B * sizeof(float);
    float *z = malloc(zb), *p = calloc(B * C, sizeof(float)), *e = calloc(B, sizeof(float)), *pd = malloc(pb), *ed = malloc(eb);
    for (size_t i = 0; i < (size_t)S * B * C; ++i) z[i] = 4.0f * frand(&seed);
    float *dz, *dp, *de;
    CHECK(hipMalloc((void**)&dz, zb)); CHECK(hipMalloc((void**)&dp, pb)); CHECK(hipMalloc((void**)&de, eb));
    CHECK(hipMemcpy(dz, z, zb, hipMemcpyHostToDevice)); CHECK(hipMemset(dp, 0, pb)); CHECK(hipMemset(de, 0, eb));
    const float omg = (float)(1 - 1e-4), goc = (float)(1e-4 / C);
    CHECK(ursa_bma_accumulate_f32(dz, dp, de, NULL, NULL, S, B, C, omg, goc, 0, st));
    CHECK(hipStreamSynchronize(st));
    oracle_bma_accumulate_f32(z, p, e, NULL, NULL, S, B, C, omg, goc, 0);
    CHECK(hipMemcpy(pd, dp, pb, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ed, de, eb, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int64_t i = 0; i < B * C; ++i) { double r = fabs(pd[i] - p[i]) / (fabs(p[i]) + 1e-8); if (r > worst) worst = r; }
    for (int64_t i = 0; i < B; ++i) { double r = fabs(ed[i] - e[i]) / (fabs(e[i]) + 1e-6); if (r > worst) worst = r; }
    if (worst > 1e-5) { printf("FAIL bma relative error %g\n", worst); return 1; }
    /* ABI v2: two chains in ONE self-advancing launch over [2, stride] slabs, control blocks built by this C host.
     * Chain k must equal the oracle's single-chain update with ctl[k]'s scalars / key / call index, for 3 steps. */
    {
        const int K = 2; const int64_t nc = 4096 + 8 + 2, stride = 4096 + 16;
        const size_t sb = (size_t)K * stride * sizeof(float);
        float *hth = malloc(sb), *hgr = malloc(sb), *hmo = malloc(sb), *hb = malloc(sb);
        for (int64_t i = 0; i < K * stride; ++i) { hth[i] = frand(&seed); hgr[i] = frand(&seed); hmo[i] = 0.1f * frand(&seed); }
        float *sth, *sgr, *smo; static ursa_step_ctl hc[2]; ursa_step_ctl* dc;
        CHECK(hipMalloc((void**)&sth, sb)); CHECK(hipMalloc((void**)&sgr, sb)); CHECK(hipMalloc((void**)&smo, sb));
        CHECK(hipMalloc((void**)&dc, sizeof hc));
        memset(hc, 0, sizeof hc);
        for (int k = 0; k < K; ++k) {
            hc[k].lr = 0.05f + 0.01f * k; hc[k].mu = k ? 0.0f : 0.9f; hc[k].c_wd = 8e-5f; hc[k].c_noise = 0.1f + 0.1f * k;
            hc[k].n_train = 50000.0f; hc[k].flags = URSA_STEP_NOISE | URSA_STEP_WD | URSA_STEP_ADVANCE; hc[k].seed = 500 + k; hc[k].step = 10 * k;
        }
        CHECK(hipMemcpy(sth, hth, sb, hipMemcpyHostToDevice)); CHECK(hipMemcpy(sgr, hgr, sb, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(smo, hmo, sb, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice));
        for (int it = 0; it < 3; ++it) {
            CHECK(ursa_sgmcmc_step_multi_f32(sth, sgr, smo, NULL, NULL, nc, K, stride, dc, st));
            for (int k = 0; k < K; ++k)
                oracle_sgmcmc_step_f32(hth + k * stride, hgr + k * stride, hc[k].mu != 0.0f ? hmo + k * stride : NULL, NULL, NULL, nc,
                                       hc[k].lr, hc[k].mu, hc[k].c_wd, hc[k].c_noise, hc[k].n_train, hc[k].seed, hc[k].step + it,
                                       URSA_STEP_NOISE | URSA_STEP_WD);
        }
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(hb, sth, sb, hipMemcpyDeviceToHost));
        if (memcmp(hb, hth, sb)) { printf("FAIL multi-chain theta differs from the oracle\n"); return 1; }
        CHECK(hipMemcpy(hb, smo, sb, hipMemcpyDeviceToHost));
        if (memcmp(hb, hmo, (size_t)stride * sizeof(float))) { printf("FAIL multi-chain momentum differs from the oracle\n"); return 1; }
        CHECK(hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost));
        int armed = 0;
        for (int k = 0; k < K; ++k) for (size_t w = 0; w < sizeof hc[k].tickets / 4; ++w) armed |= hc[k].tickets[w] != 0;
        if (hc[0].step != 3 || hc[1].step != 13 || armed) { printf("FAIL control blocks did not advance / tickets not re-armed\n"); return 1; }
    }
    /* the generator's in-register division / square root against the IEEE forms, all 2^32 inputs, on this device */
    {
        uint64_t *dm, hm[2] = {1, 1};
        CHECK(hipMalloc((void**)&dm, sizeof hm)); CHECK(hipMemset(dm, 0, sizeof hm));
        CHECK(ursa_selftest_rng_f32(dm, st)); CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(hm, dm, sizeof hm, hipMemcpyDeviceToHost));
        if (hm[0] || hm[1]) { printf("FAIL generator self-test: %llu radius / %llu logarithm mismatches\n", (unsigned long long)hm[0], (unsigned long long)hm[1]); return 1; }
    }
    /* K6: relu(bn(x)) forward + backward, [N, C, HW] = [24, 6, 36], against the oracle: the statistics and every
     * output / gate / parameter gradient are the same floats (both sides round exact double sums once) */
    {
        const int64_t N = 24, C = 6, HW = 36, tot = N * C * HW;
        float *hx = malloc(tot * 4), *hdy = malloc(tot * 4), *hy = malloc(tot * 4), *hdx = malloc(tot * 4);
        float *oy = malloc(tot * 4), *odx = malloc(tot * 4);
        float hg[6], hb[6], osm[6], osi[6], odg[6], odb[6], gsm[6], gsi[6], gdg[6], gdb[6];
        unsigned sd = 99u;
        for (int64_t i = 0; i < tot; ++i) { hx[i] = 1.5f * frand(&sd) + 0.3f; hdy[i] = frand(&sd); }
        for (int c = 0; c < C; ++c) { hg[c] = 1.0f + 0.5f * frand(&sd); hb[c] = 0.3f * frand(&sd); }
        oracle_bn_relu_fwd_f32(hx, oy, hg, hb, NULL, NULL, osm, osi, N, C, HW, 1e-5f, 0.0f, 1);
        oracle_bn_relu_bwd_f32(hx, hdy, odx, hg, hb, osm, osi, odg, odb, N, C, HW, 1);
        float *dx_, *ddy, *dy_, *ddx, *dg_, *db_, *dsm, *dsi, *ddg, *ddb, *dws, *dgate;   /* dgate: the forward's scale / shift (2 C floats) */
        CHECK(hipMalloc((void**)&dx_, tot * 4)); CHECK(hipMalloc((void**)&ddy, tot * 4));
        CHECK(hipMalloc((void**)&dy_, tot * 4)); CHECK(hipMalloc((void**)&ddx, tot * 4));
        CHECK(hipMalloc((void**)&dg_, 24)); CHECK(hipMalloc((void**)&db_, 24)); CHECK(hipMalloc((void**)&dsm, 24));
        CHECK(hipMalloc((void**)&dsi, 24)); CHECK(hipMalloc((void**)&ddg, 24)); CHECK(hipMalloc((void**)&ddb, 24)); CHECK(hipMalloc((void**)&dgate, 48));
        CHECK(hipMalloc((void**)&dws, URSA_BN_WS_FLOATS(C) * 4));
        CHECK(hipMemcpy(dx_, hx, tot * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(ddy, hdy, tot * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dg_, hg, 24, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db_, hb, 24, hipMemcpyHostToDevice));
        CHECK(ursa_bn_relu_fwd_f32(dx_, NULL, NULL, dy_, dg_, db_, NULL, NULL, dsm, dsi, dgate, dws, N, C, HW, 1e-5f, 0.0f, URSA_BN_RELU, st));
        CHECK(ursa_bn_relu_bwd_f32(dx_, ddy, NULL, ddx, dg_, db_, dsm, dsi, dgate, ddg, ddb, dws, N, C, HW, URSA_BN_RELU, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(hy, dy_, tot * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hdx, ddx, tot * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(gsm, dsm, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gsi, dsi, 24, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(gdg, ddg, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gdb, ddb, 24, hipMemcpyDeviceToHost));
        if (memcmp(gsm, osm, 24) || memcmp(gsi, osi, 24)) { printf("FAIL K6 batch statistics differ from the oracle\n"); return 1; }
        if (memcmp(hy, oy, tot * 4)) { printf("FAIL K6 forward differs from the oracle\n"); return 1; }
        if (memcmp(gdg, odg, 24) || memcmp(gdb, odb, 24)) { printf("FAIL K6 dgamma / dbeta differ from the oracle\n"); return 1; }
        if (memcmp(hdx, odx, tot * 4)) { printf("FAIL K6 dx differs from the oracle\n"); return 1; }
        /* ABI 4: the gated backward (the parity instrument) against its oracle twin: three listed gates flipped */
        {
            int32_t hidx[4] = {5, 40, 700, 0x7fffffff};                    /* ascending; INT32_MAX = padding */
            uint8_t hop[4];
            for (int k = 0; k < 3; ++k) hop[k] = !(oy[hidx[k]] > 0.0f);    /* the opposite of the gate the forward took */
            hop[3] = 1;
            oracle_bn_relu_bwd_gated_f32(hx, hdy, odx, hg, hb, osm, osi, odg, odb, N, C, HW, 1, hidx, hop, 4);
            int32_t* didx; uint8_t* dop;
            CHECK(hipMalloc((void**)&didx, sizeof hidx)); CHECK(hipMalloc((void**)&dop, sizeof hop));
            CHECK(hipMemcpy(didx, hidx, sizeof hidx, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dop, hop, sizeof hop, hipMemcpyHostToDevice));
            CHECK(ursa_bn_relu_bwd_gated_f32(dx_, ddy, NULL, ddx, dg_, db_, dsm, dsi, NULL, ddg, ddb, dws, N, C, HW, URSA_BN_RELU, didx, dop, 4, st));
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(hdx, ddx, tot * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(gdg, ddg, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gdb, ddb, 24, hipMemcpyDeviceToHost));
            if (memcmp(gdg, odg, 24) || memcmp(gdb, odb, 24) || memcmp(hdx, odx, tot * 4)) { printf("FAIL K6 gated backward differs from the oracle\n"); return 1; }
        }
        if (ursa_bn_relu_fwd_f32(dx_, NULL, NULL, dy_, dg_, db_, NULL, NULL, dsm, dsi, NULL, dws, 1, C, 1, 1e-5f, 0.0f, 0, st) != URSA_EVALUE) { printf("FAIL K6 evalue\n"); return 1; }
        if (ursa_bn_relu_fwd_f32(dx_, NULL, NULL, dy_, dg_, db_, NULL, NULL, dsm, dsi, NULL, dws, N, C, HW, 1e-5f, 0.0f, 0x10u, st) != URSA_EFLAGS) { printf("FAIL K6 eflags\n"); return 1; }
    }
    /* ABI 6: K8 (forward / input gradient) and K7 (weight gradient, both launches and the deferred pair) of one 16-channel 3x3
     * layer on small integers - every product and partial sum exact in fp32, so the device must equal the oracle bit for bit */
    {
        const int64_t N = 3, C = 16, H = 32, tot = N * C * H * H, wn = C * C * 9;
        float *hx = malloc(tot * 4), *hdy = malloc(tot * 4), *hw = malloc(wn * 4), *want = malloc(tot * 4), *got = malloc(tot * 4);
        float *wantw = malloc(wn * 4), *gotw = malloc(wn * 4);
        unsigned s2 = 99u;
        for (int64_t i = 0; i < tot; ++i) { hx[i] = floorf(3.99f * frand(&s2)); hdy[i] = floorf(2.99f * frand(&s2)); }
        for (int64_t i = 0; i < wn; ++i) hw[i] = floorf(2.99f * frand(&s2));
        float *dx_, *ddy, *dw_, *dy_, *ddw, *dws;
        const int64_t wsf = ursa_conv_wgrad_ws_floats(N, C, C, H, H, 3, 1);
        if (wsf <= 0 || !ursa_conv3x3_supported(N, C, C, H, H, 0)) { printf("FAIL K7 / K8 do not cover the 16-channel layer\n"); return 1; }
        CHECK(hipMalloc((void**)&dx_, tot * 4)); CHECK(hipMalloc((void**)&ddy, tot * 4)); CHECK(hipMalloc((void**)&dy_, tot * 4));
        CHECK(hipMalloc((void**)&dw_, wn * 4)); CHECK(hipMalloc((void**)&ddw, wn * 4)); CHECK(hipMalloc((void**)&dws, wsf * 4));
        CHECK(hipMemcpy(dx_, hx, tot * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(ddy, hdy, tot * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dw_, hw, wn * 4, hipMemcpyHostToDevice));
        CHECK(ursa_conv3x3_f32(dx_, dw_, dy_, N, C, C, H, H, 0, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(got, dy_, tot * 4, hipMemcpyDeviceToHost));
        oracle_conv3x3_f32(hx, hw, want, N, C, C, H, H, 0, 1);
        if (memcmp(got, want, tot * 4)) { printf("FAIL K8 forward differs from the oracle\n"); return 1; }
        CHECK(ursa_conv3x3_f32(ddy, dw_, dy_, N, C, C, H, H, URSA_CONV_FLIP, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(got, dy_, tot * 4, hipMemcpyDeviceToHost));
        oracle_conv3x3_f32(hdy, hw, want, N, C, C, H, H, 1, 1);
        if (memcmp(got, want, tot * 4)) { printf("FAIL K8 input gradient differs from the oracle\n"); return 1; }
        oracle_conv_wgrad_f32(hx, hdy, wantw, N, C, C, H, H, 3, 1);
        CHECK(ursa_conv_wgrad_f32(dx_, ddy, ddw, dws, wsf, N, C, C, H, H, 3, 1, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(gotw, ddw, wn * 4, hipMemcpyDeviceToHost));
        if (memcmp(gotw, wantw, wn * 4)) { printf("FAIL K7 differs from the oracle\n"); return 1; }
        CHECK(hipMemset(ddw, 0xff, wn * 4));
        CHECK(ursa_conv_wgrad_partial_f32(dx_, ddy, dws, wsf, N, C, C, H, H, 3, 1, st));
        const ursa_conv_pending pend = {dws, ddw, N, C, C, H, H, 3, 1};
        CHECK(ursa_conv_wgrad_reduce_f32(&pend, 1, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(gotw, ddw, wn * 4, hipMemcpyDeviceToHost));
        if (memcmp(gotw, wantw, wn * 4)) { printf("FAIL K7 (deferred second launch) differs from the oracle\n"); return 1; }
        if (ursa_conv3x3_f32(dx_, dw_, dy_, N, 5, C, H, H, 0, st) != URSA_EVALUE) { printf("FAIL K8 evalue\n"); return 1; }
        if (ursa_conv_wgrad_f32(dx_, ddy, ddw, dws, 16, N, C, C, H, H, 3, 1, st) != URSA_ESIZE) { printf("FAIL K7 esize\n"); return 1; }
    }
    /* ABI 7: K10 - one fused unit forward (BatchNorm + ReLU while the tile is staged, convolution, `+= residual`, statistics of the
     * result) and its backward (paired input-gradient / weight-gradient launch, then K6's dx launch) against the oracle's
     * restatement on integer-valued data: x = +-1 balanced per channel -> mean 0, variance 1, eps = 0 -> invstd 1, so every
     * product and sum is an integer and the device must equal the oracle bit for bit, sums included. Then K11's classifier launch. */
    {
        const int64_t N = 4, C = 16, H = 32, HW = H * H, tot = N * C * HW, wn = C * C * 9;
        float *hx = malloc(tot * 4), *hadd = malloc(tot * 4), *hdy = malloc(tot * 4), *hw = malloc(wn * 4), *hg = malloc(C * 4), *hb = malloc(C * 4);
        float *oy = malloc(tot * 4), *oh = malloc(tot * 4), *og = malloc(tot * 4), *odx = malloc(tot * 4), *got = malloc(tot * 4);
        float osave[64], gsave[64], odg[16], odb[16], gdg[16], gdb[16];
        double osums[32], obs[32];
        unsigned s3 = 7u;
        for (int64_t c = 0; c < C; ++c)                       /* a balanced +-1 pattern per channel: position parity xor a pseudo-random per-row flip */
            for (int64_t n = 0; n < N; ++n)
                for (int64_t j = 0; j < HW; j += 2) {
                    const float v = frand(&s3) < 0.5f ? 1.0f : -1.0f;
                    hx[(n * C + c) * HW + j] = v, hx[(n * C + c) * HW + j + 1] = -v;
                }
        for (int64_t i = 0; i < tot; ++i) { hadd[i] = floorf(4.99f * frand(&s3)) - 2.0f; hdy[i] = floorf(2.99f * frand(&s3)) - 1.0f; }
        for (int64_t i = 0; i < wn; ++i) hw[i] = floorf(2.99f * frand(&s3)) - 1.0f;
        for (int64_t c = 0; c < C; ++c) { hg[c] = 1.0f + floorf(2.99f * frand(&s3)); hb[c] = floorf(2.99f * frand(&s3)) - 1.0f; }
        oracle_preact_fwd_f32(hx, hw, hadd, oy, oh, hg, hb, NULL, NULL, osave, osums, N, C, C, H, H, 1, 0.0f, 0.0f, 1);
        oracle_preact_bwd_f32(hdy, hw, hx, osave, og, obs, N, C, C, H, H, 1);
        oracle_bn_bwd_dx_f32(hx, og, NULL, odx, hg, osave, obs, odg, odb, N, C, HW);
        int64_t geo[4], geob[4];
        if (ursa_preact_geometry(N, C, C, H, H, URSA_PREACT_BN | URSA_PREACT_STATS | URSA_PREACT_ADD, geo) != URSA_OK ||
            ursa_preact_geometry(N, C, C, H, H, URSA_CONV_FLIP | URSA_PREACT_BNBWD, geob) != URSA_OK) { printf("FAIL K10 does not cover the 16-channel unit\n"); return 1; }
        double hin[32];                                        /* the producer's partial sums of x: one line per channel, (sum, sum of squares) = (0, n) */
        for (int64_t c = 0; c < C; ++c) { hin[2 * c] = 0.0; hin[2 * c + 1] = (double)(N * HW); }
        float *dx_, *dadd, *ddy, *dw_, *dy_, *dgm, *dbt, *dsave, *dg_, *ddx, *ddgb, *dws;
        double *din, *dpart, *dpb;
        void* dscr;
        const int64_t wsf = ursa_conv_wgrad_ws_floats(N, C, C, H, H, 3, 1);
        CHECK(hipMalloc((void**)&dx_, tot * 4)); CHECK(hipMalloc((void**)&dadd, tot * 4)); CHECK(hipMalloc((void**)&ddy, tot * 4));
        CHECK(hipMalloc((void**)&dy_, tot * 4)); CHECK(hipMalloc((void**)&dg_, tot * 4)); CHECK(hipMalloc((void**)&ddx, tot * 4));
        CHECK(hipMalloc((void**)&dw_, wn * 4)); CHECK(hipMalloc((void**)&dgm, C * 4)); CHECK(hipMalloc((void**)&dbt, C * 4));
        CHECK(hipMalloc((void**)&dsave, 4 * C * 4)); CHECK(hipMalloc((void**)&ddgb, 2 * C * 4)); CHECK(hipMalloc((void**)&dws, wsf * 4));
        CHECK(hipMalloc((void**)&din, sizeof hin)); CHECK(hipMalloc((void**)&dpart, C * geo[0] * 16)); CHECK(hipMalloc((void**)&dpb, C * geob[0] * 16));
        CHECK(hipMalloc(&dscr, geo[1])); CHECK(hipMemset(dscr, 0, geo[1]));
        CHECK(hipMemcpy(dx_, hx, tot * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dadd, hadd, tot * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(ddy, hdy, tot * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dw_, hw, wn * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dgm, hg, C * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dbt, hb, C * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice));
        CHECK(ursa_preact_conv3x3_f32(dx_, dw_, dy_, din, 1, dgm, dbt, NULL, NULL, dsave, 0.0f, 0.0f, dadd, NULL, dpart, dscr, geo[1], N, C, C, H, H,
                                      URSA_PREACT_BN | URSA_PREACT_STATS | URSA_PREACT_ADD, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(got, dy_, tot * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gsave, dsave, 4 * C * 4, hipMemcpyDeviceToHost));
        if (memcmp(gsave, osave, 4 * C * 4)) { printf("FAIL K10 saved BatchNorm scalars differ from the oracle\n"); return 1; }
        if (memcmp(got, oy, tot * 4)) { printf("FAIL K10 forward unit differs from the oracle\n"); return 1; }
        {
            double* hp = malloc(C * geo[0] * 16);
            CHECK(hipMemcpy(hp, dpart, C * geo[0] * 16, hipMemcpyDeviceToHost));
            for (int64_t c = 0; c < C; ++c) {
                double a = 0.0, b = 0.0;
                for (int64_t l = 0; l < geo[0]; ++l) { a += hp[(c * geo[0] + l) * 2]; b += hp[(c * geo[0] + l) * 2 + 1]; }
                if (a != osums[2 * c] || b != osums[2 * c + 1]) { printf("FAIL K10 statistics of the result differ from the oracle\n"); return 1; }
            }
            unsigned char* hs = malloc(geo[1]);
            CHECK(hipMemcpy(hs, dscr, geo[1], hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < geo[1]; ++i) if (hs[i]) { printf("FAIL K10 scratch not zero after the launch\n"); return 1; }
        }
        CHECK(ursa_preact_bwd_pair_f32(ddy, dw_, dg_, dx_, dsave, dpb, dws, wsf, N, C, C, H, H, 0, st));
        CHECK(ursa_bn_bwd_dx_f32(dx_, dg_, NULL, ddx, dgm, dsave, dpb, (int32_t)geob[0], ddgb, ddgb + C, N, C, HW, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(got, dg_, tot * 4, hipMemcpyDeviceToHost));
        if (memcmp(got, og, tot * 4)) { printf("FAIL K10 gated input gradient differs from the oracle\n"); return 1; }
        CHECK(hipMemcpy(got, ddx, tot * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(gdg, ddgb, C * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gdb, ddgb + C, C * 4, hipMemcpyDeviceToHost));
        if (memcmp(gdg, odg, C * 4) || memcmp(gdb, odb, C * 4) || memcmp(got, odx, tot * 4)) { printf("FAIL K10 dx / dgamma / dbeta differ from the oracle\n"); return 1; }
        {   /* the weight gradient the paired launch left in ws: x operand = relu(bn(x)) = the oracle's h */
            float *wantw = malloc(wn * 4), *gotw = malloc(wn * 4), *ddw;
            CHECK(hipMalloc((void**)&ddw, wn * 4));
            const ursa_conv_pending pw = {dws, ddw, N, C, C, H, H, 3, 1};
            CHECK(ursa_conv_wgrad_reduce_f32(&pw, 1, st));
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(gotw, ddw, wn * 4, hipMemcpyDeviceToHost));
            oracle_conv_wgrad_f32(oh, hdy, wantw, N, C, C, H, H, 3, 1);
            if (memcmp(gotw, wantw, wn * 4)) { printf("FAIL K10 weight gradient (paired launch) differs from the oracle\n"); return 1; }
        }
        if (ursa_preact_geometry(N, 5, C, H, H, URSA_PREACT_BN | URSA_PREACT_STATS, geo) != URSA_EVALUE) { printf("FAIL K10 evalue\n"); return 1; }
        if (ursa_preact_conv3x3_f32(dx_, dw_, dy_, din, 1, dgm, dbt, NULL, NULL, dsave, 0.0f, 0.0f, dadd, NULL, dpart, dscr, 64, N, C, C, H, H,
                                    URSA_PREACT_BN | URSA_PREACT_STATS | URSA_PREACT_ADD, st) != URSA_ESIZE) { printf("FAIL K10 esize\n"); return 1; }
        /* ABI 8: K13 - conv1x1(relu(bn(x))), 16 -> 64 channels at 32 x 32, on the same integer-valued x / gamma / beta: the
         * statistics entry's saved block, the forward and the weight gradient against the oracle's K6 and K12 restatements composed */
        {
            const int64_t Co = 64, w1n = Co * C, ytot = N * Co * HW;
            float *hw1 = malloc(w1n * 4), *hdy1 = malloc(ytot * 4), *oh1 = malloc(tot * 4), *oy1 = malloc(ytot * 4), *gy1 = malloc(ytot * 4);
            float *odw1 = malloc(w1n * 4), *gdw1 = malloc(w1n * 4), om[16], oi[16], gsv[64];
            for (int64_t i = 0; i < w1n; ++i) hw1[i] = floorf(4.99f * frand(&s3)) - 2.0f;
            for (int64_t i = 0; i < ytot; ++i) hdy1[i] = floorf(2.99f * frand(&s3)) - 1.0f;
            oracle_bn_relu_fwd_f32(hx, oh1, hg, hb, NULL, NULL, om, oi, N, C, HW, 0.0f, 0.0f, 1);
            oracle_conv1x1_f32(oh1, hw1, oy1, N, C, Co, HW, 0);
            oracle_conv_wgrad_f32(oh1, hdy1, odw1, N, C, Co, H, H, 1, 1);
            float *dw1, *dy1, *ddy1, *dsv, *dbws, *dws1, *ddw1;
            const int64_t wsf1 = ursa_conv_wgrad_ws_floats(N, C, Co, H, H, 1, 1), bws = URSA_BN_WS_FLOATS(C);
            if (!ursa_preact_conv1x1_supported(N, C, Co, H, H) || wsf1 <= 0) { printf("FAIL K13 does not cover 16 -> 64 at 32 x 32\n"); return 1; }
            CHECK(hipMalloc((void**)&dw1, w1n * 4)); CHECK(hipMalloc((void**)&dy1, ytot * 4)); CHECK(hipMalloc((void**)&ddy1, ytot * 4));
            CHECK(hipMalloc((void**)&dsv, 4 * C * 4)); CHECK(hipMalloc((void**)&dbws, bws * 4)); CHECK(hipMalloc((void**)&dws1, wsf1 * 4));
            CHECK(hipMalloc((void**)&ddw1, w1n * 4));
            CHECK(hipMemcpy(dw1, hw1, w1n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(ddy1, hdy1, ytot * 4, hipMemcpyHostToDevice));
            CHECK(ursa_bn_stats_f32(dx_, NULL, NULL, dgm, dbt, NULL, NULL, dsv, dbws, N, C, HW, 0.0f, 0.0f, st));
            CHECK(ursa_preact_conv1x1_f32(dx_, dsv, dw1, dy1, N, C, Co, H, H, st));
            CHECK(ursa_preact_wgrad1x1_partial_f32(dx_, dsv, ddy1, dws1, wsf1, N, C, Co, H, H, st));
            ursa_conv_pending pend = {dws1, ddw1, N, C, Co, H, H, 1, 1};
            CHECK(ursa_conv_wgrad_reduce_f32(&pend, 1, st));
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(gsv, dsv, 4 * C * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gy1, dy1, ytot * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(gdw1, ddw1, w1n * 4, hipMemcpyDeviceToHost));
            for (int64_t c = 0; c < C; ++c)
                if (gsv[c] != om[c] || gsv[C + c] != oi[c] || gsv[2 * C + c] != hg[c] || gsv[3 * C + c] != hb[c]) { printf("FAIL K13 saved block differs from the oracle\n"); return 1; }
            if (memcmp(gy1, oy1, ytot * 4)) { printf("FAIL K13 forward differs from the oracle\n"); return 1; }
            if (memcmp(gdw1, odw1, w1n * 4)) { printf("FAIL K13 weight gradient differs from the oracle\n"); return 1; }
            if (ursa_preact_conv1x1_f32(dx_, NULL, dw1, dy1, N, C, Co, H, H, st) != URSA_ENULL) { printf("FAIL K13 enull\n"); return 1; }
            if (ursa_preact_conv1x1_f32(dx_, dsv, dw1, dy1, N, C, 48, H, H, st) != URSA_EVALUE) { printf("FAIL K13 evalue\n"); return 1; }
        }
        /* K11: classifier + mean cross entropy + their gradients in one launch, against the oracle (double sums) to rounding */
        {
            const int64_t Nn = 32, Cc = 64, Kk = 10;
            float *hp = malloc(Nn * Cc * 4), *hW = malloc(Kk * Cc * 4), hbias[10], oloss, gloss, *odW = malloc(Kk * Cc * 4), odbias[10], *odp = malloc(Nn * Cc * 4);
            float *gdW = malloc(Kk * Cc * 4), *gdp = malloc(Nn * Cc * 4), *ologits = malloc(Nn * Kk * 4);
            int64_t ht[32];
            for (int64_t i = 0; i < Nn * Cc; ++i) hp[i] = 2.0f * frand(&s3);
            for (int64_t i = 0; i < Kk * Cc; ++i) hW[i] = 0.6f * frand(&s3) - 0.3f;
            for (int64_t k = 0; k < Kk; ++k) hbias[k] = 0.2f * frand(&s3) - 0.1f;
            for (int64_t n = 0; n < Nn; ++n) ht[n] = (int64_t)(9.99f * frand(&s3));
            ht[3] = -100;
            oracle_fc_ce_f32(hp, hW, hbias, ht, &oloss, ologits, odW, odbias, odp, Nn, Cc, Kk, -100);
            float *dp_, *dW_, *dbias, *dloss, *ddW, *ddb, *ddp;
            int64_t* dt;
            CHECK(hipMalloc((void**)&dp_, Nn * Cc * 4)); CHECK(hipMalloc((void**)&dW_, Kk * Cc * 4)); CHECK(hipMalloc((void**)&dbias, Kk * 4));
            CHECK(hipMalloc((void**)&dloss, 4)); CHECK(hipMalloc((void**)&ddW, Kk * Cc * 4)); CHECK(hipMalloc((void**)&ddb, Kk * 4));
            CHECK(hipMalloc((void**)&ddp, Nn * Cc * 4)); CHECK(hipMalloc((void**)&dt, Nn * 8));
            CHECK(hipMemcpy(dp_, hp, Nn * Cc * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dW_, hW, Kk * Cc * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dbias, hbias, Kk * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dt, ht, Nn * 8, hipMemcpyHostToDevice));
            CHECK(ursa_fc_ce_f32(dp_, dW_, dbias, dt, dloss, NULL, ddW, ddb, ddp, Nn, Cc, Kk, -100, st));
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(&gloss, dloss, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gdW, ddW, Kk * Cc * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(gdp, ddp, Nn * Cc * 4, hipMemcpyDeviceToHost));
            float wmax = 0.f, pmax = 0.f, werr = 0.f, perr = 0.f;
            for (int64_t i = 0; i < Kk * Cc; ++i) { wmax = fmaxf(wmax, fabsf(odW[i])); werr = fmaxf(werr, fabsf(gdW[i] - odW[i])); }
            for (int64_t i = 0; i < Nn * Cc; ++i) { pmax = fmaxf(pmax, fabsf(odp[i])); perr = fmaxf(perr, fabsf(gdp[i] - odp[i])); }
            if (fabsf(gloss - oloss) > 2e-6f * fabsf(oloss) || werr > 5e-6f * wmax || perr > 5e-6f * pmax) {
                printf("FAIL K11 classifier launch: loss %g vs %g, dW err %g of %g, dp err %g of %g\n", gloss, oloss, werr, wmax, perr, pmax); return 1; }
            if (ursa_fc_ce_f32(dp_, dW_, dbias, dt, dloss, NULL, ddW, ddb, ddp, Nn, Cc, 100, -100, st) != URSA_EVALUE) { printf("FAIL K11 evalue\n"); return 1; }
        }
    }
    /* argument errors come back as codes, not crashes */
    if (ursa_sgmcmc_step_f32(NULL, NULL, NULL, NULL, NULL, 8, 0, 0, 0, 0, 1, 0, 0, 0, st) != URSA_ENULL) { printf("FAIL enull\n"); return 1; }
    if (ursa_bma_accumulate_f32(dz, dp, de, NULL, NULL, S, B, 5000, omg, goc, 0, st) != URSA_EVALUE) { printf("FAIL evalue\n"); return 1; }
    if (ursa_sgmcmc_step_multi_f32(dth, dgr, dmo, NULL, NULL, 64, 2, 62, NULL, st) != URSA_ESIZE) { printf("FAIL esize (stride < n)\n"); return 1; }
    printf("C-ABI host OK: K1 bit-equal to the oracle over 3 steps (n=%lld), 2 chains in one self-advancing launch bit-equal, "
           "generator self-test clean, K5 max relative error %.2e, "
           "K6 forward + backward + gated backward bit-equal, K7 / K8 bit-equal on integer inputs, "
           "K10 fused unit (forward, paired backward, dx) and K13 (statistics, conv1x1 of relu(bn(x)), weight gradient) bit-equal on integer inputs, "
           "K11 classifier launch within rounding\n", (long long)n, worst);
    return 0;
}
