/* ursa_oracle.c — CPU restatement of URSABench's SG-MCMC / BMA hot path (plain C, scalar loops).
 *
 * TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product (ursabench_amd/) never does.
 *
 * Pinning: every function below is checked in tests/test_oracle_golden.py against golden
 * vectors captured from the imported reference (tools/gen_golden.py -> tests/golden/ npz files).
 * K1/K2/K3 are bit-exact against the reference's torch-CPU path on injected noise; K5 is
 * within 1e-5 relative (ATen's vectorised softmax/sum order is not reproducible in scalar C).
 * HMC (oracle_leapfrog_f32) follows hamiltorch's published leapfrog: PARITY UNPINNED
 * (hamiltorch is an un-vendored, un-pinned dependency that is absent from /root/reference).
 * K6 (oracle_bn_relu_*: the networks' relu(bn(x)), models/preresnet.py:40-41,76-85,146) restates
 * torch's CPU BatchNorm + ReLU, which is what the reference executes there; pinned in
 * tests/test_oracle_golden.py against torch's CPU kernels themselves on seeded inputs (the batch
 * mean bit for bit, the outputs bit for bit in every channel whose invstd agrees).
 *
 * Build: make -C oracle   (gcc -O2 -mfma -ffp-contract=off: the two fmaf() per update are
 * the only fused operations, exactly where ATen's add(alpha=) fuses — SURVEY.md A.1).
 *
 * Citations are relative to /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

#define STEP_NOISE     0x1u
#define STEP_FIRST     0x2u
#define STEP_ZERO_GRAD 0x4u
#define STEP_WD        0x8u
#define STEP_SGD       0x10u
#define BMA_SMOOTHED   0x1u
#define LEAP_KICK      0x1u
#define LEAP_DRIFT     0x2u

/* ---------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
 * SC'11; Random123 1.x). Not in the reference: the reference draws torch.randn_like per
 * tensor from torch's global generator (optim_sghmc.py:64), a stream no flat kernel can
 * reproduce (SURVEY.md §7 hard part 1). Pinned by Random123's known-answer vectors.
 */
static inline void philox_round(uint32_t c[4], const uint32_t k[2])
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; ++r) {
        if (r) { k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u; }
        philox_round(c, k);
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

/* Deterministic fp32 ln(u) for u in (0, 1]: only +,-,*,/ and fmaf, so the HIP kernel and
 * this file agree bit for bit. Argument reduction and coefficients as in FreeBSD msun
 * e_logf.c (public-domain fdlibm lineage). */
static inline float det_logf(float u)
{
    union { float f; uint32_t i; } b; b.f = u;
    int k = (int)(b.i >> 23) - 127;
    uint32_t m = b.i & 0x007fffffu;
    if (m > 0x3504f3u) { k += 1; b.i = m | 0x3f000000u; }   /* mantissa > sqrt(2): halve */
    else               {          b.i = m | 0x3f800000u; }
    const float f = b.f - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * fmaf(w, 0.24279078841f, 0.40000972152f);
    const float t2 = z * fmaf(w, 0.28498786688f, 0.66666662693f);
    const float R = t2 + t1;
    const float hfsq = 0.5f * f * f;
    const float dk = (float)k;
    /* ln2_hi = 6.9313812256e-01, ln2_lo = 9.0580006145e-06 */
    return dk * 6.9313812256e-01f - ((hfsq - fmaf(s, hfsq + R, dk * 9.0580006145e-06f)) - f);
}

/* Deterministic sin/cos of 2*pi*t for t = (j + 0.5) * 2^-23, j a 23-bit integer.
 * Octant reduction is exact; kernels are the Cephes single-precision minimax polynomials
 * on [0, pi/4]. */
static inline void det_sincos2pi(uint32_t j, float* sn, float* cs)
{
    const float t4 = ((float)j + 0.5f) * 4.76837158203125e-07f;   /* 4*t in (0,4), exact */
    const int q = (int)t4;                                          /* quadrant 0..3 */
    float r = t4 - (float)q;                                        /* in (0,1), exact */
    const int swap = r > 0.5f;
    if (swap) r = 1.0f - r;                                         /* exact */
    const float x = r * 1.57079637050628662109375f;                 /* in [0, pi/4] */
    const float x2 = x * x;
    float ps = fmaf(x2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(x2, ps, -1.6666654611e-1f);
    const float s = fmaf(x * x2, ps, x);
    float pc = fmaf(x2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(x2, pc, 4.166664568298827e-2f);
    const float c = fmaf(x2 * x2, pc, fmaf(x2, -0.5f, 1.0f));
    const float sq = swap ? c : s, cq = swap ? s : c;               /* sin, cos of quadrant angle */
    switch (q) {
    case 0:  *sn =  sq; *cs =  cq; break;
    case 1:  *sn =  cq; *cs = -sq; break;
    case 2:  *sn = -sq; *cs = -cq; break;
    default: *sn = -cq; *cs =  sq; break;
    }
}

/* Four standard normals for element block i4 (elements 4*i4 .. 4*i4+3) of call (seed, step). */
static inline void normal4(uint64_t seed, uint64_t step, uint64_t i4, float z[4])
{
    const uint32_t ctr[4] = {(uint32_t)i4, (uint32_t)(i4 >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t x[4];
    oracle_philox4x32_10(ctr, key, x);
    for (int h = 0; h < 2; ++h) {
        /* u1 = (x + 0.5) * 2^-32 rounded once: in (0, 1]  =>  radius <= 6.76 */
        const float u1 = fmaf((float)x[2 * h], 2.3283064365386963e-10f, 1.1641532182693481e-10f);
        const float rad = sqrtf(-2.0f * det_logf(u1));
        float sn, cs;
        det_sincos2pi(x[2 * h + 1] >> 9, &sn, &cs);
        z[2 * h] = rad * cs;
        z[2 * h + 1] = rad * sn;
    }
}

void oracle_philox_normal_f32(float* out, int64_t n, uint64_t seed, uint64_t step)
{
    for (int64_t i4 = 0; 4 * i4 < n; ++i4) {
        float z[4];
        normal4(seed, step, (uint64_t)i4, z);
        for (int j = 0; j < 4 && 4 * i4 + j < n; ++j) out[4 * i4 + j] = z[j];
    }
}

/* elements [start, start+count) of the same stream (start must be a multiple of 4): lets the tests check the
 * far end of a > 2^31-element arena without generating all of it on the CPU */
void oracle_philox_normal_range_f32(float* out, int64_t start, int64_t count, uint64_t seed, uint64_t step)
{
    for (int64_t k = 0; k < count; k += 4) {
        float z[4];
        normal4(seed, step, (uint64_t)((start + k) >> 2), z);
        for (int j = 0; j < 4 && k + j < count; ++j) out[k + j] = z[j];
    }
}

/* ---------------------------------------------------------------------------------------
 * K1: optimSGHMC.step over a flat vector.   URSABench/inference/optim_sghmc.py:43-67
 */
int oracle_sgmcmc_step_f32(float* theta, float* grad, float* mom, const float* eps, float* snapshot,
                           int64_t n, float lr, float mu, float c_wd, float c_noise, float n_train,
                           uint64_t seed, uint64_t step, uint32_t flags)
{
    const float neg_lr = -lr;
    float z[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
        const float th = theta[i];
        float g = grad[i];
        if (flags & STEP_WD) g = fmaf(c_wd, th, g);                  /* :48  d_p.add(p, alpha=wd/N) */
        if (flags & STEP_SGD) {
            /* torch.optim.SGD single-tensor update (the SWA/SWAG trajectory, swa.py:41-42):
             * buf = FIRST ? clone(g~) : buf.mul_(mu).add_(g~);  p.add_(buf, alpha=-lr) */
            float b = g;
            if (mu != 0.0f) { b = (flags & STEP_FIRST) ? g : mom[i] * mu + g; mom[i] = b; }
            const float ts = fmaf(neg_lr, b, th);
            theta[i] = ts;
            if (flags & STEP_ZERO_GRAD) grad[i] = 0.0f;
            if (snapshot) snapshot[i] = ts;
            continue;
        }
        float d;
        if (mu != 0.0f) {
            float v = (flags & STEP_FIRST) ? g : mom[i];             /* :52  clone(d_p) */
            v = v * mu;                                              /* :53/:56 buf.mul_(momentum) */
            v = fmaf(neg_lr, g, v);                                  /*         .add_(d_p, alpha=-lr) */
            d = v;                                                   /* :60 */
        } else {
            d = g * neg_lr;                                          /* :62 */
        }
        if (flags & STEP_NOISE) {
            float e;
            if (eps) e = eps[i];
            else { if ((i & 3) == 0) normal4(seed, step, (uint64_t)(i >> 2), z); e = z[i & 3]; }
            d = d + (e * c_noise) / n_train;                         /* :64 */
        }
        const float tn = th + d;                                     /* :65 p.add_(d_p) */
        theta[i] = tn;
        if (mu != 0.0f) mom[i] = d;                                  /* :67 */
        if (flags & STEP_ZERO_GRAD) grad[i] = 0.0f;                  /* sghmc.py:79 zero_grad */
        if (snapshot) snapshot[i] = tn;                              /* sghmc.py:99 snapshot */
    }
    return 0;
}

/* K2: SWA._collect_model.   URSABench/inference/swa.py:81-88 */
int oracle_swag_collect_f32(float* mean, float* sq, const float* w, int64_t n, float decay, float denom)
{
    for (int64_t i = 0; i < n; ++i) {
        const float wi = w[i];
        const float m = mean[i] * decay;          /* :83 weight_mean.mul_(n/(n+1)) */
        mean[i] = m + wi / denom;                 /* :84 .add_(w/(n+1)) */
        const float s = sq[i] * decay;            /* :87 */
        sq[i] = s + (wi * wi) / denom;            /* :88 w**2/(n+1) */
    }
    return 0;
}

/* K3: diagonal SWAG draw.   URSABench/inference/swa.py:106-108 + swag.py:84-86 */
int oracle_swag_draw_f32(float* theta_out, const float* mean, const float* sq, const float* eps,
                         int64_t n, float var_clamp, float scale, uint64_t seed, uint64_t draw)
{
    float z[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
        const float m = mean[i];
        float var = sq[i] - m * m;                /* swa.py:107 */
        var = var < var_clamp ? var_clamp : var;  /* torch.clamp(min) ; NaN propagates like ATen */
        float e;
        if (eps) e = eps[i];
        else { if ((i & 3) == 0) normal4(seed, draw, (uint64_t)(i >> 2), z); e = z[i & 3]; }
        theta_out[i] = e * (sqrtf(var) * scale) + m;   /* torch.normal(mean, std): normal_().mul_(std).add_(mean) */
    }
    return 0;
}

/* K5: tasks accumulators.   URSABench/tasks/prediction.py:57-63, ood_detection.py:59-65,
 * decision_making.py:124-129, util.py:126-144 */
int oracle_bma_accumulate_f32(const float* logits, float* proba_sum, float* ent_sum, float* risk_sum,
                              const float* cost, int32_t S, int64_t B, int32_t C,
                              float one_minus_gamma, float gamma_over_c, uint32_t flags)
{
    float ps[1024];
    if (C < 1 || C > 1024) return -5;
    for (int64_t b = 0; b < B; ++b) {
        for (int32_t s = 0; s < S; ++s) {
            const float* z = logits + ((int64_t)s * B + b) * C;
            float mx = z[0];
            for (int c = 1; c < C; ++c) mx = z[c] > mx ? z[c] : mx;
            float sum = 0.0f;
            for (int c = 0; c < C; ++c) sum += expf(z[c] - mx);
            const float lse = logf(sum);
            float ent = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float p = expf((z[c] - mx) - lse);                       /* log_softmax().exp_() */
                const float q = p * one_minus_gamma + gamma_over_c;            /* util.py:134 */
                ps[c] = q;
                ent += q > 0.0f ? q * logf(q) : 0.0f;                          /* util.py:144; 0 ln 0 := 0 (gamma = 0 only) */
                proba_sum[b * C + c] += (flags & BMA_SMOOTHED) ? q : p;
            }
            if (ent_sum) ent_sum[b] += -ent;
            if (risk_sum) {                                                    /* decision_making.py:129 */
                for (int j = 0; j < C; ++j) {
                    float acc = 0.0f;
                    for (int c = 0; c < C; ++c) acc += ps[c] * cost[c * C + j];
                    risk_sum[b * C + j] += acc;
                }
            }
        }
    }
    return 0;
}

/* K4: one leapfrog sub-step (hamiltorch semantics; parity unpinned). */
int oracle_leapfrog_f32(float* theta, float* mom, const float* grad, int64_t n, float kick_coef,
                        float step_size, float inv_mass, uint32_t flags, double* kinetic_out)
{
    const float drift = step_size * inv_mass;
    double ke = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        float p = mom[i];
        if (flags & LEAP_KICK) { p = p + kick_coef * grad[i]; mom[i] = p; }
        if (flags & LEAP_DRIFT) theta[i] = theta[i] + drift * p;
        ke += (double)p * (double)p;
    }
    if (kinetic_out) *kinetic_out += 0.5 * (double)inv_mass * ke;
    return 0;
}

int oracle_sumsq_f32(const float* x, int64_t n, double* out)
{
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) acc += (double)x[i] * (double)x[i];
    *out += acc;
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * K6  relu(bn(x)), training mode      URSABench/models/preresnet.py:40-41,45-46,76-85,146;
 *     wideresnet.py:47,49,117 -> torch.nn.functional.batch_norm + relu on the CPU path.
 * torch 2.10's CPU kernel (probed, see include/ursa_hip.h K6): statistics accumulated in double,
 *     alpha = invstd * gamma ; beta' = fma(-mean, alpha, beta) ; y = fma(x, alpha, beta')
 * and its backward (native_batch_norm_backward, training):
 *     sum = sum g ; dotp = sum g (x - mean) ; dbeta = sum ; dgamma = dotp * invstd
 *     dx = (((g - sum/n) - (x - mean) * (dotp * invstd^2 / n)) * invstd) * gamma
 * with g = dy where the ReLU output is > 0 (threshold_backward). x, y, dy, dx: [N, C, HW].
 */
int oracle_bn_relu_fwd_f32(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* save_mean, float* save_invstd, int64_t N, int64_t C, int64_t HW,
                           float eps, float momentum, int relu)
{
    const double n = (double)N * (double)HW;
    for (int64_t c = 0; c < C; ++c) {
        double s1 = 0.0, s2 = 0.0;
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < HW; ++j) {
                const double d = (double)x[(i * C + c) * HW + j];
                s1 += d;
                s2 += d * d;
            }
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float meanf = (float)mean;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float alpha = invstd * gamma[c];
        const float shift = fmaf(-meanf, alpha, beta[c]);
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < HW; ++j) {
                const int64_t o = (i * C + c) * HW + j;
                const float t = fmaf(x[o], alpha, shift);
                y[o] = (relu && t < 0.0f) ? 0.0f : t;
            }
        save_mean[c] = meanf;
        save_invstd[c] = invstd;
        if (running_mean) {
            running_mean[c] = momentum * meanf + (1.0f - momentum) * running_mean[c];
            running_var[c] = momentum * (float)(var * (n / (n - 1.0))) + (1.0f - momentum) * running_var[c];
        }
    }
    return 0;
}

/* gate of element o: listed (ascending gate_idx, INT32_MAX = padding) -> gate_open, else the forward's own
 * (twin of ursa_bn_relu_bwd_gated_f32, include/ursa_hip.h: the parity instrument) */
static int oracle_gate(int computed, int64_t o, const int32_t* gate_idx, const uint8_t* gate_open, int64_t n_gates)
{
    int64_t lo = 0, hi = n_gates;
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if ((int64_t)gate_idx[mid] < o) lo = mid + 1; else hi = mid;
    }
    return (lo < n_gates && (int64_t)gate_idx[lo] == o) ? gate_open[lo] != 0 : computed;
}

int oracle_bn_relu_bwd_gated_f32(const float* x, const float* dy, float* dx, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, int64_t N,
                                 int64_t C, int64_t HW, int relu, const int32_t* gate_idx, const uint8_t* gate_open,
                                 int64_t n_gates)
{
    const double n = (double)N * (double)HW;
    for (int64_t c = 0; c < C; ++c) {
        const float mean = save_mean[c], invstd = save_invstd[c], w = gamma[c];
        const float alpha = invstd * w;
        const float shift = fmaf(-mean, alpha, beta[c]);
        double sum = 0.0, dotp = 0.0;
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < HW; ++j) {
                const int64_t o = (i * C + c) * HW + j;
                const int open = oracle_gate(fmaf(x[o], alpha, shift) > 0.0f, o, gate_idx, gate_open, n_gates);
                const float g = (relu && !open) ? 0.0f : dy[o];
                sum += (double)g;
                dotp += (double)g * ((double)x[o] - (double)mean);
            }
        const float gm = (float)(sum / n);
        const float k = (float)(dotp * (double)invstd * (double)invstd / n);
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < HW; ++j) {
                const int64_t o = (i * C + c) * HW + j;
                const int open = oracle_gate(fmaf(x[o], alpha, shift) > 0.0f, o, gate_idx, gate_open, n_gates);
                const float g = (relu && !open) ? 0.0f : dy[o];
                dx[o] = (((g - gm) - (x[o] - mean) * k) * invstd) * w;
            }
        dbeta[c] = (float)sum;
        dgamma[c] = (float)(dotp * (double)invstd);
    }
    return 0;
}

int oracle_bn_relu_bwd_f32(const float* x, const float* dy, float* dx, const float* gamma, const float* beta,
                           const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, int64_t N,
                           int64_t C, int64_t HW, int relu)
{
    const double n = (double)N * (double)HW;
    for (int64_t c = 0; c < C; ++c) {
        const float mean = save_mean[c], invstd = save_invstd[c], w = gamma[c];
        const float alpha = invstd * w;
        const float shift = fmaf(-mean, alpha, beta[c]);
        double sum = 0.0, dotp = 0.0;
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < HW; ++j) {
                const int64_t o = (i * C + c) * HW + j;
                const float g = (relu && !(fmaf(x[o], alpha, shift) > 0.0f)) ? 0.0f : dy[o];
                sum += (double)g;
                dotp += (double)g * ((double)x[o] - (double)mean);
            }
        const float gm = (float)(sum / n);
        const float k = (float)(dotp * (double)invstd * (double)invstd / n);
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < HW; ++j) {
                const int64_t o = (i * C + c) * HW + j;
                const float g = (relu && !(fmaf(x[o], alpha, shift) > 0.0f)) ? 0.0f : dy[o];
                dx[o] = (((g - gm) - (x[o] - mean) * k) * invstd) * w;
            }
        dbeta[c] = (float)sum;
        dgamma[c] = (float)(dotp * (double)invstd);
    }
    return 0;
}

/* evaluation mode: torch's CPU kernel computes invstd in float here (probed: 0 of 262,144 outputs differ) */
int oracle_bn_relu_eval_f32(const float* x, float* y, const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, int64_t N, int64_t C, int64_t HW, float eps, int relu)
{
    for (int64_t c = 0; c < C; ++c) {
        const float invstd = 1.0f / sqrtf(running_var[c] + eps);
        const float alpha = invstd * gamma[c];
        const float shift = fmaf(-running_mean[c], alpha, beta[c]);
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < HW; ++j) {
                const int64_t o = (i * C + c) * HW + j;
                const float t = fmaf(x[o], alpha, shift);
                y[o] = (relu && t < 0.0f) ? 0.0f : t;
            }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * K7  weight gradient of a k x k convolution, pad k/2    `loss.backward()` URSABench/inference/sghmc.py:80 -> ATen
 *     convolution_backward on the CPU path (oneDNN) for the nn.Conv2d(.., bias=False) layers of
 *     URSABench/models/preresnet.py:25-27,62-64,100,130-136.
 *
 *     dw[co][ci][kh][kw] = sum_{n, oh, ow} dy[n][co][oh][ow] * x[n][ci][oh*stride + kh - k/2][ow*stride + kw - k/2]
 *
 * oneDNN's summation order is not part of any contract (it changes with thread count and ISA), so the restatement takes
 * the sum in double and rounds once: the value every fp32 order approximates. tests/test_fused_conv_cpu.py pins it against
 * torch's own CPU op (float64: equal to rounding; float32: within the fp32 order's error bound).
 * x: [N, Cin, H, W], dy: [N, Cout, H/stride, W/stride], dw: [Cout, Cin, k, k].
 */
int oracle_conv_wgrad_f32(const float* x, const float* dy, float* dw, int64_t N, int64_t Cin, int64_t Cout, int64_t H,
                          int64_t W, int64_t ksize, int64_t stride)
{
    const int64_t OH = H / stride, OW = W / stride, pad = ksize / 2;
    for (int64_t co = 0; co < Cout; ++co)
        for (int64_t ci = 0; ci < Cin; ++ci)
            for (int64_t kh = 0; kh < ksize; ++kh)
                for (int64_t kw = 0; kw < ksize; ++kw) {
                    double acc = 0.0;
                    for (int64_t n = 0; n < N; ++n)
                        for (int64_t oh = 0; oh < OH; ++oh) {
                            const int64_t ih = oh * stride + kh - pad;
                            if (ih < 0 || ih >= H) continue;
                            for (int64_t ow = 0; ow < OW; ++ow) {
                                const int64_t iw = ow * stride + kw - pad;
                                if (iw < 0 || iw >= W) continue;
                                acc += (double)dy[((n * Cout + co) * OH + oh) * OW + ow] * (double)x[((n * Cin + ci) * H + ih) * W + iw];
                            }
                        }
                    dw[((co * Cin + ci) * ksize + kh) * ksize + kw] = (float)acc;
                }
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * K8  3x3 / pad 1 convolution (stride 1 or 2), no bias     `out = self.conv1(out)` URSABench/models/preresnet.py:42,47,143
 *     (F.conv2d on the CPU path, oneDNN) and, with flip, the input gradient of that layer (ATen convolution_backward).
 *
 *     y[n][o][oh][ow] = sum_{i, kh, kw} w[o][i][kh][kw] * x[n][i][oh + kh - 1][ow + kw - 1]
 *     flip: w[o][i][kh][kw] is read as W[i][o][2 - kh][2 - kw] from the layer's own [Cin, Cout, 3, 3] tensor W
 *
 * As for K7, the sum is taken in double and rounded once (oneDNN's order is not a contract); pinned against torch's CPU ops
 * in tests/test_fused_conv_cpu.py. x: [N, Cin, H, W], y: [N, Cout, H, W].
 */
int oracle_conv3x3_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                       int flip, int64_t stride)
{
    /* forward: x [N, Cin, H, W] -> y [N, Cout, H/stride, W/stride]. flip (the layer's input gradient): x = dy [N, Cin, H, W] ->
     * y = dx [N, Cout, H*stride, W*stride], w = the layer's [Cin, Cout, 3, 3] tensor: dx[i][ih][iw] = sum over (o, kh, kw) with
     * ih = stride*oh + kh - 1, iw = stride*ow + kw - 1 of w[o][i][kh][kw] * dy[o][oh][ow] */
    if (!flip) {
        const int64_t OH = H / stride, OW = W / stride;
        for (int64_t n = 0; n < N; ++n)
            for (int64_t o = 0; o < Cout; ++o)
                for (int64_t oh = 0; oh < OH; ++oh)
                    for (int64_t ow = 0; ow < OW; ++ow) {
                        double acc = 0.0;
                        for (int64_t i = 0; i < Cin; ++i)
                            for (int64_t kh = 0; kh < 3; ++kh) {
                                const int64_t ih = oh * stride + kh - 1;
                                if (ih < 0 || ih >= H) continue;
                                for (int64_t kw = 0; kw < 3; ++kw) {
                                    const int64_t iw = ow * stride + kw - 1;
                                    if (iw < 0 || iw >= W) continue;
                                    acc += (double)w[((o * Cin + i) * 3 + kh) * 3 + kw] * (double)x[((n * Cin + i) * H + ih) * W + iw];
                                }
                            }
                        y[((n * Cout + o) * OH + oh) * OW + ow] = (float)acc;
                    }
        return 0;
    }
    const int64_t IH = H * stride, IW = W * stride;
    for (int64_t n = 0; n < N; ++n)
        for (int64_t i = 0; i < Cout; ++i)
            for (int64_t ih = 0; ih < IH; ++ih)
                for (int64_t iw = 0; iw < IW; ++iw) {
                    double acc = 0.0;
                    for (int64_t o = 0; o < Cin; ++o)
                        for (int64_t kh = 0; kh < 3; ++kh) {
                            const int64_t th = ih + 1 - kh;
                            if (th < 0 || th % stride || th / stride >= H) continue;
                            for (int64_t kw = 0; kw < 3; ++kw) {
                                const int64_t tw = iw + 1 - kw;
                                if (tw < 0 || tw % stride || tw / stride >= W) continue;
                                acc += (double)w[((o * Cout + i) * 3 + kh) * 3 + kw] * (double)x[((n * Cin + o) * H + th / stride) * W + tw / stride];
                            }
                        }
                    y[((n * Cout + i) * IH + ih) * IW + iw] = (float)acc;
                }
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * K9  1x1 / stride 2 convolution, no bias, no padding (`downsample`, URSABench/models/preresnet.py:130-136) and, with flip,
 *     its input gradient (zero wherever the forward did not read). Sums in double, rounded once (as K7 / K8's restatements).
 *     forward: x [N, Cin, H, W] -> y [N, Cout, H/2, W/2], w [Cout, Cin]; flip: x = dy [N, Cin, H, W] -> y = dx [N, Cout, 2H, 2W],
 *     w = the layer's [Cin, Cout] tensor.
 */
int oracle_conv1x1s2_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                         int flip)
{
    if (!flip) {
        const int64_t OH = H / 2, OW = W / 2;
        for (int64_t n = 0; n < N; ++n)
            for (int64_t o = 0; o < Cout; ++o)
                for (int64_t oh = 0; oh < OH; ++oh)
                    for (int64_t ow = 0; ow < OW; ++ow) {
                        double acc = 0.0;
                        for (int64_t i = 0; i < Cin; ++i) acc += (double)w[o * Cin + i] * (double)x[((n * Cin + i) * H + 2 * oh) * W + 2 * ow];
                        y[((n * Cout + o) * OH + oh) * OW + ow] = (float)acc;
                    }
    } else {
        const int64_t OH = 2 * H, OW = 2 * W;
        for (int64_t n = 0; n < N; ++n)
            for (int64_t i = 0; i < Cout; ++i)
                for (int64_t ih = 0; ih < OH; ++ih)
                    for (int64_t iw = 0; iw < OW; ++iw) {
                        double acc = 0.0;
                        if (ih % 2 == 0 && iw % 2 == 0)
                            for (int64_t o = 0; o < Cin; ++o) acc += (double)w[o * Cout + i] * (double)x[((n * Cin + o) * H + ih / 2) * W + iw / 2];
                        y[((n * Cout + i) * OH + ih) * OW + iw] = (float)acc;
                    }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * K10  the pre-activation unit, restated as the composition of the restatements above     URSABench/models/preresnet.py:33-52
 *      (`out = self.bn1(x); out = self.relu(out); out = self.conv1(out)` ... `out += residual`) and the backward of those ops.
 *
 * forward:  h = bn ? relu(batch_norm(x)) : x  (oracle_bn_relu_fwd_f32: torch's CPU rounding, running statistics updated);
 *           y = conv3x3(h, w, stride) (oracle_conv3x3_f32) ; addend: y = y + addend (one fp32 add, torch's `out += residual`);
 *           sums[c] = (sum y, sum y^2) per output channel in double - the statistics the next BatchNorm starts from;
 *           save [4][Cin] = mean, invstd, alpha = invstd * gamma, beta' = fmaf(-mean, alpha, beta)  (bn only).
 * h: scratch of x's size (bn only).
 */
int oracle_preact_fwd_f32(const float* x, const float* w, const float* addend, float* y, float* h, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, float* save, double* sums, int64_t N,
                          int64_t Cin, int64_t Cout, int64_t H, int64_t W, int64_t stride, float eps, float momentum, int bn)
{
    const float* in = x;
    if (bn) {
        oracle_bn_relu_fwd_f32(x, h, gamma, beta, running_mean, running_var, save, save + Cin, N, Cin, H * W, eps, momentum, 1);
        for (int64_t c = 0; c < Cin; ++c) {
            save[2 * Cin + c] = save[Cin + c] * gamma[c];
            save[3 * Cin + c] = fmaf(-save[c], save[2 * Cin + c], beta[c]);
        }
        in = h;
    }
    oracle_conv3x3_f32(in, w, y, N, Cin, Cout, H, W, 0, stride);
    const int64_t OHW = (H / stride) * (W / stride);
    for (int64_t c = 0; c < Cout; ++c) {
        double s1 = 0.0, s2 = 0.0;
        for (int64_t n = 0; n < N; ++n)
            for (int64_t j = 0; j < OHW; ++j) {
                const int64_t o = (n * Cout + c) * OHW + j;
                if (addend) y[o] = y[o] + addend[o];
                const double d = (double)y[o];
                s1 += d;
                s2 += d * d;
            }
        sums[2 * c] = s1;
        sums[2 * c + 1] = s2;
    }
    return 0;
}

/* backward, first half: dh = conv3x3_flip(dy, w, stride) (the layer's input gradient); g = fmaf(xin, alpha, beta') > 0 ? dh : 0
 * with the forward's saved scalars (threshold_backward on the forward's own gate); sums[c] = (sum g, sum g * (xin - mean)) in
 * double: the two sums of native_batch_norm_backward. dy: [N, Cd, H, W]; xin, g: [N, Cx, H*stride, W*stride]; save: [4][Cx]. */
int oracle_preact_bwd_f32(const float* dy, const float* w, const float* xin, const float* save, float* g, double* sums,
                          int64_t N, int64_t Cd, int64_t Cx, int64_t H, int64_t W, int64_t stride)
{
    oracle_conv3x3_f32(dy, w, g, N, Cd, Cx, H, W, 1, stride);
    const int64_t HW = H * stride * W * stride;
    for (int64_t c = 0; c < Cx; ++c) {
        const float mean = save[c], alpha = save[2 * Cx + c], shift = save[3 * Cx + c];
        double sum = 0.0, dotp = 0.0;
        for (int64_t n = 0; n < N; ++n)
            for (int64_t j = 0; j < HW; ++j) {
                const int64_t o = (n * Cx + c) * HW + j;
                const float ge = fmaf(xin[o], alpha, shift) > 0.0f ? g[o] : 0.0f;
                g[o] = ge;
                sum += (double)ge;
                dotp += (double)ge * ((double)xin[o] - (double)mean);
            }
        sums[2 * c] = sum;
        sums[2 * c + 1] = dotp;
    }
    return 0;
}

/* backward, second half (native_batch_norm_backward's dx, torch's CPU association, as oracle_bn_relu_bwd_f32 above) from the
 * gated gradient and its sums; dz (or NULL) = the gradient reaching x on its other path, added last. */
int oracle_bn_bwd_dx_f32(const float* x, const float* g, const float* dz, float* dx, const float* gamma, const float* save,
                         const double* sums, float* dgamma, float* dbeta, int64_t N, int64_t C, int64_t HW)
{
    const double n = (double)N * (double)HW;
    for (int64_t c = 0; c < C; ++c) {
        const float mean = save[c], invstd = save[C + c], w = gamma[c];
        const double sum = sums[2 * c], dotp = sums[2 * c + 1];
        const float gm = (float)(sum / n);
        const float k = (float)(dotp * (double)invstd * (double)invstd / n);
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < HW; ++j) {
                const int64_t o = (i * C + c) * HW + j;
                const float r = (((g[o] - gm) - (x[o] - mean) * k) * invstd) * w;
                dx[o] = dz ? dz[o] + r : r;
            }
        dbeta[c] = (float)sum;
        dgamma[c] = (float)(dotp * (double)invstd);
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * K11  the head of a training step, restated      URSABench/models/preresnet.py:146-150 (bn -> relu -> AvgPool2d(8) -> fc) and
 *      nn.CrossEntropyLoss() (URSABench/inference/sghmc.py:38-40,76-77), with the backward of `loss.backward()` (sghmc.py:80).
 *      Sums in double, rounded once (as K7 / K8's restatements: the device's fp32 trees are compared to rounding).
 */
/* pooled[n][c] = mean over HW of relu(batch_norm(z)) (oracle_bn_relu_fwd_f32's rounding for the normalised value); h: scratch */
int oracle_bn_relu_pool_f32(const float* z, float* h, const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float* save, float* pooled, int64_t N, int64_t C, int64_t HW, float eps, float momentum)
{
    oracle_bn_relu_fwd_f32(z, h, gamma, beta, running_mean, running_var, save, save + C, N, C, HW, eps, momentum, 1);
    for (int64_t c = 0; c < C; ++c) {
        save[2 * C + c] = save[C + c] * gamma[c];
        save[3 * C + c] = fmaf(-save[c], save[2 * C + c], beta[c]);
    }
    for (int64_t n = 0; n < N; ++n)
        for (int64_t c = 0; c < C; ++c) {
            double s = 0.0;
            for (int64_t j = 0; j < HW; ++j) s += (double)h[(n * C + c) * HW + j];
            pooled[n * C + c] = (float)(s / (double)HW);
        }
    return 0;
}

/* logits = p W^T + b; loss = mean CE over rows with target != ignore_index; gradients of the loss (grad_output 1) */
int oracle_fc_ce_f32(const float* p, const float* W, const float* b, const int64_t* target, float* loss, float* logits, float* dW,
                     float* db, float* dp, int64_t N, int64_t C, int64_t K, int64_t ignore_index)
{
    double* dl = (double*)malloc(sizeof(double) * (size_t)(N * K));
    double tot = 0.0;
    int64_t cnt = 0;
    for (int64_t n = 0; n < N; ++n) {
        double l[1024], m = -INFINITY, s = 0.0;
        for (int64_t k = 0; k < K; ++k) {
            double acc = b ? (double)b[k] : 0.0;
            for (int64_t c = 0; c < C; ++c) acc += (double)p[n * C + c] * (double)W[k * C + c];
            l[k] = (double)(float)acc;                  /* the logits are fp32 numbers */
            if (logits) logits[n * K + k] = (float)acc;
            if (l[k] > m) m = l[k];
        }
        for (int64_t k = 0; k < K; ++k) s += exp(l[k] - m);
        const double lse = m + log(s);
        const int valid = target[n] != ignore_index;
        if (valid) { tot += lse - l[target[n]]; ++cnt; }
        for (int64_t k = 0; k < K; ++k) dl[n * K + k] = valid ? exp(l[k] - lse) - (k == target[n] ? 1.0 : 0.0) : 0.0;
    }
    loss[0] = (float)(tot / (double)cnt);
    for (int64_t i = 0; i < N * K; ++i) dl[i] /= (double)cnt;
    for (int64_t k = 0; k < K; ++k) {
        double sb = 0.0;
        for (int64_t n = 0; n < N; ++n) sb += dl[n * K + k];
        if (db) db[k] = (float)sb;
        for (int64_t c = 0; c < C; ++c) {
            double acc = 0.0;
            for (int64_t n = 0; n < N; ++n) acc += dl[n * K + k] * (double)p[n * C + c];
            dW[k * C + c] = (float)acc;
        }
    }
    for (int64_t n = 0; n < N; ++n)
        for (int64_t c = 0; c < C; ++c) {
            double acc = 0.0;
            for (int64_t k = 0; k < K; ++k) acc += dl[n * K + k] * (double)W[k * C + c];
            dp[n * C + c] = (float)acc;
        }
    free(dl);
    return 0;
}

/* backward of oracle_bn_relu_pool_f32: dy[n][c][j] = dpooled[n][c] / HW, then oracle_bn_relu_bwd_f32 (gate from the saved scalars);
 * dy: scratch of z's size */
int oracle_bn_relu_pool_bwd_f32(const float* z, const float* dpooled, float* dy, const float* gamma, const float* beta, const float* save,
                                float* dz, float* dgamma, float* dbeta, int64_t N, int64_t C, int64_t HW)
{
    for (int64_t n = 0; n < N; ++n)
        for (int64_t c = 0; c < C; ++c)
            for (int64_t j = 0; j < HW; ++j) dy[(n * C + c) * HW + j] = dpooled[n * C + c] / (float)HW;
    return oracle_bn_relu_bwd_f32(z, dy, dz, gamma, beta, save, save + C, dgamma, dbeta, N, C, HW, 1);
}

/* ---------------------------------------------------------------------------------------
 * K12  1x1 / stride 1 convolution, no bias (`self.conv1` / `self.conv3` of the Bottleneck blocks, URSABench/models/preresnet.py:
 *      56,62,76-87) and, with flip, its input gradient. Sums in double, rounded once (as K7 / K8 / K9's restatements).
 *      forward: x [N, Cin, HW] -> y [N, Cout, HW], w [Cout, Cin]; flip: x = dy [N, Cin, HW] -> y = dx [N, Cout, HW], w = the
 *      layer's [Cin, Cout] tensor. (The weight gradient: oracle_conv_wgrad_f32 with ksize 1, stride 1.)
 */
int oracle_conv1x1_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t HW, int flip)
{
    for (int64_t n = 0; n < N; ++n)
        for (int64_t o = 0; o < Cout; ++o)
            for (int64_t p = 0; p < HW; ++p) {
                double acc = 0.0;
                for (int64_t i = 0; i < Cin; ++i)
                    acc += (double)(flip ? w[i * Cout + o] : w[o * Cin + i]) * (double)x[(n * Cin + i) * HW + p];
                y[(n * Cout + o) * HW + p] = (float)acc;
            }
    return 0;
}
