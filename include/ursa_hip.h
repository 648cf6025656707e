/* ursa_hip.h — C ABI of the MI355X (gfx950) SG-MCMC + Bayesian-model-averaging kernels.
 *
 * This is the drop-in boundary under URSABench's Python plug-in API. The reference has no
 * native layer: each entry point below replaces an *implicit* sequence of stock torch ops
 * that the reference launches from Python. Citations are relative to /root/reference/.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *   - nothing here allocates, frees, copies to the host or synchronises: every call only
 *     enqueues work on `stream` (a hipStream_t passed as void*), so a caller may capture
 *     it into a hipGraph
 *   - return 0 on success, a negative URSA_E* for argument errors, a positive hipError_t
 *     if the launch itself failed; no C++ exception crosses the boundary
 *   - all arithmetic is IEEE fp32 with the rounding sequence of the reference's CPU path
 *     (no contraction except the explicit fused multiply-adds named below)
 */
#ifndef URSA_HIP_H
#define URSA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define URSA_ABI_VERSION 8

typedef void* ursa_stream_t; /* hipStream_t */

/* error codes (negative); positive values are hipError_t */
#define URSA_OK        0
#define URSA_ENULL    (-1) /* a required pointer is NULL */
#define URSA_ESIZE    (-2) /* negative / overflowing size */
#define URSA_EALIGN   (-3) /* pointer not 4-byte aligned */
#define URSA_EFLAGS   (-4) /* unknown flag bits or inconsistent flag/pointer combination */
#define URSA_EVALUE   (-5) /* scalar outside its domain (e.g. num_classes > URSA_BMA_MAX_CLASSES) */

/* ------------------------------------------------------------------------------------
 * K1  optimSGHMC.step            URSABench/inference/optim_sghmc.py:43-67
 *
 * One fused pass over the flat parameter arena (all parameter tensors of a chain, or of
 * several chains, laid end to end). Per element i, fp32, in this order:
 *     g~ = WD ? fmaf(c_wd, theta, g) : g                              (:47-48)
 *     mu != 0:  v0 = FIRST ? g~ : mom                                 (:51-52 clone)
 *               v  = fmaf(-lr, g~, v0 * mu)                           (:53 / :56)
 *               d  = v
 *     mu == 0:  d  = g~ * (-lr)                                       (:62)
 *     NOISE:    d  = d + (eps * c_noise) / n_train                    (:64, true division)
 *     theta = theta + d                                               (:65)
 *     mu != 0:  mom = d                                               (:67, buffer includes noise)
 * eps: if `eps` != NULL it is read from there (parity mode: noise captured from the
 * reference's generator); if NULL it is generated in registers: Philox4x32-10 keyed by
 * `seed`, counter (i/4, step), Box-Muller (see csrc/ursa_rng.h).
 * Optional fusions: ZERO_GRAD stores 0 to grad[i] (replaces optimizer.zero_grad(),
 * sghmc.py:79); snapshot != NULL also stores the new theta there (the posterior-sample
 * snapshot of sghmc.py:99, as a device-resident member-bank row).
 * HBM traffic: 20 B/param (mu != 0), 12 B/param (mu == 0); +4 each for eps / ZERO_GRAD /
 * snapshot.
 */
#define URSA_STEP_NOISE     0x1u
#define URSA_STEP_FIRST     0x2u
#define URSA_STEP_ZERO_GRAD 0x4u
#define URSA_STEP_WD        0x8u
/* SGD mode: torch.optim.SGD(momentum, weight_decay) as used by the SWA/SWAG trajectory
 * (URSABench/inference/swa.py:41-42, swag.py:55-70), dampening 0, no nesterov, no noise:
 *     g~ = WD ? fmaf(c_wd, theta, g) : g          (c_wd = weight_decay itself here)
 *     mu != 0:  b = FIRST ? g~ : mom * mu + g~ ;  mom = b        mu == 0:  b = g~
 *     theta = fmaf(-lr, b, theta)
 */
#define URSA_STEP_SGD       0x10u
#define URSA_STEP_ALLFLAGS  0x1Fu   /* flags of the scalar-argument launch */

int ursa_sgmcmc_step_f32(float* theta, float* grad, float* mom /* NULL iff mu == 0 */,
                         const float* eps /* NULL => Philox */, float* snapshot /* or NULL */,
                         int64_t n, float lr, float mu, float c_wd, float c_noise,
                         float n_train, uint64_t seed, uint64_t step, uint32_t flags,
                         ursa_stream_t stream);

/* Same update with the per-step scalars read from a DEVICE control block, so the launch
 * can sit inside a captured hipGraph and be replayed while lr / step / flags change.
 *
 * Advancing a block = step += 1, clear FIRST, and if `sched` != NULL load
 * (lr, c_noise) = sched[(step - sched_base) % sched_len] — the per-iteration cyclical
 * schedule of csghmc.py:64-72 precomputed by the host in float64 and rounded once; the host
 * sets sched_base = step when it uploads an epoch's table. In SGD mode (URSA_STEP_SGD in
 * ctl->flags; no noise there) the second column is the MOMENTUM instead: per-iteration
 * (lr, momentum) of OneCycleLR, URSABench/inference/vi_dropout.py:59-61,107. `step` is also
 * the Philox call index, so it only ever grows.
 * With URSA_STEP_ADVANCE in ctl->flags the update launch advances its own block: every
 * workgroup takes a ticket once all of its waves hold a copy of the block, and the one that
 * completes the chain's ticket tree does the advance (all others have read the block by then)
 * and re-arms the tickets — no second launch. Tickets are relaxed agent-scope fetch-adds on a
 * two-level tree: workgroup b counts on tickets[(b % 16) * 32], the last of each of those 16
 * counters counts on tickets[16 * 32]. The counters sit on separate 128-byte lines because
 * fetch-adds to ONE line serialise at ~11.5 ns each (1,024 workgroups: +10.7 us on one
 * address, +0.6 us on 16 lines; tools/exp/ticket_probe.hip) — that is why the block is 2,304
 * bytes. ursa_step_ctl_advance is the same advance as a launch of its own (one thread per
 * block of ctl[n_ctl]), for hosts that step a block without an update. `tickets` is device
 * scratch: upload it as zeros. Blocks of an array must be 128-byte aligned (so: the array). */
#define URSA_STEP_ADVANCE   0x20u
#define URSA_CTL_TICKET_LINES 16
typedef struct ursa_step_ctl {
    float lr, mu, c_wd, c_noise, n_train;
    uint32_t flags;
    uint64_t seed, step, sched_base;
    const float* sched;       /* DEVICE pointer, [sched_len][2], or NULL */
    uint32_t sched_len;
    uint32_t reserved;
    uint32_t pad[16];         /* the scalars own the first 128-byte line */
    uint32_t tickets[(URSA_CTL_TICKET_LINES + 1) * 32];
} ursa_step_ctl;              /* 2,304 bytes = 18 lines of 128 */

int ursa_sgmcmc_step_ctl_f32(float* theta, float* grad, float* mom, const float* eps,
                             float* snapshot, int64_t n, ursa_step_ctl* ctl,
                             ursa_stream_t stream);
int ursa_step_ctl_advance(ursa_step_ctl* ctl /* [n_ctl] */, int32_t n_ctl, ursa_stream_t stream);

/* K chains in ONE launch (SURVEY.md 8b `n_chains`, 8f-1; the reference steps one chain per process, one parameter
 * tensor at a time: URSABench/experiment.py:166-173 + optim_sghmc.py:43): the K independent chains that share a GPU
 * keep their vectors in [K, chain_stride] slabs — chain k's theta / grad / mom (/ eps / snapshot)
 * start at element k * chain_stride of the base pointers — and their control blocks in ctl[K].
 * Chain k is updated exactly as ursa_sgmcmc_step_ctl_f32(theta + k*chain_stride, ..., n_per_chain,
 * ctl + k) would (own lr / flags / Philox key and call index; element i of chain k draws Philox
 * lane (i, ctl[k].step) of key ctl[k].seed), bit for bit. grid = (ceil(n_per_chain/2048), K): one float4 per
 * thread, 512-thread workgroups.
 * chain_stride must be a multiple of 4 and >= n_per_chain; pointers 16-byte aligned, ctl 128-byte aligned. */
int ursa_sgmcmc_step_multi_f32(float* theta, float* grad, float* mom, const float* eps,
                               float* snapshot, int64_t n_per_chain, int32_t n_chains,
                               int64_t chain_stride, ursa_step_ctl* ctl /* [n_chains] */,
                               ursa_stream_t stream);

/* Standard-normal fill with the same Philox/Box-Muller stream as K1 (element i of call
 * (seed, step) is exactly the eps K1 would use). Used by tests and by SWAG/HMC hosts. */
int ursa_philox_normal_f32(float* out, int64_t n, uint64_t seed, uint64_t step,
                           ursa_stream_t stream);

/* Device self-test of the generator arithmetic: the Box-Muller radius sqrt(-2 ln u) is computed in registers with a
 * 6-instruction division and a 9-instruction square root that are correctly rounded on the argument ranges the
 * generator feeds them; this launch recomputes the radius (and the logarithm) of EVERY one of the 2^32 possible Philox
 * words with the compiler's general IEEE division / square root as well and adds the number of differing results to
 * mismatches[0] (radius) and mismatches[1] (logarithm) — two uint64 in DEVICE memory, zeroed by the caller. Both must
 * stay 0: that is what keeps the device's noise stream bit-identical to the CPU restatement. ~30 ms. */
int ursa_selftest_rng_f32(uint64_t* mismatches /* device, [2], zero-initialised */, ursa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K2  SWA._collect_model         URSABench/inference/swa.py:81-88
 *     mean = mean * decay + w / denom ;  sq = sq * decay + (w * w) / denom
 * with decay = float(n/(n+1.0)), denom = float(n+1.0) computed by the host in float64
 * (n = num_models_collected). Each product, quotient and sum is rounded separately.
 * HBM traffic: 20 B/param.
 */
int ursa_swag_collect_f32(float* mean, float* sq, const float* w, int64_t n, float decay,
                          float denom, ursa_stream_t stream);

/* K3  SWAG diagonal draw         URSABench/inference/swag.py:84-86 + swa.py:106-108
 *     var   = max(sq - mean*mean, var_clamp)
 *     theta = eps * (sqrt(var) * scale) + mean          (torch.normal(mean, std))
 * eps as in K1 (pointer, or Philox keyed by (seed, draw)). theta_out is a member-bank row.
 * HBM traffic: 12 B/param/member.
 */
int ursa_swag_draw_f32(float* theta_out, const float* mean, const float* sq,
                       const float* eps, int64_t n, float var_clamp, float scale,
                       uint64_t seed, uint64_t draw, ursa_stream_t stream);

/* K3 for an ensemble: std = sqrt(max(sq - mean*mean, var_clamp)) * scale is the same for every member drawn
 * from one pair of moment vectors (URSABench/inference/swag.py:131-147 draws num_samples members from them).
 * ursa_swag_std_f32 stores it once (12 B/param, once); ursa_swag_draw_std_f32 is the per-member draw
 *     theta = eps * std + mean
 * — bit-identical to ursa_swag_draw_f32 on the same inputs (the same operations, the square roots hoisted out
 * of the per-member launch). HBM traffic: 12 B/param/member. */
int ursa_swag_std_f32(float* std_out, const float* mean, const float* sq, int64_t n,
                      float var_clamp, float scale, ursa_stream_t stream);
int ursa_swag_draw_std_f32(float* theta_out, const float* mean, const float* std,
                           const float* eps /* NULL => Philox (seed, draw) */, int64_t n,
                           uint64_t seed, uint64_t draw, ursa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K5  tasks accumulators         URSABench/tasks/prediction.py:57-63,
 *                                URSABench/tasks/ood_detection.py:59-65,
 *                                URSABench/tasks/decision_making.py:124-129,
 *                                URSABench/util.py:126-144
 * logits[S, B, C] (S ensemble members evaluated on the same B rows). Per row b, for
 * s = 0..S-1 in order:
 *     p   = exp((z - max z) - log(sum exp(z - max z)))        (log_softmax().exp_())
 *     ps  = p * (1-gamma) + gamma/C                           (central_smoothing)
 *     proba_sum[b, :] += SMOOTHED ? ps : p
 *     ent_sum[b]      += -sum_c ps * log(ps)                  (compute_predictive_entropy)
 *     risk_sum[b, :]  += ps @ cost            (only if risk_sum != NULL; cost is [C, C])
 * one_minus_gamma = float(1-gamma), gamma_over_c = float(gamma*1/C) from the host.
 * ent_sum may be NULL (Decision). 1 <= C <= URSA_BMA_MAX_CLASSES.
 * Results are within 1e-5 relative of that sequence, not bit-exact: members are summed in a fixed
 * but blocked order (partial sums per member slot, folded in slot order), e = 2^(z log2e - max log2e)
 * on v_exp_f32 (<= 7e-7 relative per exponential), p = e / sum, entropy through v_log_f32; measured
 * worst case 2e-6 on proba_sum and 3.5e-6 on ent_sum (tools/exp/k5_accuracy.py). A logit of -inf
 * (masked class) gives p = 0; with gamma_over_c = 0 a probability that underflows to 0 adds 0 (not
 * NaN) to ent_sum. SMOOTHED sums are formed as (1-gamma) * sum_s p + S * gamma/C.
 * Fast paths need alignment, the generic kernel does not: C <= 16 with logits 16-byte aligned and
 * B*C % 4 == 0 (row-per-lane kernel, LDS-staged tiles); C % 4 == 0 with logits 16-byte aligned
 * (float4 class loads). Give the call the logits of as many rows as there are (e.g. the whole test
 * set): one launch amortises the ~4 us launch floor.
 * HBM traffic: 4*S*B*C read + read-modify-write of the accumulators.
 */
#define URSA_BMA_SMOOTHED 0x1u
#define URSA_BMA_MAX_CLASSES 1024

int ursa_bma_accumulate_f32(const float* logits, float* proba_sum, float* ent_sum,
                            float* risk_sum, const float* cost, int32_t S, int64_t B,
                            int32_t C, float one_minus_gamma, float gamma_over_c,
                            uint32_t flags, ursa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K4  HMC leapfrog               call site URSABench/inference/hmc.py:71-75; arithmetic is
 *     hamiltorch's (un-vendored dependency: parity unpinned). grad is
 *     d log p / d theta.
 *     KICK : mom   = mom + kick_coef * grad          (kick_coef = eps/2, eps or -eps/2)
 *     DRIFT: theta = theta + (step_size * inv_mass) * mom        (after the kick, if both)
 *     kinetic_out != NULL: kinetic_out[0] += 0.5 * inv_mass * sum(mom^2) (after the kick)
 * Reductions are deterministic (fixed grid, per-block partials in `ws`, summed in block
 * order by a second 1-block launch): ws must hold URSA_REDUCE_WS_FLOATS floats and is
 * scratch, it needs no initialisation.
 * HBM traffic: 20 B/param fused kick+drift.
 */
#define URSA_REDUCE_WS_FLOATS 2048
#define URSA_LEAP_KICK  0x1u
#define URSA_LEAP_DRIFT 0x2u

int ursa_leapfrog_f32(float* theta, float* mom, const float* grad, int64_t n,
                      float kick_coef, float step_size, float inv_mass, uint32_t flags,
                      float* kinetic_out /* 1 float, or NULL */, float* ws /* iff kinetic_out */,
                      ursa_stream_t stream);

/* out[0] += sum x[i]^2  (prior term tau/2*||theta||^2 and kinetic energy). 4 B/param. */
int ursa_sumsq_f32(const float* x, int64_t n, float* out, float* ws, ursa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K6  BatchNorm2d (+ ReLU) of the pre-activation blocks      URSABench/models/preresnet.py:40-41,
 *     45-46,76-85,146; wideresnet.py:47,49,117 (`relu(bn(x))` = torch.nn.functional.batch_norm
 *     followed by relu; backward = threshold_backward + native_batch_norm_backward).
 *     x, y, dy, dx: contiguous NCHW fp32, [N, C, HW]. float4 accesses when HW % 4 == 0 and the
 *     pointers are 16-byte aligned, 4-byte accesses otherwise (same results).
 *
 *   training forward (two launches: per-channel partial sums in double; merge + normalise):
 *     mean_c = (float) (sum x / n), var_c = sum x^2 / n - mean^2 (double; biased), invstd_c =
 *     (float) (1 / sqrt(var_c + eps)) : correctly rounded, like torch's CPU kernel (double accumulation)
 *     alpha_c = invstd_c * gamma_c ; beta'_c = fmaf(-mean_c, alpha_c, beta_c)
 *     y = fmaf(x, alpha_c, beta'_c) ; RELU: y = y < 0 ? 0 : y
 *         (the association of torch 2.10's CPU BatchNorm, bit for bit given x / mean / invstd: the
 *          ReLU gate of a pre-activation near zero must not depend on which device ran it)
 *     save_mean[c] = mean_c ; save_invstd[c] = invstd_c
 *     running_mean != NULL:  running_mean[c] = momentum * mean_c + (1 - momentum) * running_mean[c]
 *                            running_var[c]  = momentum * var_c * n/(n-1) + (1 - momentum) * running_var[c]
 *   evaluation forward (one launch): the same with mean / var = running_mean / running_var.
 *   backward (two launches; torch's CPU association, native_batch_norm_backward in training mode):
 *     g = RELU ? (fmaf(x, alpha_c, beta'_c) > 0 ? dy : 0) : dy        (the forward's gate, recomputed)
 *     sum = sum g ; dotp = sum g * (x - mean_c)                      (double)
 *     dbeta[c] = sum ; dgamma[c] = dotp * invstd_c ; gm = sum / n ; k = dotp * invstd_c^2 / n
 *     dx = (((g - gm) - (x - mean_c) * k) * invstd_c) * gamma_c       (n = N * HW)
 *   one-pass form: when a channel fits one 1,024-thread workgroup's registers (N*HW <= 32,768, 16-byte
 *   aligned float4 accesses) and there are >= 48 channels (= workgroups) the forward and the backward are
 *   ONE launch each, grid = C: the channel is read once, reduced in the workgroup and written - same
 *   arithmetic and the same floats as the two-launch form (URSA_BN_TWO_LAUNCH keeps that one).
 *   residual form (the blocks end `out += residual`, preresnet.py:49-52,87-90, and the next block's first op is
 *   relu(bn(out))): with addend != NULL the forward / evaluation launches normalise z = x + addend (one fp32 add,
 *   as torch's) and also store z to z_out - the add launch folded into the statistics pass; with dz != NULL the
 *   backward returns dx + dz, dz being the gradient that reaches z on its other path (the next residual sum or
 *   the downsampling convolution) - the accumulation autograd would otherwise run as its own add launch.
 * ws: scratch, URSA_BN_WS_FLOATS(C) floats, 16-byte aligned, no initialisation needed; partial
 * results of the first launch of a call, consumed by its second launch.
 * Deterministic (no atomics); fp32 throughout. Error vs exact arithmetic: a few ulp on y / dx
 * (tests compare with float64 BatchNorm at 2e-6 relative to the activation scale).
 * Traffic: forward 12 B/element (x twice, y once), backward 20 B/element, evaluation 8 B/element;
 * residual form: forward 20, backward 24, evaluation 16. One-pass and held forms (URSA_BN_HELD): 8 / 12 and 16 / 16.
 */
#define URSA_BN_RELU        0x1u
#define URSA_BN_TWO_LAUNCH  0x2u   /* keep the two-launch form where the one-pass / held form would apply (A/B, tests) */
#define URSA_BN_HELD        0x4u   /* OPT-IN. The caller vouches that ws from URSA_BN_WS_HELD_OFFSET_FLOATS(C) on is ZERO: the library may then run
                                      the held form - ONE launch, every input read once - on activations of >= 24 MiB (backward) / >= 48 MiB
                                      (forward; 32 MiB with an addend) whose channels do not fit one workgroup (per-channel workgroups hold
                                      their chunk in registers, exchange double partial sums through ws and wait for each other; that part
                                      of ws is zero again when the launch has drained and no other form writes there, so a ws zeroed once
                                      can be reused call after call - by ONE layer: never share it between launches that may overlap). Same
                                      floats as the two-launch form. The wait is starvation-free only while NOTHING ELSE occupies the device
                                      beside the launch (other streams, graph branches, collectives, other processes): launches that may
                                      overlap with anything must not carry the flag. The wait is bounded (~3 s); a launch that runs into the
                                      bound raises the err word in ws (float offset 512 C + 33, as uint32) and POISONS ITS OUTPUTS: y / dx of
                                      the starved pieces, save_mean / save_invstd / running statistics / dgamma / dbeta of the channel are NaN.
                                      Without the flag (the default of every caller in this repository) ws needs no initialisation and the
                                      held form is never taken. */
#define URSA_BN_ALLFLAGS    0x7u
/* [two-launch form: partial sums, C x 64 x {double, double}] [held form: its partial-sum slots, the same size; then its
 * counters, one 128-byte line each: ticket, {done, err}, one per channel] */
#define URSA_BN_WS_FLOATS(C) ((int64_t)(C) * 64 * 8 + ((int64_t)(C) + 2) * 32)
#define URSA_BN_WS_HELD_OFFSET_FLOATS(C) ((int64_t)(C) * 64 * 4)

/* save_gate (2 C floats, or NULL): alpha_c and beta'_c exactly as this forward used them, [alpha_0..alpha_{C-1}, beta'_0..beta'_{C-1}].
 * Handed to the backward as `gate`, the ReLU gate is recomputed from THESE instead of from the live gamma / beta: a parameter
 * changed in place between forward and backward by a raw-pointer launch (K1 updates the arena the parameters are views of; autograd's
 * version counters cannot see such a write) then cannot move a gate away from the one the forward took. gate == NULL: the forward's
 * expressions on the live parameters (the same bits while nobody touched them). dx's factor gamma_c is read live either way, like
 * torch's own backward. */
int ursa_bn_relu_fwd_f32(const float* x, const float* addend /* or NULL */, float* z_out /* iff addend */,
                         float* y, const float* gamma, const float* beta,
                         float* running_mean /* or NULL */, float* running_var /* or NULL */,
                         float* save_mean, float* save_invstd, float* save_gate /* 2 C floats, or NULL */, float* ws,
                         int64_t N, int64_t C, int64_t HW, float eps, float momentum, uint32_t flags, ursa_stream_t stream);

int ursa_bn_relu_eval_f32(const float* x, const float* addend /* or NULL */, float* z_out /* iff addend */,
                          float* y, const float* gamma, const float* beta,
                          const float* running_mean, const float* running_var, int64_t N, int64_t C,
                          int64_t HW, float eps, uint32_t flags, ursa_stream_t stream);

int ursa_bn_relu_bwd_f32(const float* x /* the normalised input: z_out if the forward had an addend */,
                         const float* dy, const float* dz /* or NULL */, float* dx, const float* gamma,
                         const float* beta, const float* save_mean, const float* save_invstd,
                         const float* gate /* the forward's save_gate, or NULL */,
                         float* dgamma, float* dbeta, float* ws, int64_t N, int64_t C, int64_t HW,
                         uint32_t flags, ursa_stream_t stream);

#ifdef URSA_DEBUG_KNOBS   /* parked experiment, NOT part of the product ABI: only csrc/libursa_hip_knobs.so exports these */
/* The same forward / backward that ALSO store their output channels-last (NHWC, [N, HW, C]) as a second tensor: y_nhwc /
 * dx_nhwc hold the same floats as y / dx (dx + dz in the residual form). Why: MIOpen's fastest weight-gradient kernel for the
 * benchmark networks works on NHWC operands and transposes NCHW ones itself (15-20 % of a training step's kernel time);
 * both operands of every 3x3 weight gradient are K6 outputs, so K6 hands them over ready-made (ursabench_amd/fused_conv.py;
 * replaces nothing in the reference, whose stock ops leave the layout to the backend). Needs HW % 4 == 0, C % 4 == 0
 * and 16-byte aligned pointers (URSA_EVALUE otherwise: the caller falls back to the plain entry points); always the two
 * launches of the two-launch form (no one-pass / held form); the gated backward has no twin. The second launch works on four
 * channels at a time: its four waves merge the four channels' partial sums side by side (same lane order as the plain
 * form: same floats), a thread loads one float4 from each of the four channel rows, transposes the 4x4 block in registers
 * and stores four NCHW float4 as before plus four NHWC float4 (channels 4k..4k+3 of four consecutive positions). */
int ursa_bn_relu_fwd_nhwc_f32(const float* x, const float* addend /* or NULL */, float* z_out /* iff addend */, float* y,
                              float* y_nhwc, const float* gamma, const float* beta, float* running_mean /* or NULL */,
                              float* running_var /* or NULL */, float* save_mean, float* save_invstd, float* ws,
                              int64_t N, int64_t C, int64_t HW, float eps, float momentum, uint32_t flags,
                              ursa_stream_t stream);
int ursa_bn_relu_bwd_nhwc_f32(const float* x, const float* dy, const float* dz /* or NULL */, float* dx, float* dx_nhwc,
                              const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                              float* dgamma, float* dbeta, float* ws, int64_t N, int64_t C, int64_t HW, uint32_t flags,
                              ursa_stream_t stream);
#endif /* URSA_DEBUG_KNOBS */

/* The same backward with the ReLU gates of LISTED elements given instead of recomputed: a parity instrument, not a
 * production launch (binary search per element; always the two-launch kernels; URSA_BN_RELU required).
 *   gate_idx[0..n_gates): ascending element offsets into the [N, C, HW] tensor (N*C*HW < 2^31 - 1); entries equal to
 *                         INT32_MAX are padding (a fixed-capacity buffer re-filled between hipGraph replays);
 *   gate_open[k] != 0:    dy passes at element gate_idx[k]; == 0: it is blocked; unlisted elements: the forward's gate.
 * Why: the inputs of these layers are convolution outputs, and MIOpen's and oneDNN's convolutions differ in the last
 * bits, so a pre-activation within ~1e-6 of zero opens its gate on one device and not on the other - for any BatchNorm
 * arithmetic. Such an element changes nothing in the forward pass and O(dy) in its gradient. With the reference CPU
 * run's gates listed for the pre-activations it computed within 1e-4 of zero, GPU and CPU evaluate the same
 * piecewise-linear function and north_star's 1e-5 criterion becomes a statement about the implementation, at any batch
 * size and over several steps (tests/test_gate_parity_gpu.py, bench.py `parity`). Replaces nothing in the reference
 * (URSABench/models/preresnet.py:40-41 has one device); oracle twin: oracle_bn_relu_bwd_gated_f32. */
int ursa_bn_relu_bwd_gated_f32(const float* x, const float* dy, const float* dz /* or NULL */, float* dx,
                               const float* gamma, const float* beta, const float* save_mean,
                               const float* save_invstd, const float* gate /* or NULL */, float* dgamma, float* dbeta, float* ws, int64_t N,
                               int64_t C, int64_t HW, uint32_t flags, const int32_t* gate_idx,
                               const uint8_t* gate_open, int64_t n_gates, ursa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K7  weight gradient of the ResNets' convolutions      `loss.backward()` URSABench/inference/sghmc.py:80 (and the same
 *     call of every sampler) -> ATen convolution_backward for the nn.Conv2d(.., bias=False) layers of
 *     URSABench/models/preresnet.py:25-27,100,130-136
 *
 *     dw[co][ci][kh][kw] = sum_{n, oh, ow} dy[n][co][oh][ow] * x[n][ci][oh*stride + kh - p][ow*stride + kw - p]   (zero padded)
 *
 * ksize 3 (p = 1) or 1 (p = 0). x: [N, Cin, H, W], dy: [N, Cout, H/stride, W/stride], dw: [Cout, Cin, ksize, ksize], all
 * contiguous NCHW fp32, x / dy / ws 16-byte aligned. Exact fp32 on v_mfma_f32_16x16x4_f32 (every product rounded once, fma
 * chains); the batch x position sum is split over workgroups (K slices, partial copies of dW in `ws`) and reduced in a FIXED
 * order by a second launch (no atomics: the same inputs give the same bits every run). Nothing transposed, nothing zero-filled.
 * Shapes covered (any N; H = W): every convolution of the CIFAR pre-activation ResNets with BasicBlocks -
 *     3x3 stride 1: (Cin, Cout, H) in {(3, 16, 32), (16, 16, 32), (32, 32, 16), (64, 64, 8)}
 *     3x3 stride 2: (16, 32, 32), (32, 64, 16)            1x1 stride 2: (16, 32, 32), (32, 64, 16)
 *     1x1 stride 1 (K12, the Bottleneck networks): (64, 16, 32) (16, 64, 32) (128, 32, 16) (32, 128, 16) (256, 64, 8) (64, 256, 8)
 *                                                  (64, 32, 32) (128, 64, 16)
 * ursa_conv_wgrad_ws_floats() returns the scratch `ws` must hold for a shape, 0 when the shape is not covered (the caller
 * then keeps the stock weight gradient; the launches return URSA_EVALUE).
 *   ursa_conv_wgrad_f32          both launches.
 *   ursa_conv_wgrad_partial_f32  the first launch only: `ws` then holds the K slices' partial sums ...
 *   ursa_conv_wgrad_reduce_f32   ... and this takes the second launch for `n` such layers at once (one launch per 48 layers):
 *                                a backward pass costs one launch per layer plus one. Every item is checked before anything
 *                                is launched; `ws` of every item must stay untouched between its two launches.
 * Algorithmic HBM traffic: 4 B x (N*Cin*H*W + N*Cout*OH*OW + Cout*Cin*k*k); the partials (ws) are written and read once more.
 */
typedef struct ursa_conv_pending {
    const float* ws;   /* the partial sums ursa_conv_wgrad_partial_f32 left for this layer */
    float* dw;         /* [Cout, Cin, ksize, ksize] */
    int64_t N, Cin, Cout, H, W;
    int32_t ksize, stride;
} ursa_conv_pending;

int64_t ursa_conv_wgrad_ws_floats(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int32_t ksize, int32_t stride);
int ursa_conv_wgrad_f32(const float* x, const float* dy, float* dw, float* ws, int64_t ws_floats, int64_t N,
                        int64_t Cin, int64_t Cout, int64_t H, int64_t W, int32_t ksize, int32_t stride, ursa_stream_t stream);
int ursa_conv_wgrad_partial_f32(const float* x, const float* dy, float* ws, int64_t ws_floats, int64_t N, int64_t Cin,
                                int64_t Cout, int64_t H, int64_t W, int32_t ksize, int32_t stride, ursa_stream_t stream);
int ursa_conv_wgrad_reduce_f32(const ursa_conv_pending* items, int32_t n, ursa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K8  forward / input gradient of the 3x3 convolutions      `out = self.conv1(out)` URSABench/models/
 *     preresnet.py:42,47,143 and the input-gradient half of `loss.backward()` (URSABench/inference/sghmc.py:80)
 *
 *     y[n][o][oh][ow] = sum_{i, kh, kw} w[o][i][kh][kw] * x[n][i][s oh + kh - 1][s ow + kw - 1]      (zero padded, no bias)
 *     s = 1, or 2 with URSA_CONV_STRIDE2 (x: [N, Cin, H, W] -> y: [N, Cout, H/s, W/s]).
 *     URSA_CONV_FLIP: the input gradient of such a layer: x = dy [N, Cin', H, W] with Cin' = the layer's OUTPUT channels
 *                     (H = W = dy's size), y = dx [N, Cout', s H, s W] with Cout' = its INPUT channels, w = the layer's own
 *                     [Cin', Cout', 3, 3] weight tensor (Cin / Cout below = Cin' / Cout').
 *
 * Contiguous NCHW fp32, x / y / w 16-byte aligned (URSA_EALIGN otherwise; w is read through 16-byte loads); w contiguous. One launch; exact fp32 on v_mfma_f32_16x16x4_f32: each output
 * is four interleaved fma chains over (channel group, tap), added pairwise - a direct convolution, no Winograd transform. Shapes covered
 * (any N, H = W):   stride 1 (Cin, Cout, H): (3, 16, 32) forward only; (16, 16, 32), (32, 32, 16), (64, 64, 8) both forms
 *                   stride 2 forward: (16, 32, 32), (32, 64, 16);   stride 2 flipped (Cin', Cout', H of dy): (32, 16, 16), (64, 32, 8)
 * ursa_conv3x3_supported() says whether a (shape, flags) is covered (otherwise URSA_EVALUE).
 * Algorithmic HBM traffic: 4 B x (elements of x + elements of y + Cout*Cin*9).
 */
#define URSA_CONV_FLIP    0x1u
#define URSA_CONV_STRIDE2 0x2u
int ursa_conv3x3_supported(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, uint32_t flags);
int ursa_conv3x3_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                     uint32_t flags, ursa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K9  the 1x1 / stride 2 shortcut convolutions (`downsample`, URSABench/models/preresnet.py:130-136), forward and input
 *     gradient; no bias, no padding.
 *
 *     y[n][o][oh][ow] = sum_i w[o][i] * x[n][i][2 oh][2 ow]                 x: [N, Cin, H, W] -> y: [N, Cout, H/2, W/2]
 *     URSA_CONV_FLIP:   the input gradient: x = dy [N, Cin', H, W] (Cin' = the layer's OUTPUT channels, H = W = dy's size),
 *                       y = dx [N, Cout', 2H, 2W] (Cout' = the layer's INPUT channels), w = the layer's own [Cin', Cout']
 *                       tensor; dx is zero wherever the forward did not read x.
 * Plain fp32 fma chains (ascending input channel), one launch. Shapes covered (any N, H = W): forward (Cin, Cout, H) in
 * {(16, 32, 32), (32, 64, 16)}; flipped (Cin', Cout', H) in {(32, 16, 16), (64, 32, 8)} - the same two layers.
 * Algorithmic HBM traffic: forward 4 B x (N*Cin*H*W/2 + N*Cout*H*W/4) (only even rows of x are read, in whole cache lines);
 * flipped 4 B x (N*Cin'*H*W + 4*N*Cout'*H*W).
 */
int ursa_conv1x1s2_supported(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, uint32_t flags);
int ursa_conv1x1s2_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                       uint32_t flags, ursa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K12  the 1x1 / stride 1 convolutions of the Bottleneck pre-activation ResNets (`self.conv1`, `self.conv3`, the stride-1
 *      `downsample`: URSABench/models/preresnet.py:56,62,76-87,130-136), forward and input gradient; no bias, no padding.
 *
 *     y[n][o][p] = sum_i w[o][i] x[n][i][p]                                   x: [N, Cin, H, W] -> y: [N, Cout, H, W]
 *     URSA_CONV_FLIP:   the input gradient: x = dy [N, Cin', H, W] (Cin' = the layer's OUTPUT channels), y = dx [N, Cout', H, W]
 *                       (Cout' = the layer's INPUT channels), w = the layer's own [Cin', Cout'] tensor.
 * A GEMM on the NCHW planes as they lie (per image Y = W X): exact fp32 on v_mfma_f32_16x16x4_f32, fma chains over the input
 * channels in ascending groups of four; nothing is transposed (MIOpen runs these layers as NCHW->NHWC transposes + an implicit /
 * rocBLAS GEMM + a transpose back). Shapes covered (any N, H = W; (Cin, Cout, H) of the launch, either direction):
 *     (64, 16, 32) (16, 64, 32) (128, 32, 16) (32, 128, 16) (256, 64, 8) (64, 256, 8) (16, 16, 32) (64, 32, 32) (32, 64, 32)
 *     (128, 64, 16) (64, 128, 16)          - every 1x1 / stride 1 layer of PreResNet-164 and its input gradient.
 * Weight gradients of these layers: ursa_conv_wgrad_* with ksize 1, stride 1 (K7's two launches; the first is K12's GEMM over
 * batch x positions, its partial sums in K7's tile order) - covered: every such layer except (16, 16, 32).
 * Algorithmic HBM traffic: 4 B x (elements of x + elements of y + Cout*Cin).
 */
int ursa_conv1x1_supported(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, uint32_t flags);
int ursa_conv1x1_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                     uint32_t flags, ursa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K13  `conv1x1(relu(bn(x)))` of the Bottleneck pre-activation unit without storing the normalised activation
 *      URSABench/models/preresnet.py:70-87 (`out = self.bn1(x); out = self.relu(out); out = self.conv1(out)` and the same around
 *      bn3 / conv3) and their part of the backward pass inside hamiltorch's potential gradient (URSABench/inference/hmc.py:71-75).
 *  ursa_bn_stats_f32              K6's first forward launch + the merge of its partial sums, ALONE: batch statistics of x (addend:
 *                                 of z = x + addend, stored to z_out - the previous block's `out += residual`), the running
 *                                 statistics update, and save[4][C] = (mean, invstd, scale, shift) with
 *                                 scale = invstd * gamma, shift = fma(-mean, scale, beta): bit for bit what
 *                                 ursa_bn_relu_fwd_f32's two-launch form computes; no y. ws: URSA_BN_WS_FLOATS(C) floats.
 *  ursa_preact_conv1x1_f32        y = conv1x1(relu(fma(x, scale, shift))) - K12's forward launch applying K6's own expression to
 *                                 the rows of x as it stages them (bn_save: the block above): the bits of ursa_bn_relu_fwd_f32
 *                                 followed by ursa_conv1x1_f32, with x read once and relu(bn(x)) never written
 *                                 (at [1024, 64, 32, 32]: 536 MB less traffic per unit). Shapes: ursa_conv1x1_f32's, forward only.
 *  ursa_preact_wgrad1x1_partial_f32   the first launch of the layer's weight gradient (ursa_conv_wgrad_partial_f32 with ksize 1,
 *                                 stride 1) recomputing relu(fma(x, scale, shift)) the same way; same scratch, same slices, K7's
 *                                 second launch unchanged.
 * The backward of the BatchNorm itself stays ursa_bn_relu_bwd_f32 (it needs x and the saved block only).
 */
int ursa_bn_stats_f32(const float* x, const float* addend /* or NULL */, float* z_out /* iff addend */, const float* gamma,
                      const float* beta, float* running_mean /* or NULL */, float* running_var, float* save /* [4][C] */, float* ws,
                      int64_t N, int64_t C, int64_t HW, float eps, float momentum, ursa_stream_t stream);
int ursa_preact_conv1x1_supported(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W);
int ursa_preact_conv1x1_f32(const float* x, const float* bn_save, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout,
                            int64_t H, int64_t W, ursa_stream_t stream);
int ursa_preact_wgrad1x1_partial_f32(const float* x, const float* bn_save, const float* dy, float* ws, int64_t ws_floats, int64_t N,
                                     int64_t Cin, int64_t Cout, int64_t H, int64_t W, ursa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K14  the backward of `conv1x1(relu(bn(x)))` where the layer narrows (a Bottleneck block's conv1: 64 -> 16, 128 -> 32, 256 -> 64,
 *      64 -> 32, 128 -> 64): input gradient of the convolution + the BatchNorm's backward without the convolution's input gradient
 *      dh in memory. K6's backward needs the whole batch's two sums before it can produce any dx, and dh is the widest tensor of
 *      the unit (4 x dy): instead of writing it once and reading it twice, the memory-bound flipped GEMM runs twice.
 *      (dy: [N, Cd, H, W]; w: the layer's [Cd, Cx, 1, 1]; x: the BatchNorm's input [N, Cx, H, W]; bn_save: ursa_bn_stats_f32's block)
 *  ursa_preact_conv1x1_bwd_nl        the workgroups of these launches for a shape = the partial sums per channel; 0 = not covered.
 *  ursa_preact_conv1x1_bwd_sums_f32  dh = conv^T(dy, w) in registers, the ReLU gate from fma(x, scale, shift) > 0, and per workgroup
 *                                    (sum g, sum g * (x - mean)) in double -> out_partial [Cx][nl]: K6's first backward launch's sums.
 *  ursa_bn_bwd_coef_f32              their merge (fixed order) -> coef [3][Cx] = (gm, kk, gamma), dgamma, dbeta - the scalars of
 *                                    ursa_bn_relu_bwd_f32's second launch.
 *  ursa_preact_conv1x1_bwd_dx_f32    dh again (same bits), gated, and K6's expression
 *                                    dx = (((g - gm) - (x - mean) * kk) * invstd) * gamma (+ dz: the gradient reaching the residual
 *                                    sum on its other path), stored.
 * Traffic at [1024, 64, 32, 32] <- 16 channels: 335 + 871 MB against 335 (K12) + 536 + 1,072 (K6's two launches).
 */
int64_t ursa_preact_conv1x1_bwd_nl(int64_t N, int64_t Cd, int64_t Cx, int64_t H, int64_t W);
int ursa_preact_conv1x1_bwd_sums_f32(const float* dy, const float* w, const float* x, const float* bn_save, double* out_partial,
                                     int64_t N, int64_t Cd, int64_t Cx, int64_t H, int64_t W, ursa_stream_t stream);
int ursa_bn_bwd_coef_f32(const double* partial, int64_t nl, const float* bn_save, const float* gamma, float* coef /* [3][C] */,
                         float* dgamma, float* dbeta, int64_t n_per_channel, int64_t C, ursa_stream_t stream);
int ursa_preact_conv1x1_bwd_dx_f32(const float* dy, const float* w, const float* x, const float* bn_save, const float* coef,
                                   const float* dz /* or NULL */, float* dx, int64_t N, int64_t Cd, int64_t Cx, int64_t H, int64_t W,
                                   ursa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K10  the pre-activation unit of the BasicBlock ResNets, one launch each way      URSABench/models/preresnet.py:33-52
 *      (`out = bn1(x); relu; conv1; bn2; relu; conv2; out += residual`) and the backward of those ops in `loss.backward()`
 *      (URSABench/inference/sghmc.py:80).
 *
 * ursa_preact_conv3x3_f32 is K8's launch (same arithmetic, same shapes) with up to two things folded in, chosen by flags
 * (| URSA_CONV_FLIP / URSA_CONV_STRIDE2 as in K8):
 *   URSA_PREACT_BN     the tensor convolved is relu(bn(x)) in training mode: x's batch statistics arrive as per-channel partial
 *                      sums in double (in_partial [Cin][in_nl][2] = (sum x, sum x^2), in_nl <= 16: the out_partial of the
 *                      launch that produced x); every workgroup merges them (K6's tree: the doubles K6's own merge gives) and
 *                      finishes as K6 does - mean_c = (float)(s1/n), invstd_c = (float)(1/sqrt(s2/n - mean^2 + eps)), alpha_c =
 *                      invstd_c * gamma_c, beta'_c = fmaf(-mean_c, alpha_c, beta_c), staged value = max(fmaf(x, alpha_c,
 *                      beta'_c), 0) (NaN stays), padding stays zero - so relu(bn(x)) is never stored. One workgroup also writes
 *                      bn_save [4][Cin] = mean, invstd, alpha, beta' (what the backward launches take) and, if given, updates
 *                      running_mean / running_var exactly as ursa_bn_relu_fwd_f32 does.
 *   URSA_PREACT_STATS  (forward forms; required there) out_partial [Cout][nl][2] receives per-channel (sum, sum of squares)
 *                      of what is stored to y, in double, as nl <= 16 partial sums: the in_partial of the next unit / of
 *                      ursa_bn_apply_f32. With URSA_PREACT_ADD what is stored (and summed) is y + aux (`out += residual`;
 *                      aux: y's shape) - one fp32 add per element, as torch's.
 *   URSA_PREACT_BNBWD  (with URSA_CONV_FLIP; required there) the launch is the input gradient of the convolution AND the
 *                      first half of the backward of the BatchNorm + ReLU in front of it: aux = that BatchNorm's input (the
 *                      result's shape), aux_bn_save = its bn_save; stored to y: g = fmaf(aux, alpha_c, beta'_c) > 0 ? dh : 0
 *                      (the forward's gate recomputed from its saved scalars, as K6's backward); out_partial [Cout'][nl][2] =
 *                      (sum g, sum g * (aux - mean_c)) in double, one pair per workgroup. ursa_bn_bwd_dx_f32 finishes: dx, dgamma, dbeta.
 * Sums are deterministic (a fixed summation order whatever the arrival order; no atomics on the data):
 *   forward forms: the grid.x workgroups of a channel are cut into <= 16 lines of consecutive indices; every workgroup takes a
 *     ticket on its line's counter when it STARTS, stores its sums to its own slot of `scratch` and exits - except the one that
 *     drew the line's last ticket: it knows every other workgroup of the line is running (a running workgroup of this launch
 *     never waits for anything, so waiting for THEIR sums cannot starve, whatever else shares the device), polls their slots,
 *     adds them in ascending order with its own, stores the line's partial sum, zeroes the slots and re-arms the counter.
 *     `scratch` (ursa_preact_geometry()[1] bytes, 128-byte aligned) must be ZERO before its first launch and is zero again when a
 *     launch has drained: zero it once, then reuse it for the same layer (never for two launches that may overlap). A poll that
 *     runs out (never in a correct run) raises the error word in scratch (uint32 at byte offset ursa_preact_geometry()[3]) and
 *     makes the sums NaN.
 *   input-gradient forms (BNBWD): no hand-over inside the launch - every workgroup stores its sums at out_partial[channel][its
 *     grid.x index] (nl = workgroups per channel, up to a few hundred; scratch may be NULL) and ursa_bn_bwd_dx_f32, which is per
 *     channel anyway, adds its channel's nl partial sums in ascending order.
 * Shapes (any N, H = W): forward stride 1 (Cin, Cout, H): (3, 16, 32) without BN / ADD (the stem); (16, 16, 32), (32, 32, 16),
 * (64, 64, 8) with BN, with or without ADD; forward stride 2: (16, 32, 32), (32, 64, 16) with BN; flipped (+ BNBWD): the five
 * input-gradient forms K8 covers. ursa_preact_geometry: out[0] = nl, out[1] = scratch bytes, out[2] = workgroups per channel,
 * out[3] = byte offset of the error word in scratch; URSA_EVALUE = not covered.
 * ursa_preact_wgrad_partial_f32: K7's first launch with x = max(fmaf(x, alpha, beta'), 0) taken while the tile is staged
 * (bn_save of the BatchNorm in front of the layer); same scratch size, same second launch (ursa_conv_wgrad_reduce_f32).
 * ursa_bn_apply_f32 (nl <= 64) / ursa_bn_bwd_dx_f32 (nl <= 65,536): K6's second launches alone, fed by such partial sums: the BatchNorm that
 * ends the network, and the `dx` half of every BatchNorm backward. g is already gated. save: [4][C] as bn_save.
 * Traffic per unit and direction: the convolution's own (x + y + w) + 4 B x elements of aux; K6's 12 / 20 B per element are gone.
 */
#define URSA_PREACT_BN      0x10u
#define URSA_PREACT_STATS   0x20u
#define URSA_PREACT_ADD     0x40u
#define URSA_PREACT_BNBWD   0x80u
#define URSA_PREACT_EVAL    0x100u  /* with URSA_PREACT_BN [| URSA_PREACT_ADD | URSA_CONV_STRIDE2]: EVALUATION mode - the staged tensor is
                                       max(fmaf(x, alpha_c, beta'_c), 0) with alpha_c = gamma_c / sqrtf(running_var_c + eps) and beta'_c =
                                       fmaf(-running_mean_c, alpha_c, beta_c) (ursa_bn_relu_eval_f32's expressions, bit for bit); nothing is
                                       saved, no sums are taken (out_partial / scratch / in_partial / bn_save may be NULL), the running
                                       statistics are only read. One launch per bn -> relu -> conv unit of an ensemble member's forward
                                       (URSABench/tasks/prediction.py:56-58 `model(x)` in eval mode) instead of K6's launch + the convolution */
#define URSA_PREACT_ALLFLAGS 0x1F3u
int ursa_preact_geometry(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, uint32_t flags, int64_t* out /* [4] */);
int ursa_preact_conv3x3_f32(const float* x, const float* w, float* y,
                            const double* in_partial, int32_t in_nl, const float* gamma, const float* beta,
                            float* running_mean /* or NULL */, float* running_var /* or NULL */, float* bn_save /* [4][Cin] */,
                            float eps, float momentum,
                            const float* aux /* ADD: addend; BNBWD: the BatchNorm input */, const float* aux_bn_save /* BNBWD */,
                            double* out_partial, void* scratch, int64_t scratch_bytes,
                            int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, uint32_t flags, ursa_stream_t stream);
int ursa_preact_wgrad_partial_f32(const float* x, const float* bn_save, const float* dy, float* ws, int64_t ws_floats,
                                  int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int32_t stride, ursa_stream_t stream);
/* The two convolutions of a unit's backward pass in ONE launch: ursa_preact_conv3x3_f32(URSA_CONV_FLIP | URSA_PREACT_BNBWD) and
 * ursa_preact_wgrad_partial_f32 read the same output gradient dy and depend on nothing of each other; here their workgroups
 * alternate inside one grid (even index: input gradient, odd: weight gradient). The same workgroup programs on the same operands:
 * g, out_partial and ws hold exactly what the two launches would have left (bit for bit); one launch's fixed cost instead of two,
 * and on the 16-channel layers the two kinds of workgroups run side by side on a CU.
 * dy: [N, Cd, H, W]; w: the layer's [Cd, Cx, 3, 3]; x: the input of the BatchNorm in front of the layer, [N, Cx, sH, sW] (s = 2 with
 * URSA_CONV_STRIDE2); bn_save: that BatchNorm's [4][Cx]; g: [N, Cx, sH, sW]; out_partial: [Cx][nl][2] doubles, nl as
 * ursa_preact_geometry(N, Cd, Cx, H, W, FLIP | BNBWD | stride) says; ws: ursa_conv_wgrad_ws_floats(N, Cx, Cd, sH, sW, 3, s) floats. */
int ursa_preact_bwd_pair_f32(const float* dy, const float* w, float* g, const float* x, const float* bn_save, double* out_partial,
                             float* ws, int64_t ws_floats, int64_t N, int64_t Cd, int64_t Cx, int64_t H, int64_t W,
                             uint32_t flags /* URSA_CONV_STRIDE2 */, ursa_stream_t stream);
int ursa_bn_apply_f32(const float* x, float* y, const double* partial, int32_t nl, const float* gamma, const float* beta,
                      float* running_mean /* or NULL */, float* running_var /* or NULL */, float* save /* [4][C] */,
                      int64_t N, int64_t C, int64_t HW, float eps, float momentum, uint32_t flags /* URSA_BN_RELU */,
                      ursa_stream_t stream);
int ursa_bn_bwd_dx_f32(const float* x, const float* g, const float* dz /* or NULL */, float* dx, const float* gamma,
                       const float* save /* [4][C] */, const double* partial, int32_t nl, float* dgamma, float* dbeta,
                       int64_t N, int64_t C, int64_t HW, ursa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K11  the head of a training step      URSABench/models/preresnet.py:146-150 (`x = self.bn(x); x = self.relu(x); x = self.avgpool(x);
 *      x = x.view(x.size(0), -1); x = self.fc(x)`), the samplers' loss nn.CrossEntropyLoss() (URSABench/inference/sghmc.py:38-40,
 *      76-77) and the backward of all of it in `loss.backward()` (sghmc.py:80): stock PyTorch-ROCm runs ~13 launches of ~5 us on
 *      128 x 64 x 8 x 8 numbers; here three.
 *   ursa_bn_relu_pool_f32      pooled[n][c] = mean over the 8 x 8 map of max(fmaf(z, alpha_c, beta'_c), 0): training-mode BatchNorm (its
 *                              statistics from `partial`, the out_partial of the K10 launch that produced z; scalars, `save` [4][C] and
 *                              running statistics exactly as ursa_bn_apply_f32) + ReLU + AvgPool2d(8) without storing the activation.
 *                              HW must be 64 (URSA_EVALUE otherwise: the caller keeps ursa_bn_apply_f32 + its own pooling). fp32 sums
 *                              in a fixed tree; the division by 64 is exact.
 *   ursa_fc_ce_f32             ONE workgroup: logits = pooled W^T + b, loss = mean cross entropy over the rows whose target is not
 *                              ignore_index, and the gradients dW [K][C], db [K], dpooled [N][C] of that loss (grad_output 1). logits
 *                              may be NULL. fma chains in ascending index order; expf / logf of the device library (<= 1 ulp).
 *                              Covered: N <= 256, K <= 16, K*C <= 2048, N*C <= 8192 (ursa_fc_ce_supported). A target outside [0, K)
 *                              other than ignore_index makes the loss and the gradients NaN (torch: device-side assert).
 *   ursa_bn_relu_pool_bwd_f32  the backward of ursa_bn_relu_pool_f32: dz, dgamma, dbeta from dpooled (the gradient of every position of
 *                              map (n, c) is dpooled[n][c] / 64 where the forward's gate was open; then K6's backward arithmetic,
 *                              one workgroup per channel, the channel in registers: N <= 128, HW = 64).
 * Results are within fp32 rounding of the stock sequence (different summation trees), not bit-equal to it; oracle twins:
 * oracle_bn_relu_pool_f32, oracle_fc_ce_f32, oracle_bn_relu_pool_bwd_f32 (sums in double, rounded once).
 */
int ursa_bn_relu_pool_f32(const float* z, const double* partial, int32_t nl, const float* gamma, const float* beta,
                          float* running_mean /* or NULL */, float* running_var /* or NULL */, float* save /* [4][C] */,
                          float* pooled /* [N][C] */, int64_t N, int64_t C, int64_t HW, float eps, float momentum, ursa_stream_t stream);
int ursa_fc_ce_supported(int64_t N, int64_t C, int64_t K);
int ursa_fc_ce_f32(const float* pooled, const float* W, const float* b /* or NULL */, const int64_t* target, float* loss /* [1] */,
                   float* logits /* [N][K] or NULL */, float* dW, float* db /* iff b */, float* dpooled, int64_t N, int64_t C, int64_t K,
                   int64_t ignore_index, ursa_stream_t stream);
int ursa_bn_relu_pool_bwd_f32(const float* z, const float* dpooled, const float* gamma, const float* save /* [4][C] */, float* dz,
                              float* dgamma, float* dbeta, int64_t N, int64_t C, int64_t HW, ursa_stream_t stream);

/* ------------------------------------------------------------------------------------ */
int ursa_abi_version(void);
const char* ursa_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* URSA_HIP_H */
