// ursa_head.hip — K11: the classifier end of a training step as ONE launch: `self.fc(x)` (URSABench/models/preresnet.py:149-150),
// the cross entropy of the samplers' loss (`model_loss = 'multi_class_linear_output'` -> nn.CrossEntropyLoss(), URSABench/
// inference/sghmc.py:38-40,76-77: mean over the batch) AND the backward of both (the `loss.backward()` of sghmc.py:80 as far as
// the pooled features): stock PyTorch-ROCm runs these as 3 rocBLAS GEMMs, log_softmax / nll_loss forward and backward, a bias
// reduction and two fills - ten launches of ~5 us on 128 x 64 x 10 numbers. Here: one workgroup of 1,024 threads, everything in LDS.
//
//   logits[n][k] = b[k] + sum_c p[n][c] W[k][c]                                  (fma chain over c, ascending)
//   lse_n = m_n + log(sum_k exp(logits[n][k] - m_n)),  m_n = max_k logits[n][k]
//   loss = (sum over rows with target != ignore_index of (lse_n - logits[n][target_n])) / count       (rows summed in a fixed tree)
//   dlogits[n][k] = (exp(logits[n][k] - m_n) / sum_n - [k == target_n]) / count  (0 for ignored rows)
//   dW[k][c] = sum_n dlogits[n][k] p[n][c] ; db[k] = sum_n dlogits[n][k] ; dp[n][c] = sum_k dlogits[n][k] W[k][c]   (fma chains, ascending)
// A target outside [0, K) that is not ignore_index makes the loss NaN (torch raises a device-side assert there).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ursa_hip.h"

namespace {

constexpr int kHeadThreads = 1024;   // one workgroup, 16 waves: the whole head lives in one CU's LDS
constexpr int kHeadMaxNC = 8192;     // floats of the pooled features held in LDS
constexpr int kHeadMaxK = 16;
constexpr int kHeadMaxKC = 2048;     // floats of the weight matrix held in LDS
constexpr int kHeadMaxN = 256;

// LDS layouts are chosen per phase so that a wave either reads ONE address (broadcast) or consecutive banks:
//   ps[n][c] at a row pitch of C + 1 (rows n, n + 1, ... of one column sit in different banks), ws[k][c], ls[k][n].
__global__ __launch_bounds__(kHeadThreads) void k_fc_ce(const float* __restrict__ p, const float* __restrict__ W, const float* __restrict__ b,
                                                        const int64_t* __restrict__ target, float* __restrict__ loss,
                                                        float* __restrict__ logits_out, float* __restrict__ dW, float* __restrict__ db,
                                                        float* __restrict__ dp, int N, int C, int K, int64_t ignore_index)
{
    __shared__ float ps[kHeadMaxNC + kHeadMaxN];         // [N][C + 1]
    __shared__ float ws[kHeadMaxKC];                     // [K][C]
    __shared__ float ls[kHeadMaxN * kHeadMaxK];          // logits, then dlogits: [K][N]
    __shared__ float red[kHeadThreads / 64][3];
    __shared__ float rowmax[kHeadMaxN], rowtgt[kHeadMaxN];
    __shared__ float sh_inv;
    const int tid = threadIdx.x, P = C + 1;
    for (int i = tid; i < N * C; i += kHeadThreads) ps[(i / C) * P + i % C] = p[i];
    for (int i = tid; i < K * C; i += kHeadThreads) ws[i] = W[i];
    __syncthreads();
    for (int i = tid; i < K * N; i += kHeadThreads) {      // a wave: one k (weights broadcast), 64 consecutive rows n
        const int k = i / N, n = i % N;
        float acc = b ? b[k] : 0.f;
        const float* pr = ps + n * P;
        const float* wr = ws + k * C;
        for (int c = 0; c < C; ++c) acc = fmaf(pr[c], wr[c], acc);
        ls[i] = acc;
        if (logits_out) logits_out[n * K + k] = acc;
    }
    __syncthreads();
    // log-sum-exp with the exponentials spread over all threads: per row the maximum and the target's logit (one thread per row), per
    // (class, row) e = exp(logit - max) (one thread each), per row the sum, its logarithm and the row's loss; dlogits = e / sum - onehot
    // before the 1 / count factor (rows that do not count: zeros)
    for (int n = tid; n < N; n += kHeadThreads) {
        const int64_t t = target[n];
        float m = ls[n];
        for (int k = 1; k < K; ++k) m = fmaxf(m, ls[k * N + n]);
        rowmax[n] = m;
        const bool ok = t != ignore_index && t >= 0 && t < K;
        rowtgt[n] = ok ? ls[(int)t * N + n] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < K * N; i += kHeadThreads) ls[i] = expf(ls[i] - rowmax[i % N]);
    __syncthreads();
    float my_loss = 0.f, my_cnt = 0.f, my_bad = 0.f;
    for (int n = tid; n < N; n += kHeadThreads) {
        const int64_t t = target[n];
        const bool valid = t != ignore_index;
        const bool bad = valid && (t < 0 || t >= K);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += ls[k * N + n];
        if (valid && !bad) my_loss += (rowmax[n] + logf(s)) - rowtgt[n];
        my_cnt += valid ? 1.f : 0.f;
        my_bad += bad ? 1.f : 0.f;
        rowmax[n] = (valid && !bad) ? 1.0f / s : 0.f;          // (reused: the row's 1 / sum, 0 for rows that do not count)
    }
    __syncthreads();
    for (int i = tid; i < K * N; i += kHeadThreads) {
        const int k = i / N, n = i % N;
        const float r = rowmax[n];
        ls[i] = r != 0.f ? ls[i] * r - ((int64_t)k == target[n] ? 1.f : 0.f) : 0.f;
    }
    // rows are summed in a fixed order: the wave's shuffle tree, then the waves in ascending order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        my_loss += __shfl_xor(my_loss, off);
        my_cnt += __shfl_xor(my_cnt, off);
        my_bad += __shfl_xor(my_bad, off);
    }
    if ((tid & 63) == 0) { red[tid >> 6][0] = my_loss; red[tid >> 6][1] = my_cnt; red[tid >> 6][2] = my_bad; }
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f, cnt = 0.f, bad = 0.f;
        for (int w = 0; w < kHeadThreads / 64; ++w) { tot += red[w][0]; cnt += red[w][1]; bad += red[w][2]; }
        const float inv = bad != 0.f ? __builtin_nanf("") : 1.0f / cnt;     // (no row counts: 0 / 0 = NaN, as torch)
        loss[0] = tot * inv;
        sh_inv = inv;
    }
    __syncthreads();
    const float inv = sh_inv;
    for (int i = tid; i < K * N; i += kHeadThreads) ls[i] = ls[i] * inv;
    __syncthreads();
    for (int i = tid; i < K * C; i += kHeadThreads) {      // dW[k][c]: a wave: one k (dlogits broadcast), consecutive c
        const int k = i / C, c = i % C;
        float acc = 0.f;
        const float* lr = ls + k * N;
        for (int n = 0; n < N; ++n) acc = fmaf(lr[n], ps[n * P + c], acc);
        dW[i] = acc;
    }
    if (db && tid >= kHeadThreads - 64) {                   // the last wave: db[k] = sum_n dlogits[n][k]
        for (int k = tid - (kHeadThreads - 64); k < K; k += 64) {
            float acc = 0.f;
            for (int n = 0; n < N; ++n) acc += ls[k * N + n];
            db[k] = acc;
        }
    }
    for (int i = tid; i < N * C; i += kHeadThreads) {      // dp[n][c]
        const int n = i / C, c = i % C;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(ls[k * N + n], ws[k * C + c], acc);
        dp[i] = acc;
    }
}

}  // namespace

extern "C" int ursa_fc_ce_supported(int64_t N, int64_t C, int64_t K) {
    return N >= 1 && N <= kHeadMaxN && K >= 1 && K <= kHeadMaxK && C >= 1 && K * C <= kHeadMaxKC && N * C <= kHeadMaxNC;
}

extern "C" int ursa_fc_ce_f32(const float* pooled, const float* W, const float* b, const int64_t* target, float* loss, float* logits,
                              float* dW, float* db, float* dpooled, int64_t N, int64_t C, int64_t K, int64_t ignore_index,
                              ursa_stream_t stream) {
    if (!pooled || !W || !target || !loss || !dW || !dpooled) return URSA_ENULL;
    if ((b == nullptr) != (db == nullptr)) return URSA_ENULL;
    if (N < 1 || C < 1 || K < 1) return URSA_ESIZE;
    if (!ursa_fc_ce_supported(N, C, K)) return URSA_EVALUE;
    if (((uintptr_t)pooled | (uintptr_t)W | (uintptr_t)b | (uintptr_t)loss | (uintptr_t)logits | (uintptr_t)dW | (uintptr_t)db | (uintptr_t)dpooled) & 3 ||
        (uintptr_t)target & 7)
        return URSA_EALIGN;
    hipLaunchKernelGGL(k_fc_ce, dim3(1), dim3(kHeadThreads), 0, (hipStream_t)stream, pooled, W, b, target, loss, logits, dW, db, dpooled, (int)N,
                       (int)C, (int)K, ignore_index);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? URSA_OK : (int)e;
}
