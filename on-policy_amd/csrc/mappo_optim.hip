// K13: gradient clipping + Adam for one network as two launches (hidden work of a PPO update that is not a network):
// the reference clips each network's gradients to a global L2 norm and takes one Adam step per minibatch
// (onpolicy/algorithms/r_mappo/r_mappo.py:146-167: clip_grad_norm_(max_grad_norm) or get_gard_norm, optimizer.step();
// rMAPPOPolicy.py:31-37: torch.optim.Adam(lr, eps = opti_eps, weight_decay)).  In PyTorch that is ~8 launches per
// network (foreach norms, stack, norm, clamp, multiply, the fused Adam kernel, the step counters) -- 16 of the ~70 tiny
// launches of an update, each ~4.5 us on an idle queue: 3 % of a step on one rank's shard of an 8-GPU north-star job and
// 10 % of the 25 ms config-2 step.  Here: one launch sums the squares (and advances the step counters), one applies
// clip coefficient + Adam to every tensor of the network.  Same arithmetic as torch.optim.Adam (no amsgrad, no
// maximize) on float32 tensors; the optimiser's own state tensors (exp_avg, exp_avg_sq, step) are updated in place, so
// torch.optim.Adam.state_dict() / lr schedules keep working.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mappo_hip.h"
#include "mappo_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 256;          // Adam: the second launch folds the first one's partials in ONE block
constexpr int kMaxSumBlocks = 1024;      // ValueNorm / minibatch sums (13 M elements per call at the north star)

struct Desc {
    mappo_adam_t a;
    long long total;
};

__device__ __forceinline__ float block_sum(float v, float* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < kThreads / 64; ++w) t += sh[w];
    return t;       // valid on thread 0
}

// prefix[t] = elements of tensors 0 .. t - 1, in LDS; -> tensor of flat element e
__device__ __forceinline__ int tensor_of(const long long* prefix, int n, long long e) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (prefix[mid] <= e) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ void __launch_bounds__(kThreads) adam_norm_kernel(Desc d) {
    __shared__ long long prefix[MAPPO_ADAM_MAX_TENSORS + 1];
    __shared__ float sh[kThreads / 64];
    if (threadIdx.x == 0) {
        long long s = 0;
        for (int t = 0; t < d.a.n; ++t) {
            prefix[t] = s;
            s += d.a.numel[t];
        }
        prefix[d.a.n] = s;
    }
    __syncthreads();
    const long long per = (d.total + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < d.total ? lo + per : d.total;
    float s = 0.f;
    for (long long e = lo + threadIdx.x; e < hi; e += kThreads) {
        const int t = tensor_of(prefix, d.a.n, e);
        const float g = d.a.grad[t][e - prefix[t]];
        s += g * g;
    }
    const float tot = block_sum(s, sh);
    if (threadIdx.x == 0) d.a.workspace[blockIdx.x] = tot;
    // the step counters advance here, so that every block of the second launch reads the new value
    if (blockIdx.x == 0 && threadIdx.x < d.a.n) d.a.step[threadIdx.x][0] += 1.f;
}

__global__ void __launch_bounds__(kThreads) adam_step_kernel(Desc d, int n_partials) {
    __shared__ long long prefix[MAPPO_ADAM_MAX_TENSORS + 1];
    __shared__ float sh[kThreads / 64];
    __shared__ float coef_s;
    // per tensor: step size lr / (1 - beta1^step) and sqrt(1 - beta2^step) (two double pow() per ELEMENT made this launch
    // 30 us for a 40 k-parameter network)
    __shared__ float step_size_s[MAPPO_ADAM_MAX_TENSORS], sqrt_bc2_s[MAPPO_ADAM_MAX_TENSORS];
    if (threadIdx.x == 0) {
        long long s = 0;
        for (int t = 0; t < d.a.n; ++t) {
            prefix[t] = s;
            s += d.a.numel[t];
        }
        prefix[d.a.n] = s;
    }
    if (threadIdx.x < d.a.n) {
        const double step = (double)d.a.step[threadIdx.x][0];
        const float bc1 = (float)(1.0 - pow(d.a.beta1, step)), bc2 = (float)(1.0 - pow(d.a.beta2, step));
        const double lr = d.a.lr_device != nullptr ? d.a.lr_device[0] : d.a.lr;      // (device copy: a captured launch follows lr_decay)
        step_size_s[threadIdx.x] = (float)(lr / bc1);
        sqrt_bc2_s[threadIdx.x] = sqrtf(bc2);
    }
    float p = threadIdx.x < n_partials ? d.a.workspace[threadIdx.x] : 0.f;      // (n_partials <= kThreads, fixed order)
    __syncthreads();
    const float sq = block_sum(p, sh);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(sq);
        float c = 1.f;
        if (d.a.max_grad_norm > 0.0) {                  // torch.nn.utils.clip_grad_norm_: max_norm / (total + 1e-6), <= 1
            c = (float)d.a.max_grad_norm / (norm + 1e-6f);
            c = c > 1.f ? 1.f : c;                      // (NaN stays NaN, like torch.clamp: a non-finite norm poisons every gradient)
        }
        coef_s = c;
        if (blockIdx.x == 0 && d.a.grad_norm != nullptr) d.a.grad_norm[0] = norm;
    }
    __syncthreads();
    const float coef = coef_s;
    const long long per = (d.total + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < d.total ? lo + per : d.total;
    for (long long e = lo + threadIdx.x; e < hi; e += kThreads) {
        const int t = tensor_of(prefix, d.a.n, e);
        const long long i = e - prefix[t];
        // (torch's fused Adam kernel: the hyper-parameters are doubles, so the products with them are formed in double and
        // rounded to float32 once)
        float g = d.a.grad[t][i] * coef;
        d.a.grad[t][i] = g;                             // the clipped gradient stays in .grad, as clip_grad_norm_ leaves it
        float w = d.a.param[t][i];
        if (d.a.weight_decay != 0.0) g = (float)(g + w * d.a.weight_decay);
        float m = d.a.exp_avg[t][i], v = d.a.exp_avg_sq[t][i];
        m = (float)(d.a.beta1 * m + (1.0 - d.a.beta1) * g);
        v = (float)(d.a.beta2 * v + (1.0 - d.a.beta2) * g * g);
        d.a.exp_avg[t][i] = m;
        d.a.exp_avg_sq[t][i] = v;
        const float denom = (float)(sqrtf(v) / sqrt_bc2_s[t] + d.a.eps);
        d.a.param[t][i] = w - step_size_s[t] * m / denom;
    }
}

// ---------------------------------------------------------------- ValueNorm update + de-normalisation scalars ----
// ValueNorm.update (onpolicy/utils/valuenorm.py:39-55: batch mean and mean of squares folded into the debiased running
// moments with weight beta) followed by running_mean_var / the [sigma, mu] pair the loss and the GAE scan read
// (valuenorm.py:32-37): ~17 launches and an n-element temporary (x ** 2) in PyTorch per minibatch.  Launch 1: per-block
// float64 partial sums of x and x^2; launch 2 (one block): batch moments -> the three statistics in place -> [sigma, mu].
__global__ void __launch_bounds__(kThreads) vn_sums_kernel(const float* x, long long n, double* partials) {
    __shared__ double sh[2][kThreads / 64];
    double s = 0.0, q = 0.0;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const double v = x[i];
        s += v;
        q += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off);
        q += __shfl_down(q, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wave] = s;
        sh[1][wave] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) {
            a += sh[0][w];
            b += sh[1][w];
        }
        partials[2 * blockIdx.x] = a;
        partials[2 * blockIdx.x + 1] = b;
    }
}

__global__ void vn_fold_kernel(const double* partials, int n_partials, double count, const float* batch_moments,
                               double weight, float eps, float* m1, float* m2, float* d, float* denorm) {
    float mean, mean_sq;
    if (batch_moments != nullptr) {         // data parallel: the all-reduced moments of the global minibatch
        mean = batch_moments[0];
        mean_sq = batch_moments[1];
    } else {                                // (one wave: lane l adds partials l, l + 64, ..., then the lanes are folded)
        double a = 0.0, b = 0.0;
        for (int i = threadIdx.x; i < n_partials; i += 64) {
            a += partials[2 * i];
            b += partials[2 * i + 1];
        }
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off);
            b += __shfl_down(b, off);
        }
        mean = (float)(a / count);
        mean_sq = (float)(b / count);
    }
    if (threadIdx.x != 0) return;
    // m <- w m + (1 - w) E[.] in float32 with the Python float weight, like the in-place tensor ops of the reference
    const float w = (float)weight, u = (float)(1.0 - weight);
    const float n1 = m1[0] * w + mean * u, n2 = m2[0] * w + mean_sq * u, nd = d[0] * w + u;
    m1[0] = n1;
    m2[0] = n2;
    d[0] = nd;
    const float dd = nd > eps ? nd : eps;
    const float mu = n1 / dd;
    float var = n2 / dd - mu * mu;
    var = var > 1e-2f ? var : 1e-2f;
    denorm[0] = sqrtf(var);
    denorm[1] = mu;
}

// ---------------------------------------------------------------- what a minibatch needs before its loss is formed ----
// Denominators of the masked means (r_mappo.py:135-139 policy loss, :84-87 value loss: sum of active_masks, or the row
// count) and the batch moments of the returns for the ValueNorm / PopArt update (r_mappo.py:65) -- in a data-parallel job the
// GLOBAL ones, so the four sums are all-reduced between the two halves below.  In PyTorch this was ~13 launches per update
// (27 with the data-parallel bookkeeping): at an 8-GPU shard of the north star 3 % of the step.
// Launch 1 + 2: float64 sums of active_masks, returns, returns^2 (per block, then in block order) -> [sum active, rows,
// sum returns, sum returns^2].
__global__ void __launch_bounds__(kThreads) mb_sums_kernel(const float* active, const float* returns, long long n,
                                                           double* partials) {
    __shared__ double sh[3][kThreads / 64];
    double sa = 0.0, s = 0.0, q = 0.0;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const double v = returns[i];
        sa += (double)active[i];
        s += v;
        q += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        sa += __shfl_down(sa, off);
        s += __shfl_down(s, off);
        q += __shfl_down(q, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wave] = sa;
        sh[1][wave] = s;
        sh[2][wave] = q;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double a = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) a += sh[threadIdx.x][w];
        partials[3 * blockIdx.x + threadIdx.x] = a;
    }
}
// one wave per sum: lane l adds partials l, l + 64, ..., then the lanes are folded (fixed order; one thread walking the
// partials one load at a time took 29 us)
__global__ void mb_sums_finish_kernel(const double* partials, int n_partials, double rows, double* sums) {
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;        // q < 3
    double a = 0.0;
    for (int i = lane; i < n_partials; i += 64) a += partials[3 * i + q];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if (lane == 0) sums[q == 0 ? 0 : q + 1] = a;
    if (threadIdx.x == 0) sums[1] = rows;
}
// Launch 3: out = [1 / global policy denominator, 1 / global value denominator,
//                  1 / local policy denominator (twice), 1 / local value denominator, 1 / local rows,
//                  mean, mean of squares of the returns over the global minibatch]
__global__ void mb_scales_kernel(const double* local, const double* global, int policy_masked, int value_masked, float* out) {
    if (threadIdx.x != 0) return;
    const double gp = policy_masked ? global[0] : global[1], gv = value_masked ? global[0] : global[1];
    const double lp = policy_masked ? local[0] : local[1], lv = value_masked ? local[0] : local[1];
    out[0] = (float)(1.0 / gp);
    out[1] = (float)(1.0 / gv);
    out[2] = (float)(1.0 / lp);
    out[3] = (float)(1.0 / lp);
    out[4] = (float)(1.0 / lv);
    out[5] = (float)(1.0 / local[1]);
    out[6] = (float)(global[2] / global[1]);
    out[7] = (float)(global[3] / global[1]);
}

// ---------------------------------------------------------------- input LayerNorm folded into the first Linear ----
// A trunk that reads standardised rows x^ (mappo_standardize_rows) evaluates Linear(LayerNorm(x)) as x^ W'^T + b' with
// W'[f][k] = W[f][k] gamma[k], b'[f] = b[f] + sum_k W[f][k] beta[k] (mlp.py:47-48 feature_norm, :20 fc1).  Forward: one
// block per output feature writes its row of W' (zero columns up to ld) and b'[f].  Backward: one thread per input
// column k walks the 64 features: dW[f][k] = dW'[f][k] gamma[k] + db'[f] beta[k], dgamma[k] = sum_f dW'[f][k] W[f][k],
// dbeta[k] = sum_f db'[f] W[f][k]; db = db'.  (As tensor ops: 3 launches forward, 7 backward, per network and update.)
__global__ void __launch_bounds__(kThreads) fold_fwd_kernel(const float* W, const float* b, const float* gamma, const float* beta,
                                                            int out_f, int din, int ld, float* Wf, float* bf) {
    __shared__ float sh[kThreads / 64];
    const int f = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < ld; k += kThreads) {
        float w = 0.f;
        if (k < din) {
            w = W[(long long)f * din + k];
            s += w * beta[k];
            w *= gamma[k];
        }
        Wf[(long long)f * ld + k] = w;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < kThreads / 64; ++w) t += sh[w];
        bf[f] = b[f] + t;
    }
}
__global__ void __launch_bounds__(kThreads) fold_bwd_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ dWf,
                                                            const float* __restrict__ dbf, int out_f, int din, int ld,
                                                            float* __restrict__ dW, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= din) return;
    const float g = gamma[k], be = beta[k];
    float sg = 0.f, sb = 0.f;
#pragma unroll 8
    for (int f = 0; f < out_f; ++f) {       // (no aliasing + unrolled: the loads of eight features in flight, sums in order)
        const float w = W[(long long)f * din + k], dwf = dWf[(long long)f * ld + k], d = dbf[f];
        dW[(long long)f * din + k] = dwf * g + d * be;
        sg += dwf * w;
        sb += d * w;
    }
    dgamma[k] = sg;
    dbeta[k] = sb;
}

}  // namespace

extern "C" int mappo_fold_input_norm_forward(const float* w, const float* b, const float* gamma, const float* beta, int out_features,
                                             int din, int ld, float* w_folded, float* b_folded, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!w || !b || !gamma || !beta || !w_folded || !b_folded) return MAPPO_E_NULL;
    if (out_features <= 0 || din <= 0 || ld < din) return MAPPO_E_SHAPE;
    hipLaunchKernelGGL(fold_fwd_kernel, dim3((unsigned)out_features), dim3(kThreads), 0, stream, w, b, gamma, beta, out_features,
                       din, ld, w_folded, b_folded);
    return (int)hipGetLastError();
}

extern "C" int mappo_fold_input_norm_backward(const float* w, const float* gamma, const float* beta, const float* dw_folded,
                                              const float* db_folded, int out_features, int din, int ld, float* dw,
                                              float* dgamma, float* dbeta, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!w || !gamma || !beta || !dw_folded || !db_folded || !dw || !dgamma || !dbeta) return MAPPO_E_NULL;
    if (out_features <= 0 || din <= 0 || ld < din) return MAPPO_E_SHAPE;
    hipLaunchKernelGGL(fold_bwd_kernel, dim3((unsigned)((din + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, w, gamma, beta,
                       dw_folded, db_folded, out_features, din, ld, dw, dgamma, dbeta);
    return (int)hipGetLastError();
}

extern "C" int64_t mappo_minibatch_sums_workspace_doubles(void) { return 3 * kMaxSumBlocks; }

extern "C" int mappo_minibatch_sums(const float* active_masks, const float* returns, int64_t n, double* sums,
                                    double* workspace, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!active_masks || !returns || !sums || !workspace) return MAPPO_E_NULL;
    if (n <= 0) return MAPPO_E_SHAPE;
    long long blocks = (n + kThreads * 16 - 1) / (kThreads * 16);
    if (blocks > kMaxSumBlocks) blocks = kMaxSumBlocks;
    hipLaunchKernelGGL(mb_sums_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, active_masks, returns, (long long)n,
                       workspace);
    hipLaunchKernelGGL(mb_sums_finish_kernel, dim3(1), dim3(192), 0, stream, (const double*)workspace, (int)blocks, (double)n,
                       sums);
    return (int)hipGetLastError();
}

extern "C" int mappo_minibatch_scales(const double* local_sums, const double* global_sums, int policy_masked, int value_masked,
                                      float* out, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!local_sums || !global_sums || !out) return MAPPO_E_NULL;
    hipLaunchKernelGGL(mb_scales_kernel, dim3(1), dim3(64), 0, stream, local_sums, global_sums, policy_masked, value_masked, out);
    return (int)hipGetLastError();
}

extern "C" int64_t mappo_valuenorm_workspace_doubles(void) { return 2 * kMaxSumBlocks; }

extern "C" int mappo_valuenorm_update(const float* x, int64_t n, const float* batch_moments, double weight, float eps,
                                      float* running_mean, float* running_mean_sq, float* debiasing_term, float* denorm,
                                      double* workspace, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!running_mean || !running_mean_sq || !debiasing_term || !denorm) return MAPPO_E_NULL;
    if (!batch_moments && (!x || !workspace)) return MAPPO_E_NULL;
    if (!batch_moments && n <= 0) return MAPPO_E_SHAPE;
    long long blocks = 0;
    if (!batch_moments) {
        blocks = (n + kThreads * 16 - 1) / (kThreads * 16);
        if (blocks > kMaxSumBlocks) blocks = kMaxSumBlocks;
        hipLaunchKernelGGL(vn_sums_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, x, (long long)n, workspace);
    }
    hipLaunchKernelGGL(vn_fold_kernel, dim3(1), dim3(64), 0, stream, (const double*)workspace, (int)blocks, (double)n,
                       batch_moments, weight, eps, running_mean, running_mean_sq, debiasing_term, denorm);
    return (int)hipGetLastError();
}

extern "C" int64_t mappo_adam_workspace_floats(void) { return kMaxBlocks; }

extern "C" int mappo_clip_adam(const mappo_adam_t* a, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!a || !a->workspace) return MAPPO_E_NULL;
    if (a->n <= 0 || a->n > MAPPO_ADAM_MAX_TENSORS) return MAPPO_E_SHAPE;
    Desc d;
    d.a = *a;
    d.total = 0;
    for (int t = 0; t < a->n; ++t) {
        if (!a->param[t] || !a->grad[t] || !a->exp_avg[t] || !a->exp_avg_sq[t] || !a->step[t]) return MAPPO_E_NULL;
        if (a->numel[t] <= 0) return MAPPO_E_SHAPE;
        d.total += a->numel[t];
    }
    long long blocks = (d.total + kThreads * 8 - 1) / (kThreads * 8);
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipLaunchKernelGGL(adam_norm_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, d);
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, d, (int)blocks);
    return (int)hipGetLastError();
}
