// K7: the clipped-surrogate PPO loss of R_MAPPO.ppo_update for a Discrete action head, forward and
// backward in one pass over the minibatch rows (reference onpolicy/algorithms/r_mappo/r_mappo.py:52-89
// cal_value_loss, :119-153 policy loss / entropy; utils/distributions.py FixedCategorical;
// utils/act.py:115-170 evaluate_actions).
//
// In the framework the loss is ~100 elementwise / reduction launches over [rows, 1] tensors per
// minibatch span, each latency-bound.  Every term is a per-row function whose normalising denominators
// (number of rows, sum of the active masks) are known before the forward pass, so the gradient of the
// total loss w.r.t. the head's logits and the critic's values can be written in the same pass that
// evaluates the loss: one thread per row, ~100 B of traffic per row.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mappo_hip.h"
#include "mappo_internal.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// huber_loss / mse_loss of onpolicy/utils/util.py:16-23 and their derivatives w.r.t. the error
__device__ __forceinline__ float loss_of(float e, float delta, bool huber) {
    if (!huber) return e * e / 2.f;
    float ae = fabsf(e);
    return ae <= delta ? e * e / 2.f : delta * (ae - delta / 2.f);
}
__device__ __forceinline__ float dloss_of(float e, float delta, bool huber) {
    if (!huber) return e;
    float ae = fabsf(e);
    return ae <= delta ? e : (e > 0.f ? delta : -delta);
}

// STAGED: the [256, n_actions] logits / availability tile of a workgroup goes through LDS with coalesced
// global accesses (a thread's own row is n_actions floats at a stride of n_actions: 20-byte pieces for the
// 5 actions of MPE), and the gradient tile goes back the same way.  Rows sit at an odd word stride in LDS.
// RK > 0 (heads of <= RK actions -- MPE's 5): a row's masked logits and probabilities stay in registers between the passes,
// one exponential per action instead of three (round 4: at 5 actions the launch was instruction-bound -- 15 expf and 10
// integer divisions per row -- at 0.37 of the HBM rate).
template <bool HAS_AVAIL, bool STAGED, int RK>
__global__ void __launch_bounds__(256) ppo_loss_kernel(mappo_ppo_loss_t a) {
    extern __shared__ float lds[];
    const bool huber = a.flags & MAPPO_LOSS_HUBER;
    const bool clipped_value = a.flags & MAPPO_LOSS_CLIPPED_VALUE;
    const bool p_active = a.flags & MAPPO_LOSS_POLICY_ACTIVE_MASKS;
    const bool v_active = a.flags & MAPPO_LOSS_VALUE_ACTIVE_MASKS;
    const int na = a.n_actions;
    const float inv_dp = a.inv_denoms[0], inv_dv = a.inv_denoms[1];
    float nmean = 0.f, nstd = 1.f;
    if (a.norm != nullptr) {  // {sigma, mu}
        nstd = a.norm[0];
        nmean = a.norm[1];
    }
    double s_policy = 0.0, s_entropy = 0.0, s_value = 0.0, s_ratio = 0.0;
    const int stride = na | 1;
    float* s_lg = lds;
    float* s_av = lds + 256 * stride;
    // element e = threadIdx.x + 256 j of a staged tile is (row, action) = (e / na, e % na): stepped, not divided
    const int r0 = threadIdx.x / na, k0 = threadIdx.x - r0 * na, dr = 256 / na, dk = 256 - dr * na;

    for (long long base = (long long)blockIdx.x * 256; base < a.rows; base += (long long)gridDim.x * 256) {
        const long long i = base + threadIdx.x;
        const bool live = i < a.rows;
        if (STAGED && a.logits != nullptr) {
            const long long left = a.rows - base;
            const int tile = (int)(left < 256 ? left : 256) * na;
            const float* gl = a.logits + base * na;
            const float* ga = HAS_AVAIL ? a.available + base * na : nullptr;
            for (int e = threadIdx.x, r = r0, k = k0; e < tile; e += 256) {
                s_lg[r * stride + k] = gl[e];
                if (HAS_AVAIL) s_av[r * stride + k] = ga[e];
                r += dr;
                k += dk;
                if (k >= na) {
                    k -= na;
                    ++r;
                }
            }
            __syncthreads();
        }
        const float am = (live && a.active != nullptr) ? a.active[i] : 1.f;
        // ---------------- actor: log-softmax over the (masked) logits, entropy, surrogate
        if (live && a.logits != nullptr) {
            const float* lg = STAGED ? s_lg + threadIdx.x * stride : a.logits + i * na;
            const float* av = !HAS_AVAIL ? nullptr : STAGED ? s_av + threadIdx.x * stride : a.available + i * na;
            constexpr int NR = RK > 0 ? RK : 1;
            float lr[NR], pr[NR];           // RK > 0: masked logits / probabilities of this row
            float mx = -INFINITY, lse;
            const int act = (int)a.actions[i];
            float ent = 0.f, logp_a = 0.f;
            if (RK > 0) {
#pragma unroll
                for (int k = 0; k < RK; ++k) {
                    lr[k] = k < na ? ((HAS_AVAIL && av[k] == 0.f) ? -1e10f : lg[k]) : -INFINITY;
                    mx = fmaxf(mx, lr[k]);
                }
                float se = 0.f;
#pragma unroll
                for (int k = 0; k < RK; ++k) {
                    pr[k] = k < na ? expf(lr[k] - mx) : 0.f;
                    se += pr[k];
                }
                lse = mx + logf(se);
                const float inv_se = 1.f / se;
#pragma unroll
                for (int k = 0; k < RK; ++k) {
                    if (k < na) {
                        const float lp = lr[k] - lse;
                        pr[k] *= inv_se;
                        ent -= pr[k] * lp;
                        if (k == act) logp_a = lp;
                    }
                }
            } else {
                for (int k = 0; k < na; ++k) {
                    float l = (HAS_AVAIL && av[k] == 0.f) ? -1e10f : lg[k];  // distributions.py: masked logits
                    mx = fmaxf(mx, l);
                }
                float se = 0.f;
                for (int k = 0; k < na; ++k) {
                    float l = (HAS_AVAIL && av[k] == 0.f) ? -1e10f : lg[k];
                    se += expf(l - mx);
                }
                lse = mx + logf(se);
                for (int k = 0; k < na; ++k) {
                    float l = (HAS_AVAIL && av[k] == 0.f) ? -1e10f : lg[k];
                    float lp = l - lse;
                    float p = expf(lp);
                    ent -= p * lp;
                    if (k == act) logp_a = lp;
                }
            }
            const float ratio = expf(logp_a - a.old_logp[i]);  // r_mappo.py:129
            const float adv = a.adv[i];
            const float lo = 1.f - a.clip, hi = 1.f + a.clip;
            const float surr1 = ratio * adv;
            const float surr2 = fminf(fmaxf(ratio, lo), hi) * adv;
            float s = fminf(surr1, surr2);
            const bool inside = ratio >= lo && ratio <= hi;
            float ds_dratio = (inside || surr1 < surr2) ? adv : 0.f;  // min / clamp sub-gradients as autograd takes them
            if (a.factor != nullptr) {                                 // happo_trainer.py:137-141
                s *= a.factor[i];
                ds_dratio *= a.factor[i];
            }
            const float wp = p_active ? am : 1.f;
            s_policy += (double)(-s * wp);
            s_entropy += (double)(ent * wp);
            s_ratio += (double)ratio;
            if (a.dlogits != nullptr) {
                // d/dl_j of [ -s - c_ent * H ] * wp / D_p;  dlogp_a/dl_j = [j == a] - p_j,  dH/dl_j = -p_j (logp_j + H)
                const float g_logp = -ds_dratio * ratio;
                const float scale = wp * inv_dp;
                float* dl = STAGED ? s_lg + threadIdx.x * stride : a.dlogits + i * na;  // in place: read, then written
                if (RK > 0) {
#pragma unroll
                    for (int k = 0; k < RK; ++k) {
                        if (k < na) {
                            const bool masked = HAS_AVAIL && av[k] == 0.f;
                            const float lp = lr[k] - lse;
                            const float g = g_logp * ((k == act ? 1.f : 0.f) - pr[k]) + a.entropy_coef * pr[k] * (lp + ent);
                            dl[k] = masked ? 0.f : g * scale;
                        }
                    }
                } else
                for (int k = 0; k < na; ++k) {
                    bool masked = HAS_AVAIL && av[k] == 0.f;
                    float l = masked ? -1e10f : lg[k];
                    float lp = l - lse;
                    float p = expf(lp);
                    float g = g_logp * ((k == act ? 1.f : 0.f) - p) + a.entropy_coef * p * (lp + ent);
                    dl[k] = masked ? 0.f : g * scale;  // torch.where routes no gradient to masked logits
                }
            }
        }
        if (STAGED && a.logits != nullptr && a.dlogits != nullptr) {
            __syncthreads();
            const long long left = a.rows - base;
            const int tile = (int)(left < 256 ? left : 256) * na;
            float* gd = a.dlogits + base * na;
            for (int e = threadIdx.x, r = r0, k = k0; e < tile; e += 256) {
                gd[e] = s_lg[r * stride + k];
                r += dr;
                k += dk;
                if (k >= na) {
                    k -= na;
                    ++r;
                }
            }
        }
        if (STAGED && a.logits != nullptr) __syncthreads();   // the tile is free for the next iteration
        // ---------------- critic: clipped value loss against the (normalised) return
        if (live && a.values != nullptr) {
            const float v = a.values[i], vp = a.value_preds[i];
            const float target = a.norm != nullptr ? (a.returns[i] - nmean) / nstd : a.returns[i];
            const float d = v - vp;
            const float vpc = vp + fminf(fmaxf(d, -a.clip), a.clip);  // r_mappo.py:62-63
            const float e_c = target - vpc, e_o = target - v;
            const float l_c = loss_of(e_c, a.huber_delta, huber), l_o = loss_of(e_o, a.huber_delta, huber);
            const float g_o = -dloss_of(e_o, a.huber_delta, huber);
            const float g_c = (d >= -a.clip && d <= a.clip) ? -dloss_of(e_c, a.huber_delta, huber) : 0.f;
            float vl = l_o, g = g_o;
            if (clipped_value) {  // torch.max: the larger branch, both halves on a tie
                if (l_c > l_o) {
                    vl = l_c;
                    g = g_c;
                } else if (l_c == l_o) {
                    g = 0.5f * (g_o + g_c);
                }
            }
            const float wv = v_active ? am : 1.f;
            s_value += (double)(vl * wv);
            if (a.dvalues != nullptr) a.dvalues[i] = a.value_loss_coef * g * wv * inv_dv;
        }
    }

    __shared__ double red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s_policy = wave_sum_d(s_policy);
    s_entropy = wave_sum_d(s_entropy);
    s_value = wave_sum_d(s_value);
    s_ratio = wave_sum_d(s_ratio);
    if (lane == 0) {
        red[wave][0] = s_policy;
        red[wave][1] = s_entropy;
        red[wave][2] = s_value;
        red[wave][3] = s_ratio;
    }
    __syncthreads();
    if (threadIdx.x < 4 && a.sums != nullptr) {
        double t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(a.sums + threadIdx.x, t);
    }
}

}  // namespace

extern "C" int mappo_ppo_loss_f32(const mappo_ppo_loss_t* args, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!args) return MAPPO_E_NULL;
    mappo_ppo_loss_t a = *args;
    if (!a.inv_denoms) return MAPPO_E_NULL;
    if (!a.logits && !a.values) return MAPPO_E_NULL;
    if (a.logits && (!a.actions || !a.old_logp || !a.adv)) return MAPPO_E_NULL;
    if (a.values && (!a.value_preds || !a.returns)) return MAPPO_E_NULL;
    if (a.rows <= 0 || (a.logits && a.n_actions <= 0)) return MAPPO_E_SHAPE;
    if (a.flags & ~15u) return MAPPO_E_FLAGS;
    long long blocks = (a.rows + 255) / 256;
    if (blocks > mappo::kCUs * 8) blocks = mappo::kCUs * 8;
    const bool avail = a.logits && a.available;
    const size_t lds = a.logits ? (size_t)(avail ? 2 : 1) * 256 * (a.n_actions | 1) * sizeof(float) : 0;
    // Staged through LDS up to 150 KB (gfx950: 160 KB per CU; above 64 KB the kernel has to be granted it): Hanabi's 48
    // actions need 100 KB for the two [256, 49] tiles.  Unstaged, a thread walks its own 192-byte row and the launch ran at
    // 80 GB/s (4.9 ms per 683 k-row span, 6.5 % of the Hanabi-shaped step, profiles/r02_bench_hanabi_kernel_stats.csv).
    const bool staged = a.logits && lds <= 150 * 1024;
    dim3 grid((unsigned)blocks), block(256);
    if (staged) {
        if (lds > 48 * 1024) {
            const void* fn = avail ? reinterpret_cast<const void*>(&ppo_loss_kernel<true, true, 0>)
                                   : reinterpret_cast<const void*>(&ppo_loss_kernel<false, true, 0>);
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        const bool small = a.n_actions <= 8;
        if (avail && small) hipLaunchKernelGGL((ppo_loss_kernel<true, true, 8>), grid, block, lds, stream, a);
        else if (avail) hipLaunchKernelGGL((ppo_loss_kernel<true, true, 0>), grid, block, lds, stream, a);
        else if (small) hipLaunchKernelGGL((ppo_loss_kernel<false, true, 8>), grid, block, lds, stream, a);
        else hipLaunchKernelGGL((ppo_loss_kernel<false, true, 0>), grid, block, lds, stream, a);
    } else {
        if (avail) hipLaunchKernelGGL((ppo_loss_kernel<true, false, 0>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((ppo_loss_kernel<false, false, 0>), grid, block, 0, stream, a);
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// K14: the rollout side of the Categorical head as one launch -- availability masking, normalised logits, one action per
// row and its log-probability (reference onpolicy/algorithms/utils/distributions.py:14-28 FixedCategorical.sample /
// log_probs, :55-68 Categorical.forward; act.py:44-60).  As framework ops that is ~15 launches on [rows, n_actions]
// tensors (where / logsumexp / softmax / multinomial with its asserts / gather), 4-5 us each inside the captured rollout
// step.  Sampling is torch.multinomial's own rule for one draw: argmax_i p_i / q_i with q ~ Exponential(1) -- the noise is
// drawn by the caller from torch's generator (graph-safe philox), so the distribution is exactly the framework's.
// One thread per row; n_actions <= 64.
__global__ void __launch_bounds__(256) categorical_sample_kernel(const float* logits, const float* avail, const float* noise,
                                                                 long long* actions, float* logp, long long rows, int na) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* lg = logits + r * na;
    const float* av = avail ? avail + r * na : nullptr;
    const float* q = noise + r * na;
    float mx = -INFINITY;
    for (int i = 0; i < na; ++i) {
        const float x = (av && av[i] == 0.f) ? -1e10f : lg[i];
        mx = fmaxf(mx, x);
    }
    float se = 0.f;
    for (int i = 0; i < na; ++i) {
        const float x = (av && av[i] == 0.f) ? -1e10f : lg[i];
        se += expf(x - mx);
    }
    const float lse = mx + logf(se);
    int best = 0;
    float best_v = -1.f, best_l = 0.f;
    for (int i = 0; i < na; ++i) {
        const float x = (av && av[i] == 0.f) ? -1e10f : lg[i];
        const float l = x - lse;                // normalised logit = log p_i
        const float v = expf(l) / q[i];         // p_i / q_i (first maximum wins, like argmax)
        if (v > best_v) {
            best_v = v;
            best = i;
            best_l = l;
        }
    }
    actions[r] = best;
    logp[r] = best_l;
}

extern "C" int mappo_categorical_sample(const float* logits, const float* available, const float* noise, int64_t* actions,
                                        float* log_probs, int64_t rows, int n_actions, mappo_stream_t stream_) {
    if (!logits || !noise || !actions || !log_probs) return MAPPO_E_NULL;
    if (rows <= 0 || n_actions <= 0 || n_actions > 64) return MAPPO_E_SHAPE;
    hipLaunchKernelGGL(categorical_sample_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), logits, available, noise,
                       reinterpret_cast<long long*>(actions), log_probs, (long long)rows, n_actions);
    return (int)hipGetLastError();
}
