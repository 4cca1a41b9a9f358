// K1 -- GAE / discounted-return scan over the time-major rollout buffer, for gfx950.
//
// Replaces SharedReplayBuffer.compute_returns (reference onpolicy/utils/shared_buffer.py:179-262)
// and the advantage line of R_MAPPO.train (onpolicy/algorithms/r_mappo/r_mappo.py:179-182).
//
// The recurrence g_t = delta_t + (gamma*lambda*m_{t+1}) * g_{t+1} is serial along t and
// independent across the C = N*A columns, and the buffer is time-major ([T, C] rows), so
// lanes run along columns (coalesced rows) and every lane walks t = T-1 .. 0 with its
// carry in a register.  The arithmetic is the reference's float32 operation order with FMA
// contraction disabled => bit-identical to numpy.
//
// The scan is HBM-bound (16..24 B per (t, column) element, ~6 flops), and at the
// north-star size there are only C = 32768 columns = 512 wavefronts for 256 CUs, so
// bandwidth comes from memory-level parallelism per wave, not from occupancy:
//   * "strip" kernels: a workgroup owns a strip of W columns; all its lanes stream
//     [TC rows x W cols] tiles of every input field with 16-byte loads into registers
//     (the prefetch stage), drop them into LDS (the transpose stage: load lanes are
//     (row, 4-col group), compute lanes are columns), and one wave walks the tile
//     backwards in time out of LDS while the next tile's loads are in flight.
//   * "column" kernel: one lane per column with plain 4-byte loads; used for any C / any
//     alignment and for the non-GAE modes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mappo_hip.h"
#include "mappo_internal.h"

#pragma clang fp contract(off)

namespace {

struct GaeArgs {
    const float* rewards;
    float* value_preds;
    const float* next_value;
    const float* masks;
    const float* bad;
    float* returns;
    const float* denorm;
    float* adv;
    const float* active;
    double* partials;
    long long partial_rows;  // rows of `partials`; rows >= gridDim.x are zero-filled by block 0
    int T;
    long long C;
    float gamma;
    float gl;  // float32(gamma_f64 * lambda_f64), shared_buffer.py:239
};

// One backward step of the reference's GAE branches (shared_buffer.py:192-203 PTL,
// :236-240 / :255-258 no PTL).  dv1 = D(v_{t+1}) carried from the previous step.
template <bool PTL, bool DENORM>
__device__ __forceinline__ float gae_step(float r, float v0, float m1, float bad1, float sigma,
                                          float mu, float gamma, float gl, float& dv1, float& g,
                                          float& dv0_out) {
    float dv0 = v0;
    if (DENORM) {
        float s = v0 * sigma;  // valuenorm.py:75: x * sqrt(var) + mean, two roundings
        dv0 = s + mu;
    }
    float gv = gamma * dv1;
    float gvm = gv * m1;
    float rp = r + gvm;
    float delta = rp - dv0;
    float carry;
    if (PTL && DENORM) {
        float x = gl * g;  // shared_buffer.py:195: gamma*lambda*gae*mask
        carry = x * m1;
    } else {
        float x = gl * m1;  // shared_buffer.py:201,239,257: gamma*lambda*mask*gae
        carry = x * g;
    }
    g = delta + carry;
    if (PTL) g = g * bad1;
    float ret = g + dv0;
    dv1 = dv0;
    dv0_out = dv0;
    return ret;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}


// Rows of the partials array that no workgroup owns are zeroed by workgroup 0, so that the
// fixed-size reduction needs no separate memset launch.
__device__ __forceinline__ void zero_unowned_partials(double* partials, long long rows) {
    if (partials != nullptr && blockIdx.x == 0) {
        for (long long r = (long long)gridDim.x + threadIdx.x; r < rows; r += blockDim.x) {
            partials[r * 3 + 0] = 0.0;
            partials[r * 3 + 1] = 0.0;
            partials[r * 3 + 2] = 0.0;
        }
    }
}

// ------------------------------------------------------------------ strip kernel ----
// W      columns per strip (multiple of 4, <= 64)
// NWAVES waves per workgroup (wave 0 walks the recurrence; LOADERS selects who streams)
// TC     time steps per tile
// NBUF   register prefetch depth in tiles (1 or 2): tiles k+1 .. k+NBUF are in flight while
//        tile k is walked
// WLOAD  whether the walker wave also issues loads (false: waves 1.. are pure producers)
template <int W, int NWAVES, int TC, int NBUF, bool WLOAD, bool PTL, bool DENORM>
__global__ void __launch_bounds__(NWAVES * 64) gae_strip_kernel(GaeArgs a) {
    constexpr int NT = NWAVES * 64;
    constexpr int NL = WLOAD ? NT : NT - 64;    // loader threads
    constexpr int V = W / 4;                    // float4 per tile row
    constexpr int NVEC = TC * V;                // float4 per field per tile
    constexpr int PER = (NVEC + NL - 1) / NL;   // float4 per loader thread per field
    constexpr int NFMAX = 5;                    // r, v, m, bad, active
    static_assert(W % 4 == 0 && W <= 64, "strip width");
    static_assert(WLOAD || NWAVES > 1, "pure-walker needs producer waves");
    static_assert(NBUF == 1 || NBUF == 2, "prefetch depth");

    extern __shared__ float4 lds4[];            // [slots][TC][V], slots = fields in use
    float* ldsf = reinterpret_cast<float*>(lds4);

    const int tid = threadIdx.x;
    const long long col0 = (long long)blockIdx.x * W;
    const long long C = a.C;
    const int T = a.T;
    const bool has_act = a.active != nullptr;
    const bool has_adv = a.adv != nullptr;
    const bool loader = WLOAD || tid >= 64;
    const int ltid = WLOAD ? tid : tid - 64;    // index among loader threads

    // field f: base pointer (masks / bad_masks are read at row t+1) and LDS slot
    const float* fbase[NFMAX] = {a.rewards, a.value_preds, a.masks + C, PTL ? a.bad + C : nullptr,
                                 a.active};
    const int act_slot = PTL ? 4 : 3;

    float4 pre[NBUF][NFMAX][PER];

#define MAPPO_LOAD_TILE(B, TBASE)                                                               \
    if (loader) {                                                                               \
        _Pragma("unroll") for (int f = 0; f < NFMAX; ++f) {                                     \
            if (f == 3 && !PTL) continue;                                                       \
            if (f == 4 && !has_act) continue;                                                   \
            _Pragma("unroll") for (int p = 0; p < PER; ++p) {                                   \
                int i = ltid + p * NL;                                                          \
                int row = i / V;                                                                \
                int c4 = i - row * V;                                                           \
                int t = (TBASE) + row;                                                          \
                long long lc = col0 + c4 * 4;                                                   \
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);                                   \
                if (i < NVEC && t >= 0 && lc < C)                                               \
                    val = *reinterpret_cast<const float4*>(fbase[f] + (long long)t * C + lc);   \
                pre[B][f][p] = val;                                                             \
            }                                                                                   \
        }                                                                                       \
    }
#define MAPPO_STASH_TILE(B)                                                                     \
    if (loader) {                                                                               \
        _Pragma("unroll") for (int f = 0; f < NFMAX; ++f) {                                     \
            if (f == 3 && !PTL) continue;                                                       \
            if (f == 4 && !has_act) continue;                                                   \
            const int slot = f == 4 ? act_slot : f;                                             \
            _Pragma("unroll") for (int p = 0; p < PER; ++p) {                                   \
                int i = ltid + p * NL;                                                          \
                if (i < NVEC) lds4[slot * NVEC + i] = pre[B][f][p];                             \
            }                                                                                   \
        }                                                                                       \
    }

    const int lane = tid;  // only wave 0 walks: lane == tid there
    const long long col = col0 + lane;
    const bool walker = (tid < 64);
    const bool live = walker && lane < W && col < C;

    float sigma = 1.f, mu = 0.f;
    if (DENORM) {
        sigma = a.denorm[0];
        mu = a.denorm[1];
    }
    const float gamma = a.gamma, gl = a.gl;

    float g = 0.f, dv1 = 0.f;
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    if (live) {
        float nv = a.next_value[col];
        a.value_preds[(long long)T * C + col] = nv;  // shared_buffer.py:187,218
        dv1 = nv;
        if (DENORM) {
            float s = nv * sigma;
            dv1 = s + mu;
        }
    }

    const int nch = (T + TC - 1) / TC;
    MAPPO_LOAD_TILE(0, T - TC)
    if (NBUF == 2) {
        if (nch > 1) MAPPO_LOAD_TILE(NBUF - 1, T - 2 * TC)
    }
    for (int k0 = 0; k0 < nch; k0 += NBUF) {
#pragma unroll
        for (int b = 0; b < NBUF; ++b) {
            const int k = k0 + b;
            if (k >= nch) break;
            const int tbase = T - (k + 1) * TC;
            MAPPO_STASH_TILE(b)
            __syncthreads();
            if (k + NBUF < nch) MAPPO_LOAD_TILE(b, tbase - NBUF * TC)  // in flight during the walk(s)
            if (walker) {
                const int lc = lane < W ? lane : 0;  // lanes beyond the strip stay in bounds
                const float* lr = ldsf + 0 * NVEC * 4 + lc;
                const float* lv = ldsf + 1 * NVEC * 4 + lc;
                const float* lm = ldsf + 2 * NVEC * 4 + lc;
                const float* lb = ldsf + 3 * NVEC * 4 + lc;
                const float* la = ldsf + act_slot * NVEC * 4 + lc;
                const int slo = tbase < 0 ? -tbase : 0;
#pragma unroll 8
                for (int s = TC - 1; s >= slo; --s) {
                    float r = lr[s * W], v0 = lv[s * W], m1 = lm[s * W];
                    float bad1 = PTL ? lb[s * W] : 1.f;
                    float dv0;
                    float ret = gae_step<PTL, DENORM>(r, v0, m1, bad1, sigma, mu, gamma, gl, dv1, g, dv0);
                    if (live) {
                        long long o = (long long)(tbase + s) * C + col;
                        a.returns[o] = ret;
                        if (has_adv) {
                            float adv = ret - dv0;  // r_mappo.py:180 (from the rounded return)
                            a.adv[o] = adv;
                            float am = has_act ? la[s * W] : 1.f;
                            if (am != 0.f) {
                                double d = (double)adv;
                                s1 += d;
                                s2 += d * d;
                                cnt += 1.0;
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
#undef MAPPO_LOAD_TILE
#undef MAPPO_STASH_TILE

    zero_unowned_partials(a.partials, a.partial_rows);
    if (walker && a.partials != nullptr) {
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        cnt = wave_sum(cnt);
        if (lane == 0) {
            double* p = a.partials + (long long)blockIdx.x * 3;
            p[0] = s1;
            p[1] = s2;
            p[2] = cnt;
        }
    }
}

// ----------------------------------------------------------------- column kernel ----
// One lane per column, any C, all seven reference branches.
template <bool GAE, bool PTL, bool DENORM>
__global__ void __launch_bounds__(64) gae_column_kernel(GaeArgs a) {
    const long long C = a.C;
    const int T = a.T;
    const long long col = (long long)blockIdx.x * 64 + threadIdx.x;
    const bool live = col < C;
    const bool has_act = a.active != nullptr;
    const bool has_adv = a.adv != nullptr;

    float sigma = 1.f, mu = 0.f;
    if (DENORM) {
        sigma = a.denorm[0];
        mu = a.denorm[1];
    }
    const float gamma = a.gamma, gl = a.gl;
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;

    if (live) {
        float nv = a.next_value[col];
        float g = 0.f, dv1 = 0.f, ret1 = nv;
        if (GAE) {
            a.value_preds[(long long)T * C + col] = nv;
            dv1 = nv;
            if (DENORM) {
                float s = nv * sigma;
                dv1 = s + mu;
            }
        } else {
            a.returns[(long long)T * C + col] = nv;  // shared_buffer.py:205,260
        }
        constexpr int U = 4;
        for (int thi = T - 1; thi >= 0; thi -= U) {
            float r[U], v0[U], m1[U], bad1[U], am[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int t = thi - u;
                bool ok = t >= 0;
                long long o = (long long)(ok ? t : 0) * C + col;
                r[u] = a.rewards[o];
                v0[u] = a.value_preds[o];
                m1[u] = a.masks[o + C];
                bad1[u] = PTL ? a.bad[o + C] : 1.f;
                am[u] = has_act ? a.active[o] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int t = thi - u;
                if (t < 0) break;
                long long o = (long long)t * C + col;
                float ret, dv0;
                if (GAE) {
                    ret = gae_step<PTL, DENORM>(r[u], v0[u], m1[u], bad1[u], sigma, mu, gamma, gl,
                                                dv1, g, dv0);
                } else {
                    dv0 = v0[u];
                    if (DENORM) {
                        float s = v0[u] * sigma;
                        dv0 = s + mu;
                    }
                    float x = ret1 * gamma;  // shared_buffer.py:208,213,262
                    float y = x * m1[u];
                    float z = y + r[u];
                    if (PTL) {
                        float zb = z * bad1[u];
                        float ob = 1.f - bad1[u];
                        float w = ob * dv0;
                        ret = zb + w;
                    } else {
                        ret = z;
                    }
                    ret1 = ret;
                }
                a.returns[o] = ret;
                if (has_adv) {
                    float adv = ret - dv0;
                    a.adv[o] = adv;
                    if (am[u] != 0.f) {
                        double d = (double)adv;
                        s1 += d;
                        s2 += d * d;
                        cnt += 1.0;
                    }
                }
            }
        }
    }
    zero_unowned_partials(a.partials, a.partial_rows);
    if (a.partials != nullptr) {
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        cnt = wave_sum(cnt);
        if (threadIdx.x == 0) {
            double* p = a.partials + (long long)blockIdx.x * 3;
            p[0] = s1;
            p[1] = s2;
            p[2] = cnt;
        }
    }
}

// ------------------------------------------------------- K5: moments and stats ----
__global__ void __launch_bounds__(256) adv_reduce_kernel(const double* partials, long long rows,
                                                          double* sums) {
    __shared__ double sh[3][256];
    double acc[3] = {0.0, 0.0, 0.0};
    for (long long r = threadIdx.x; r < rows; r += 256) {
        acc[0] += partials[r * 3 + 0];
        acc[1] += partials[r * 3 + 1];
        acc[2] += partials[r * 3 + 2];
    }
    for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride)
            for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + stride];
        __syncthreads();
    }
    if (threadIdx.x < 3) sums[threadIdx.x] = sh[threadIdx.x][0];
}

__global__ void adv_stats_kernel(const double* sums, float* stats) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double n = sums[2];
        double mean = sums[0] / n;  // n == 0 -> NaN, like np.nanmean of an all-NaN array
        double var = sums[1] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[0] = (float)mean;
        stats[1] = (float)sqrt(var);
    }
}

__global__ void __launch_bounds__(256) adv_normalize_kernel(const float* adv, const float* stats,
                                                             float* out, long long n) {
    const float mean = stats[0];
    const float den = stats[1] + 1e-5f;  // r_mappo.py:187
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long step = (long long)gridDim.x * 256;
    for (; i < n; i += step) out[i] = (adv[i] - mean) / den;
}

constexpr int kAdvBlocks = 1024;

// r_mappo.py:179-186 as one grid-stride pass: advantages + per-block partial moments.
__global__ void __launch_bounds__(256) advantages_kernel(const float* ret, const float* vp,
                                                          const float* denorm, const float* active,
                                                          float* adv, double* partials, long long n,
                                                          long long partial_rows) {
    zero_unowned_partials(partials, partial_rows);
    float sigma = 1.f, mu = 0.f;
    const bool dn = denorm != nullptr;
    if (dn) {
        sigma = denorm[0];
        mu = denorm[1];
    }
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    const long long step = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
        float dv = vp[i];
        if (dn) {
            float s = dv * sigma;
            dv = s + mu;
        }
        float a = ret[i] - dv;
        adv[i] = a;
        float am = active ? active[i] : 1.f;
        if (am != 0.f) {
            double d = (double)a;
            s1 += d;
            s2 += d * d;
            cnt += 1.0;
        }
    }
    __shared__ double sh[3][4];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    cnt = wave_sum(cnt);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh[0][w] = s1;
        sh[1][w] = s2;
        sh[2][w] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < 3 && partials)
        partials[(long long)blockIdx.x * 3 + threadIdx.x] =
            ((sh[threadIdx.x][0] + sh[threadIdx.x][1]) + sh[threadIdx.x][2]) + sh[threadIdx.x][3];
}

int g_variant = 0;

template <int W, int NWAVES, int TC, int NBUF = 1, bool WLOAD = true>
hipError_t launch_strip(const GaeArgs& a, unsigned flags, hipStream_t stream) {
    const bool ptl = flags & MAPPO_GAE_PROPER_TIME_LIMITS;
    const bool den = flags & MAPPO_GAE_DENORM;
    const int slots = 3 + (ptl ? 1 : 0) + (a.active ? 1 : 0);
    const size_t lds = (size_t)slots * TC * W * sizeof(float);
    dim3 grid((unsigned)((a.C + W - 1) / W)), block(NWAVES * 64);
    if (ptl && den)
        hipLaunchKernelGGL((gae_strip_kernel<W, NWAVES, TC, NBUF, WLOAD, true, true>), grid, block, lds, stream, a);
    else if (ptl)
        hipLaunchKernelGGL((gae_strip_kernel<W, NWAVES, TC, NBUF, WLOAD, true, false>), grid, block, lds, stream, a);
    else if (den)
        hipLaunchKernelGGL((gae_strip_kernel<W, NWAVES, TC, NBUF, WLOAD, false, true>), grid, block, lds, stream, a);
    else
        hipLaunchKernelGGL((gae_strip_kernel<W, NWAVES, TC, NBUF, WLOAD, false, false>), grid, block, lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_column(const GaeArgs& a, unsigned flags, hipStream_t stream) {
    const bool gae = flags & MAPPO_GAE_USE_GAE;
    const bool ptl = flags & MAPPO_GAE_PROPER_TIME_LIMITS;
    const bool den = flags & MAPPO_GAE_DENORM;
    dim3 grid((unsigned)((a.C + 63) / 64)), block(64);
#define MAPPO_COL(G, P, D) \
    hipLaunchKernelGGL((gae_column_kernel<G, P, D>), grid, block, 0, stream, a)
    if (gae) {
        if (ptl && den) MAPPO_COL(true, true, true);
        else if (ptl) MAPPO_COL(true, true, false);
        else if (den) MAPPO_COL(true, false, true);
        else MAPPO_COL(true, false, false);
    } else {
        if (ptl && den) MAPPO_COL(false, true, true);
        else if (ptl) MAPPO_COL(false, true, false);
        else if (den) MAPPO_COL(false, false, true);  // D() only enters the fused advantages here
        else MAPPO_COL(false, false, false);
    }
#undef MAPPO_COL
    return hipGetLastError();
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int64_t mappo_gae_partial_rows(int64_t C) {
    if (C <= 0) return 0;
    int64_t r = (C + 15) / 16;
    return r < kAdvBlocks ? kAdvBlocks : r;
}

extern "C" int mappo_advantages_f32(const float* returns, const float* value_preds,
                                    const float* denorm, const float* active_masks,
                                    float* advantages, double* adv_partials, int T, int64_t C,
                                    mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!returns || !value_preds || !advantages) return MAPPO_E_NULL;
    if (T <= 0 || C <= 0) return MAPPO_E_SHAPE;
    const long long n = (long long)T * C;
    long long blocks = (n + 256 * 8 - 1) / (256 * 8);
    if (blocks > kAdvBlocks) blocks = kAdvBlocks;
    hipLaunchKernelGGL(advantages_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, returns,
                       value_preds, denorm, active_masks, advantages, adv_partials, n,
                       (long long)mappo_gae_partial_rows(C));
    return (int)hipGetLastError();
}

extern "C" int mappo_gae_set_variant(int variant) {
    int old = g_variant;
    g_variant = variant;
    return old;
}

extern "C" int mappo_gae_f32(const float* rewards, float* value_preds, const float* next_value,
                             const float* masks, const float* bad_masks, float* returns,
                             const float* denorm, float* advantages, const float* active_masks,
                             double* adv_partials, int T, int64_t C, double gamma,
                             double gae_lambda, unsigned flags, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!rewards || !value_preds || !next_value || !masks || !returns) return MAPPO_E_NULL;
    if ((flags & MAPPO_GAE_PROPER_TIME_LIMITS) && !bad_masks) return MAPPO_E_NULL;
    if ((flags & MAPPO_GAE_DENORM) && !denorm) return MAPPO_E_NULL;
    if (flags & ~7u) return MAPPO_E_FLAGS;
    if (T <= 0 || C <= 0) return MAPPO_E_SHAPE;
    if (adv_partials && !advantages) return MAPPO_E_NULL;
    if (active_masks && !advantages) return MAPPO_E_NULL;
    const void* ptrs[] = {rewards, value_preds, next_value, masks, bad_masks, returns,
                          denorm,  advantages,  active_masks};
    for (const void* p : ptrs)
        if (p && (reinterpret_cast<uintptr_t>(p) & 3u)) return MAPPO_E_ALIGN;

    GaeArgs a;
    a.rewards = rewards;
    a.value_preds = value_preds;
    a.next_value = next_value;
    a.masks = masks;
    a.bad = (flags & MAPPO_GAE_PROPER_TIME_LIMITS) ? bad_masks : nullptr;
    a.returns = returns;
    a.denorm = (flags & MAPPO_GAE_DENORM) ? denorm : nullptr;
    a.adv = advantages;
    a.active = active_masks;
    a.partials = adv_partials;
    a.partial_rows = mappo_gae_partial_rows(C);
    a.T = T;
    a.C = C;
    a.gamma = (float)gamma;
    a.gl = (float)(gamma * gae_lambda);

    const bool strip_ok = (flags & MAPPO_GAE_USE_GAE) && (C % 4 == 0) && aligned16(rewards) &&
                          aligned16(value_preds) && aligned16(masks) &&
                          (!a.bad || aligned16(bad_masks)) && (!a.active || aligned16(active_masks));
    int variant = g_variant;
    if (!strip_ok) variant = 99;
    if (variant == 0) variant = (C >= 64 * 256) ? 2 : (C >= 32 * 256 ? 3 : 6);

    hipError_t e;
    switch (variant) {
        case 1: e = launch_strip<64, 1, 32>(a, flags, stream); break;
        case 2: e = launch_strip<64, 4, 32>(a, flags, stream); break;
        case 3: e = launch_strip<32, 1, 32>(a, flags, stream); break;
        case 5: e = launch_strip<64, 2, 32>(a, flags, stream); break;
        case 6: e = launch_strip<16, 1, 64>(a, flags, stream); break;
        case 10: e = launch_strip<64, 4, 32, 2>(a, flags, stream); break;
        case 11: e = launch_strip<64, 4, 64, 1>(a, flags, stream); break;
        case 12: e = launch_strip<64, 8, 64, 1>(a, flags, stream); break;
        case 13: e = launch_strip<64, 8, 32, 2>(a, flags, stream); break;
        case 14: e = launch_strip<64, 4, 16, 2>(a, flags, stream); break;
        case 15: e = launch_strip<64, 4, 32, 1, false>(a, flags, stream); break;
        case 16: e = launch_strip<64, 4, 32, 2, false>(a, flags, stream); break;
        case 17: e = launch_strip<64, 8, 64, 2, false>(a, flags, stream); break;
        case 18: e = launch_strip<32, 4, 64, 2>(a, flags, stream); break;
        case 19: e = launch_strip<64, 8, 64, 2>(a, flags, stream); break;
        default: e = launch_column(a, flags, stream); break;
    }
    return (int)e;
}

extern "C" int mappo_adv_reduce(const double* partials, int64_t rows, double* sums,
                                mappo_stream_t stream_) {
    if (!partials || !sums) return MAPPO_E_NULL;
    if (rows <= 0) return MAPPO_E_SHAPE;
    hipLaunchKernelGGL(adv_reduce_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       partials, (long long)rows, sums);
    return (int)hipGetLastError();
}

extern "C" int mappo_adv_stats(const double* sums, float* stats, mappo_stream_t stream_) {
    if (!sums || !stats) return MAPPO_E_NULL;
    hipLaunchKernelGGL(adv_stats_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_),
                       sums, stats);
    return (int)hipGetLastError();
}

extern "C" int mappo_adv_normalize(const float* adv, const float* stats, float* out, int64_t n,
                                   mappo_stream_t stream_) {
    if (!adv || !stats || !out) return MAPPO_E_NULL;
    if (n <= 0) return MAPPO_E_SHAPE;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(adv_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), adv, stats, out, (long long)n);
    return (int)hipGetLastError();
}
