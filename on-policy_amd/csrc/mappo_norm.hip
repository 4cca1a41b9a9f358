// K6 -- row LayerNorm (forward / backward) for the narrow feature widths of the MAPPO networks.
//
// The actor / critic trunks apply nn.LayerNorm after every Linear and to the raw observation
// (reference onpolicy/algorithms/utils/mlp.py:17-22,47-53, rnn.py:22,79).  With hidden sizes of 64
// and minibatches of 10^6..10^7 rows this is pure HBM streaming ([rows, D] in, [rows, D] out), but
// PyTorch-ROCm's generic kernels assign one workgroup per row and reach ~0.5 TB/s on D = 64: at the
// north-star size they are 54 % of the whole update (profiles/r01_bench_ns_kernel_stats.csv).
//
// Here a row is owned by LPR lanes (LPR = power of two <= 64 covering D / VEC units), so a wave
// handles 64 / LPR rows at once with 16-byte accesses; statistics are two-pass in registers (mean,
// then centred sum of squares) and reduced with lane shuffles inside the LPR-lane group.  The
// backward keeps the weight / bias gradient partials of the lanes' columns in registers across the
// grid-stride row loop and writes one [D] partial per workgroup (deterministic, no atomics).
//
//   y      = (x - mean) * rstd * w + b,  mean / rstd over the last dimension, biased variance
//   dx     = rstd * (g - mean_D(g) - xhat * mean_D(g * xhat)),  g = dy * w, xhat = (x - mean) * rstd
//   dw     = sum_rows dy * xhat,  db = sum_rows dy
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mappo_hip.h"
#include "mappo_internal.h"

namespace {

constexpr int kThreads = 256;

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int VEC> struct Unit;
template <> struct Unit<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Unit<1> { typedef float type; };

template <int VEC>
__device__ __forceinline__ float elem(const typename Unit<VEC>::type& u, int k) {
    if constexpr (VEC == 1) return u;
    else return u[k];
}
template <int VEC>
__device__ __forceinline__ void set_elem(typename Unit<VEC>::type& u, int k, float v) {
    if constexpr (VEC == 1) u = v;
    else u[k] = v;
}

// Optional activation fused in front of the normalisation (the trunks are Linear -> act ->
// LayerNorm, ref mlp.py:17-22): ACT 0 = none, 1 = tanh, 2 = relu.
template <int ACT>
__device__ __forceinline__ float act_fwd(float z) {
    if constexpr (ACT == 1) return tanhf(z);
    else if constexpr (ACT == 2) return z > 0.f ? z : 0.f;
    else return z;
}
// derivative given the pre-activation z and the activation value a
template <int ACT>
__device__ __forceinline__ float act_grad(float z, float a) {
    if constexpr (ACT == 1) return 1.f - a * a;
    else if constexpr (ACT == 2) return z > 0.f ? 1.f : 0.f;
    else return 1.f;
}

// ------------------------------------------------------------------------ forward ----
template <int VEC, int LPR, int EPL, int ACT>
__global__ void __launch_bounds__(kThreads) ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pre,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd,
                                                          long long M, int D, float eps) {
    using U = typename Unit<VEC>::type;
    constexpr int RPB = kThreads / LPR;            // rows per workgroup pass
    const int lane = threadIdx.x % LPR;
    const int rib = threadIdx.x / LPR;
    const int units = D / VEC;
    const float invD = 1.0f / (float)D;

    U wv[EPL], bv[EPL], pv[EPL];   // LayerNorm weight / bias, bias of the preceding Linear (or 0)
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        int u = lane + e * LPR;
        wv[e] = U(0.f);
        bv[e] = U(0.f);
        pv[e] = U(0.f);
        if (u < units) {
            wv[e] = reinterpret_cast<const U*>(w)[u];
            bv[e] = reinterpret_cast<const U*>(b)[u];
            if (pre != nullptr) pv[e] = reinterpret_cast<const U*>(pre)[u];
        }
    }
    for (long long row = (long long)blockIdx.x * RPB + rib; row < M; row += (long long)gridDim.x * RPB) {
        const U* xr = reinterpret_cast<const U*>(x + row * D);
        U xv[EPL];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            int u = lane + e * LPR;
            xv[e] = U(0.f);
            if (u < units) {
                xv[e] = __builtin_nontemporal_load(xr + u);
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    set_elem<VEC>(xv[e], k, act_fwd<ACT>(elem<VEC>(xv[e], k) + elem<VEC>(pv[e], k)));
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) s += elem<VEC>(xv[e], k);
        }
        const float mu = group_sum<LPR>(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            int u = lane + e * LPR;
            if (u < units) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float d = elem<VEC>(xv[e], k) - mu;
                    q += d * d;
                }
            }
        }
        const float var = group_sum<LPR>(q) * invD;
        const float r = 1.0f / sqrtf(var + eps);
        U* yr = reinterpret_cast<U*>(y + row * D);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            int u = lane + e * LPR;
            if (u < units) {
                U o;
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    set_elem<VEC>(o, k, (elem<VEC>(xv[e], k) - mu) * r * elem<VEC>(wv[e], k) + elem<VEC>(bv[e], k));
                yr[u] = o;
            }
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = r;
        }
    }
}

// ------------------------------------------------ gather + standardise (K3/K4 + K6) ----
// dst[r, :] = (x - mean(x)) * rsqrt(var(x) + eps) with x = src[source_row(r), :]: the parameter-free
// half of the input LayerNorm, done while the sampler copies the row anyway.
template <int VEC, int LPR, int EPL>
__global__ void __launch_bounds__(kThreads) gather_std_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              long long M, int D, mappo::RowMap map,
                                                              unsigned first_only, float eps) {
    using U = typename Unit<VEC>::type;
    constexpr int RPB = kThreads / LPR;
    constexpr int RI = EPL <= 2 ? 2 : 1;           // independent rows per iteration (loads in flight)
    const int lane = threadIdx.x % LPR;
    const int rib = threadIdx.x / LPR;
    const int units = D / VEC;
    const float invD = 1.0f / (float)D;
    const long long stride = (long long)gridDim.x * RPB;
    for (long long row0 = (long long)blockIdx.x * RPB + rib; row0 < M; row0 += stride * RI) {
        U xv[RI][EPL];
        float s[RI];
        // issue the index lookups and row loads of all RI rows before touching any of them
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const long long row = row0 + i * stride;
            s[i] = 0.f;
            const bool ok = row < M;
            const unsigned srow = ok ? mappo::source_row(map, first_only, (unsigned)row) : 0u;
            const U* xr = reinterpret_cast<const U*>(x + (long long)srow * D);
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                int u = lane + e * LPR;
                xv[i][e] = U(0.f);
                if (ok && u < units) xv[i][e] = __builtin_nontemporal_load(xr + u);
            }
        }
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const long long row = row0 + i * stride;
            if (row >= M) break;
#pragma unroll
            for (int e = 0; e < EPL; ++e)
#pragma unroll
                for (int k = 0; k < VEC; ++k) s[i] += elem<VEC>(xv[i][e], k);
            const float mu = group_sum<LPR>(s[i]) * invD;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                int u = lane + e * LPR;
                if (u < units) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float d = elem<VEC>(xv[i][e], k) - mu;
                        q += d * d;
                    }
                }
            }
            const float r = 1.0f / sqrtf(group_sum<LPR>(q) * invD + eps);
            U* yr = reinterpret_cast<U*>(y + row * D);
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                int u = lane + e * LPR;
                if (u < units) {
                    U o;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) set_elem<VEC>(o, k, (elem<VEC>(xv[i][e], k) - mu) * r);
                    __builtin_nontemporal_store(o, yr + u);
                }
            }
        }
    }
}

// ----------------------------------------------------------------------- backward ----
template <int VEC, int LPR, int EPL, bool NEED_DX, int ACT>
__global__ void __launch_bounds__(kThreads) ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ w, float* __restrict__ dx,
                                                          float* __restrict__ pw, float* __restrict__ pb,
                                                          long long M, int D, const float* __restrict__ pre,
                                                          float* __restrict__ pp) {
    using U = typename Unit<VEC>::type;
    constexpr int RPB = kThreads / LPR;
    extern __shared__ float red[];                 // [2 or 3][RPB][D]
    const int lane = threadIdx.x % LPR;
    const int rib = threadIdx.x / LPR;
    const int units = D / VEC;
    const float invD = 1.0f / (float)D;

    U wv[EPL], aw[EPL], ab[EPL], pv[EPL], ap[EPL];   // ap: column sums of dx = gradient of the Linear bias `pre`
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        int u = lane + e * LPR;
        wv[e] = U(0.f);
        aw[e] = U(0.f);
        ab[e] = U(0.f);
        pv[e] = U(0.f);
        ap[e] = U(0.f);
        if (u < units) {
            wv[e] = reinterpret_cast<const U*>(w)[u];
            if (pre != nullptr) pv[e] = reinterpret_cast<const U*>(pre)[u];
        }
    }
    for (long long row = (long long)blockIdx.x * RPB + rib; row < M; row += (long long)gridDim.x * RPB) {
        const U* xr = reinterpret_cast<const U*>(x + row * D);
        const U* gr = reinterpret_cast<const U*>(dy + row * D);
        const float mu = mean[row], r = rstd[row];
        U xh[EPL], gv[EPL], ag[EPL];   // xhat, dy*w, activation derivative
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            int u = lane + e * LPR;
            xh[e] = U(0.f);
            gv[e] = U(0.f);
            ag[e] = U(1.f);
            if (u < units) {
                U xv = __builtin_nontemporal_load(xr + u);
                U dv = __builtin_nontemporal_load(gr + u);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float zin = elem<VEC>(xv, k) + elem<VEC>(pv[e], k);
                    float av = act_fwd<ACT>(zin);
                    if (ACT != 0) set_elem<VEC>(ag[e], k, act_grad<ACT>(zin, av));
                    float h = (av - mu) * r;
                    float d = elem<VEC>(dv, k);
                    float g = d * elem<VEC>(wv[e], k);
                    set_elem<VEC>(xh[e], k, h);
                    set_elem<VEC>(gv[e], k, g);
                    s1 += g;
                    s2 += g * h;
                    set_elem<VEC>(aw[e], k, elem<VEC>(aw[e], k) + d * h);
                    set_elem<VEC>(ab[e], k, elem<VEC>(ab[e], k) + d);
                }
            }
        }
        if (NEED_DX) {
            const float m1 = group_sum<LPR>(s1) * invD;
            const float m2 = group_sum<LPR>(s2) * invD;
            U* dr = reinterpret_cast<U*>(dx + row * D);
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                int u = lane + e * LPR;
                if (u < units) {
                    U o;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float v = r * (elem<VEC>(gv[e], k) - m1 - elem<VEC>(xh[e], k) * m2) * elem<VEC>(ag[e], k);
                        set_elem<VEC>(o, k, v);
                        set_elem<VEC>(ap[e], k, elem<VEC>(ap[e], k) + v);
                    }
                    dr[u] = o;
                }
            }
        }
    }
    // workgroup partials of dw / db: rows-in-block -> LDS -> column sums
    float* rw = red;
    float* rb = red + (size_t)RPB * D;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        int u = lane + e * LPR;
        if (u < units) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                rw[(size_t)rib * D + u * VEC + k] = elem<VEC>(aw[e], k);
                rb[(size_t)rib * D + u * VEC + k] = elem<VEC>(ab[e], k);
            }
        }
    }
    float* rp = red + (size_t)2 * RPB * D;
    if (NEED_DX && pp != nullptr) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            int u = lane + e * LPR;
            if (u < units) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) rp[(size_t)rib * D + u * VEC + k] = elem<VEC>(ap[e], k);
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += kThreads) {
        float sw = 0.f, sb = 0.f;
#pragma unroll 4
        for (int q = 0; q < RPB; ++q) {
            sw += rw[(size_t)q * D + c];
            sb += rb[(size_t)q * D + c];
        }
        pw[(size_t)blockIdx.x * D + c] = sw;
        pb[(size_t)blockIdx.x * D + c] = sb;
        if (NEED_DX && pp != nullptr) {
            float sp = 0.f;
#pragma unroll 4
            for (int q = 0; q < RPB; ++q) sp += rp[(size_t)q * D + c];
            pp[(size_t)blockIdx.x * D + c] = sp;
        }
    }
}

// Column sums of the per-workgroup partials: 64 columns x 16 row groups per workgroup, fixed order.
// blockIdx.y selects the array (weight / bias / Linear-bias partials), so one launch reduces all of them.
constexpr int kReduceThreads = 1024;
struct ReduceArgs {
    const float* src[3];
    float* dst[3];
};
__global__ void __launch_bounds__(kReduceThreads) ln_reduce_kernel(ReduceArgs a, int nblk, int D) {
    __shared__ float sm[16][64];
    const float* __restrict__ src = blockIdx.y == 0 ? a.src[0] : blockIdx.y == 1 ? a.src[1] : a.src[2];
    float* __restrict__ dst = blockIdx.y == 0 ? a.dst[0] : blockIdx.y == 1 ? a.dst[1] : a.dst[2];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < D) {
#pragma unroll 8
        for (int q = g; q < nblk; q += 16) acc += src[(size_t)q * D + c];
    }
    sm[g][cl] = acc;
    __syncthreads();
    if (g == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sm[q][cl];
        dst[c] = t;
    }
}

struct Shape {
    int vec, lpr, epl;
};

// VEC = 4 when rows are 16-byte aligned, LPR = power of two covering the units (<= 64),
// EPL = units per lane rounded up to an instantiated size; epl == 0 -> unsupported width.
Shape pick_shape(const void* a, const void* b, int D) {
    Shape s;
    s.vec = (D % 4 == 0 && mappo::aligned_to(a, 16) && mappo::aligned_to(b, 16)) ? 4 : 1;
    int units = D / s.vec;
    int lpr = 4;
    while (lpr < units && lpr < 64) lpr <<= 1;
    s.lpr = lpr;
    int epl = (units + lpr - 1) / lpr;
    const int sizes[] = {1, 2, 4, 8, 16, 24};
    s.epl = 0;
    for (int z : sizes)
        if (epl <= z) {
            s.epl = z;
            break;
        }
    if (s.vec == 4 && s.epl > 8) s.epl = 0;
    return s;
}

int ln_grid(long long M, int lpr) {
    long long rpb = kThreads / lpr;
    long long blocks = (M + rpb - 1) / rpb;
    long long cap = (long long)mappo::kCUs * 8;
    return (int)(blocks < cap ? blocks : cap);
}

#define LN_FOR_SHAPES(X)                                                                         \
    X(4, 4, 1) X(4, 8, 1) X(4, 16, 1) X(4, 32, 1) X(4, 64, 1) X(4, 64, 2) X(4, 64, 4) X(4, 64, 8) \
    X(1, 4, 1) X(1, 8, 1) X(1, 16, 1) X(1, 32, 1) X(1, 64, 1) X(1, 64, 2) X(1, 64, 4) X(1, 64, 8) \
    X(1, 64, 16) X(1, 64, 24)

}  // namespace

int mappo::gather_standardize(const float* src, float* dst, int width, long long rows_out,
                              const mappo::RowMap& map, unsigned first_only, float eps, hipStream_t stream) {
    if (rows_out <= 0 || width <= 0) return MAPPO_E_SHAPE;
    Shape s = pick_shape(src, dst, width);
    if (s.epl == 0) return MAPPO_E_SHAPE;
    dim3 grid(ln_grid(rows_out, s.lpr)), block(kThreads);
#define LN_LAUNCH_GS(V, L, E)                                                                    \
    if (s.vec == V && s.lpr == L && s.epl == E) {                                                \
        hipLaunchKernelGGL((gather_std_kernel<V, L, E>), grid, block, 0, stream, src, dst, rows_out, width, \
                           map, first_only, eps);                                                \
        return (int)hipGetLastError();                                                           \
    }
    LN_FOR_SHAPES(LN_LAUNCH_GS)
#undef LN_LAUNCH_GS
    return MAPPO_E_SHAPE;
}

extern "C" int mappo_layernorm_max_blocks(void) { return mappo::kCUs * 8; }

extern "C" int mappo_bias_act_layernorm_fwd(const float* x, const float* pre_bias, const float* weight,
                                            const float* bias, float* y, float* mean, float* rstd, int64_t M,
                                            int D, float eps, int act, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !bias || !y || !mean || !rstd) return MAPPO_E_NULL;
    if (M <= 0 || D <= 0) return MAPPO_E_SHAPE;
    Shape s = pick_shape(x, y, D);
    if (s.vec == 4 && !(mappo::aligned_to(weight, 16) && mappo::aligned_to(bias, 16) &&
                        (!pre_bias || mappo::aligned_to(pre_bias, 16))))
        s = pick_shape((void*)1, (void*)1, D);
    if (s.epl == 0) return MAPPO_E_SHAPE;
    dim3 grid(ln_grid(M, s.lpr)), block(kThreads);
    if (act < 0 || act > 2) return MAPPO_E_FLAGS;
#define LN_LAUNCH_FWD(V, L, E)                                                                   \
    if (s.vec == V && s.lpr == L && s.epl == E) {                                                \
        if (act == 0)                                                                            \
            hipLaunchKernelGGL((ln_fwd_kernel<V, L, E, 0>), grid, block, 0, stream, x, pre_bias, weight, bias, y, mean, \
                               rstd, (long long)M, D, eps);                                      \
        else if (act == 1)                                                                       \
            hipLaunchKernelGGL((ln_fwd_kernel<V, L, E, 1>), grid, block, 0, stream, x, pre_bias, weight, bias, y, mean, \
                               rstd, (long long)M, D, eps);                                      \
        else                                                                                     \
            hipLaunchKernelGGL((ln_fwd_kernel<V, L, E, 2>), grid, block, 0, stream, x, pre_bias, weight, bias, y, mean, \
                               rstd, (long long)M, D, eps);                                      \
        return (int)hipGetLastError();                                                           \
    }
    LN_FOR_SHAPES(LN_LAUNCH_FWD)
#undef LN_LAUNCH_FWD
    return MAPPO_E_SHAPE;
}

extern "C" int mappo_act_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y,
                                       float* mean, float* rstd, int64_t M, int D, float eps, int act,
                                       mappo_stream_t stream) {
    return mappo_bias_act_layernorm_fwd(x, nullptr, weight, bias, y, mean, rstd, M, D, eps, act, stream);
}

extern "C" int mappo_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y,
                                   float* mean, float* rstd, int64_t M, int D, float eps,
                                   mappo_stream_t stream) {
    return mappo_bias_act_layernorm_fwd(x, nullptr, weight, bias, y, mean, rstd, M, D, eps, 0, stream);
}

extern "C" int mappo_bias_act_layernorm_bwd(const float* dy, const float* x, const float* pre_bias,
                                            const float* mean, const float* rstd, const float* weight, float* dx,
                                            float* dweight, float* dbias, float* dpre_bias, float* partials,
                                            int64_t M, int D, int act, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!dy || !x || !mean || !rstd || !weight || !dweight || !dbias || !partials) return MAPPO_E_NULL;
    if (dpre_bias && !dx) return MAPPO_E_NULL;        // the Linear-bias gradient is the column sum of dx
    if (M <= 0 || D <= 0) return MAPPO_E_SHAPE;
    if (dpre_bias && D > 1024) return MAPPO_E_SHAPE;  // third LDS partial: keep the workgroup within 64 KB
    Shape s = pick_shape(x, dy, D);
    if (s.vec == 4 && !(mappo::aligned_to(weight, 16) && (!dx || mappo::aligned_to(dx, 16)) &&
                        (!pre_bias || mappo::aligned_to(pre_bias, 16))))
        s = pick_shape((void*)1, (void*)1, D);
    if (s.epl == 0) return MAPPO_E_SHAPE;
    const int nblk = ln_grid(M, s.lpr);
    float* pw = partials;
    float* pb = partials + (size_t)mappo::kCUs * 8 * D;
    float* pp = dpre_bias ? partials + (size_t)2 * mappo::kCUs * 8 * D : nullptr;
    const size_t lds = (size_t)(dpre_bias ? 3 : 2) * (kThreads / s.lpr) * D * sizeof(float);
    dim3 grid(nblk), block(kThreads);
    bool launched = false;
    if (act < 0 || act > 2) return MAPPO_E_FLAGS;
#define LN_BWD_ONE(V, L, E, DX, A)                                                               \
    hipLaunchKernelGGL((ln_bwd_kernel<V, L, E, DX, A>), grid, block, lds, stream, dy, x, mean, rstd, weight, \
                       dx, pw, pb, (long long)M, D, pre_bias, pp)
#define LN_LAUNCH_BWD(V, L, E)                                                                   \
    if (!launched && s.vec == V && s.lpr == L && s.epl == E) {                                   \
        if (dx) {                                                                                \
            if (act == 0) LN_BWD_ONE(V, L, E, true, 0);                                          \
            else if (act == 1) LN_BWD_ONE(V, L, E, true, 1);                                     \
            else LN_BWD_ONE(V, L, E, true, 2);                                                   \
        } else {                                                                                 \
            if (act == 0) LN_BWD_ONE(V, L, E, false, 0);                                         \
            else if (act == 1) LN_BWD_ONE(V, L, E, false, 1);                                    \
            else LN_BWD_ONE(V, L, E, false, 2);                                                  \
        }                                                                                        \
        launched = true;                                                                         \
    }
    LN_FOR_SHAPES(LN_LAUNCH_BWD)
#undef LN_LAUNCH_BWD
    if (!launched) return MAPPO_E_SHAPE;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    ReduceArgs ra;
    ra.src[0] = pw;
    ra.src[1] = pb;
    ra.src[2] = pp;
    ra.dst[0] = dweight;
    ra.dst[1] = dbias;
    ra.dst[2] = dpre_bias;
    hipLaunchKernelGGL(ln_reduce_kernel, dim3((D + 63) / 64, dpre_bias ? 3 : 2), dim3(kReduceThreads), 0, stream, ra,
                       nblk, D);
    return (int)hipGetLastError();
}

extern "C" int mappo_act_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                                       const float* weight, float* dx, float* dweight, float* dbias,
                                       float* partials, int64_t M, int D, int act, mappo_stream_t stream) {
    return mappo_bias_act_layernorm_bwd(dy, x, nullptr, mean, rstd, weight, dx, dweight, dbias, nullptr, partials,
                                        M, D, act, stream);
}

extern "C" int mappo_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                                   const float* weight, float* dx, float* dweight, float* dbias,
                                   float* partials, int64_t M, int D, mappo_stream_t stream) {
    return mappo_bias_act_layernorm_bwd(dy, x, nullptr, mean, rstd, weight, dx, dweight, dbias, nullptr, partials,
                                        M, D, 0, stream);
}
