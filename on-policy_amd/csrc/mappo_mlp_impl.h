// Fused actor / critic trunk for hidden width 64 (K9): sampler gather + input standardisation + Linear + act + LayerNorm
// chain + head in one forward kernel; the matching backward as two kernels (row-parallel chain, first-layer weight
// gradient as a gather-fused split-K GEMM).  Reference maths: onpolicy/algorithms/utils/mlp.py:6-58 (MLPLayer / MLPBase:
// [LayerNorm] -> (Linear -> Tanh|ReLU -> LayerNorm) x (1 + layer_N)), act.py:44-60 / distributions.py:55-68 (Categorical
// head = Linear), r_actor_critic.py:147-175 (v_out = Linear(hidden, 1)), rows drawn by the samplers of
// onpolicy/utils/shared_buffer.py:340-400 (feed-forward) and :499-608 (recurrent chunks).
//
// Why: at the north-star size the update was bound by [rows, 64] activation round trips between separate GEMM, bias,
// activation and LayerNorm launches (47 % library GEMMs + 23 % LayerNorm passes + 17 % gathers of a 528 ms step).  Here a
// row tile stays in registers from the gathered observation to the head's output; the only per-row HBM traffic is the
// observation row itself, two saved [64] pre-activation rows and the head output.  All products run on the f32 MFMA
// (v_mfma_f32_32x32x2_f32: exact float32 fma chains, 64 flop / clk / SIMD), so this path is MFMA-bound, not HBM-bound.
//
// This header is the whole implementation, written against a handful of primitives (prim::mfma32, prim::xhalf,
// prim::lds, __syncthreads, MAPPO_LAUNCH).  mappo_mlp.hip binds them to gfx950; tests/simt/ binds them to a host SIMT
// emulator so that the fragment-layout logic is checked without a GPU (test infrastructure, never linked into the
// product library).
//
// Orientation of every product: D[feature][row] = W[feature][k] . X[k][row] -- weights are the MFMA A operand
// (lane & 31 = output feature), activations the B operand (lane & 31 = row), so a lane owns ONE row and, after each
// layer, holds 32 of its 64 features in accumulator registers: slot s = 16 t + v (t = feature tile, v = register) of
// half-wave h = lane >> 5 is feature
//     f(h, s) = 32 t + (v & 3) + 8 (v >> 2) + 4 h.
// The next layer contracts over features, i.e. over (h, s): accumulator registers feed the next MFMA's B operand
// directly (step s takes slot s from both half-waves) and the weights are staged in LDS in the matching order
// ("permuted" arrays below).  LayerNorm statistics are 32 in-lane terms plus one exchange with lane ^ 32.
#ifndef MAPPO_MLP_IMPL_H
#define MAPPO_MLP_IMPL_H

#include "../../include/mappo_hip.h"

namespace mlp {

constexpr int kH = 64;          // hidden width
constexpr int kKC = 64;         // first-layer k chunk (256-byte pieces of an observation row)
constexpr int kXS = kKC + 4;    // LDS row stride of the chunk tiles (floats): 16-lane ds_read_b128 groups hit 64 banks
constexpr int kTR = 128;        // rows per workgroup tile = 4 waves x 32
constexpr int kWS = 68;         // LDS row stride of a permuted 64-wide weight row
constexpr int kTS = 36;         // LDS row stride of the [feature][32 rows] transposes of the backward
constexpr int kThreads = 256;
constexpr int kFwdGridCap = 512;     // 2 workgroups per CU
constexpr int kBwdGridCap = 256;     // 1 workgroup per CU (108 KB of LDS)
constexpr int kDw1Rows = 32;         // rows per iteration of the first-layer weight-gradient kernel
constexpr int kDw1Slab = 384;        // k columns per workgroup of that kernel (6 accumulator tiles per wave)
constexpr int kDw1GridCap = 512;

__host__ __device__ __forceinline__ int feat_of(int h, int s) {
    return 32 * (s >> 4) + (s & 3) + 8 * ((s & 15) >> 2) + 4 * h;
}

// ---------------------------------------------------------------- flat parameter-gradient layout ----
//   [w1: 64 * din] [per layer l: bias 64 | ln weight 64 | ln bias 64] [per hidden layer l >= 1: w 64 * 64]
//   [head weight out * 64] [head bias out]
__host__ __device__ __forceinline__ long long g_vec(int din, int l) { return 64LL * din + 192LL * l; }
__host__ __device__ __forceinline__ long long g_w2(int din, int L, int l) { return 64LL * din + 192LL * L + 4096LL * (l - 1); }
__host__ __device__ __forceinline__ long long g_wh(int din, int L) { return 64LL * din + 192LL * L + 4096LL * (L - 1); }
__host__ __device__ __forceinline__ long long g_total(int din, int L, int out) { return g_wh(din, L) + 65LL * out; }
// the chain kernel's per-wave partial row is the same layout without w1
__host__ __device__ __forceinline__ long long p_main(int L, int out) { return 192LL * L + 4096LL * (L - 1) + 65LL * out; }

struct RowSrc {
    const float* src;
    const float* stats;
    const long long* idx;
    long long rows;
    long long mb;
    int chunk_len, T, N, A;
    int din;
};

// source row (time-major [T, N, A] row space) of launch row r  (shared_buffer.py:379-396 / :554-604)
__device__ __forceinline__ long long source_row(const RowSrc& m, long long r) {
    if (m.idx == nullptr) return r;
    if (m.chunk_len <= 0) return m.idx[r];
    const long long l = r / m.mb, j = r - l * m.mb;
    const long long f = m.idx[j] * m.chunk_len + l;
    const long long at = (long long)m.A * m.T;
    const long long n = f / at, rem = f - n * at;
    const long long ag = rem / m.T, t = rem - ag * m.T;
    return (t * m.N + n) * m.A + ag;
}

// 4 consecutive floats of a row starting at column k (rows are only 4-byte aligned for odd widths: the unaligned
// 16-byte load is legal on gfx950), zero beyond the row's end
__device__ __forceinline__ v4 load4_guard(const float* p, int remaining) {
    v4 r = {0.f, 0.f, 0.f, 0.f};
    if (remaining >= 4) {
        r = *reinterpret_cast<const v4u*>(p);
    } else {
        if (remaining > 0) r[0] = p[0];
        if (remaining > 1) r[1] = p[1];
        if (remaining > 2) r[2] = p[2];
    }
    return r;
}

__device__ __forceinline__ float act_fn(float z, int act) { return act == 1 ? tanhf(z) : (act == 2 ? fmaxf(z, 0.f) : z); }
// derivative from the activation output a (tanh) / the pre-activation z (ReLU)
__device__ __forceinline__ float act_grad(float z, float a, int act) {
    return act == 1 ? 1.f - a * a : (act == 2 ? (z > 0.f ? 1.f : 0.f) : 1.f);
}

struct Net {
    int din, L, act, out;
    float eps;
    const float* w1;
    const float* bias[3];
    const float* ln_g[3];
    const float* ln_b[3];
    const float* w2[2];
    const float* wh;
    const float* bh;
};

// ---------------------------------------------------------------- LDS layouts (float offsets) ----
struct FwdLds {
    int vec, w2p, whp, bh, xt, wt, total;
};
__host__ __device__ __forceinline__ FwdLds fwd_lds(int L, int out) {
    FwdLds o;
    o.vec = 0;                                // [L][bias | g | beta][64]
    o.w2p = o.vec + 192 * L;                  // [L - 1][2][32][kWS]
    o.whp = o.w2p + (L - 1) * 2 * 32 * kWS;   // [out][64] permuted
    o.bh = o.whp + out * 64;                  // [out]
    o.xt = (o.bh + out + 3) & ~3;             // [kTR][kXS]
    o.wt = o.xt + kTR * kXS;                  // [64][kXS]
    o.total = o.wt + 64 * kXS;
    return o;
}

// Parameters that every tile needs, staged once per workgroup.  w2p[l-1][t][i][h * 32 + s] = W2_l[32 t + i][f(h, s)]:
// the A operand (output feature 32 t + i on lane i) of the step that consumes slot s of the previous layer's registers.
__device__ __forceinline__ void stage_fwd_params(const Net& n, float* lds, const FwdLds& o) {
    const int tid = threadIdx.x;
    for (int e = tid; e < 192 * n.L; e += kThreads) {
        const int l = e / 192, q = (e - 192 * l) >> 6, c = e & 63;
        const float* p = q == 0 ? n.bias[l] : (q == 1 ? n.ln_g[l] : n.ln_b[l]);
        lds[o.vec + e] = p[c];
    }
    for (int l = 1; l < n.L; ++l)
        for (int e = tid; e < 64 * 64; e += kThreads) {
            const int fo = e >> 6, hs = e & 63;          // output feature, (h, s)
            const int k = feat_of(hs >> 5, hs & 31);
            lds[o.w2p + (l - 1) * 2 * 32 * kWS + fo * kWS + hs] = n.w2[l - 1][fo * 64 + k];
        }
    for (int e = tid; e < n.out * 64; e += kThreads) {
        const int oo = e >> 6, hs = e & 63;
        lds[o.whp + e] = n.wh[oo * 64 + feat_of(hs >> 5, hs & 31)];
    }
    for (int e = tid; e < n.out; e += kThreads) lds[o.bh + e] = n.bh[e];
}

// One layer's tail on the accumulators: z = acc + bias (optionally kept for saving), a = act(z), LayerNorm over the
// row's 64 features (mlp.py:17-22).  In: acc[2] (this lane's 32 slots).  Out: hreg[32] = LayerNorm output.
template <bool KEEPZ>
__device__ __forceinline__ void layer_tail(const f32x16* acc, const float* vec /* bias | g | beta in LDS */, int h,
                                           int act, float eps, float* hreg, float* zreg) {
    float a[32];
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4 b = *reinterpret_cast<const v4*>(vec + 32 * t + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int s = 16 * t + 4 * q + e;
                const float z = acc[t][4 * q + e] + b[e];
                if (KEEPZ) zreg[s] = z;
                a[s] = act_fn(z, act);
                sum += a[s];
            }
        }
    sum += prim::xhalf(sum);
    const float mean = sum * (1.f / 64.f);
    float var = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        a[s] -= mean;
        var += a[s] * a[s];
    }
    var += prim::xhalf(var);
    const float rstd = 1.f / sqrtf(var * (1.f / 64.f) + eps);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4 g = *reinterpret_cast<const v4*>(vec + 64 + 32 * t + 8 * q + 4 * h);
            const v4 be = *reinterpret_cast<const v4*>(vec + 128 + 32 * t + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int s = 16 * t + 4 * q + e;
                hreg[s] = a[s] * rstd * g[e] + be[e];
            }
        }
}

// registers (slot order) -> a [rows, 64] row in HBM: slots 4q .. 4q+3 of tile t are 4 consecutive features
__device__ __forceinline__ void store_row64(float* dst_row, const float* reg, int h) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = reg[16 * t + 4 * q + e];
            *reinterpret_cast<v4*>(dst_row + 32 * t + 8 * q + 4 * h) = o;
        }
}
__device__ __forceinline__ void load_row64(const float* src_row, float* reg, int h) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4 o = *reinterpret_cast<const v4*>(src_row + 32 * t + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) reg[16 * t + 4 * q + e] = o[e];
        }
}

// acc[t] = sum over (h, s) of Wp[t][lane & 31][h * 32 + s] * reg[s]   (a 64 -> 64 product on registers)
__device__ __forceinline__ void dense64(const float* wp, int c, int h, const float* reg, f32x16* acc) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    const float* w0 = wp + c * kWS + 32 * h;
    const float* w1 = wp + (32 + c) * kWS + 32 * h;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const v4 a0 = *reinterpret_cast<const v4*>(w0 + 4 * q);
        const v4 a1 = *reinterpret_cast<const v4*>(w1 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0] = prim::mfma32(a0[e], reg[4 * q + e], acc[0]);
            acc[1] = prim::mfma32(a1[e], reg[4 * q + e], acc[1]);
        }
    }
}

// ================================================================== forward ====
struct FwdArgs {
    RowSrc rs;
    Net net;
    float* y;
    float* z[3];
};

__global__ void __launch_bounds__(kThreads) mlp_fwd_kernel(FwdArgs a) {
    float* lds = prim::lds();
    const Net& n = a.net;
    const FwdLds o = fwd_lds(n.L, n.out);
    stage_fwd_params(n, lds, o);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
    const int xr = tid >> 4, xq = tid & 15;          // staging role: row within a group of 16, 16-byte piece of the chunk
    const int din = n.din;
    const int nch = (din + kKC - 1) / kKC;
    const long long rows = a.rs.rows;
    const long long ntiles = (rows + kTR - 1) / kTR;
    float* xt = lds + o.xt;
    float* wt = lds + o.wt;

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row0 = tile * kTR;
        // the 8 rows this thread stages: source offsets and standardisation constants
        const float* xsrc[8];
        float mu[8], rsd[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            long long r = row0 + xr + 16 * p;
            if (r >= rows) r = rows - 1;
            const long long sr = source_row(a.rs, r);
            xsrc[p] = a.rs.src + sr * din;
            mu[p] = 0.f;
            rsd[p] = 1.f;
            if (a.rs.stats != nullptr) {
                mu[p] = a.rs.stats[2 * sr];
                rsd[p] = a.rs.stats[2 * sr + 1];
            }
        }
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

        v4 xv[8], wv[4];
        auto load_chunk = [&](int kc) {
            const int k = kc * kKC + 4 * xq;
#pragma unroll
            for (int p = 0; p < 8; ++p) xv[p] = load4_guard(xsrc[p] + k, din - k);
#pragma unroll
            for (int p = 0; p < 4; ++p) wv[p] = load4_guard(n.w1 + (long long)(xr + 16 * p) * din + k, din - k);
        };
        load_chunk(0);
        for (int kc = 0; kc < nch; ++kc) {
            __syncthreads();        // the previous chunk's operands (and the previous tile's) have been consumed
            {
                const int k = kc * kKC + 4 * xq;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    v4 x = xv[p];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = (k + e < din) ? (x[e] - mu[p]) * rsd[p] : 0.f;
                    *reinterpret_cast<v4*>(xt + (xr + 16 * p) * kXS + 4 * xq) = x;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) *reinterpret_cast<v4*>(wt + (xr + 16 * p) * kXS + 4 * xq) = wv[p];
            }
            __syncthreads();
            if (kc + 1 < nch) load_chunk(kc + 1);     // in flight during this chunk's MFMAs
            const float* xa = xt + (32 * wave + c) * kXS + 32 * h;
            const float* w0 = wt + c * kXS + 32 * h;
            const float* w1 = wt + (32 + c) * kXS + 32 * h;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const v4 b = *reinterpret_cast<const v4*>(xa + 4 * q);
                const v4 a0 = *reinterpret_cast<const v4*>(w0 + 4 * q);
                const v4 a1 = *reinterpret_cast<const v4*>(w1 + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0] = prim::mfma32(a0[e], b[e], acc[0]);
                    acc[1] = prim::mfma32(a1[e], b[e], acc[1]);
                }
            }
        }
        // ---- the rest of the network on this lane's row
        const long long row = row0 + 32 * wave + c;
        const bool ok = row < rows;
        float hreg[32], zreg[32];
        for (int l = 0; l < n.L; ++l) {
            if (l > 0) dense64(lds + o.w2p + (l - 1) * 2 * 32 * kWS, c, h, hreg, acc);
            if (a.z[l] != nullptr) {
                layer_tail<true>(acc, lds + o.vec + 192 * l, h, n.act, n.eps, hreg, zreg);
                if (ok) store_row64(a.z[l] + row * 64, zreg, h);
            } else {
                layer_tail<false>(acc, lds + o.vec + 192 * l, h, n.act, n.eps, hreg, zreg);
            }
        }
        if (n.out == 0) {
            if (ok) store_row64(a.y + row * 64, hreg, h);
        } else {
            for (int oo = 0; oo < n.out; ++oo) {
                const float* wp = lds + o.whp + oo * 64 + 32 * h;
                float p = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v4 w = *reinterpret_cast<const v4*>(wp + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) p += w[e] * hreg[4 * q + e];
                }
                p += prim::xhalf(p);
                if (ok && h == (oo & 1)) a.y[row * n.out + oo] = p + lds[o.bh + oo];
            }
        }
    }
}

// ================================================================== backward: row-parallel chain ====
// Per 32-row wave tile: recompute act / LayerNorm of every layer from the saved pre-activations, walk the chain
// backwards in registers, write d loss / d z of the first layer to HBM (consumed by the weight-gradient kernel below)
// and accumulate every other parameter gradient: hidden-layer weights in accumulator registers (MFMA over the rows of
// the tile, operands transposed through wave-private LDS scratch), vectors and the head in per-wave LDS accumulators.
struct BwdLds {
    int vec, w2t, whp, scratch, acc, scratch_per_wave, acc_per_wave, total;
};
__host__ __device__ __forceinline__ BwdLds bwd_lds(int L, int out) {
    BwdLds o;
    o.vec = 0;                                    // [L][bias | g | beta][64]
    o.w2t = o.vec + 192 * L;                      // [L - 1][2][32][kWS]: transposed + permuted hidden weights
    o.whp = o.w2t + (L - 1) * 2 * 32 * kWS;       // [out][64] permuted head weights
    o.scratch = (o.whp + out * 64 + 3) & ~3;
    o.scratch_per_wave = 2 * 64 * kTS + ((out * 32 + 3) & ~3);   // TA | TB | DY[out][32]
    o.acc = o.scratch + 4 * o.scratch_per_wave;
    o.acc_per_wave = 192 * L + 65 * out;          // vectors per layer | head weight [out][64] | head bias [out]
    o.total = o.acc + 4 * o.acc_per_wave;
    return o;
}

struct BwdArgs {
    RowSrc rs;          // only rows is used here
    Net net;
    const float* z[3];
    const float* dy;    // [rows, out] (head) or [rows, 64] (out == 0)
    float* dz1;         // [rows, 64]
    float* partials;    // [gridDim.x * 4][p_main]
};

// row sums of a [64][32] transpose: lane = feature
__device__ __forceinline__ float rowsum32(const float* trow) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const v4 t = *reinterpret_cast<const v4*>(trow + 4 * q);
        s += (t[0] + t[1]) + (t[2] + t[3]);
    }
    return s;
}

// registers (slot order, this lane's row = column c) -> T[feature][c]
__device__ __forceinline__ void put_transposed(float* T, const float* reg, int c, int h) {
#pragma unroll
    for (int s = 0; s < 32; ++s) T[feat_of(h, s) * kTS + c] = reg[s];
}

template <int L>
__global__ void __launch_bounds__(kThreads) mlp_bwd_kernel(BwdArgs a) {
    float* lds = prim::lds();
    const Net& n = a.net;
    const int out = n.out;
    const BwdLds o = bwd_lds(L, out);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
    // ---- parameters
    for (int e = tid; e < 192 * L; e += kThreads) {
        const int l = e / 192, q = (e - 192 * l) >> 6, cc = e & 63;
        const float* p = q == 0 ? n.bias[l] : (q == 1 ? n.ln_g[l] : n.ln_b[l]);
        lds[o.vec + e] = p[cc];
    }
    // w2t[l-1][t][i][h * 32 + s] = W2_l[f(h, s)][32 t + i]: A operand of dX = W^T dZ (lane i = input feature)
    for (int l = 1; l < L; ++l)
        for (int e = tid; e < 64 * 64; e += kThreads) {
            const int ki = e >> 6, hs = e & 63;
            lds[o.w2t + (l - 1) * 2 * 32 * kWS + ki * kWS + hs] = n.w2[l - 1][feat_of(hs >> 5, hs & 31) * 64 + ki];
        }
    for (int e = tid; e < out * 64; e += kThreads) {
        const int oo = e >> 6, hs = e & 63;
        lds[o.whp + e] = n.wh[oo * 64 + feat_of(hs >> 5, hs & 31)];
    }
    float* TA = lds + o.scratch + wave * o.scratch_per_wave;
    float* TB = TA + 64 * kTS;
    float* DY = TB + 64 * kTS;
    float* vacc = lds + o.acc + wave * o.acc_per_wave;      // [L][db | dg | dbeta][64] | dwh[out][64] | dbh[out]
    for (int e = lane; e < o.acc_per_wave; e += 64) vacc[e] = 0.f;
    // hidden layers 1 .. L - 1: tile 2 t + t' = (output feature tile t, input feature tile t').  L is a template
    // parameter so that every index into this array is a compile-time constant (registers, not scratch memory)
    constexpr int NW = L > 1 ? L - 1 : 1;
    f32x16 dw2[NW][4];
#pragma unroll
    for (int l = 0; l < NW; ++l)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) dw2[l][t][v] = 0.f;
    __syncthreads();

    const long long rows = a.rs.rows;
    const long long ntiles = (rows + kTR - 1) / kTR;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row = tile * kTR + 32 * wave + c;
        const bool ok = row < rows;
        const long long rrow = ok ? row : rows - 1;
        float dh[32];       // gradient w.r.t. the output of the layer being processed (slot order)
        float zr[32], nh[32], hr[32];
        float rstd;
        // ---- top layer forward quantities
        auto recompute = [&](int l) {
            load_row64(a.z[l] + rrow * 64, zr, h);
            float av[32];       // the saved z already contains the bias: act + LayerNorm restated on it
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                av[s] = act_fn(zr[s], n.act);
                sum += av[s];
            }
            sum += prim::xhalf(sum);
            const float mean = sum * (1.f / 64.f);
            float var = 0.f;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                av[s] -= mean;
                var += av[s] * av[s];
            }
            var += prim::xhalf(var);
            rstd = 1.f / sqrtf(var * (1.f / 64.f) + n.eps);
            const float* vec = lds + o.vec + 192 * l;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4 g = *reinterpret_cast<const v4*>(vec + 64 + 32 * t + 8 * q + 4 * h);
                    const v4 be = *reinterpret_cast<const v4*>(vec + 128 + 32 * t + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int s = 16 * t + 4 * q + e;
                        nh[s] = av[s] * rstd;
                        hr[s] = nh[s] * g[e] + be[e];
                    }
                }
        };
        recompute(L - 1);
        // ---- head
        if (out == 0) {
            load_row64(a.dy + rrow * 64, dh, h);
            if (!ok) {
#pragma unroll
                for (int s = 0; s < 32; ++s) dh[s] = 0.f;
            }
        } else {
#pragma unroll
            for (int s = 0; s < 32; ++s) dh[s] = 0.f;
            put_transposed(TA, hr, c, h);
            for (int oo = 0; oo < out; ++oo) {
                const float d = ok ? a.dy[row * out + oo] : 0.f;
                if (h == 0) DY[oo * 32 + c] = d;
                const float* wp = lds + o.whp + oo * 64 + 32 * h;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v4 w = *reinterpret_cast<const v4*>(wp + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dh[4 * q + e] += w[e] * d;
                }
            }
            __syncthreads();
            // lane = feature: d loss / d Wh[o][lane] += sum over the tile's rows of dy[row][o] * h[lane][row]
            {
                float hrow[32];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v4 t = *reinterpret_cast<const v4*>(TA + lane * kTS + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hrow[4 * q + e] = t[e];
                }
                for (int oo = 0; oo < out; ++oo) {
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const v4 d = *reinterpret_cast<const v4*>(DY + oo * 32 + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) s += d[e] * hrow[4 * q + e];
                    }
                    vacc[192 * L + oo * 64 + lane] += s;
                }
                if (lane < out) vacc[192 * L + 64 * out + lane] += rowsum32(DY + lane * 32);
            }
            __syncthreads();
        }
        // ---- layers, top down
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            const float* vec = lds + o.vec + 192 * l;
            float* va = vacc + 192 * l;
            // LayerNorm backward: d beta = sum dh, d gamma = sum dh * nhat; then d a, d z
            float dn[32];
            float m1 = 0.f, m2 = 0.f;
            {
                float prod[32];
#pragma unroll
                for (int s = 0; s < 32; ++s) prod[s] = dh[s] * nh[s];
                put_transposed(TA, dh, c, h);
                put_transposed(TB, prod, c, h);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4 g = *reinterpret_cast<const v4*>(vec + 64 + 32 * t + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int s = 16 * t + 4 * q + e;
                        dn[s] = dh[s] * g[e];
                        m1 += dn[s];
                        m2 += dn[s] * nh[s];
                    }
                }
            m1 += prim::xhalf(m1);
            m2 += prim::xhalf(m2);
            m1 *= (1.f / 64.f);
            m2 *= (1.f / 64.f);
            float dz[32];
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const float da = rstd * (dn[s] - m1 - nh[s] * m2);
                // act output: a = nhat / rstd + mean is not kept; tanh' = 1 - tanh(z)^2 is re-evaluated from z
                const float av = act_fn(zr[s], n.act);
                dz[s] = da * act_grad(zr[s], av, n.act);
            }
            __syncthreads();
            va[128 + lane] += rowsum32(TA + lane * kTS);      // d beta
            va[64 + lane] += rowsum32(TB + lane * kTS);       // d gamma
            __syncthreads();
            put_transposed(TA, dz, c, h);
            if (l == 0) {
                if (ok) store_row64(a.dz1 + row * 64, dz, h);
                __syncthreads();
                va[lane] += rowsum32(TA + lane * kTS);        // d bias
                __syncthreads();
                break;
            }
            // hidden layer l >= 1: its input is the output of layer l - 1
            recompute(l - 1);     // overwrites zr / nh / hr / rstd with layer l - 1's
            put_transposed(TB, hr, c, h);
            __syncthreads();
            va[lane] += rowsum32(TA + lane * kTS);            // d bias
            // dW[f][k] += sum over rows dz[f][row] * hin[k][row]: A = TA (lane = f), B = TB (lane = k), rows 16 h + s
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4 a0 = *reinterpret_cast<const v4*>(TA + c * kTS + 16 * h + 4 * q);
                const v4 a1 = *reinterpret_cast<const v4*>(TA + (32 + c) * kTS + 16 * h + 4 * q);
                const v4 b0 = *reinterpret_cast<const v4*>(TB + c * kTS + 16 * h + 4 * q);
                const v4 b1 = *reinterpret_cast<const v4*>(TB + (32 + c) * kTS + 16 * h + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dw2[l - 1][0] = prim::mfma32(a0[e], b0[e], dw2[l - 1][0]);
                    dw2[l - 1][1] = prim::mfma32(a0[e], b1[e], dw2[l - 1][1]);
                    dw2[l - 1][2] = prim::mfma32(a1[e], b0[e], dw2[l - 1][2]);
                    dw2[l - 1][3] = prim::mfma32(a1[e], b1[e], dw2[l - 1][3]);
                }
            }
            // d hin = W^T dz
            f32x16 dx[2];
            dense64(lds + o.w2t + (l - 1) * 2 * 32 * kWS, c, h, dz, dx);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) dh[16 * t + v] = dx[t][v];
            __syncthreads();
        }
    }
    // ---- flush this wave's partial sums
    float* prow = a.partials + ((long long)blockIdx.x * 4 + wave) * p_main(L, out);
    for (int e = lane; e < 192 * L; e += 64) prow[e] = vacc[e];
    for (int e = lane; e < 65 * out; e += 64) prow[192 * L + 4096 * (L - 1) + e] = vacc[192 * L + e];
#pragma unroll
    for (int l = 1; l < L; ++l)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int f = 32 * t + (v & 3) + 8 * (v >> 2) + 4 * h;
                    const int k = 32 * tp + c;
                    prow[192 * L + 4096 * (l - 1) + f * 64 + k] = dw2[l - 1][2 * t + tp][v];
                }
}

// ================================================================== backward: first-layer weight gradient ====
// dW1[f][k] = sum over rows dz1[row][f] * xhat[row][k] with xhat gathered and standardised on the fly: a split-K GEMM
// (K = rows) whose B operand is read through the sampler's index list.  A workgroup owns a slab of <= 384 k columns
// (blockIdx.y) and a strided set of 32-row tiles; wave w owns k tiles w, w + 4, w + 8 of the slab x both feature tiles.
struct Dw1Args {
    RowSrc rs;
    const float* dz1;
    float* partials;    // [gridDim.x][64 * din]
};

__global__ void __launch_bounds__(kThreads) mlp_dw1_kernel(Dw1Args a) {
    float* lds = prim::lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
    const int din = a.rs.din;
    const int k0 = blockIdx.y * kDw1Slab;
    const int kw = (din - k0 < kDw1Slab ? ((din - k0 + 31) / 32) * 32 : kDw1Slab);    // slab width, multiple of 32
    const int xs = kw + 4;
    const int pieces = kw / 4;                    // 16-byte pieces per row of the slab
    const int per_thread = (kDw1Rows * pieces + kThreads - 1) / kThreads;   // <= 12
    float* xt = lds;                              // [32][xs]
    float* dzt = lds + kDw1Rows * (kDw1Slab + 4); // [32][kWS]
    long long* srow = reinterpret_cast<long long*>(dzt + kDw1Rows * kWS);   // [2][32] source rows of a tile
    float* sst = reinterpret_cast<float*>(srow + 2 * kDw1Rows);             // [2][32][2] their (mean, rstd)
    const long long rows = a.rs.rows;
    const long long ntiles = (rows + kDw1Rows - 1) / kDw1Rows;
    const int ntk = kw / 32;

    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][t][v] = 0.f;

    auto rowinfo = [&](long long tile, int slot) {
        if (tid < kDw1Rows) {
            long long r = tile * kDw1Rows + tid;
            if (r >= rows) r = rows - 1;
            const long long sr = source_row(a.rs, r);
            srow[slot * kDw1Rows + tid] = sr;
            float m = 0.f, s = 1.f;
            if (a.rs.stats != nullptr) {
                m = a.rs.stats[2 * sr];
                s = a.rs.stats[2 * sr + 1];
            }
            sst[(slot * kDw1Rows + tid) * 2] = m;
            sst[(slot * kDw1Rows + tid) * 2 + 1] = s;
        }
    };
    v4 xv[12], dv[2];
    float mu[12], rsd[12];
    auto load_tile = [&](long long tile, int slot) {
#pragma unroll
        for (int p = 0; p < 12; ++p) {
            const int e = tid + kThreads * p;
            xv[p] = v4{0.f, 0.f, 0.f, 0.f};
            mu[p] = 0.f;
            rsd[p] = 1.f;
            if (p < per_thread && e < kDw1Rows * pieces) {
                const int r = e / pieces, q = e - r * pieces;
                const int k = k0 + 4 * q;
                xv[p] = load4_guard(a.rs.src + srow[slot * kDw1Rows + r] * din + k, din - k);
                mu[p] = sst[(slot * kDw1Rows + r) * 2];
                rsd[p] = sst[(slot * kDw1Rows + r) * 2 + 1];
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int e = tid + kThreads * p;       // 32 rows x 16 pieces
            const int r = e >> 4, q = e & 15;
            const long long gr = tile * kDw1Rows + r;
            dv[p] = gr < rows ? *reinterpret_cast<const v4*>(a.dz1 + gr * 64 + 4 * q) : v4{0.f, 0.f, 0.f, 0.f};
        }
    };

    long long tile = blockIdx.x;
    int slot = 0;
    if (tile < ntiles) rowinfo(tile, slot);
    __syncthreads();
    if (tile < ntiles) load_tile(tile, slot);
    for (; tile < ntiles; tile += gridDim.x) {
        const long long next = tile + gridDim.x;
        if (next < ntiles) rowinfo(next, slot ^ 1);
        __syncthreads();            // previous tile's operands consumed; next tile's row info visible
#pragma unroll
        for (int p = 0; p < 12; ++p) {
            const int e = tid + kThreads * p;
            if (p < per_thread && e < kDw1Rows * pieces) {
                const int r = e / pieces, q = e - r * pieces;
                const int k = k0 + 4 * q;
                v4 x = xv[p];
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) x[ee] = (k + ee < din) ? (x[ee] - mu[p]) * rsd[p] : 0.f;
                *reinterpret_cast<v4*>(xt + r * xs + 4 * q) = x;
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int e = tid + kThreads * p;
            *reinterpret_cast<v4*>(dzt + (e >> 4) * kWS + 4 * (e & 15)) = dv[p];
        }
        __syncthreads();
        if (next < ntiles) load_tile(next, slot ^ 1);
        slot ^= 1;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int r = 16 * h + s;
            const float a0 = dzt[r * kWS + c], a1 = dzt[r * kWS + 32 + c];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int nt = wave + 4 * i;
                if (nt < ntk) {
                    const float b = xt[r * xs + 32 * nt + c];
                    acc[i][0] = prim::mfma32(a0, b, acc[i][0]);
                    acc[i][1] = prim::mfma32(a1, b, acc[i][1]);
                }
            }
        }
    }
    float* prow = a.partials + (long long)blockIdx.x * 64 * din;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int nt = wave + 4 * i;
        const int k = k0 + 32 * nt + c;
        if (nt < ntk && k < din) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int f = 32 * t + (v & 3) + 8 * (v >> 2) + 4 * h;
                    prow[(long long)f * din + k] = acc[i][t][v];
                }
        }
    }
}

// out[e] = sum over n partial rows (row stride `stride`); deterministic order
__global__ void __launch_bounds__(kThreads) mlp_reduce_kernel(const float* partials, long long n, long long stride,
                                                              long long count, float* out) {
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < count; e += (long long)gridDim.x * kThreads) {
        float s = 0.f;
        for (long long r = 0; r < n; ++r) s += partials[r * stride + e];
        out[e] = s;
    }
}

// ================================================================== per-row input statistics ====
// stats[r] = {mean, 1 / sqrt(var + eps)} of src[r, :] (population variance, like nn.LayerNorm: mlp.py:47-48).  The
// observation fields of the rollout buffer do not change during the ppo epochs, so this runs once per train() and the
// trunk kernels standardise rows on the fly from 8 bytes per row.  16 lanes per row.
__global__ void __launch_bounds__(kThreads) row_stats_kernel(const float* src, long long rows, int D, float eps,
                                                             float* stats) {
    const int sub = threadIdx.x & 15;
    const long long groups = ((long long)gridDim.x * kThreads) >> 4;
    const long long first = ((long long)blockIdx.x * kThreads + threadIdx.x) >> 4;
    const long long trips = (rows + groups - 1) / groups;       // the same for every lane: sum16 is a wave collective
    for (long long it = 0; it < trips; ++it) {
        const long long r = first + it * groups;
        const bool ok = r < rows;
        const float* p = src + (ok ? r : rows - 1) * D;
        float s = 0.f;
        for (int k = 4 * sub; k < D; k += 64) {
            const v4 x = load4_guard(p + k, D - k);
            s += (x[0] + x[1]) + (x[2] + x[3]);
        }
        s = prim::sum16(s);
        const float mean = s / (float)D;
        float q = 0.f;
        for (int k = 4 * sub; k < D; k += 64) {
            const v4 x = load4_guard(p + k, D - k);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k + e < D) {
                    const float d = x[e] - mean;
                    q += d * d;
                }
        }
        q = prim::sum16(q);
        if (ok && sub == 0) {
            stats[2 * r] = mean;
            stats[2 * r + 1] = 1.f / sqrtf(q / (float)D + eps);
        }
    }
}

// ================================================================== host side ====
inline bool net_ok(const mappo_mlp_t* m) {
    if (m->n_layers < 1 || m->n_layers > MAPPO_MLP_MAX_LAYERS || m->din <= 0 || m->out < 0 || m->out > 64) return false;
    if (m->act < 0 || m->act > 2) return false;
    return true;
}

inline int fill(const mappo_mlp_t* m, RowSrc& rs, Net& n) {
    if (!m || !m->src || !m->w1) return MAPPO_E_NULL;
    if (!net_ok(m) || m->rows <= 0) return MAPPO_E_SHAPE;
    if (m->chunk_len > 0 && (!m->idx || m->mb <= 0 || m->T <= 0 || m->N <= 0 || m->A <= 0 ||
                             m->rows != m->mb * (int64_t)m->chunk_len))
        return MAPPO_E_SHAPE;
    rs.src = m->src;
    rs.stats = m->row_stats;
    rs.idx = reinterpret_cast<const long long*>(m->idx);
    rs.rows = m->rows;
    rs.mb = m->mb;
    rs.chunk_len = m->chunk_len;
    rs.T = m->T;
    rs.N = m->N;
    rs.A = m->A;
    rs.din = m->din;
    n.din = m->din;
    n.L = m->n_layers;
    n.act = m->act;
    n.out = m->out;
    n.eps = m->ln_eps;
    n.w1 = m->w1;
    for (int l = 0; l < 3; ++l) {
        n.bias[l] = m->bias[l];
        n.ln_g[l] = m->ln_g[l];
        n.ln_b[l] = m->ln_b[l];
        if (l < m->n_layers && (!m->bias[l] || !m->ln_g[l] || !m->ln_b[l])) return MAPPO_E_NULL;
    }
    for (int l = 0; l < 2; ++l) {
        n.w2[l] = m->w2[l];
        if (l + 1 < m->n_layers && !m->w2[l]) return MAPPO_E_NULL;
    }
    n.wh = m->wh;
    n.bh = m->bh;
    if (m->out > 0 && (!m->wh || !m->bh)) return MAPPO_E_NULL;
    return 0;
}

inline long long ceil_div(long long a, long long b) { return (a + b - 1) / b; }

inline int forward(const mappo_mlp_t* m, hipStream_t stream) {
    FwdArgs a;
    int code = fill(m, a.rs, a.net);
    if (code) return code;
    if (!m->y) return MAPPO_E_NULL;
    a.y = m->y;
    for (int l = 0; l < 3; ++l) a.z[l] = l < m->n_layers ? m->z[l] : nullptr;
    const FwdLds o = fwd_lds(m->n_layers, m->out);
    long long grid = ceil_div(m->rows, kTR);
    if (grid > kFwdGridCap) grid = kFwdGridCap;
    MAPPO_LAUNCH(mlp_fwd_kernel, (unsigned)grid, kThreads, (size_t)o.total * 4, stream, a);
    return MAPPO_LAUNCH_ERROR();
}

inline long long workspace_floats(int din, int n_layers, int out) {
    return (long long)kBwdGridCap * 4 * p_main(n_layers, out) + (long long)kDw1GridCap * 64 * din;
}

inline int backward(const mappo_mlp_t* m, hipStream_t stream) {
    BwdArgs b;
    int code = fill(m, b.rs, b.net);
    if (code) return code;
    if (!m->dy || !m->dz1 || !m->workspace || !m->grads) return MAPPO_E_NULL;
    for (int l = 0; l < 3; ++l) {
        b.z[l] = l < m->n_layers ? m->z[l] : nullptr;
        if (l < m->n_layers && !m->z[l]) return MAPPO_E_NULL;
    }
    const int L = m->n_layers, out = m->out, din = m->din;
    b.dy = m->dy;
    b.dz1 = m->dz1;
    b.partials = m->workspace;
    const BwdLds o = bwd_lds(L, out);
    long long grid = ceil_div(m->rows, kTR);
    if (grid > kBwdGridCap) grid = kBwdGridCap;
    if (L == 1) {
        MAPPO_LAUNCH(mlp_bwd_kernel<1>, (unsigned)grid, kThreads, (size_t)o.total * 4, stream, b);
    } else if (L == 2) {
        MAPPO_LAUNCH(mlp_bwd_kernel<2>, (unsigned)grid, kThreads, (size_t)o.total * 4, stream, b);
    } else {
        MAPPO_LAUNCH(mlp_bwd_kernel<3>, (unsigned)grid, kThreads, (size_t)o.total * 4, stream, b);
    }
    code = MAPPO_LAUNCH_ERROR();
    if (code) return code;

    Dw1Args d;
    d.rs = b.rs;
    d.dz1 = m->dz1;
    d.partials = m->workspace + (long long)kBwdGridCap * 4 * p_main(L, out);
    long long gx = ceil_div(m->rows, kDw1Rows);
    if (gx > kDw1GridCap) gx = kDw1GridCap;
    const int gy = (int)ceil_div(din, kDw1Slab);
    if (gx * gy > kDw1GridCap) gx = kDw1GridCap / gy > 0 ? kDw1GridCap / gy : 1;
    const size_t dw1_lds = ((size_t)kDw1Rows * (kDw1Slab + 4) + kDw1Rows * kWS) * 4 + 2 * kDw1Rows * 8 + 2 * kDw1Rows * 8;
    MAPPO_LAUNCH(mlp_dw1_kernel, dim3((unsigned)gx, (unsigned)gy), kThreads, dw1_lds, stream, d);
    code = MAPPO_LAUNCH_ERROR();
    if (code) return code;

    // every slab's workgroups write disjoint k columns of their partial row; rows of unused workgroups do not exist
    const long long pm = p_main(L, out);
    MAPPO_LAUNCH(mlp_reduce_kernel, (unsigned)ceil_div(64LL * din, kThreads), kThreads, 0, stream, d.partials, gx,
                 64LL * din, 64LL * din, m->grads);
    MAPPO_LAUNCH(mlp_reduce_kernel, (unsigned)ceil_div(pm, kThreads), kThreads, 0, stream, b.partials, grid * 4, pm, pm,
                 m->grads + 64LL * din);
    return MAPPO_LAUNCH_ERROR();
}

inline int row_stats(const float* src, long long rows, int D, float eps, float* stats, hipStream_t stream) {
    if (!src || !stats) return MAPPO_E_NULL;
    if (rows <= 0 || D <= 0) return MAPPO_E_SHAPE;
    long long grid = ceil_div(rows * 16, kThreads);
    if (grid > 256 * 8) grid = 256 * 8;
    MAPPO_LAUNCH(row_stats_kernel, (unsigned)grid, kThreads, 0, stream, src, rows, D, eps, stats);
    return MAPPO_LAUNCH_ERROR();
}

}  // namespace mlp
#endif
