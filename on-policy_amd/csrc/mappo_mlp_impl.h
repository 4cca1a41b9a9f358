// Fused actor / critic trunk for hidden width 64 (K9): sampler gather + input standardisation + Linear + act + LayerNorm
// chain + head in one forward kernel; the matching backward as two kernels (row-parallel chain, first-layer weight
// gradient as a gather-fused split-K GEMM in three forms: direct-to-LDS loads for wide aligned inputs, row-split for
// narrow ones, loader waves for the rest).  Reference maths: onpolicy/algorithms/utils/mlp.py:6-58 (MLPLayer / MLPBase:
// [LayerNorm] -> (Linear -> Tanh|ReLU -> LayerNorm) x (1 + layer_N)), act.py:44-60 / distributions.py:55-68 (Categorical
// head = Linear), r_actor_critic.py:147-175 (v_out = Linear(hidden, 1)), rows drawn by the samplers of
// onpolicy/utils/shared_buffer.py:340-400 (feed-forward) and :499-608 (recurrent chunks).
//
// Why: at the north-star size the update was bound by [rows, 64] activation round trips between separate GEMM, bias,
// activation and LayerNorm launches (47 % library GEMMs + 23 % LayerNorm passes + 17 % gathers of a 528 ms step).  Here a
// row tile stays in registers from the gathered observation to the head's output; the only per-row HBM traffic is the
// observation row itself, the saved normalised activations ([64] per row and layer) and the head output.  All products run on the f32 MFMA
// (v_mfma_f32_32x32x2_f32: exact float32 fma chains, 64 flop / clk / SIMD), so this path is MFMA-bound, not HBM-bound.
//
// This header is the whole implementation, written against a handful of primitives (prim::mfma32, prim::xhalf,
// prim::lds, prim::load_lds16 / load_lds4 / wait_lds_loads for the direct-to-LDS loads, prim::wave_sync,
// __syncthreads, MAPPO_LAUNCH).  mappo_mlp.hip binds them to gfx950; tests/simt/ binds them to a host SIMT
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

#include <stdio.h>
#include <stdlib.h>

#include "../../include/mappo_hip.h"

// (the rest of the library is built with -ffp-contract=off because the GAE scan must round like numpy; nothing here is
// compared bit for bit with a CPU evaluation, and fused multiply-adds are a quarter of these kernels' VALU instructions)
#ifndef MAPPO_MLP_NO_CONTRACT
#pragma clang fp contract(fast)
#endif

namespace mlp {

constexpr int kH = 64;          // hidden width
constexpr int kKC = 32;         // first-layer k chunk of the forward (128-byte pieces of an observation row)
constexpr int kXS = kKC + 4;    // LDS row stride of its chunk tiles (floats): 16-byte slot (9 row + piece) mod 16, so 16
                                // consecutive rows at one piece, or rows r and r + 8 at all 8 pieces, cover all 64 banks
constexpr int kTR = 128;        // rows per workgroup tile = 4 waves x 32
constexpr int kWS = 68;         // LDS row stride of a permuted 64-wide weight row
constexpr int kTS = 36;         // LDS row stride of the [feature][32 rows] transposes of the backward
constexpr int kThreads = 256;
constexpr int kPipeThreads = 512;    // weight-gradient kernel: 4 compute waves + 4 loader waves
constexpr int kDepth = 3;            // tile loads a loader thread of that kernel keeps in flight (register buffers)
constexpr int kFwdThreads = 512;     // forward kernel: 4 compute waves + 4 loader waves (waves go to the SIMDs round robin:
                                     // every SIMD gets one compute and one loader wave of each workgroup)
constexpr int kFwdDepth = 3;         // chunk loads a loader thread of the forward keeps in flight
constexpr int kFwdGridCap = 512;     // 2 workgroups per CU (each 2 LDS stages of 27 KB + 19 KB of parameters; <= 128 VGPRs):
                                     // while one workgroup's wave on a SIMD runs a layer tail (VALU), the other's feeds the MFMA
constexpr int kBwdGridCap = 256;     // 1 workgroup per CU
constexpr int kDw1Rows = 32;         // rows per iteration of the first-layer weight-gradient kernel
constexpr int kDw1Slab = 384;        // k columns per workgroup of that kernel (6 accumulator tiles per wave)
constexpr int kDw1GridCap = 256;

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

// The rows of a launch: row r reads row srow[r] of `src`.  The table is the sampler's row map (shared_buffer.py:379-396
// rows mode, :554-604 chunk mode) resolved once per minibatch by rowtab_kernel (int32, padded to the 128-row tile with
// copies of the last row so that no kernel clamps an index): the hot kernels do one coalesced table load per tile instead
// of a dependent idx chain with 64-bit divisions.  Rows are used as they are: a network with an input LayerNorm reads
// from a standardised copy of the observation field (standardize_rows_kernel, once per train()).
struct RowSrc {
    const float* src;
    const int* srow;        // [rows128]
    long long rows;
    int din;
};
__host__ __device__ __forceinline__ long long rows128(long long rows) { return (rows + 127) & ~127LL; }

struct RowMapArgs {
    const long long* idx;
    long long rows, mb;
    int chunk_len, T, N, A;
    int* srow;
};

__global__ void __launch_bounds__(kThreads) rowtab_kernel(RowMapArgs m) {
    const long long padded = rows128(m.rows);
    for (long long rr = (long long)blockIdx.x * kThreads + threadIdx.x; rr < padded; rr += (long long)gridDim.x * kThreads) {
        const long long r = rr < m.rows ? rr : m.rows - 1;
        long long sr = r;
        if (m.idx != nullptr) {
            if (m.chunk_len <= 0) {
                sr = m.idx[r];
            } else {
                const long long l = r / m.mb, j = r - l * m.mb;
                const long long f = m.idx[j] * m.chunk_len + l;
                const long long at = (long long)m.A * m.T;
                const long long n = f / at, rem = f - n * at;
                const long long ag = rem / m.T, t = rem - ag * m.T;
                sr = (t * m.N + n) * m.A + ag;
            }
        }
        m.srow[rr] = (int)sr;
    }
}

// 4 consecutive floats of a row of `din` floats starting at column k, zero beyond the row's end, in two halves so that
// the software pipelines below can separate issue from use.  Rows are only 4-byte aligned for odd widths (the unaligned
// 16-byte load is legal on gfx950).  piece_at(): where to load from -- a piece that sticks out of the row is fetched
// from the row's last four floats (din >= 4) -- always exactly ONE load instruction and no branch, because the compiler's
// wait counts are only exact if every path issues the same loads.  shift4(): applied when the data is consumed; `sft` =
// k - piece_at(k) is 0 for an interior piece, 1..3 on the row's tail, >= 4 past the end.
__device__ __forceinline__ int piece_at(int k, int din) { return k > din - 4 ? din - 4 : k; }
__device__ __forceinline__ v4 shift4(v4 v, int sft) {
    if (sft == 0) return v;
    v4 r;
    r[0] = sft == 1 ? v[1] : sft == 2 ? v[2] : sft == 3 ? v[3] : 0.f;
    r[1] = sft == 1 ? v[2] : sft == 2 ? v[3] : 0.f;
    r[2] = sft == 1 ? v[3] : 0.f;
    r[3] = 0.f;
    return r;
}
// (guarded variant for code outside the pipelines)
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

// tanh on the hardware exp2 / rcp (the library tanhf was 60 % of the forward kernel's time: 64 calls per row).
// Round 4: tanh(x) = sign(x) (1 - e) / (1 + e) with e = exp(-2 |x|) in (0, 1] -- 7 instructions, two of them transcendental.
// Absolute error <= ~1.5e-7 everywhere (1 ulp of exp2 and of rcp on values <= 1, one rounding of 1 - e); for |x| << 1 the
// RELATIVE error grows like 6e-8 / |x| (1 - e cancels) -- harmless here: the activation feeds a LayerNorm, which subtracts a
// mean and scales by O(1), so only absolute errors propagate, and the backward differentiates through the activation OUTPUT
// (1 - a^2).  The counters of round 4 show VALU and MFMA cycles of a SIMD adding up (the matrix pipe's busy share is
// MFMA / (MFMA + VALU) to within 10 % for the MFMA-heavy critic AND the VALU-heavy actor), so every instruction of the 128 tanh
// per row is paid for in full.  MAPPO_TANH_POLY keeps rounds 2-3's form: 1 - 2 / (exp(2|x|) + 1) above 0.25 and the odd Taylor
// polynomial up to x^7 below, selected per element (16 instructions; 4e-7 relative everywhere).
__device__ __forceinline__ float fast_tanh(float x) {
#ifdef MAPPO_TANH_POLY
    const float ax = fabsf(x), x2 = x * x;
    const float e = prim::exp2_fast(ax * 2.8853900817779268f);
    const float big = 1.f - 2.f * prim::rcp_fast(e + 1.f);
    const float small = ax * (1.f + x2 * (-0.33333333333f + x2 * (0.13333333333f + x2 * -0.05396825397f)));
    return copysignf(ax < 0.25f ? small : big, x);
#else
    const float e = prim::exp2_fast(fabsf(x) * -2.8853900817779268f);
    return copysignf((1.f - e) * prim::rcp_fast(1.f + e), x);
#endif
}
// The activation is a template parameter of the kernels: a run-time switch inside the unrolled per-feature loops costs
// several scalar branches per element (measured: 13 k of a tile's 20 k epilogue cycles).
template <int ACT>
__device__ __forceinline__ float act_fn(float z) {
    return ACT == 1 ? fast_tanh(z) : (ACT == 2 ? fmaxf(z, 0.f) : z);
}
// Two elements at a time: gfx950 has packed f32 multiply / add / fma (two lanes' worth of work per issue slot), but left to
// itself the compiler packs only part of a layer tail (round 4, from the ISA: scalar adds for 1 + e and for both LayerNorm
// sums, the sum of squares as packed multiplies + 32 scalar adds, an IEEE square root and division per row: ~450 vector
// instructions per 32-feature tail, ~350 with these forms).  exp2(-|m|) takes its sign handling as source modifiers.
typedef float f2 __attribute__((ext_vector_type(2)));
// Streaming accesses of the K9 kernels (build-time bit mask MAPPO_K9_NT; round 6 default 30).  Every saved activation and every
// dz1 row is touched once per launch and the next touch is a launch and >= 7 GB of other traffic away, so nothing of this is
// worth a place in L2 / the Infinity Cache:
//    2 = the forward's saved activations / statistics as non-temporal stores     (forward launch -1.3 %)
//    4 = the direct-to-LDS loads of the first-layer weight-gradient kernels with the nt bit
//    8 = the chain's saved-activation loads non-temporal                         (4 + 8: backward call -0.8 %)
//   16 = the chain's dz1 rows as non-temporal stores
//    1 = the forward's input rows as non-temporal loads -- NOT set: + 24 % on the forward launch (a lane loads 16-byte pieces
//        of 32 different rows and needs L2 to merge the eight pieces of a 128-byte line it asks for one after the other)
// North star 200.5-201.5 -> 198.1-199.3 ms with 30 on one box, config 2 -3.5 %, config 3 / recurrent north star / the
// 512-thread shard -1 %, the 64-thread SMAC shard unchanged (profiles/r06_ab_k9_streaming_hints.json).
#ifndef MAPPO_K9_NT
#define MAPPO_K9_NT 30
#endif
template <int BIT, typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
    if ((MAPPO_K9_NT) & BIT) return __builtin_nontemporal_load(p);
    return *p;
}
template <int BIT, typename T>
__device__ __forceinline__ void st_stream(T* p, T v) {
    if ((MAPPO_K9_NT) & BIT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <int ACT>
__device__ __forceinline__ f2 act_fn2(f2 z) {
    if (ACT == 1) {
#ifdef MAPPO_TANH_POLY
        return f2{fast_tanh(z[0]), fast_tanh(z[1])};
#else
        const f2 m = z * 2.8853900817779268f;
        const f2 e = {prim::exp2_fast(-fabsf(m[0])), prim::exp2_fast(-fabsf(m[1]))};
        const f2 num = 1.f - e, den = 1.f + e;
        const f2 r = {prim::rcp_fast(den[0]), prim::rcp_fast(den[1])};
        const f2 t = num * r;
        return f2{copysignf(t[0], z[0]), copysignf(t[1], z[1])};
#endif
    }
    if (ACT == 2) return __builtin_elementwise_max(z, f2{0.f, 0.f});
    return z;
}
// LayerNorm statistics of this lane's row from its 32 activations a[] (the other 32 sit in the partner half-wave): on
// return a[] holds a - mean.  Two partial sums each (even / odd slots), packed.
__device__ __forceinline__ void ln_stats32(float* a, float eps, float& mean, float& rstd) {
    f2 s2 = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 32; s += 2) s2 += f2{a[s], a[s + 1]};
    float sum = s2[0] + s2[1];
    sum += prim::xhalf(sum);
    mean = sum * (1.f / 64.f);
    f2 v2 = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 32; s += 2) {
        const f2 d = f2{a[s], a[s + 1]} - mean;
        a[s] = d[0];
        a[s + 1] = d[1];
        v2 += d * d;
    }
    float var = v2[0] + v2[1];
    var += prim::xhalf(var);
    rstd = prim::rsq_fast(var * (1.f / 64.f) + eps);
}
// derivative from the activation output a (tanh) / the pre-activation z (ReLU)
template <int ACT>
__device__ __forceinline__ float act_grad(float z, float a) {
    return ACT == 1 ? 1.f - a * a : (ACT == 2 ? (z > 0.f ? 1.f : 0.f) : 1.f);
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
    int vec, w2p, whp, bh, stage, total;
};
constexpr int kStageX = kTR * kXS;              // floats: [128 rows][32 k], row stride kXS
constexpr int kStageW = 64 * kXS;               // first-layer weight chunk [64 features][32 k], same stride
constexpr int kStage = kStageX + kStageW;
__host__ __device__ __forceinline__ FwdLds fwd_lds(int L, int out) {
    FwdLds o;
    o.vec = 0;                                // [L][bias | g | beta][64]
    o.w2p = o.vec + 192 * L;                  // [L - 1][2][32][kWS]
    o.whp = o.w2p + (L - 1) * 2 * 32 * kWS;   // [out][64] permuted
    o.bh = o.whp + out * 64;                  // [out]
    o.stage = (o.bh + out + 3) & ~3;          // [2][kStage]
    o.total = o.stage + 2 * kStage;
    return o;
}

// Parameters that every tile needs, staged once per workgroup.  w2p[l-1][t][i][h * 32 + s] = W2_l[32 t + i][f(h, s)]:
// the A operand (output feature 32 t + i on lane i) of the step that consumes slot s of the previous layer's registers.
__device__ __forceinline__ void stage_fwd_params(const Net& n, float* lds, const FwdLds& o) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int e = tid; e < 192 * n.L; e += nthr) {
        const int l = e / 192, q = (e - 192 * l) >> 6, c = e & 63;
        const float* p = q == 0 ? n.bias[l] : (q == 1 ? n.ln_g[l] : n.ln_b[l]);
        lds[o.vec + e] = p[c];
    }
    for (int l = 1; l < n.L; ++l)
        for (int e = tid; e < 64 * 64; e += nthr) {
            const int fo = e >> 6, hs = e & 63;          // output feature, (h, s)
            const int k = feat_of(hs >> 5, hs & 31);
            lds[o.w2p + (l - 1) * 2 * 32 * kWS + fo * kWS + hs] = n.w2[l - 1][fo * 64 + k];
        }
    for (int e = tid; e < n.out * 64; e += nthr) {
        const int oo = e >> 6, hs = e & 63;
        lds[o.whp + e] = n.wh[oo * 64 + feat_of(hs >> 5, hs & 31)];
    }
    for (int e = tid; e < n.out; e += nthr) lds[o.bh + e] = n.bh[e];
}

// One layer's tail on the accumulators: z = acc + bias (optionally kept for saving), a = act(z), LayerNorm over the
// row's 64 features (mlp.py:17-22).  In: acc[2] (this lane's 32 slots).  Out: hreg[32] = LayerNorm output.
// KEEP: what the backward needs of this layer is the NORMALISED activation nhat (-> nreg) and the row's (mean, rstd):
// the activation output is a = nhat / rstd + mean, its derivative follows from a (tanh: 1 - a^2, ReLU: a > 0), so the
// backward never re-evaluates tanh or the LayerNorm statistics (a third of its instructions when it saved z instead).
template <bool KEEP, int ACT>
__device__ __forceinline__ void layer_tail(const f32x16* acc, const float* vec /* bias | g | beta in LDS */, int h,
                                           float eps, float* hreg, float* ztile /* KEEP: wave-uniform, see load_frag64 */,
                                           int lane, float& mean_out, float& rstd_out) {
    float a[32];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4 b = *reinterpret_cast<const v4*>(vec + 32 * t + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const int s = 16 * t + 4 * q + e;
                const f2 av = act_fn2<ACT>(f2{acc[t][4 * q + e], acc[t][4 * q + e + 1]} + f2{b[e], b[e + 1]});
                a[s] = av[0];
                a[s + 1] = av[1];
            }
        }
    float mean, rstd;
    if (ACT != 1) {
        // (this kernel runs at 128 registers: the packed form's aligned register pairs cost the ReLU / identity instances a
        // spill, so they keep one element per instruction here)
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) sum += a[s];
        sum += prim::xhalf(sum);
        mean = sum * (1.f / 64.f);
        float var = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            a[s] -= mean;
            var += a[s] * a[s];
        }
        var += prim::xhalf(var);
        rstd = prim::rsq_fast(var * (1.f / 64.f) + eps);
    } else {
        ln_stats32(a, eps, mean, rstd);
    }
    mean_out = mean;
    rstd_out = rstd;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4 g = *reinterpret_cast<const v4*>(vec + 64 + 32 * t + 8 * q + 4 * h);
            const v4 be = *reinterpret_cast<const v4*>(vec + 128 + 32 * t + 8 * q + 4 * h);
            v4 nh;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int s = 16 * t + 4 * q + e;
                nh[e] = a[s] * rstd;
                hreg[s] = nh[e] * g[e] + be[e];
            }
            // (fragment order, see zfrag(): block 4 t + q) stored as they are produced
            if (KEEP) *reinterpret_cast<v4*>(ztile + 4 * lane + 256 * (4 * t + q)) = nh;
        }
}

// The saved activations z[l] travel between the forward and the backward kernel only, so they are kept in the order the
// registers hold them ("fragment order"): per group of 32 consecutive launch rows (one wave tile) 8 blocks of 1 KB,
// block 4 t + q = slots 16 t + 4 q .. + 3 of all 64 lanes, lane (h, c) at 16 (32 h + c) bytes.  Every store / load
// instruction of a wave then moves 1 KB of consecutive memory (row-major [rows, 64] made each of them touch 32 separate
// 32-byte segments: a quarter of the backward kernel's time).  The buffers hold rows128(rows) rows.
// A wave addresses its tile as (wave-uniform) z + 2048 * (first row / 32) plus 4 * lane: lane = 32 h + (row & 31).
__device__ __forceinline__ void load_frag64(const float* ztile /* z + 2048 * (tile row / 32), wave-uniform */, int lane,
                                            float* reg) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const v4 o = ld_stream<8>(reinterpret_cast<const v4*>(ztile + 4 * lane + 256 * b));
#pragma unroll
        for (int e = 0; e < 4; ++e) reg[4 * b + e] = o[e];
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
    v4 a0n = *reinterpret_cast<const v4*>(w0), a1n = *reinterpret_cast<const v4*>(w1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const v4 a0 = a0n, a1 = a1n;
        if (q < 7) {        // next group's operands in flight behind this group's MFMAs
            a0n = *reinterpret_cast<const v4*>(w0 + 4 * q + 4);
            a1n = *reinterpret_cast<const v4*>(w1 + 4 * q + 4);
        }
        prim::sched_fence();
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
    float* z[3];        // saved normalised activations per layer, rows128(rows) rows in fragment order (NULL: inference)
    float* st[3];       // saved {mean, rstd} per layer, [rows128(rows), 2]
    long long* dbg;     // tuning hook (mappo_mlp_set_debug): cycle stamps of workgroup 0's first iterations, or NULL
    int flags;          // tuning hook (MAPPO_MLP_FLAGS): 1 = compute waves keep the default priority
};

// Workgroup = 4 compute waves (one per SIMD: MFMA + the layer tails of 32 rows each) + 4 loader waves that do nothing
// but move data: global -> registers (kFwdDepth chunk loads in flight per thread, across tile boundaries) -> one of two
// LDS stages.  One barrier per chunk hands a stage over in both directions.  Two such workgroups share a CU: their
// barriers are independent, so the MFMA stream of one runs under the layer tails (VALU, transcendental, store work) of
// the other.
typedef int i4 __attribute__((ext_vector_type(4)));
// One chunk in flight in a loader thread's registers + what the buffer's NEXT load needs: the source rows are fetched
// right after this chunk's loads were issued, i.e. three issues ahead of their use, so that waiting for them never
// waits for younger data loads (the wait counter is in order).
struct ChunkBuf {
    v4 xv[4], wv[2];
    int sft;
    i4 rws;             // source rows of the 4 rows this thread serves in the tile of the buffer's next chunk
};

// ALIGNED: din % 4 == 0 -- a 16-byte piece is either inside the row or past its end, so the loaders need no tail shifting
// (their VALU work shares the SIMD's issue slots and the chip's power budget with the MFMA stream).
template <int ACT, bool ALIGNED>
__global__ void __launch_bounds__(kFwdThreads, 4) mlp_fwd_kernel(FwdArgs a) {
    float* lds = prim::lds();
    const Net& n = a.net;
    const FwdLds o = fwd_lds(n.L, n.out);
    stage_fwd_params(n, lds, o);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
    const int din = n.din;
    const int nch = (din + kKC - 1) / kKC;
    const long long rows = a.rs.rows;
    const long long ntiles = (rows + kTR - 1) / kTR;
    const long long my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long long n_it = my_tiles * nch;
    float* stage0 = lds + o.stage;

    if (wave >= 4) {
        // ------------------------------------------------------------ loader
        // thread = (group xg of 4 consecutive rows, 16-byte piece xq of the chunk): one wave instruction reads 8 rows x
        // 128 contiguous bytes and the thread's 4 source rows are one 16-byte table load.  Row groups are numbered so
        // that the two groups of a 16-lane store differ by 8 rows (LDS banks, see kXS); same for the 2 weight rows.
        const int lt = tid - 256, xg = lt >> 3, xq = lt & 7;
        const int xr0 = 16 * (xg >> 2) + 8 * (xg & 1) + 4 * ((xg >> 1) & 1);
        const int wr0 = ((xg >> 1) & 7) + 8 * (xg & 1) + 16 * (xg >> 4);      // + 32 p
        // row offset of the tile of a pipeline position, clamped to the last tile (chunks issued past the end are never
        // stored; rows are padded to the 128-row tile, so every table index is valid).  Positions are tracked as (tile,
        // chunk) counters: no division in the loop.
        auto tile_row0 = [&](long long ti) {
            if (ti >= my_tiles) ti = my_tiles - 1;
            return (blockIdx.x + ti * gridDim.x) * kTR;
        };
        auto fetch_rows = [&](long long ti, ChunkBuf& B) {
            B.rws = *reinterpret_cast<const i4*>(a.rs.srow + tile_row0(ti) + xr0);
        };
        long long it_tile = 0, ft_tile = 0;     // tile of the next issue / of the issue kFwdDepth later
        int it_kc = 0, ft_kc = 0;               // its chunk within the tile
        auto advance = [&](long long& t, int& kc) {
            if (++kc == nch) {
                kc = 0;
                ++t;
            }
        };
        // every issue is exactly 4 + 2 + 1 loads, whatever the chunk
        auto issue = [&](ChunkBuf& B) {
            const int k = it_kc * kKC + 4 * xq, kk = piece_at(k, din);
            B.sft = k - kk;
            const i4 rw = B.rws;
            // (no conditional loads here: the compiler's wait counts are exact only if every path issues the same number)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                B.xv[p] = *reinterpret_cast<const v4u*>(a.rs.src + (long long)rw[p] * din + kk);
#pragma unroll
            for (int p = 0; p < 2; ++p)
                B.wv[p] = *reinterpret_cast<const v4u*>(n.w1 + (long long)(wr0 + 32 * p) * din + kk);
            fetch_rows(ft_tile, B);
            advance(it_tile, it_kc);
            advance(ft_tile, ft_kc);
        };
        auto store = [&](long long m, const ChunkBuf& B) {
            float* st = stage0 + (m & 1) * kStage;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                *reinterpret_cast<v4*>(st + (xr0 + p) * kXS + 4 * xq) =
                    ALIGNED ? (B.sft ? v4{0.f, 0.f, 0.f, 0.f} : B.xv[p]) : shift4(B.xv[p], B.sft);
#pragma unroll
            for (int p = 0; p < 2; ++p)
                *reinterpret_cast<v4*>(st + kStageX + (wr0 + 32 * p) * kXS + 4 * xq) =
                    ALIGNED ? (B.sft ? v4{0.f, 0.f, 0.f, 0.f} : B.wv[p]) : shift4(B.wv[p], B.sft);
        };
        ChunkBuf B0, B1, B2;
        if (n_it > 0) {
            fetch_rows(ft_tile, B0);
            advance(ft_tile, ft_kc);
            fetch_rows(ft_tile, B1);
            advance(ft_tile, ft_kc);
            fetch_rows(ft_tile, B2);
            advance(ft_tile, ft_kc);
            issue(B0);
            issue(B1);
            issue(B2);
            store(0, B0);
            issue(B0);
        }
        // iteration j (after its barrier): chunk j + 1 leaves its buffer for the stage the compute waves released at the
        // barrier, chunk j + 4 takes the buffer over
        const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && lt == 0;
        long long j = 0;
        for (; j + 3 <= n_it; j += 3) {
            __syncthreads();
            if (stamp && j < 60) {
                a.dbg[256 + 4 * j] = prim::clock();
                prim::wait_loads_14();
                a.dbg[256 + 4 * j + 3] = prim::clock();
            }
            store(j + 1, B1);
            if (stamp && j < 60) a.dbg[256 + 4 * j + 1] = prim::clock();
            issue(B1);
            if (stamp && j < 60) a.dbg[256 + 4 * j + 2] = prim::clock();
            __syncthreads();
            if (stamp && j < 60) a.dbg[256 + 4 * j + 4] = prim::clock();
            store(j + 2, B2);
            issue(B2);
            if (stamp && j < 60) a.dbg[256 + 4 * j + 6] = prim::clock();
            __syncthreads();
            if (stamp && j < 60) a.dbg[256 + 4 * j + 8] = prim::clock();
            store(j + 3, B0);
            issue(B0);
            if (stamp && j < 60) a.dbg[256 + 4 * j + 10] = prim::clock();
        }
        if (j < n_it) {
            __syncthreads();
            store(j + 1, B1);
            if (j + 1 < n_it) {
                __syncthreads();
                store(j + 2, B2);
            }
        }
        return;
    }
    // ---------------------------------------------------------------- compute
    // the loader waves issue a few hundred VALU / LDS-store instructions per chunk; without a priority they take issue
    // slots from the MFMA stream (measured 85 instead of 64 cycles per MFMA)
    if (!(a.flags & 1)) prim::set_priority_high();
    const int wave_u = prim::uniform(wave);
    const bool cstamp = a.dbg != nullptr && blockIdx.x == 0 && tid == 0;
    // (the first-layer accumulators are zeroed before the loop and at the end of every tile's tail: a test on "first chunk
    // of a tile" inside the loop compiled to 32 conditional moves per chunk)
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    long long ti = 0;
    int kc = 0;
    for (long long j = 0; j < n_it; ++j) {
        if (cstamp && j < 60) a.dbg[4 * j] = prim::clock();
        __syncthreads();
        if (cstamp && j < 60) a.dbg[4 * j + 1] = prim::clock();
        const float* st = stage0 + (j & 1) * kStage;
        const float* xa = st + (32 * wave + c) * kXS + 16 * h;
        const float* w0 = st + kStageX + c * kXS + 16 * h;
        const float* w1 = st + kStageX + (32 + c) * kXS + 16 * h;
        // The MFMA stream of a chunk: 32 MFMAs and 12 operand reads, nothing else -- a wave issues its instructions one
        // after the other, and every VALU instruction in this loop would cost ~8 of the 64 cycles an MFMA occupies
        // (tools/probes/probe_mfma.hip).  The operand reads of group q + 1 are issued before the MFMAs of group q.
        v4 bn = *reinterpret_cast<const v4*>(xa);
        v4 a0n = *reinterpret_cast<const v4*>(w0);
        v4 a1n = *reinterpret_cast<const v4*>(w1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4 b = bn, a0 = a0n, a1 = a1n;
            if (q < 3) {
                bn = *reinterpret_cast<const v4*>(xa + 4 * q + 4);
                a0n = *reinterpret_cast<const v4*>(w0 + 4 * q + 4);
                a1n = *reinterpret_cast<const v4*>(w1 + 4 * q + 4);
            }
            prim::sched_fence();        // keep the compiler from sinking these reads down to their first use
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = prim::mfma32(a0[e], b[e], acc[0]);
                acc[1] = prim::mfma32(a1[e], b[e], acc[1]);
            }
        }
        if (cstamp && j < 60) a.dbg[4 * j + 2] = prim::clock();
        if (++kc < nch) continue;
        kc = 0;
        // ---- the rest of the network on this lane's row.  Rows past the end of the launch are copies of the last row
        // (row table padding); z and the statistics are padded to the tile, so only the output store is conditional.
        const long long wrow0 = (blockIdx.x + ti * gridDim.x) * kTR + 32 * wave_u;      // first row of this wave's 32
        const long long row = wrow0 + c;
        ++ti;
        float hreg[32], mean, rstd;
        for (int l = 0; l < n.L; ++l) {
            if (l > 0) dense64(lds + o.w2p + (l - 1) * 2 * 32 * kWS, c, h, hreg, acc);
            if (a.z[l] != nullptr) {
                layer_tail<true, ACT>(acc, lds + o.vec + 192 * l, h, n.eps, hreg, a.z[l] + wrow0 * 64, lane, mean, rstd);
                *reinterpret_cast<f2*>(a.st[l] + 2 * wrow0 + 2 * c) = f2{mean, rstd};      // both half-waves hold the same pair
            } else {
                layer_tail<false, ACT>(acc, lds + o.vec + 192 * l, h, n.eps, hreg, nullptr, lane, mean, rstd);
            }
        }
        // (dead lanes hold the last row's values bit for bit -- its padding copies -- and store them to the last row)
        const long long yrow = row < rows ? row : rows - 1;
        if (n.out == 0) {
            store_row64(a.y + yrow * 64, hreg, h);
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
                a.y[yrow * n.out + oo] = p + lds[o.bh + oo];
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    }
}

// ================================================================== forward, version 3 (aligned rows, din <= 448) ====
// Same maths and the same saved activations as mlp_fwd_kernel; built around what the counters of round 3 say limits that
// kernel (matrix pipe 59-63 % busy): its loader waves.  Every LDS store / vector load a loader wave issues waits for a gap
// in the MFMA operand traffic of the compute wave on its SIMD (~400 cycles each beside an MFMA stream against ~60 alone),
// 13 of them per 32-column chunk, so two workgroups' loaders deliver a chunk every 4.5-5 k cycles against 2 x 2.05 k
// cycles of MFMA work.  Here nothing stands between HBM and the B operand:
//   * the B operand (activations: lane = row) comes STRAIGHT from global memory into registers: lane (c, h) loads the four
//     16-byte pieces 8 q + 4 h .. + 3 of its row's 32-column chunk -- element e of piece q is exactly the k index
//     32 kc + 8 q + 4 h + e that MFMA step (q, e) contracts over (columns in order, 8 per group q: a row's last chunk
//     runs only the groups that hold real columns, template parameter NQL).  A wave instruction touches 32 rows x 2
//     adjacent pieces; the four instructions of a chunk are issued back to back and cover whole 128-byte lines of the
//     (16-byte aligned) rows, so the lines are fetched once.  No LDS hop, no loader waves, no barriers: three chunks in
//     flight per wave in 48 registers, issued by the wave that consumes them right behind the MFMA block that freed the
//     buffer;
//   * the WHOLE first-layer weight matrix sits in LDS (64 x din floats: 96 KB for the 384-wide critic; 16-byte pieces
//     XOR-swizzled so that the A-operand reads are conflict free), staged once per workgroup;
//   * ONE workgroup of 8 waves per CU = two waves per SIMD with 256 registers each: while one is in its layer tails
//     (VALU, transcendentals, stores) the other streams MFMAs -- what the two workgroups per CU of mlp_fwd_kernel are for;
//   * no LDS reads in the layer tails: the bias is the accumulators' initial value (per-lane constants in registers), and
//     the LayerNorm's gamma / beta are folded into the weights that consume its output when they are staged
//     (W' = W diag(gamma), b' = b + W beta: hidden layers and head), so a tail is bias-free activation + statistics +
//     the fragment-order store of nhat, and nhat itself is the next layer's B operand;
//   * a head of >= 3 outputs runs on the MFMA too (outputs zero-padded to one 32-feature tile: 32 MFMAs and 8 operand reads
//     whatever the width, instead of 8 reads + 32 multiply-adds + a lane exchange per output).
constexpr int kF3MaxDin = 448;              // 14 chunks x 8 KB of first-layer weights (SMAC's padded 436-wide critic input)
constexpr int kF3MinDin = 4;                // every aligned width.  Until the last chunk of a row was shortened to its real columns
                                            // (NQL) inputs of <= 128 columns stayed on the loader / compute kernel: width 48
                                            // 0.775 against 0.745 ms per 2.6 M rows; with NQL 0.648 against 0.676, steps
                                            // -0.9 % (north star) / -1.3 % (config 3), profiles/r04_ab_forward.json call_12
constexpr int kF3GridCap = 256;             // one workgroup per CU

struct Fwd3Lds {
    int vec, w2p, whp, bh, w1, total;
};
__host__ __device__ __forceinline__ Fwd3Lds fwd3_lds(int L, int out, int nch, bool hid6 = false) {
    Fwd3Lds o;
    o.vec = 0;                                  // [L][bias' | g | beta][64]  (bias' = the folded bias of layers >= 1)
    o.w2p = o.vec + 192 * L;                    // [L - 1][2][32][kWS], gamma of the layer below folded in
                                                // (HID6, L = 2: three bf16 planes [plane][tile][k16 step][lane] x 16 B = 6144 floats)
    o.whp = o.w2p + (hid6 ? 6144 : (L - 1) * 2 * 32 * kWS);     // [32][kWS] permuted head weights (rows >= out are zero), gamma folded in
    o.bh = o.whp + 32 * kWS;                    // [32] folded head bias
    o.w1 = (o.bh + 32 + 3) & ~3;                // [nch][64 features][32 k], 16-byte pieces swizzled within a row
    o.total = o.w1 + nch * 2048;
    return o;
}
inline bool fwd3_takes(int din, int L, int out) {
    // (single-layer trunks -- layer_N = 0 -- stay on the loader / compute kernel: the one-layer tanh instance of version 3
    // spilled 148 bytes per lane and no shipped configuration uses it)
    return din % 4 == 0 && din >= kF3MinDin && din <= kF3MaxDin && out <= 32 && L >= 2 && L <= 3;
}

// The tail of one layer on accumulators that already hold z = W x + b: a = act(z), statistics, nhat in place (-> reg),
// fragment-order store of nhat (KEEP).  No LDS traffic.
template <bool KEEP, int ACT>
__device__ __forceinline__ void layer_tail_nhat(const f32x16* acc, float eps, float* reg, float* ztile, int lane,
                                                float& mean_out, float& rstd_out) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
            const f2 av = act_fn2<ACT>(f2{acc[t][v], acc[t][v + 1]});
            reg[16 * t + v] = av[0];
            reg[16 * t + v + 1] = av[1];
        }
    float mean, rstd;
    ln_stats32(reg, eps, mean, rstd);
    mean_out = mean;
    rstd_out = rstd;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        v4 nh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            reg[4 * b + e] *= rstd;
            nh[e] = reg[4 * b + e];
        }
        if (KEEP) st_stream<2>(reinterpret_cast<v4*>(ztile + 4 * lane + 256 * b), nh);     // block 4 t + q = slots 16 t + 4 q ..
    }
}

struct XBuf {
    v4 x[4];
};

// Eight waves: two per SIMD with 256 registers each -- three chunks in flight, the first layer's bias in registers.  (A
// 12-wave form -- three per SIMD, 168 registers, two chunks in flight -- was measured twice in round 4, for the wide critic
// inputs and for the narrow actor inputs: 0-1 % slower both times, profiles/r04_ab_forward.json; removed.)
template <int V>
struct mlp_int {
    static constexpr int value = V;
};
// NQL: groups of 8 real columns in a row's LAST chunk (1 .. 4).  MFMA step (q, e) of a chunk contracts k = 8 q + 4 h + e, so
// a row's columns are consumed in order, 8 per group q, and the last chunk runs only its NQL groups (round 4: the 48 wide
// actor input of the north star spent 16 of its 64 first-layer MFMAs per tile on the zero padding of its second chunk, the
// 18 wide one of config 3 8 of 32).  A template parameter, so that the shortened chunk is straight-line code.
__device__ __forceinline__ void split3(const float* x, bf8& p1, bf8& p2, bf8& p3);     // (with mlp_fwd4_kernel below)
// HID6 (two-layer trunks; option bit 4096: emulator-green at the end of round 4, device A / B pending): the hidden layer in the
// six-term bf16 form of mlp_fwd4_kernel -- its weights as three bf16 planes in LDS, the normalised activations split in the wave.
template <int L, int ACT, int NQL, bool HID6 = false>
__global__ void __launch_bounds__(64 * 8) mlp_fwd3_kernel(FwdArgs a) {
    static_assert(!HID6 || L == 2, "the six-term hidden layer is built for two-layer trunks");
    constexpr int kF3Waves = 8;
    constexpr bool BIAS_REGS = true;
    float* lds = prim::lds();
    const Net& n = a.net;
    const int din = n.din, out = n.out;
    const int nch = (din + 31) / 32;
    const Fwd3Lds o = fwd3_lds(L, out, nch, HID6);
    const int tid = threadIdx.x, lane = tid & 63, wave = prim::uniform(tid >> 6), c = lane & 31, h = lane >> 5;
    constexpr int kThr = 64 * kF3Waves;
    // ---- parameters, once per workgroup.  vec[l] = [bias of layer l with beta of layer l - 1 folded in | g | beta]
    for (int e = tid; e < 192 * L; e += kThr) {
        const int l = e / 192, q = (e - 192 * l) >> 6, f = e & 63;
        float v;
        if (q == 0) {
            v = n.bias[l][f];
            if (l > 0)
                for (int k = 0; k < 64; ++k) v += n.w2[l - 1][f * 64 + k] * n.ln_b[l - 1][k];
        } else {
            v = q == 1 ? n.ln_g[l][f] : n.ln_b[l][f];
        }
        lds[o.vec + e] = v;
    }
    // w2p[l-1][t][i][h * 32 + s] = gamma_{l-1}[k] W_l[32 t + i][k], k = f(h, s): A operand of the step that consumes slot s
    if (HID6) {
        for (int e = tid; e < 512; e += kThr) {
            const int t = e >> 8, cc = (e >> 3) & 31, hh = (e >> 2) & 1, j = e & 3;
            float w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = feat_of(hh, 8 * j + i);
                w[i] = n.w2[0][(32 * t + cc) * 64 + k] * n.ln_g[0][k];
            }
            bf8 p1, p2, p3;
            split3(w, p1, p2, p3);
            float* base = lds + o.w2p + (t * 4 + j) * 256 + (32 * hh + cc) * 4;
            *reinterpret_cast<bf8*>(base) = p1;
            *reinterpret_cast<bf8*>(base + 2048) = p2;
            *reinterpret_cast<bf8*>(base + 4096) = p3;
        }
    } else {
    for (int l = 1; l < L; ++l)
        for (int e = tid; e < 64 * 64; e += kThr) {
            const int fo = e >> 6, hs = e & 63;
            const int k = feat_of(hs >> 5, hs & 31);
            lds[o.w2p + (l - 1) * 2 * 32 * kWS + fo * kWS + hs] = n.w2[l - 1][fo * 64 + k] * n.ln_g[l - 1][k];
        }
    }
    for (int e = tid; e < 32 * 64; e += kThr) {
        const int oo = e >> 6, hs = e & 63;
        const int k = feat_of(hs >> 5, hs & 31);
        lds[o.whp + oo * kWS + hs] = oo < out ? n.wh[oo * 64 + k] * n.ln_g[L - 1][k] : 0.f;
    }
    for (int e = tid; e < 32; e += kThr) {
        float v = 0.f;
        if (e < out) {
            v = n.bh[e];
            for (int k = 0; k < 64; ++k) v += n.wh[e * 64 + k] * n.ln_b[L - 1][k];
        }
        lds[o.bh + e] = v;
    }
    // first-layer weights: piece p (k = 32 kc + 4 p ..) of feature row f at slot p ^ ((f >> 1) & 7); zero beyond din
    for (int e = tid; e < nch * 512; e += kThr) {
        const int kc = e >> 9, f = (e >> 3) & 63, p = e & 7;
        const int k = 32 * kc + 4 * p;
        v4 w = {0.f, 0.f, 0.f, 0.f};
        if (k < din) w = *reinterpret_cast<const v4u*>(n.w1 + (long long)f * din + k);       // (din % 4 == 0)
        *reinterpret_cast<v4*>(lds + o.w1 + kc * 2048 + f * 32 + 4 * (p ^ ((f >> 1) & 7))) = w;
    }
    __syncthreads();
    // per-lane constants: the (folded) bias of every layer in accumulator order -- the accumulators start from it
    // (layer 0 only: its accumulators are started in the chunk loop; the hidden layers' start values are read from LDS in
    // the tail, eight 16-byte reads each -- with all of them in registers the three-layer instances spilled)
    float biasr[32];
    if (BIAS_REGS) {
#pragma unroll
        for (int s = 0; s < 32; ++s) biasr[s] = lds[o.vec + feat_of(h, s)];
    }

    const long long rows = a.rs.rows;
    const long long ntiles = rows128(rows) / 32;        // (z / statistics are padded to the 128-row tile, like the table)
    const long long gw = (long long)blockIdx.x * kF3Waves + wave, nw = (long long)gridDim.x * kF3Waves;
    const long long my_tiles = gw < ntiles ? (ntiles - gw + nw - 1) / nw : 0;
    if (my_tiles == 0) return;
    // reader side: float offset of piece P = 2 q + h (columns 8 q + 4 h ..) of feature row c inside a [64][32] weight chunk
    int off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) off[q] = c * 32 + 4 * ((2 * q + h) ^ ((c >> 1) & 7));
    auto tile_of = [&](long long m) {           // launch tile of this wave's m-th tile (past the end: the last one again)
        if (m >= my_tiles) m = my_tiles - 1;
        return gw + m * nw;
    };
    // issue side: position = (local tile, chunk) as counters; the row of the tile being issued and of the one after it.
    // The lane's read pointer xp walks its row one chunk (32 floats) per issue, so that the four 16-byte loads of a chunk
    // are one 64-bit address and four immediate offsets (round 4: the per-load clamp of k -- four vector instructions per
    // load -- is taken only by the last chunk of a row whose width is not a multiple of 32).
    long long it_m = 0;
    int it_kc = 0;
    const bool ragged = (din & 31) != 0;
    const float* row_it = a.rs.src + (long long)a.rs.srow[tile_of(0) * 32 + c] * din;
    const float* xp = row_it + 4 * h;
    int sr_next = a.rs.srow[tile_of(1) * 32 + c];
    auto issue = [&](XBuf& B) {                 // always exactly 4 loads (+ 1 table load per tile)
        if (ragged && it_kc == nch - 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int k = 32 * it_kc + 8 * q + 4 * h;
                if (k > din - 4) k = din - 4;   // a piece past the row's end: finite data against zero weights (or never used)
                B.x[q] = ld_stream<1>(reinterpret_cast<const v4u*>(row_it + k));
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) B.x[q] = ld_stream<1>(reinterpret_cast<const v4u*>(xp + 8 * q));
        }
        xp += 32;
        if (++it_kc == nch) {
            it_kc = 0;
            ++it_m;
            row_it = a.rs.src + (long long)sr_next * din;
            xp = row_it + 4 * h;
            sr_next = a.rs.srow[tile_of(it_m + 1) * 32 + c];
        }
    };
    f32x16 acc[2];
    auto mfma_groups = [&](const XBuf& B, int kc, auto ngroups) {
        constexpr int NQ = decltype(ngroups)::value;
        const float* wt = lds + o.w1 + kc * 2048;
        v4 a0n = *reinterpret_cast<const v4*>(wt + off[0]);
        v4 a1n = *reinterpret_cast<const v4*>(wt + 1024 + off[0]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const v4 a0 = a0n, a1 = a1n;
            if (q + 1 < NQ) {
                a0n = *reinterpret_cast<const v4*>(wt + off[q + 1 < NQ ? q + 1 : 0]);
                a1n = *reinterpret_cast<const v4*>(wt + 1024 + off[q + 1 < NQ ? q + 1 : 0]);
            }
            prim::sched_fence();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = prim::mfma32(a0[e], B.x[q][e], acc[0]);
                acc[1] = prim::mfma32(a1[e], B.x[q][e], acc[1]);
            }
        }
    };
    auto mfma_chunk = [&](const XBuf& B, int kc) { mfma_groups(B, kc, mlp_int<4>{}); };
    auto mfma_last = [&](const XBuf& B, int kc) { mfma_groups(B, kc, mlp_int<NQL>{}); };
    auto init_acc = [&](int l) {
        if (BIAS_REGS && l == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = biasr[16 * t + v];
        } else {        // slots 16 t + 4 q .. + 3 are features 32 t + 8 q + 4 h .. + 3: eight 16-byte reads
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4 b = *reinterpret_cast<const v4*>(lds + o.vec + 192 * l + 32 * t + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][4 * q + e] = b[e];
                }
        }
    };

    XBuf B0, B1, B2;
    issue(B0);
    issue(B1);
    issue(B2);
    // tuning hook (mappo_mlp_set_debug): shader-clock stamps of the first tiles of two waves that share SIMD 0 of
    // workgroup 0 (waves 0 and 4): [tile start, chunk loop done, tail done] at dbg[64 w + 4 m ..]
    const bool cstamp = a.dbg != nullptr && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0;
    // consume a chunk, then refill its buffer with the chunk DEPTH positions ahead (past the end: the last tile again)
#define MAPPO_F3_STEP(B, KC) do { mfma_chunk(B, KC); issue(B); } while (0)
#define MAPPO_F3_LAST(B, KC) do { mfma_last(B, KC); issue(B); } while (0)
    for (long long m = 0; m < my_tiles;) {
        if (cstamp && m < 15) a.dbg[64 * (wave >> 2) + 4 * m] = prim::clock();
        // The ring of chunk buffers with STATIC names: the loop body covers DEPTH chunks, so no buffer is ever copied
        // (round 4: with a run-time ring index the compiler merged the buffers through 16 64-bit register moves per chunk).
        // A chunk count that is not a multiple of DEPTH rotates the names once per tile instead.
        // (the accumulators start from the first layer's bias HERE, right in front of the tile's first MFMAs, which then
        // take the bias registers as their C operand: no register moves)
        init_acc(0);
        int kc = 0;
        if (NQL == 4) {                         // every chunk is a full one
            for (; kc + 3 <= nch; kc += 3) {
                MAPPO_F3_STEP(B0, kc);
                MAPPO_F3_STEP(B1, kc + 1);
                MAPPO_F3_STEP(B2, kc + 2);
            }
            if (nch - kc == 1) {
                MAPPO_F3_STEP(B0, kc);
                const XBuf t = B0;
                B0 = B1;
                B1 = B2;
                B2 = t;
            } else if (nch - kc == 2) {
                MAPPO_F3_STEP(B0, kc);
                MAPPO_F3_STEP(B1, kc + 1);
                const XBuf t = B2;
                B2 = B1;
                B1 = B0;
                B0 = t;
            }
        } else {
            for (; kc + 3 < nch; kc += 3) {     // (strictly before the row's last chunk)
                MAPPO_F3_STEP(B0, kc);
                MAPPO_F3_STEP(B1, kc + 1);
                MAPPO_F3_STEP(B2, kc + 2);
            }
            if (nch - kc == 1) {
                MAPPO_F3_LAST(B0, kc);
                const XBuf t = B0;
                B0 = B1;
                B1 = B2;
                B2 = t;
            } else if (nch - kc == 2) {
                MAPPO_F3_STEP(B0, kc);
                MAPPO_F3_LAST(B1, kc + 1);
                const XBuf t = B2;
                B2 = B1;
                B1 = B0;
                B0 = t;
            } else {
                MAPPO_F3_STEP(B0, kc);
                MAPPO_F3_STEP(B1, kc + 1);
                MAPPO_F3_LAST(B2, kc + 2);
            }
        }
        if (cstamp && m < 15) a.dbg[64 * (wave >> 2) + 4 * m + 1] = prim::clock();
        // ---- the rest of the network on this lane's row (rows past the end of the launch are copies of the last row:
        // row-table padding; z and the statistics are padded to the tile, only the output store is conditional)
        const long long tile = gw + m * nw;
        const long long m_done = m;
        ++m;
        const long long row = tile * 32 + c;
        float nh[32], mean, rstd;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            if (l > 0 && HID6) {
                init_acc(l);
                const float* wt = lds + o.w2p + lane * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf8 wa[3][2];
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int t = 0; t < 2; ++t) wa[p][t] = *reinterpret_cast<const bf8*>(wt + p * 2048 + (t * 4 + j) * 256);
                    bf8 b1, b2, b3;
                    split3(nh + 8 * j, b1, b2, b3);
                    prim::sched_fence();
                    acc[0] = prim::mfma_bf16(wa[0][0], b3, acc[0]);
                    acc[1] = prim::mfma_bf16(wa[0][1], b3, acc[1]);
                    acc[0] = prim::mfma_bf16(wa[2][0], b1, acc[0]);
                    acc[1] = prim::mfma_bf16(wa[2][1], b1, acc[1]);
                    acc[0] = prim::mfma_bf16(wa[1][0], b2, acc[0]);
                    acc[1] = prim::mfma_bf16(wa[1][1], b2, acc[1]);
                    acc[0] = prim::mfma_bf16(wa[0][0], b2, acc[0]);
                    acc[1] = prim::mfma_bf16(wa[0][1], b2, acc[1]);
                    acc[0] = prim::mfma_bf16(wa[1][0], b1, acc[0]);
                    acc[1] = prim::mfma_bf16(wa[1][1], b1, acc[1]);
                    acc[0] = prim::mfma_bf16(wa[0][0], b1, acc[0]);
                    acc[1] = prim::mfma_bf16(wa[0][1], b1, acc[1]);
                }
            } else if (l > 0) {
                // acc = b' + (gamma (.) W) nhat, accumulators starting from the folded bias
                init_acc(l);
                const float* wp = lds + o.w2p + (l - 1) * 2 * 32 * kWS;
                const float* w0 = wp + c * kWS + 32 * h;
                const float* w1 = wp + (32 + c) * kWS + 32 * h;
                v4 a0n = *reinterpret_cast<const v4*>(w0), a1n = *reinterpret_cast<const v4*>(w1);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v4 a0 = a0n, a1 = a1n;
                    if (q < 7) {
                        a0n = *reinterpret_cast<const v4*>(w0 + 4 * q + 4);
                        a1n = *reinterpret_cast<const v4*>(w1 + 4 * q + 4);
                    }
                    prim::sched_fence();
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0] = prim::mfma32(a0[e], nh[4 * q + e], acc[0]);
                        acc[1] = prim::mfma32(a1[e], nh[4 * q + e], acc[1]);
                    }
                }
            }
            if (a.z[l] != nullptr) {
                layer_tail_nhat<true, ACT>(acc, n.eps, nh, a.z[l] + tile * 2048, lane, mean, rstd);
                *reinterpret_cast<f2*>(a.st[l] + 2 * row) = f2{mean, rstd};     // both half-waves hold the same pair
            } else {
                layer_tail_nhat<false, ACT>(acc, n.eps, nh, nullptr, lane, mean, rstd);
            }
        }
        const long long yrow = row < rows ? row : rows - 1;
        if (out == 0) {
            // trunk only (features for the GRU): the LayerNorm's affine half applied here
            float hreg[32];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4 g = *reinterpret_cast<const v4*>(lds + o.vec + 192 * (L - 1) + 64 + 32 * t + 8 * q + 4 * h);
                    const v4 be = *reinterpret_cast<const v4*>(lds + o.vec + 192 * (L - 1) + 128 + 32 * t + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hreg[16 * t + 4 * q + e] = nh[16 * t + 4 * q + e] * g[e] + be[e];
                }
            store_row64(a.y + yrow * 64, hreg, h);
        } else if (out <= 2) {
            for (int oo = 0; oo < out; ++oo) {
                const float* wp = lds + o.whp + oo * kWS + 32 * h;
                float pr = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v4 w = *reinterpret_cast<const v4*>(wp + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) pr += w[e] * nh[4 * q + e];
                }
                pr += prim::xhalf(pr);
                a.y[yrow * out + oo] = pr + lds[o.bh + oo];
            }
        } else {
            // head on the MFMA: D[output i][row] = sum over (h, s) whp[i][h * 32 + s] * nhat[s]; lane (c, h) ends up with
            // outputs (v & 3) + 8 (v >> 2) + 4 h of its row in register v
            f32x16 ah;
#pragma unroll
            for (int v = 0; v < 16; ++v) ah[v] = 0.f;
            const float* w0 = lds + o.whp + c * kWS + 32 * h;
            v4 a0n = *reinterpret_cast<const v4*>(w0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const v4 a0 = a0n;
                if (q < 7) a0n = *reinterpret_cast<const v4*>(w0 + 4 * q + 4);
                prim::sched_fence();
#pragma unroll
                for (int e = 0; e < 4; ++e) ah = prim::mfma32(a0[e], nh[4 * q + e], ah);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int oo = (v & 3) + 8 * (v >> 2) + 4 * h;
                if (oo < out) a.y[yrow * out + oo] = ah[v] + lds[o.bh + oo];
            }
        }
        if (cstamp && m_done < 15) a.dbg[64 * (wave >> 2) + 4 * m_done + 2] = prim::clock();
    }
#undef MAPPO_F3_STEP
#undef MAPPO_F3_LAST
}

// ------------------------------------------------------------------ forward, version 4 (option bit 64: opt-in) ----
// The first layer on the bf16 matrix pipe with float32 products.  v_mfma_f32_32x32x16_bf16 runs at 16 x the FLOP rate of
// v_mfma_f32_32x32x2_f32, and a float32 value is EXACTLY the sum of three bf16 values (x = x1 + x2 + x3: 8 + 8 + 8
// significand bits, each term the round-to-nearest bf16 of what the terms before it left).  A float32 product is then
//     x y = x1 y1 + (x1 y2 + x2 y1) + (x2 y2 + x1 y3 + x3 y1) + [x2 y3 + x3 y2 + x3 y3: < 2^-24 |x y|, dropped]
// -- six bf16 x bf16 products (each exact in float32), accumulated in float32 by the matrix core, smallest terms first.
// What is dropped is below float32's own rounding of the product.  Measured on the MI355X (tools/probes/probe_bf16_split.hip,
// profiles/r04_probe_bf16_split.json): error against a float64 sum at K = 384, relative to the result's rms: max 1.3e-6 /
// rms 2.8e-7 for the six terms, 2.3e-6 / 3.5e-7 for the float32 MFMA chain the other kernels use, 2.6e-6 / 3.6e-7 for a
// float32 loop on the CPU; three terms (1.5e-5) are NOT float32 arithmetic and are not offered.  Cost of a k = 16 step of
// two 32-feature tiles per SIMD: 434 ns on the float32 MFMA (16 instructions), 251 ns this way (12 MFMAs + the 3-way split
// of the wave's 8 activation values: ~36 vector instructions that add to the MFMA time like every VALU instruction does).
// Structure: version 3's (operands straight from global memory into registers, resident first-layer weights, no
// barriers), with three changes the arithmetic asks for:
//   * the first-layer weights sit in LDS as THREE bf16 planes, split once per workgroup: 24 KB per 64-column chunk,
//     144 KB for the 384-wide critic input -- which leaves no room for the hidden layer's weights next to them, so
//   * ONE wave per SIMD (4 waves, up to 512 registers each) and the hidden layer's A operands (64 floats per lane), both
//     folded biases (32 + 32) in REGISTERS: the hidden layer and the tails read no LDS at all;
//   * chunks of 64 columns (four k = 16 steps: lane (c, g) loads columns 16 s + 8 g .. + 7 of its row for step s, two
//     16-byte pieces), three in flight per wave = the bytes version 3's two waves per SIMD keep in flight.
// Two-layer trunks, aligned widths up to 384.  The hidden layer, the head and the tails are version 3's (float32 MFMA).
constexpr int kF4MaxDin = 384;              // 6 chunks x 24 KB of first-layer weight planes
struct Fwd4Lds {
    int vec, whp, bh, w1, total;
};
__host__ __device__ __forceinline__ Fwd4Lds fwd4_lds(int nsc) {
    Fwd4Lds o;
    o.vec = 0;                                  // [bias of layer 0 | folded bias of layer 1][64]
    o.whp = 128;                                // [32][kWS] permuted head weights (rows >= out are zero), gamma folded in
    o.bh = o.whp + 32 * kWS;                    // [32] folded head bias
    o.w1 = (o.bh + 32 + 3) & ~3;                // [nsc][plane 3][step 4][tile 2][lane 64] x 16 bytes (8 bf16: k = 16 s + 8 g ..)
    o.total = o.w1 + nsc * 6144;
    return o;
}
// Narrow inputs stay on version 3 unless option bit 128 (tests) asks for every width: one wave per SIMD with a few k = 16 steps
// per tile is all tails and latency (width 48: 0.762 against 0.674 ms per 2.6 M rows; the 384-wide critic input: 1.372
// against 1.566 ms, profiles/r04_ab_forward_v4.json)
constexpr int kF4MinDin = 128;
inline bool fwd4_takes(int din, int L, int out, bool any_width) {
    return din % 4 == 0 && din >= (any_width ? 4 : kF4MinDin) && din <= kF4MaxDin && out <= 32 && L == 2;
}

// x[e] = p1[e] + p2[e] + p3[e] exactly (round-to-nearest conversions; the residuals are exact float32 subtractions)
__device__ __forceinline__ void split3(const float* x, bf8& p1, bf8& p2, bf8& p3) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        p1[e] = (__bf16)x[e];
        const float r1 = x[e] - (float)p1[e];
        p2[e] = (__bf16)r1;
        p3[e] = (__bf16)(r1 - (float)p2[e]);
    }
}

struct XBuf8 {
    v4 x[8];
};

// NSL: k = 16 steps of a row's LAST 64-column chunk that hold real columns (1 .. 4)
// HID6 (option bit 2048 with 64; emulator-green, device A / B pending at the end of round 4): the hidden layer in the six-term
// form too -- its weights as three bf16 planes in the registers the float32 operands occupied (96 instead of 64 + the 32 of
// the folded bias, which is read from LDS instead: eight 16-byte reads per tile), the normalised activations split in the
// wave: 48 MFMAs of 8 passes + ~145 split instructions instead of 64 MFMAs of 16 passes.
template <int ACT, int NSL, bool HID6 = false>
__global__ void __launch_bounds__(64 * 4, 1) mlp_fwd4_kernel(FwdArgs a) {
    constexpr int kF4Waves = 4;
    float* lds = prim::lds();
    const Net& n = a.net;
    const int din = n.din, out = n.out;
    const int nsc = (din + 63) / 64;
    const Fwd4Lds o = fwd4_lds(nsc);
    const int tid = threadIdx.x, lane = tid & 63, wave = prim::uniform(tid >> 6), c = lane & 31, h = lane >> 5;
    constexpr int kThr = 64 * kF4Waves;
    // ---- parameters, once per workgroup
    for (int e = tid; e < 128; e += kThr) {
        const int l = e >> 6, f = e & 63;
        float v = n.bias[l][f];
        if (l > 0)
            for (int k = 0; k < 64; ++k) v += n.w2[0][f * 64 + k] * n.ln_b[0][k];
        lds[o.vec + e] = v;
    }
    for (int e = tid; e < 32 * 64; e += kThr) {
        const int oo = e >> 6, hs = e & 63;
        const int k = feat_of(hs >> 5, hs & 31);
        lds[o.whp + oo * kWS + hs] = oo < out ? n.wh[oo * 64 + k] * n.ln_g[1][k] : 0.f;
    }
    for (int e = tid; e < 32; e += kThr) {
        float v = 0.f;
        if (e < out) {
            v = n.bh[e];
            for (int k = 0; k < 64; ++k) v += n.wh[e * 64 + k] * n.ln_b[1][k];
        }
        lds[o.bh + e] = v;
    }
    // first-layer weights: the 8 values k = 64 sc + 16 s + 8 g .. + 7 of feature row f = 32 t + cc, split into three bf16
    // planes, at piece 32 g + cc of block (plane, s, t) -- the piece lane (cc, g) reads as its A operand; zero beyond din
    for (int e = tid; e < nsc * 512; e += kThr) {
        const int sc = e >> 9, s = (e >> 7) & 3, f = (e >> 1) & 63, g = e & 1;
        const int k = 64 * sc + 16 * s + 8 * g;
        float w[8];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            v4 t = {0.f, 0.f, 0.f, 0.f};
            if (k + 4 * j < din) t = *reinterpret_cast<const v4u*>(n.w1 + (long long)f * din + k + 4 * j);    // (din % 4 == 0)
#pragma unroll
            for (int i = 0; i < 4; ++i) w[4 * j + i] = t[i];
        }
        bf8 p1, p2, p3;
        split3(w, p1, p2, p3);
        float* base = lds + o.w1 + sc * 6144 + (s * 2 + (f >> 5)) * 256 + (32 * g + (f & 31)) * 4;
        *reinterpret_cast<bf8*>(base) = p1;
        *reinterpret_cast<bf8*>(base + 2048) = p2;
        *reinterpret_cast<bf8*>(base + 4096) = p3;
    }
    __syncthreads();
    // per-lane constants (registers for the whole launch): both biases in accumulator order, the hidden layer's A operands
    // w2r[t][s] = gamma_0[k] W_1[32 t + c][k], k = f(h, s) -- the operand of the step that consumes slot s
    float biasr[32], bias1r[HID6 ? 1 : 32], w2r[2][HID6 ? 1 : 32];
    bf8 w2q[3][2][HID6 ? 4 : 1];        // (HID6) [plane][feature tile][k = 16 step j: slots 8 j .. 8 j + 7]
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        biasr[s] = lds[o.vec + feat_of(h, s)];
        if (!HID6) bias1r[s] = lds[o.vec + 64 + feat_of(h, s)];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (HID6) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = feat_of(h, 8 * j + i);
                    w[i] = n.w2[0][(32 * t + c) * 64 + k] * n.ln_g[0][k];
                }
                split3(w, w2q[0][t][HID6 ? j : 0], w2q[1][t][HID6 ? j : 0], w2q[2][t][HID6 ? j : 0]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int k = feat_of(h, s);
                w2r[t][HID6 ? 0 : s] = n.w2[0][(32 * t + c) * 64 + k] * n.ln_g[0][k];
            }
        }
    }

    const long long rows = a.rs.rows;
    const long long ntiles = rows128(rows) / 32;
    const long long gw = (long long)blockIdx.x * kF4Waves + wave, nw = (long long)gridDim.x * kF4Waves;
    const long long my_tiles = gw < ntiles ? (ntiles - gw + nw - 1) / nw : 0;
    if (my_tiles == 0) return;
    auto tile_of = [&](long long m) {
        if (m >= my_tiles) m = my_tiles - 1;
        return gw + m * nw;
    };
    long long it_m = 0;
    int it_kc = 0;
    const bool ragged = (din & 63) != 0;
    const float* row_it = a.rs.src + (long long)a.rs.srow[tile_of(0) * 32 + c] * din;
    const float* xp = row_it + 8 * h;
    int sr_next = a.rs.srow[tile_of(1) * 32 + c];
    auto issue = [&](XBuf8& B) {                // always exactly 8 loads (+ 1 table load per tile)
        if (ragged && it_kc == nsc - 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int k = 64 * it_kc + 16 * (q >> 1) + 8 * h + 4 * (q & 1);
                if (k > din - 4) k = din - 4;   // a piece past the row's end: finite data against zero weights (or never used)
                B.x[q] = ld_stream<1>(reinterpret_cast<const v4u*>(row_it + k));
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) B.x[q] = ld_stream<1>(reinterpret_cast<const v4u*>(xp + 16 * (q >> 1) + 4 * (q & 1)));
        }
        xp += 64;
        if (++it_kc == nsc) {
            it_kc = 0;
            ++it_m;
            row_it = a.rs.src + (long long)sr_next * din;
            xp = row_it + 8 * h;
            sr_next = a.rs.srow[tile_of(it_m + 1) * 32 + c];
        }
    };
    f32x16 acc[2];
    auto mfma_steps = [&](const XBuf8& B, int kc, auto nsteps) {
        constexpr int NS = decltype(nsteps)::value;
        const float* wt = lds + o.w1 + kc * 6144 + lane * 4;
        bf8 an[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t) an[p][t] = *reinterpret_cast<const bf8*>(wt + p * 2048 + t * 256);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            bf8 af[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) af[p][t] = an[p][t];
            if (s + 1 < NS) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        an[p][t] = *reinterpret_cast<const bf8*>(wt + p * 2048 + ((s + 1 < NS ? s + 1 : 0) * 2 + t) * 256);
            }
            prim::sched_fence();
            float r[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[i] = B.x[2 * s][i];
                r[4 + i] = B.x[2 * s + 1][i];
            }
            bf8 b1, b2, b3;
            split3(r, b1, b2, b3);
            // smallest terms first
            acc[0] = prim::mfma_bf16(af[0][0], b3, acc[0]);
            acc[1] = prim::mfma_bf16(af[0][1], b3, acc[1]);
            acc[0] = prim::mfma_bf16(af[2][0], b1, acc[0]);
            acc[1] = prim::mfma_bf16(af[2][1], b1, acc[1]);
            acc[0] = prim::mfma_bf16(af[1][0], b2, acc[0]);
            acc[1] = prim::mfma_bf16(af[1][1], b2, acc[1]);
            acc[0] = prim::mfma_bf16(af[0][0], b2, acc[0]);
            acc[1] = prim::mfma_bf16(af[0][1], b2, acc[1]);
            acc[0] = prim::mfma_bf16(af[1][0], b1, acc[0]);
            acc[1] = prim::mfma_bf16(af[1][1], b1, acc[1]);
            acc[0] = prim::mfma_bf16(af[0][0], b1, acc[0]);
            acc[1] = prim::mfma_bf16(af[0][1], b1, acc[1]);
        }
    };
    auto mfma_chunk = [&](const XBuf8& B, int kc) { mfma_steps(B, kc, mlp_int<4>{}); };
    auto mfma_last = [&](const XBuf8& B, int kc) { mfma_steps(B, kc, mlp_int<NSL>{}); };

    XBuf8 B0, B1, B2;
    issue(B0);
    issue(B1);
    issue(B2);
    const bool cstamp = a.dbg != nullptr && blockIdx.x == 0 && wave == 0 && lane == 0;
#define MAPPO_F4_STEP(B, KC) do { mfma_chunk(B, KC); issue(B); } while (0)
#define MAPPO_F4_LAST(B, KC) do { mfma_last(B, KC); issue(B); } while (0)
    for (long long m = 0; m < my_tiles;) {
        if (cstamp && m < 15) a.dbg[4 * m] = prim::clock();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = biasr[16 * t + v];
        int kc = 0;
        // (the ring of chunk buffers with static names, like version 3)
        if (NSL == 4) {
            for (; kc + 3 <= nsc; kc += 3) {
                MAPPO_F4_STEP(B0, kc);
                MAPPO_F4_STEP(B1, kc + 1);
                MAPPO_F4_STEP(B2, kc + 2);
            }
            if (nsc - kc == 1) {
                MAPPO_F4_STEP(B0, kc);
                const XBuf8 t = B0;
                B0 = B1;
                B1 = B2;
                B2 = t;
            } else if (nsc - kc == 2) {
                MAPPO_F4_STEP(B0, kc);
                MAPPO_F4_STEP(B1, kc + 1);
                const XBuf8 t = B2;
                B2 = B1;
                B1 = B0;
                B0 = t;
            }
        } else {
            for (; kc + 3 < nsc; kc += 3) {
                MAPPO_F4_STEP(B0, kc);
                MAPPO_F4_STEP(B1, kc + 1);
                MAPPO_F4_STEP(B2, kc + 2);
            }
            if (nsc - kc == 1) {
                MAPPO_F4_LAST(B0, kc);
                const XBuf8 t = B0;
                B0 = B1;
                B1 = B2;
                B2 = t;
            } else if (nsc - kc == 2) {
                MAPPO_F4_STEP(B0, kc);
                MAPPO_F4_LAST(B1, kc + 1);
                const XBuf8 t = B2;
                B2 = B1;
                B1 = B0;
                B0 = t;
            } else {
                MAPPO_F4_STEP(B0, kc);
                MAPPO_F4_STEP(B1, kc + 1);
                MAPPO_F4_LAST(B2, kc + 2);
            }
        }
        if (cstamp && m < 15) a.dbg[4 * m + 1] = prim::clock();
        const long long tile = gw + m * nw;
        const long long m_done = m;
        ++m;
        const long long row = tile * 32 + c;
        float nh[32], mean, rstd;
        // ---- layer 0's tail
        if (a.z[0] != nullptr) {
            layer_tail_nhat<true, ACT>(acc, n.eps, nh, a.z[0] + tile * 2048, lane, mean, rstd);
            st_stream<2>(reinterpret_cast<f2*>(a.st[0] + 2 * row), f2{mean, rstd});
        } else {
            layer_tail_nhat<false, ACT>(acc, n.eps, nh, nullptr, lane, mean, rstd);
        }
        // ---- the hidden layer: acc = b' + (gamma (.) W) nhat, every operand in registers
        if (HID6) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4 b = *reinterpret_cast<const v4*>(lds + o.vec + 64 + 32 * t + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][4 * q + e] = b[e];
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                constexpr int J0 = 0;
                const int jj = HID6 ? j : J0;
                bf8 b1, b2, b3;
                split3(nh + 8 * j, b1, b2, b3);
                acc[0] = prim::mfma_bf16(w2q[0][0][jj], b3, acc[0]);
                acc[1] = prim::mfma_bf16(w2q[0][1][jj], b3, acc[1]);
                acc[0] = prim::mfma_bf16(w2q[2][0][jj], b1, acc[0]);
                acc[1] = prim::mfma_bf16(w2q[2][1][jj], b1, acc[1]);
                acc[0] = prim::mfma_bf16(w2q[1][0][jj], b2, acc[0]);
                acc[1] = prim::mfma_bf16(w2q[1][1][jj], b2, acc[1]);
                acc[0] = prim::mfma_bf16(w2q[0][0][jj], b2, acc[0]);
                acc[1] = prim::mfma_bf16(w2q[0][1][jj], b2, acc[1]);
                acc[0] = prim::mfma_bf16(w2q[1][0][jj], b1, acc[0]);
                acc[1] = prim::mfma_bf16(w2q[1][1][jj], b1, acc[1]);
                acc[0] = prim::mfma_bf16(w2q[0][0][jj], b1, acc[0]);
                acc[1] = prim::mfma_bf16(w2q[0][1][jj], b1, acc[1]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = bias1r[HID6 ? 0 : 16 * t + v];
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                acc[0] = prim::mfma32(w2r[0][HID6 ? 0 : s], nh[s], acc[0]);
                acc[1] = prim::mfma32(w2r[1][HID6 ? 0 : s], nh[s], acc[1]);
            }
        }
        if (a.z[1] != nullptr) {
            layer_tail_nhat<true, ACT>(acc, n.eps, nh, a.z[1] + tile * 2048, lane, mean, rstd);
            st_stream<2>(reinterpret_cast<f2*>(a.st[1] + 2 * row), f2{mean, rstd});
        } else {
            layer_tail_nhat<false, ACT>(acc, n.eps, nh, nullptr, lane, mean, rstd);
        }
        const long long yrow = row < rows ? row : rows - 1;
        if (out == 0) {
            // trunk only (features for the GRU): the LayerNorm's affine half applied here (parameters straight from memory)
            float hreg[32];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4 g = *reinterpret_cast<const v4u*>(n.ln_g[1] + 32 * t + 8 * q + 4 * h);
                    const v4 be = *reinterpret_cast<const v4u*>(n.ln_b[1] + 32 * t + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hreg[16 * t + 4 * q + e] = nh[16 * t + 4 * q + e] * g[e] + be[e];
                }
            store_row64(a.y + yrow * 64, hreg, h);
        } else if (out <= 2) {
            for (int oo = 0; oo < out; ++oo) {
                const float* wp = lds + o.whp + oo * kWS + 32 * h;
                float pr = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v4 w = *reinterpret_cast<const v4*>(wp + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) pr += w[e] * nh[4 * q + e];
                }
                pr += prim::xhalf(pr);
                a.y[yrow * out + oo] = pr + lds[o.bh + oo];
            }
        } else {
            f32x16 ah;
#pragma unroll
            for (int v = 0; v < 16; ++v) ah[v] = 0.f;
            const float* w0 = lds + o.whp + c * kWS + 32 * h;
            v4 a0n = *reinterpret_cast<const v4*>(w0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const v4 a0 = a0n;
                if (q < 7) a0n = *reinterpret_cast<const v4*>(w0 + 4 * q + 4);
                prim::sched_fence();
#pragma unroll
                for (int e = 0; e < 4; ++e) ah = prim::mfma32(a0[e], nh[4 * q + e], ah);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int oo = (v & 3) + 8 * (v >> 2) + 4 * h;
                if (oo < out) a.y[yrow * out + oo] = ah[v] + lds[o.bh + oo];
            }
        }
        if (cstamp && m_done < 15) a.dbg[4 * m_done + 2] = prim::clock();
    }
#undef MAPPO_F4_STEP
#undef MAPPO_F4_LAST
}

// ================================================================== backward: row-parallel chain ====
struct BwdArgs {
    RowSrc rs;          // rows; DW1: the source matrix and the row table too
    Net net;
    const float* z[3];  // saved normalised activations / {mean, rstd} of every layer (forward kernel)
    const float* st[3];
    const float* dy;    // [rows, out] (head) or [rows, 64] (out == 0)
    float* dz1;         // [rows128(rows), 64]  (DW1: not written)
    float* partials;    // [gridDim.x][r_total]
    float* p1;          // DW1: [gridDim.x][64 * din] first-layer weight-gradient sums of the workgroups
    unsigned* ticket;   // the call's ticket word (last float4 of the workspace): zeroed here, drawn from by mlp_tail_kernel
    long long* dbg;     // tuning hook (mappo_mlp_set_debug): cycle stamps of workgroup 0's first tiles at [1024 ...], or NULL
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

// The chain kernel.  Its first version (round 2) transposed dh, dh * nhat, dz and the layer input of every layer through two
// scratch tiles per wave and prefetched a whole tile in registers (416 registers, 23 k cycles per tile).  This one keeps
// most of that per-tile LDS / VALU work out of the loop.  Measured on gfx950 (cycle stamps, round 3): an LDS
// instruction costs a wave ~30 cycles of issue whatever it moves, a second wave per SIMD does not hide them (its MFMA
// stream starves the partner's LDS / vector-memory instructions: every phase of a tile just took twice as long), and
// global-load latency (3 - 6 k cycles under load) was exposed three times per tile.  So: ONE wave per SIMD with the whole
// register file, as few LDS instructions as possible, every global load issued a tile (or a layer) ahead.
//
// (1) Only the weight-gradient operands are transposed.  With h_l = nhat_l * gamma_l + beta_l the input of hidden layer
//     l + 1 and dz_{l+1} the gradient at its pre-activation, accumulate per tile only
//         G_{l+1}[f][k] = sum_rows dz_{l+1}[row][f] * nhat_l[row][k]      (MFMA; both operands transposed through LDS)
//         db_{l+1}[f]   = sum_rows dz_{l+1}[row][f]
//     and derive, once per launch after the cross-workgroup reduction (mlp_finish_kernel):
//         dW_{l+1}[f][k] = gamma_l[k] G[f][k] + beta_l[k] db[f]
//         dgamma_l[k]    = sum_f W_{l+1}[f][k] G[f][k]       (= sum_rows dh_l * nhat_l with dh_l = W^T dz)
//         dbeta_l[k]     = sum_f W_{l+1}[f][k] db[f]         (= sum_rows dh_l)
//     -- the same for the head (Gh[o][k] = sum_rows dy[row][o] nhat_{L-1}[row][k], dbh[o] = sum_rows dy[row][o]).  The
//     per-tile transposes of dh and dh * nhat of every layer with their row sums are gone.  Only a trunk without a head (out == 0: the features feed a GRU) still transposes its top layer.
// (2) Sums over rows that are not MFMA operands stay in ROW layout (lane = row position, one register per slot) across all
//     tiles of the wave and are reduced over the lanes once, at the end: the bias gradients db_l (32 registers per hidden
//     layer) and, for heads of up to HR outputs (template parameter: 1 = value head, 5 = the MPE action heads), Gh
//     (32 HR registers, out x 32 fused multiply-adds per tile, no LDS at all).  Wider heads go through LDS: dy tile,
//     transposed nhat, lane = feature dot products.
// (3) gamma is folded into the staged weights: dnhat_l = (gamma_l (.) W_{l+1}^T) dz_{l+1} comes straight out of the MFMA;
//     the head's dnhat = (gamma (.) Wh^T) dy is an MFMA too (K = out, B operand = the dy values as loaded).
// (4) One transposed scratch tile per wave: the A operands of the weight-gradient MFMAs (dz^T) are parked in 32 registers
//     while nhat^T takes the tile over.  Between its last use in a tile and its first use in the next the tile receives
//     the next tile's top-layer activations by direct-to-LDS loads (no registers); dy and the row statistics of the next
//     tile are prefetched in registers, the lower layers' activations are loaded one layer ahead of their use.
// (5) Waves own 32-row tiles individually (no workgroup tile, no barrier in the loop); the waves' sums are added through
//     LDS once at the end, so the partial buffer has one row per workgroup.
constexpr int kB2Waves = 4;

constexpr int kSS = 68;          // LDS row stride of that staging tile: 16-byte slot (17 row + piece) mod 16 -- rows c .. c + 15 at
                                // one piece (the writes) and one row's 16 pieces (the reads) both cover all 64 banks
struct Bwd2Lds {
    int gam, w2t, whg, wave0, dy, hacc, stg, t1, per_wave, total;
};
constexpr int kB2W2Six = 3 * 2 * 4 * 256;   // floats of a hidden layer's weights as three bf16 planes (SIX): [plane][tile][k16 step][lane] x 16 B
template <int NW>
__host__ __device__ __forceinline__ Bwd2Lds bwd2_lds(int L, int out, bool six = false, bool dw1 = false) {
    Bwd2Lds o;
    const int outp = (out + 1) & ~1;                 // head rows padded to a whole MFMA k step
    o.gam = 0;                                       // [64] LayerNorm weight of the top layer (out == 0)
    o.w2t = 64;                                      // [L - 1][64][kWS]: gamma-scaled, transposed + permuted hidden weights
    o.whg = o.w2t + (L - 1) * (six ? kB2W2Six : 64 * kWS);    // [outp][64]: gamma-scaled head weights (zero row for odd out)
    o.wave0 = (o.whg + outp * 64 + 3) & ~3;
    o.dy = 64 * kTS;                                 // per wave: T[64][kTS] | DY[out][32] | head sums [out][64] + [out] | S
    o.hacc = o.dy + ((out * 32 + 3) & ~3);
    o.stg = o.hacc + ((65 * out + 3) & ~3);          // S[32][kSS]: row-major staging of the dz1 tile (coalesced stores)
    o.t1 = o.stg + 32 * kSS;                         // DW1: S holds the xhat tile [2][32][32], T1[64][kTS] the transposed dz1
    o.per_wave = o.t1 + (dw1 ? 64 * kTS : 0);
    o.total = o.wave0 + NW * o.per_wave;
    const int red = o.wave0 + (NW / 2) * 4096;       // the end-of-kernel reduction parks NW / 2 accumulator sets here
    if (o.total < red) o.total = red;
    return o;
}
// raw per-workgroup sums: [db_l 64 x L] [top dgamma 64 | top dbeta 64 (out == 0)] [G_l 4096 x (L - 1)]
// [Gh out x 64] [dbh out]
__host__ __device__ __forceinline__ long long r_g(int L, int l) { return 64LL * L + 128 + 4096LL * (l - 1); }
__host__ __device__ __forceinline__ long long r_gh(int L) { return 64LL * L + 128 + 4096LL * (L - 1); }
// (a multiple of 4 floats: the tail kernel adds the workgroups' rows with 16-byte loads; the padding is written as zeros)
__host__ __device__ __forceinline__ long long r_total(int L, int out) { return (r_gh(L) + 65LL * out + 3) & ~3LL; }

// d loss / d nhat (in dn) -> d loss / d z (in dn) of one layer on this lane's row: LayerNorm backward (mlp.py:17-22)
// and the activation's derivative from the saved normalised activations (see layer_tail)
template <int ACT>
__device__ __forceinline__ void ln_act_backward(float* dn, const float* nh, float mean, float rstd) {
    // (two elements per instruction -- packed f32 add / multiply / fma -- and two partial sums per mean; see act_fn2)
    f2 a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 32; s += 2) {
        const f2 d = {dn[s], dn[s + 1]}, nv = {nh[s], nh[s + 1]};
        a1 += d;
        a2 += d * nv;
    }
    float m1 = a1[0] + a1[1], m2 = a2[0] + a2[1];
    m1 += prim::xhalf(m1);
    m2 += prim::xhalf(m2);
    m1 *= (1.f / 64.f);
    m2 *= (1.f / 64.f);
    const float sd = prim::rcp_fast(rstd);
    const float thr = (0.f - mean) * rstd;      // every zero of a ReLU row maps to this nhat (the forward's own operations)
#pragma unroll
    for (int s = 0; s < 32; s += 2) {
        const f2 d = {dn[s], dn[s + 1]}, nv = {nh[s], nh[s + 1]};
        const f2 t = (d - m1) - nv * m2;
        f2 dact = {rstd, rstd};
        if (ACT == 1) {
            const f2 av = nv * sd + mean;                // the activation output of the forward pass
            dact = (1.f - av * av) * rstd;
        } else if (ACT == 2) {
            dact = f2{nv[0] > thr ? rstd : 0.f, nv[1] > thr ? rstd : 0.f};
        }
        const f2 r = t * dact;
        dn[s] = r[0];
        dn[s + 1] = r[1];
    }
}

// SIX (opt-in, option bit 512; two-layer trunks): the two 64 x 64 products of a tile -- dnhat = (gamma (.) W^T) dz and
// G += dz^T nhat -- on the bf16 matrix pipe, every float32 product from six bf16 x bf16 terms of exact three-way splits
// (see mlp_fwd4_kernel): 2 x 48 MFMAs of 8 passes instead of 2 x 64 of 16, + ~430 split instructions per tile.  The staged
// weights are split once per workgroup (three planes in LDS); dz, nhat and their transposes are split in the wave.
//
// DW1 (round 5; six-term two-layer trunks whose input is at most 64 wide -- the actors of the MPE configurations): the
// first-layer weight gradient G1[f][k] = sum_rows dz1[row][f] xhat[row][k] is accumulated HERE, the way the hidden layer's
// G is, instead of writing dz1 [rows, 64] for a second kernel that reads it back together with xhat: the xhat tile of
// the next 32 rows (through the sampler's row table) comes in by direct-to-LDS loads a tile ahead, dz1 is transposed
// through a second scratch tile (T is already receiving the next tile's activations), 48 more MFMAs per tile, 64 more
// accumulator registers.  The launch moves 512 B per row less (dz1 written + read) and mlp_dw1_rows_kernel is not launched.
template <int L, int ACT, int HR, bool SIX = false, bool DW1 = false>
__global__ void __launch_bounds__(64 * kB2Waves, 1) mlp_bwd_kernel(BwdArgs a) {
    static_assert(!SIX || L == 2, "the six-term form is built for two-layer trunks");
    static_assert(!DW1 || SIX, "the fused first-layer weight gradient is built on the six-term form");
    constexpr int NW = kB2Waves;
    float* lds = prim::lds();
    const Net& n = a.net;
    const int out = n.out, outp = (out + 1) & ~1;
    const Bwd2Lds o = bwd2_lds<NW>(L, out, SIX, DW1);
    const int tid = threadIdx.x, lane = tid & 63, wave = prim::uniform(tid >> 6), c = lane & 31, h = lane >> 5;
    constexpr int kThr = 64 * NW;
    // the tail kernel of THIS call (two launches further down the same stream) counts its finished blocks in a word of the
    // call's own workspace: calls on different streams do not share a counter, and a call that was cut short cannot leave
    // a stale count behind for the next one
    if (blockIdx.x == 0 && tid == 0) *a.ticket = 0u;
    // ---- parameters
    for (int e = tid; e < 64; e += kThr) lds[o.gam + e] = n.ln_g[L - 1][e];
    // w2t[l-1][ki][h * 32 + s] = gamma_{l-1}[ki] * W_l[f(h, s)][ki]: A operand (lane = input feature ki) of
    // dnhat_{l-1} = (gamma (.) W^T) dz
    if (SIX) {
        // piece (plane, tile t, step j, lane (cc, hh)): the 8 values w2t[32 t + cc][32 hh + 8 j ..] as bf16
        for (int e = tid; e < 512; e += kThr) {
            const int t = e >> 8, cc = (e >> 3) & 31, hh = (e >> 2) & 1, j = e & 3;
            const int ki = 32 * t + cc;
            float w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = n.w2[0][feat_of(hh, 8 * j + i) * 64 + ki] * n.ln_g[0][ki];
            bf8 p1, p2, p3;
            split3(w, p1, p2, p3);
            float* base = lds + o.w2t + (t * 4 + j) * 256 + (32 * hh + cc) * 4;
            *reinterpret_cast<bf8*>(base) = p1;
            *reinterpret_cast<bf8*>(base + 2048) = p2;
            *reinterpret_cast<bf8*>(base + 4096) = p3;
        }
    } else {
        for (int l = 1; l < L; ++l)
            for (int e = tid; e < 64 * 64; e += kThr) {
                const int ki = e >> 6, hs = e & 63;
                lds[o.w2t + (l - 1) * 64 * kWS + ki * kWS + hs] =
                    n.w2[l - 1][feat_of(hs >> 5, hs & 31) * 64 + ki] * n.ln_g[l - 1][ki];
            }
    }
    for (int e = tid; e < outp * 64; e += kThr) {
        const int oo = e >> 6, f = e & 63;
        lds[o.whg + e] = oo < out ? n.wh[oo * 64 + f] * n.ln_g[L - 1][f] : 0.f;
    }
    float* T = lds + o.wave0 + wave * o.per_wave;
    float* DY = T + o.dy;
    float* hacc = T + o.hacc;       // [out][64] sums of dy * nhat | [out] sums of dy (lane = feature / lane = o)
    float* S = T + o.stg;
    float* T1 = T + o.t1;           // (DW1)
    for (int e = lane; e < 65 * out; e += 64) hacc[e] = 0.f;
    constexpr int NG = L > 1 ? L - 1 : 1;
    f32x16 G[NG][4];                // hidden layer l: tile 2 t + t' = (output feature tile t, input feature tile t')
    f32x16 G1[DW1 ? 4 : 1];         // DW1: the first layer's, tile 2 t + t' (input features 32 t' + c < din)
    if (DW1) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) G1[DW1 ? t : 0][v] = 0.f;
    }
    const int din = a.rs.din;
    // Bias gradients = column sums of dz_l, taken where the values pass through registers in a column-friendly layout anyway
    // (round 4; before: 32 row-layout registers per layer, which the compiler kept in the AGPR half -- three instructions per
    // add -- or, for three layers, an extra transpose per layer).  Layers >= 1: the parked A operands of the G MFMAs (lane =
    // feature c / 32 + c, 16 rows of this half-wave).  Layer 0: the reads of the row-major staging tile in flush_dz1 (lane =
    // 4 features x every fourth row).
    float dbs[L][2], db[L];
    v4 db0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < L; ++l) db[l] = dbs[l][0] = dbs[l][1] = 0.f;
#pragma unroll
    for (int l = 0; l < NG; ++l)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) G[l][t][v] = 0.f;
    constexpr int NH = HR > 0 ? HR : 1;
    float ghr[NH][32], dbhr[NH];    // row layout: head sums (HR > 0)
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        dbhr[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) ghr[i][s] = 0.f;
    }
    float dgt = 0.f, dbt = 0.f;     // lane = feature: LayerNorm gradients of the top layer (out == 0)
    // HR == 0 and a head of <= kHQ outputs: Gh (lane = feature) and dbh (lane = o) stay in registers too
    constexpr int kHQ = 8;
    float hsum[kHQ], dbq = 0.f;
#pragma unroll
    for (int i = 0; i < kHQ; ++i) hsum[i] = 0.f;
    __syncthreads();
    // A operands of the head's dnhat MFMAs (constant over the tiles): step j takes outputs o = 2 j + h
    constexpr int kDyQ = 8;
    constexpr int NJ = HR > 0 ? (HR + 1) / 2 : kDyQ;
    float wa[NJ][2];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int oo = 2 * j + h < outp ? 2 * j + h : 0;
        wa[j][0] = 2 * j + h < outp ? lds[o.whg + oo * 64 + c] : 0.f;
        wa[j][1] = 2 * j + h < outp ? lds[o.whg + oo * 64 + 32 + c] : 0.f;
    }

    const long long rows = a.rs.rows;
    const long long ntiles = (rows + 31) / 32;
    const long long gw = (long long)blockIdx.x * NW + wave, nw = (long long)gridDim.x * NW;
    // ---- what is fetched a tile ahead
    auto prefetch_top = [&](long long t) {          // top layer's nhat -> T (direct-to-LDS: no registers)
        if (t < ntiles) {
#pragma unroll
            for (int bq = 0; bq < 8; ++bq) prim::load_lds16(a.z[L - 1] + t * 2048 + 4 * lane + 256 * bq, T + 256 * bq);
        }
    };
    float dyv[NH];                  // HR > 0: this row's dy[0 .. out)
    float dyq[kDyQ];                // HR == 0, out > 0: this lane's dy values o = 2 j + h, j < kDyQ (wider heads: loaded at use)
    float dhn[32];                  // out == 0: this row's gradient at the trunk's output
    f2 stn;                         // the top layer's {mean, rstd} of this row
    // (raw values: the selects on `live` / `o < out` happen where the values are used, one iteration later.  Next to
    // the load the compiler turns such a select into a branch around the load and follows it with a full vmcnt(0) wait --
    // the latency of everything the tile has in flight)
    bool livn = false;
    auto prefetch_row = [&](long long t) {
        long long r = t * 32 + c;
        if (t >= ntiles) r = rows - 1;              // (past the last tile: loaded, never used)
        stn = *reinterpret_cast<const f2*>(a.st[L - 1] + 2 * r);
        livn = r < rows;
        const long long rr = livn ? r : rows - 1;
        if (HR > 0) {
#pragma unroll
            for (int i = 0; i < HR; ++i) dyv[i] = a.dy[rr * out + (i < out ? i : 0)];
        } else if (out > 0) {
#pragma unroll
            for (int j = 0; j < kDyQ; ++j) dyq[j] = a.dy[rr * out + (2 * j + h < out ? 2 * j + h : 0)];
        } else {
            load_row64(a.dy + rr * 64, dhn, h);
        }
    };
    // dz1 [rows128(rows), 64] row-major (what the weight-gradient kernels read): staged in S by layer 0, stored by whole
    // rows (4 per wave instruction; straight from the registers an instruction would touch 32 rows x 32 bytes)
    long long staged = -1;
    auto flush_dz1 = [&]() {
        if (staged >= 0) {
            v4 sv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) sv[i] = *reinterpret_cast<const v4*>(S + (4 * i + (lane >> 4)) * kSS + 4 * (lane & 15));
#pragma unroll
            for (int i = 0; i < 8; ++i) db0 += sv[i];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                st_stream<16>(reinterpret_cast<v4*>(a.dz1 + (staged * 32 + 4 * i + (lane >> 4)) * 64 + 4 * (lane & 15)), sv[i]);
            prim::wave_sync();
        }
    };
    // DW1: the xhat tile of 32 rows -> S as [k tile t'][row][32] (what the B operands of the G1 MFMAs read lane-consecutively):
    // instruction (t', q) moves rows 8 q .. 8 q + 7, lane = (row 8 q + lane / 8, 16-byte piece lane % 8 of the k tile)
    int srt[4] = {0, 0, 0, 0};      // source rows of this lane's four rows of the NEXT tile (fetched a tile ahead of the loads)
    auto fetch_tab = [&](long long t) {
        if (t >= ntiles) t = ntiles - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) srt[q] = a.rs.srow[t * 32 + 8 * q + (lane >> 3)];
    };
    auto prefetch_x = [&](long long t) {
        if (t < ntiles) {
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                if (32 * tp < din) {
                    int k = 32 * tp + 4 * (lane & 7);
                    if (k + 4 > din) k = din - 4;       // past the row's end: its last piece again (columns that are never stored)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        prim::load_lds16(a.rs.src + (long long)srt[q] * din + k, S + (tp * 4 + q) * 256);
                }
            }
        }
    };
    prefetch_row(gw);
    prefetch_top(gw);
    if (DW1) {
        fetch_tab(gw);
        prefetch_x(gw);
    }
    const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && tid == 0;
    int n_stamp = 0;
#define MAPPO_B2_STAMP(k) if (stamp && n_stamp < 12) a.dbg[1024 + 16 * n_stamp + (k)] = prim::clock()
    for (long long tile = gw; tile < ntiles; tile += nw) {
        const long long row = tile * 32 + c;
        const bool ok = row < rows;
        float nh[32], dn[32], nxt[32];
        const bool liv = livn;      // this tile's prefetched row is a row of the launch
        MAPPO_B2_STAMP(0);
        // The prefetch into T has landed.  It is the youngest vector-memory instruction in flight -- a counted wait
        // that lets younger stores pass is not safe: stores and loads complete out of order with respect to each other
        // -- which is why the previous tile's dz1 leaves its staging tile only now, and before this tile's own loads
        // are issued (the counter would wait for them too).
        prim::wait_lds_loads<0>();
        load_frag64(T, lane, nh);
        if (DW1) fetch_tab(tile + nw);
        else flush_dz1();
        // (z and the statistics are padded to the 128-row tile: rows past the end repeat the last row and meet dy = 0)
        f2 st = stn, stx = stn;
        if (L > 1) {
            load_frag64(a.z[L > 1 ? L - 2 : 0] + tile * 2048, lane, nxt);
            stx = *reinterpret_cast<const f2*>(a.st[L > 1 ? L - 2 : 0] + 2 * row);
        }
        prim::wave_sync();          // every lane has its copy before T is written again
        MAPPO_B2_STAMP(1);
        if (out > 0) {
            // ---- head.  dnhat[f][row] = sum_o (gamma (.) Wh)[o][f] dy[row][o] on the MFMA: this lane's dy values
            // (o = 2 j + h) are the B operand
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
            if (HR > 0) {
#pragma unroll
                for (int i = 0; i < HR; ++i) dyv[i] = (liv && i < out) ? dyv[i] : 0.f;
#pragma unroll
                for (int j = 0; 2 * j < HR; ++j) {
                    if (2 * j < out) {
                        const float d1 = 2 * j + 1 < HR ? dyv[2 * j + 1 < HR ? 2 * j + 1 : 0] : 0.f;
                        const float d = h ? d1 : dyv[2 * j];
                        acc[0] = prim::mfma32(wa[j][0], d, acc[0]);
                        acc[1] = prim::mfma32(wa[j][1], d, acc[1]);
                    }
                }
                // head sums in row layout: no LDS
#pragma unroll
                for (int i = 0; i < HR; ++i) {
                    dbhr[i] += dyv[i];
#pragma unroll
                    for (int s = 0; s < 32; ++s) ghr[i][s] += dyv[i] * nh[s];
                }
            } else {
#pragma unroll
                for (int j = 0; j < kDyQ; ++j) {
                    if (2 * j < out) {
                        const int oo = 2 * j + h;
                        const float d = (liv && oo < out) ? dyq[j] : 0.f;
                        if (oo < out) DY[oo * 32 + c] = d;
                        acc[0] = prim::mfma32(wa[j < NJ ? j : 0][0], d, acc[0]);
                        acc[1] = prim::mfma32(wa[j < NJ ? j : 0][1], d, acc[1]);
                    }
                }
                for (int j = kDyQ; 2 * j < out; ++j) {
                    const int oo = 2 * j + h;
                    const float d = (ok && oo < out) ? a.dy[row * out + oo] : 0.f;
                    if (oo < out) DY[oo * 32 + c] = d;
                    const float* wp = lds + o.whg + oo * 64 + c;
                    acc[0] = prim::mfma32(wp[0], d, acc[0]);
                    acc[1] = prim::mfma32(wp[32], d, acc[1]);
                }
                // head sums, lane = feature k: Gh[o][k] += sum_rows dy[row][o] nhat[row][k]
                put_transposed(T, nh, c, h);
                prim::wave_sync();        // DY and T of this wave are complete
                float hrow[32];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v4 t = *reinterpret_cast<const v4*>(T + lane * kTS + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hrow[4 * q + e] = t[e];
                }
                auto dot_o = [&](int oo) {      // (packed: two partial sums, 16 fused multiply-adds per output)
                    f2 sm = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const v4 d = *reinterpret_cast<const v4*>(DY + oo * 32 + 4 * q);
                        sm += f2{d[0], d[1]} * f2{hrow[4 * q], hrow[4 * q + 1]};
                        sm += f2{d[2], d[3]} * f2{hrow[4 * q + 2], hrow[4 * q + 3]};
                    }
                    return sm[0] + sm[1];
                };
                if (out <= kHQ) {       // (unrolled: the reads of output o + 1 are in flight behind the products of o)
#pragma unroll
                    for (int oo = 0; oo < kHQ; ++oo)
                        if (oo < out) hsum[oo] += dot_o(oo);
                } else {
                    for (int oo = 0; oo < out; ++oo) hacc[oo * 64 + lane] += dot_o(oo);
                }
                if (lane < out) dbq += rowsum32(DY + lane * 32);
                prim::wave_sync();
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) dn[16 * t + v] = acc[t][v];
        } else {
            // ---- no head: dy is the gradient at the trunk's output h = nhat * gamma + beta
            float dh[32];
#pragma unroll
            for (int s = 0; s < 32; ++s) dh[s] = liv ? dhn[s] : 0.f;
            put_transposed(T, dh, c, h);
            prim::wave_sync();
            dbt += rowsum32(T + lane * kTS);
            prim::wave_sync();
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4 g = *reinterpret_cast<const v4*>(lds + o.gam + 32 * t + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int s = 16 * t + 4 * q + e;
                        dn[s] = dh[s] * g[e];
                        dh[s] *= nh[s];
                    }
                }
            put_transposed(T, dh, c, h);
            prim::wave_sync();
            dgt += rowsum32(T + lane * kTS);
            prim::wave_sync();
        }
        prefetch_row(tile + nw);
        if (L == 1) prefetch_top(tile + nw);     // (T's last use of this tile)
        MAPPO_B2_STAMP(3);
        // ---- layers, top down
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            ln_act_backward<ACT>(dn, nh, st[0], st[1]);      // dn now holds dz_l
            MAPPO_B2_STAMP(l == L - 1 ? 4 : 10);
            if (l == 0 && DW1) {
                // ---- G1 += dz1^T xhat (see the kernel's head): A operands from the transposed dz1, B operands from the
                // xhat tile that landed in S before this tile began
                put_transposed(T1, dn, c, h);
                prim::wave_sync();
                v4 a0[4], a1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a0[q] = *reinterpret_cast<const v4*>(T1 + c * kTS + 16 * h + 4 * q);
                    a1[q] = *reinterpret_cast<const v4*>(T1 + (32 + c) * kTS + 16 * h + 4 * q);
                }
                {
                    f2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        p0 += f2{a0[q][0], a0[q][1]} + f2{a0[q][2], a0[q][3]};
                        p1 += f2{a1[q][0], a1[q][1]} + f2{a1[q][2], a1[q][3]};
                    }
                    dbs[0][0] += p0[0] + p0[1];
                    dbs[0][1] += p1[0] + p1[1];
                }
                const bool wide = din > 32;     // (uniform) the second k tile holds columns
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float va[2][8], vb[2][8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        va[0][i] = a0[2 * j + (i >> 2)][i & 3];
                        va[1][i] = a1[2 * j + (i >> 2)][i & 3];
                        vb[0][i] = S[(16 * h + 8 * j + i) * 32 + c];
                        vb[1][i] = wide ? S[1024 + (16 * h + 8 * j + i) * 32 + c] : 0.f;
                    }
                    bf8 A[2][3], B[2][3];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        split3(va[t], A[t][0], A[t][1], A[t][2]);
                        split3(vb[t], B[t][0], B[t][1], B[t][2]);
                    }
#pragma unroll
                    for (int ta = 0; ta < 2; ++ta) {
                        f32x16& g0 = G1[DW1 ? 2 * ta : 0];
                        g0 = prim::mfma_bf16(A[ta][0], B[0][2], g0);
                        g0 = prim::mfma_bf16(A[ta][2], B[0][0], g0);
                        g0 = prim::mfma_bf16(A[ta][1], B[0][1], g0);
                        g0 = prim::mfma_bf16(A[ta][0], B[0][1], g0);
                        g0 = prim::mfma_bf16(A[ta][1], B[0][0], g0);
                        g0 = prim::mfma_bf16(A[ta][0], B[0][0], g0);
                        if (wide) {
                            f32x16& g1 = G1[DW1 ? 2 * ta + 1 : 0];
                            g1 = prim::mfma_bf16(A[ta][0], B[1][2], g1);
                            g1 = prim::mfma_bf16(A[ta][2], B[1][0], g1);
                            g1 = prim::mfma_bf16(A[ta][1], B[1][1], g1);
                            g1 = prim::mfma_bf16(A[ta][0], B[1][1], g1);
                            g1 = prim::mfma_bf16(A[ta][1], B[1][0], g1);
                            g1 = prim::mfma_bf16(A[ta][0], B[1][0], g1);
                        }
                    }
                }
                prim::wave_sync();          // every lane has read the xhat tile: the next one may land
                prefetch_x(tile + nw);
            } else if (l == 0) {
                // -> the row-major staging tile
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = dn[16 * t + 4 * q + e];
                        *reinterpret_cast<v4*>(S + c * kSS + 32 * t + 8 * q + 4 * h) = v;
                    }
                prim::wave_sync();
                staged = tile;          // (stored at the top of the next iteration, see flush_dz1)
            } else {
                put_transposed(T, dn, c, h);
                // dnhat_{l-1} = (gamma (.) W^T) dz: the last use of dz in row order
                f32x16 dx[2];
                if (SIX) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int v = 0; v < 16; ++v) dx[t][v] = 0.f;
                    const float* wt = lds + o.w2t + lane * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bf8 wa[3][2];
#pragma unroll
                        for (int p = 0; p < 3; ++p)
#pragma unroll
                            for (int t = 0; t < 2; ++t)
                                wa[p][t] = *reinterpret_cast<const bf8*>(wt + p * 2048 + (t * 4 + j) * 256);
                        bf8 b1, b2, b3;
                        split3(dn + 8 * j, b1, b2, b3);
                        dx[0] = prim::mfma_bf16(wa[0][0], b3, dx[0]);
                        dx[1] = prim::mfma_bf16(wa[0][1], b3, dx[1]);
                        dx[0] = prim::mfma_bf16(wa[2][0], b1, dx[0]);
                        dx[1] = prim::mfma_bf16(wa[2][1], b1, dx[1]);
                        dx[0] = prim::mfma_bf16(wa[1][0], b2, dx[0]);
                        dx[1] = prim::mfma_bf16(wa[1][1], b2, dx[1]);
                        dx[0] = prim::mfma_bf16(wa[0][0], b2, dx[0]);
                        dx[1] = prim::mfma_bf16(wa[0][1], b2, dx[1]);
                        dx[0] = prim::mfma_bf16(wa[1][0], b1, dx[0]);
                        dx[1] = prim::mfma_bf16(wa[1][1], b1, dx[1]);
                        dx[0] = prim::mfma_bf16(wa[0][0], b1, dx[0]);
                        dx[1] = prim::mfma_bf16(wa[0][1], b1, dx[1]);
                    }
                } else {
                    dense64(lds + o.w2t + (l - 1) * 64 * kWS, c, h, dn, dx);
                }
                MAPPO_B2_STAMP(5);
                prim::wave_sync();
                // A operands of G += dz^T nhat (lane = output feature, rows 16 h + ..): parked in registers while the
                // input of this layer, nhat of layer l - 1 (loaded one layer ahead), takes the tile over
                v4 a0[4], a1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a0[q] = *reinterpret_cast<const v4*>(T + c * kTS + 16 * h + 4 * q);
                    a1[q] = *reinterpret_cast<const v4*>(T + (32 + c) * kTS + 16 * h + 4 * q);
                }
                {
                    f2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        p0 += f2{a0[q][0], a0[q][1]} + f2{a0[q][2], a0[q][3]};
                        p1 += f2{a1[q][0], a1[q][1]} + f2{a1[q][2], a1[q][3]};
                    }
                    dbs[l][0] += p0[0] + p0[1];
                    dbs[l][1] += p1[0] + p1[1];
                }
                prim::wave_sync();
                MAPPO_B2_STAMP(6);
                put_transposed(T, nxt, c, h);
                prim::wave_sync();
                MAPPO_B2_STAMP(7);
                if (SIX) {
                    // k = 16 rows per step: step j of lane (c, h) contracts rows 16 h + 8 j .. + 7 (the same rows on both sides)
                    f32x16* Gl = G[0];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float va[2][8], vb[2][8];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const v4 b0 = *reinterpret_cast<const v4*>(T + c * kTS + 16 * h + 8 * j + 4 * q);
                            const v4 b1 = *reinterpret_cast<const v4*>(T + (32 + c) * kTS + 16 * h + 8 * j + 4 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                va[0][4 * q + e] = a0[2 * j + q][e];
                                va[1][4 * q + e] = a1[2 * j + q][e];
                                vb[0][4 * q + e] = b0[e];
                                vb[1][4 * q + e] = b1[e];
                            }
                        }
                        bf8 A[2][3], B[2][3];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            split3(va[t], A[t][0], A[t][1], A[t][2]);
                            split3(vb[t], B[t][0], B[t][1], B[t][2]);
                        }
#pragma unroll
                        for (int ta = 0; ta < 2; ++ta) {
                            f32x16& g0 = Gl[2 * ta], &g1 = Gl[2 * ta + 1];
                            g0 = prim::mfma_bf16(A[ta][0], B[0][2], g0);
                            g1 = prim::mfma_bf16(A[ta][0], B[1][2], g1);
                            g0 = prim::mfma_bf16(A[ta][2], B[0][0], g0);
                            g1 = prim::mfma_bf16(A[ta][2], B[1][0], g1);
                            g0 = prim::mfma_bf16(A[ta][1], B[0][1], g0);
                            g1 = prim::mfma_bf16(A[ta][1], B[1][1], g1);
                            g0 = prim::mfma_bf16(A[ta][0], B[0][1], g0);
                            g1 = prim::mfma_bf16(A[ta][0], B[1][1], g1);
                            g0 = prim::mfma_bf16(A[ta][1], B[0][0], g0);
                            g1 = prim::mfma_bf16(A[ta][1], B[1][0], g1);
                            g0 = prim::mfma_bf16(A[ta][0], B[0][0], g0);
                            g1 = prim::mfma_bf16(A[ta][0], B[1][0], g1);
                        }
                    }
                } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4 b0 = *reinterpret_cast<const v4*>(T + c * kTS + 16 * h + 4 * q);
                    const v4 b1 = *reinterpret_cast<const v4*>(T + (32 + c) * kTS + 16 * h + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        G[l > 0 ? l - 1 : 0][0] = prim::mfma32(a0[q][e], b0[e], G[l > 0 ? l - 1 : 0][0]);
                        G[l > 0 ? l - 1 : 0][1] = prim::mfma32(a0[q][e], b1[e], G[l > 0 ? l - 1 : 0][1]);
                        G[l > 0 ? l - 1 : 0][2] = prim::mfma32(a1[q][e], b0[e], G[l > 0 ? l - 1 : 0][2]);
                        G[l > 0 ? l - 1 : 0][3] = prim::mfma32(a1[q][e], b1[e], G[l > 0 ? l - 1 : 0][3]);
                    }
                }
                }
                MAPPO_B2_STAMP(8);
                prim::wave_sync();          // T's last use of this tile (l == 1): the next tile's prefetch may land
                if (l == 1) {
                    // (the compiler does not count the prefetch's loads: a wait for one of ITS older loads placed after
                    // this point would wait for the prefetch too.  Settle them here, where they have long arrived.)
                    float m0 = stx[0], m1 = stx[1];
                    prim::pin(m0);
                    prim::pin(m1);
                    stx = f2{m0, m1};
                    if (DW1) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) prim::pin(srt[q]);
                    }
                    prefetch_top(tile + nw);
                }
#pragma unroll
                for (int s = 0; s < 32; ++s) nh[s] = nxt[s];
                st = stx;
                if (l >= 2) {
                    load_frag64(a.z[l >= 2 ? l - 2 : 0] + tile * 2048, lane, nxt);
                    stx = *reinterpret_cast<const f2*>(a.st[l >= 2 ? l - 2 : 0] + 2 * row);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) dn[16 * t + v] = dx[t][v];
                MAPPO_B2_STAMP(9);
            }
        }
        MAPPO_B2_STAMP(15);
        ++n_stamp;
    }
#undef MAPPO_B2_STAMP
    prim::wait_lds_loads<0>();
    if (!DW1) flush_dz1();
    // ---- the column sums -> lane = feature
#pragma unroll
    for (int l = DW1 ? 0 : 1; l < L; ++l) {
        float sa = dbs[l][0], sb = dbs[l][1];
        sa += prim::xhalf(sa);
        sb += prim::xhalf(sb);
        db[l] = h ? sb : sa;
    }
    prim::wave_sync();
    if (!DW1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) T[64 * (lane >> 4) + 4 * (lane & 15) + e] = db0[e];
        prim::wave_sync();
        db[0] = (T[lane] + T[64 + lane]) + (T[128 + lane] + T[192 + lane]);
        prim::wave_sync();
    }
    if (HR == 0 && out > 0) {
        if (out <= kHQ) {
#pragma unroll
            for (int oo = 0; oo < kHQ; ++oo)
                if (oo < out) hacc[oo * 64 + lane] = hsum[oo];
        }
        if (lane < out) hacc[64 * out + lane] = dbq;
    }
    if (HR > 0) {
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            if (i < out) {
                prim::wave_sync();
                put_transposed(T, ghr[i], c, h);
                prim::wave_sync();
                hacc[i * 64 + lane] = rowsum32(T + lane * kTS);
                prim::wave_sync();
                if (h == 0) T[c] = dbhr[i];         // both half-waves hold the same rows' dy
                prim::wave_sync();
                if (lane == 0) hacc[64 * out + i] = rowsum32(T);
            }
        }
    }
    // ---- add the waves' sums through LDS and write this workgroup's partial row
    float* prow = a.partials + (long long)blockIdx.x * r_total(L, out);
    __syncthreads();                // every wave is done with its scratch
    {
        // vectors: [wave][L + 2][64] in the waves' own T tiles (64 * kTS >= (L + 2) * 64)
#pragma unroll
        for (int l = 0; l < L; ++l) T[64 * l + lane] = db[l];
        T[64 * L + lane] = dgt;
        T[64 * (L + 1) + lane] = dbt;
        __syncthreads();
        for (int e = tid; e < (L + 2) * 64; e += kThr) {
            float sm = 0.f;
            for (int w = 0; w < NW; ++w) sm += lds[o.wave0 + w * o.per_wave + e];
            prow[e] = sm;
        }
        for (int e = tid; e < (int)(r_total(L, out) - r_gh(L)); e += kThr) {
            float sm = 0.f;
            if (e < 65 * out)
                for (int w = 0; w < NW; ++w) sm += lds[o.wave0 + w * o.per_wave + o.hacc + e];
            prow[r_gh(L) + e] = sm;
        }
    }
    float* red = lds + o.wave0;     // [NW / 2][4096]: slot i of set w at w * 4096 + i * 64 + lane
#pragma unroll
    for (int l = 1; l < L; ++l) {
#pragma unroll
        for (int half = NW / 2; half >= 1; half >>= 1) {
            __syncthreads();
            if (wave >= half && wave < 2 * half) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) red[(wave - half) * 4096 + (16 * t + v) * 64 + lane] = G[l - 1][t][v];
            }
            __syncthreads();
            if (wave < half) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) G[l - 1][t][v] += red[wave * 4096 + (16 * t + v) * 64 + lane];
            }
        }
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int f = 32 * t + (v & 3) + 8 * (v >> 2) + 4 * h;
                        prow[r_g(L, l) + f * 64 + 32 * tp + c] = G[l - 1][2 * t + tp][v];
                    }
        }
    }
    if (DW1) {
        // the first layer's sums: same tree; the workgroup's row of the first-layer partials is what mlp_dw1_rows_kernel wrote
#pragma unroll
        for (int half = NW / 2; half >= 1; half >>= 1) {
            __syncthreads();
            if (wave >= half && wave < 2 * half) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) red[(wave - half) * 4096 + (16 * t + v) * 64 + lane] = G1[DW1 ? t : 0][v];
            }
            __syncthreads();
            if (wave < half) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) G1[DW1 ? t : 0][v] += red[wave * 4096 + (16 * t + v) * 64 + lane];
            }
        }
        if (wave == 0) {
            float* prow1 = a.p1 + (long long)blockIdx.x * 64 * din;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int f = 32 * t + (v & 3) + 8 * (v >> 2) + 4 * h;
                        if (32 * tp + c < din) prow1[(long long)f * din + 32 * tp + c] = G1[DW1 ? 2 * t + tp : 0][v];
                    }
        }
    }
}

// The parameter gradients of the chain from the reduced raw sums R (see mlp_bwd_kernel): one workgroup.
struct FinishArgs {
    Net net;
    const float* R;
    float* grads;
};
__device__ __forceinline__ void finish_grads(const FinishArgs& a) {
    const Net& n = a.net;
    const int L = n.L, out = n.out, din = n.din, tid = threadIdx.x;
    for (int l = 1; l < L; ++l) {
        // (the outputs alias none of the inputs: lets the compiler keep several iterations' loads in flight)
        const float* __restrict__ G = a.R + r_g(L, l);
        const float* __restrict__ dbl = a.R + 64 * l;
        const float* __restrict__ g = n.ln_g[l - 1];
        const float* __restrict__ be = n.ln_b[l - 1];
        const float* __restrict__ W = n.w2[l - 1];
        float* __restrict__ dW = a.grads + g_w2(din, L, l);
        for (int e = tid; e < 4096; e += kThreads) dW[e] = g[e & 63] * G[e] + be[e & 63] * dbl[e >> 6];
        if (tid < 64) {
            float sg = 0.f, sb = 0.f;
#pragma unroll 16
            for (int f = 0; f < 64; ++f) {      // (unrolled: sixteen rows' loads in flight, the sums stay in order)
                sg += W[f * 64 + tid] * G[f * 64 + tid];
                sb += W[f * 64 + tid] * dbl[f];
            }
            a.grads[g_vec(din, l - 1) + 64 + tid] = sg;
            a.grads[g_vec(din, l - 1) + 128 + tid] = sb;
            a.grads[g_vec(din, l) + tid] = dbl[tid];
        }
    }
    if (tid < 64) a.grads[g_vec(din, 0) + tid] = a.R[tid];       // first layer's bias: column sums of dz1
    const float* g = n.ln_g[L - 1];
    const float* be = n.ln_b[L - 1];
    if (out > 0) {
        const float* Gh = a.R + r_gh(L);
        const float* dbh = Gh + 64 * out;
        float* dWh = a.grads + g_wh(din, L);
        for (int e = tid; e < 64 * out; e += kThreads) dWh[e] = g[e & 63] * Gh[e] + be[e & 63] * dbh[e >> 6];
        for (int e = tid; e < out; e += kThreads) dWh[64 * out + e] = dbh[e];
        if (tid < 64) {
            float sg = 0.f, sb = 0.f;
#pragma unroll 8
            for (int oo = 0; oo < out; ++oo) {
                sg += n.wh[oo * 64 + tid] * Gh[oo * 64 + tid];
                sb += n.wh[oo * 64 + tid] * dbh[oo];
            }
            a.grads[g_vec(din, L - 1) + 64 + tid] = sg;
            a.grads[g_vec(din, L - 1) + 128 + tid] = sb;
        }
    } else if (tid < 64) {
        a.grads[g_vec(din, L - 1) + 64 + tid] = a.R[64 * L + tid];
        a.grads[g_vec(din, L - 1) + 128 + tid] = a.R[64 * L + 64 + tid];
    }
}

// The tail of a backward call as ONE launch (three in round 2/3a: 40 us of launches per call, 2 % of a step on an 8-GPU
// shard): blocks [0, nb1) add the first-layer weight-gradient partials into `grads`, blocks [nb1, nb1 + nb2) the chain's
// partial rows into the raw sums R; the block that finishes last (a ticket word in the call's workspace, zeroed by the
// chain kernel of the same call) derives the remaining parameter gradients from R.
__device__ __forceinline__ void reduce_chunk(const float* partials, long long n, long long stride, long long count,
                                             float* out, long long chunk, float* sh);
__device__ __forceinline__ void reduce_chunk4(const float* partials, long long n, long long stride, long long count,
                                              float* out, long long chunk, float* sh);
struct TailArgs {
    FinishArgs fin;             // fin.R = reduced raw sums (written here), fin.grads
    const float* p1;            // [n1][c1] first-layer partials -> grads[0 .. c1)
    long long n1, c1;
    const float* p2;            // [n2][c2] chain partials -> R
    long long n2, c2;
    float* raw;
    unsigned* ticket;
    int nb1, nb2;
    int wide1;                  // p1 / grads are 16-byte aligned and c1 is a multiple of 4: blocks of 128 elements
    int wide2;                  // the same for p2 / raw
};
__global__ void __launch_bounds__(kThreads) mlp_tail_kernel(TailArgs a) {
    float* sh = prim::lds();        // [8][128] | ticket
    const int b = blockIdx.x;
    if (b < a.nb1) {
        if (a.wide1) reduce_chunk4(a.p1, a.n1, a.c1, a.c1, a.fin.grads, b, sh);
        else reduce_chunk(a.p1, a.n1, a.c1, a.c1, a.fin.grads, b, sh);
    } else if (a.wide2) {
        reduce_chunk4(a.p2, a.n2, a.c2, a.c2, a.raw, b - a.nb1, sh);
    } else {
        reduce_chunk(a.p2, a.n2, a.c2, a.c2, a.raw, b - a.nb1, sh);
    }
    prim::fence();                  // this block's sums are visible device-wide before its ticket is drawn
    __syncthreads();
    if (threadIdx.x == 0) sh[1024] = prim::i2f((int)prim::ticket(a.ticket));
    __syncthreads();
    if (prim::f2i(sh[1024]) != a.nb1 + a.nb2 - 1) return;
    prim::fence();
    finish_grads(a.fin);
}

// ================================================================== backward: first-layer weight gradient ====
// dW1[f][k] = sum over rows dz1[row][f] * xhat[row][k] with xhat gathered and standardised on the fly: a split-K GEMM
// (K = rows) whose B operand is read through the sampler's row table.  A workgroup owns a slab of <= 384 k columns
// (blockIdx.y) and a strided set of 32-row tiles.  Same role split as the forward: 4 loader waves keep kDepth tiles in
// flight in registers (a thread always serves the same row of a tile, so standardisation needs 2 registers) and fill
// one of two LDS stages; compute wave w owns k tiles w, w + 4, w + 8 of the slab x both feature tiles (96 accumulator
// registers) and reads operands lane-consecutively (ds_read_b32, no padding needed).
struct Dw1Args {
    RowSrc rs;
    const float* dz1;
    float* partials;    // [gridDim.x][64 * din]
    long long* dbg;     // tuning hook (mappo_mlp_set_debug): cycle stamps of workgroup 0's first tiles at [512 ...], or NULL
};
constexpr int kDw1StageX = kDw1Rows * kDw1Slab;                        // floats
constexpr int kDw1Stage = kDw1StageX + kDw1Rows * 64;                  // + dz1 tile [32][64]

// (same discipline as ChunkBuf: loaded values are either stored to LDS kDepth iterations later or used as an address)
struct TileBuf {
    v4 xv[12], dv[2];
    int live;               // 0: the row in flight lies past the last row (its dz1 is dropped at the store)
    int sr_n;               // source row of the row this buffer loads next (fetched three issues ahead)
};

__global__ void __launch_bounds__(kPipeThreads) mlp_dw1_kernel(Dw1Args a) {
    float* lds = prim::lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
    const int din = a.rs.din;
    const int k0 = blockIdx.y * kDw1Slab;
    const int kw = (din - k0 < kDw1Slab ? ((din - k0 + 31) / 32) * 32 : kDw1Slab);    // slab width, multiple of 32
    const long long rows = a.rs.rows;
    const long long ntiles = (rows + kDw1Rows - 1) / kDw1Rows;
    const long long n_it = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;

    if (wave >= 4) {
        // ------------------------------------------------------------ loader: thread = (row of the tile, 1 of 8 lanes)
        const int lt = tid - 256, lr = lt >> 3, sub = lt & 7;
        const int np = kw >> 5;                  // 16-byte pieces per thread: piece sub + 8 p, p < np (<= 12)
        long long issued = 0;
            auto row_of = [&](long long m) {         // launch row this thread serves in its m-th tile (clamped past the end)
            if (m >= n_it) m = n_it - 1;
            return (blockIdx.x + m * gridDim.x) * kDw1Rows + lr;     // < rows128
        };
        auto fetch_row = [&](long long m, TileBuf& B) { B.sr_n = a.rs.srow[row_of(m)]; };
        // every issue is exactly 12 + 2 + 1 loads
        auto issue = [&](TileBuf& B) {
            const float* xrow = a.rs.src + (long long)B.sr_n * din;
            long long gr = row_of(issued);
            B.live = gr < rows;
            if (gr >= rows) gr = rows - 1;
#pragma unroll
            for (int p = 0; p < 12; ++p) {
                const int pp = p < np ? p : np - 1;          // unused pieces re-read the last one (dropped at the store)
                B.xv[p] = *reinterpret_cast<const v4u*>(xrow + piece_at(k0 + 4 * (sub + 8 * pp), din));
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) B.dv[p] = *reinterpret_cast<const v4*>(a.dz1 + gr * 64 + 4 * (sub + 8 * p));
            fetch_row(issued + kDepth, B);
            ++issued;
        };
        auto store = [&](long long m, const TileBuf& B) {
            float* st = lds + (m & 1) * kDw1Stage;
#pragma unroll
            for (int p = 0; p < 12; ++p)
                if (p < np) {
                    const int kl = 4 * (sub + 8 * p);
                    *reinterpret_cast<v4*>(st + lr * kw + kl) = shift4(B.xv[p], k0 + kl - piece_at(k0 + kl, din));
                }
#pragma unroll
            for (int p = 0; p < 2; ++p)
                *reinterpret_cast<v4*>(st + kDw1StageX + lr * 64 + 4 * (sub + 8 * p)) = B.live ? B.dv[p] : v4{0.f, 0.f, 0.f, 0.f};
        };
        TileBuf B0, B1, B2;
        if (n_it > 0) {
            fetch_row(0, B0);
            fetch_row(1, B1);
            fetch_row(2, B2);
            issue(B0);
            issue(B1);
            issue(B2);
            store(0, B0);
            issue(B0);
        }
        long long j = 0;
        for (; j + 3 <= n_it; j += 3) {
            __syncthreads();
            store(j + 1, B1);
            issue(B1);
            __syncthreads();
            store(j + 2, B2);
            issue(B2);
            __syncthreads();
            store(j + 3, B0);
            issue(B0);
        }
        if (j < n_it) {
            __syncthreads();
            store(j + 1, B1);
            if (j + 1 < n_it) {
                __syncthreads();
                store(j + 2, B2);
            }
        }
        return;
    }
    // ---------------------------------------------------------------- compute
    prim::set_priority_high();
    const int ntk = kw / 32;
    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][t][v] = 0.f;
    for (long long j = 0; j < n_it; ++j) {
        __syncthreads();
        const float* xt = lds + (j & 1) * kDw1Stage;
        const float* dzt = xt + kDw1StageX;
        // MFMA-only stream (see the forward kernel): operand reads of step s + 1 are issued before the MFMAs of step s
        // (= rows 16 h + s of the tile)
        float a0n, a1n, bn[3];
        auto rd_step = [&](int st, float& ra0, float& ra1, float* rb) {
            const int r = 16 * h + st;
            ra0 = dzt[r * 64 + c];
            ra1 = dzt[r * 64 + 32 + c];
#pragma unroll
            for (int i = 0; i < 3; ++i) rb[i] = (wave + 4 * i < ntk) ? xt[r * kw + 32 * (wave + 4 * i) + c] : 0.f;
        };
        rd_step(0, a0n, a1n, bn);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a0 = a0n, a1 = a1n, b0 = bn[0], b1 = bn[1], b2 = bn[2];
            if (s < 15) rd_step(s + 1, a0n, a1n, bn);
            prim::sched_fence();
            if (wave < ntk) {
                acc[0][0] = prim::mfma32(a0, b0, acc[0][0]);
                acc[0][1] = prim::mfma32(a1, b0, acc[0][1]);
            }
            if (wave + 4 < ntk) {
                acc[1][0] = prim::mfma32(a0, b1, acc[1][0]);
                acc[1][1] = prim::mfma32(a1, b1, acc[1][1]);
            }
            if (wave + 8 < ntk) {
                acc[2][0] = prim::mfma32(a0, b2, acc[2][0]);
                acc[2][1] = prim::mfma32(a1, b2, acc[2][1]);
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

// ---- the same product for din % 4 == 0, without loader waves and without registers in the load path
// Measured on gfx950 (tools/probes/probe_lds_store.hip): while a wave streams v_mfma_f32_32x32x2_f32 back to back, an
// LDS store or a global load issued by ANOTHER wave of that SIMD gets through only about once per 400 cycles (every
// instruction that sends VGPRs out of the SIMD waits for a gap in the MFMA operand traffic), and one issued by the MFMA
// wave itself costs about one MFMA slot.  The loader-wave version above needs 14 loads + 14 LDS stores per 96 MFMAs on
// every SIMD and runs at half the MFMA rate.  Here each wave fetches its OWN operands with direct-to-LDS loads
// (global_load_lds_dwordx4: 64 lanes x 16 B land in 1 KB of consecutive LDS; no VGPR data, no ds_write), issued in one
// group between two tiles' MFMA streams: 6 for its three 32-column k tiles of a 16-row tile + 1 for its quarter of the
// shared dz1 tile + 1 for the sampler's row table, per 48 MFMAs.  kD2Slots LDS slots per wave: the loads of tile m + 3
// are issued before the MFMAs of tile m; the wait at the top of an iteration is counted (everything but the two youngest
// groups), one barrier per tile publishes the dz1 quarters.  One workgroup of 4 waves per CU (120 KB of LDS).
constexpr int kD2Rows = 16;
constexpr int kD2Slots = 4;
constexpr int kD2GridCap = 256;
#ifndef MAPPO_D2_MIN_WIDTH
#define MAPPO_D2_MIN_WIDTH 128                   // (tuning: tools/ab_build.sh; 152 columns: 2.01 ms per 2.6 M rows against 2.16)
#endif
constexpr int kD2MinWidth = MAPPO_D2_MIN_WIDTH;  // (historic tuning knob of the loader / direct cross-over)
constexpr int kD2RowsMaxWidth = 192;             // up to six k tiles: every wave owns all of them (mlp_dw1_rows_kernel)
// A wave owns up to MAXNT k tiles of its workgroup's slab of 128 MAXNT columns.  MAXNT = 3 (384-column slabs, 120 KB of LDS)
// or 4 (512 columns, 152 KB): the launcher takes 4 where that saves a slab -- widths in (384, 512] had a second slab of a
// few k tiles whose workgroups idled most of the launch (SMAC's 436-wide critic input: 49 instead of 100 TFLOP/s).
constexpr int d2_xslot(int maxnt) { return maxnt * kD2Rows * 32; }        // floats: [k tile i < MAXNT][16 rows][32]
constexpr int kD2DzSlot = kD2Rows * 64;          // floats: [16 rows][64 features], shared by the 4 waves
constexpr int d2_lds_slots(int maxnt, int slots) { return 4 * slots * d2_xslot(maxnt) + slots * kD2DzSlot; }
constexpr int d2_lds(int maxnt) { return d2_lds_slots(maxnt, kD2Slots); }   // floats, + the row-table rings:
constexpr int kD2TabRing = 2 * kD2Slots;         // 256-byte slots per wave
inline int d2_maxnt(int din) { return (din + 511) / 512 < (din + 383) / 384 ? 4 : 3; }

// the 8 MFMA steps of one 16-row tile (step s contracts rows s and 8 + s) for a wave that owns NT k tiles
template <int NT>
__device__ __forceinline__ void dw1_tile_steps(const float* xt, const float* dzt, int c, int h, f32x16 (*acc)[2]) {
    constexpr int NB = NT > 0 ? NT : 1;
    float a0n, a1n, bn[NB];
    auto rd_step = [&](int st, float& ra0, float& ra1, float* rb) {
        const int r = 8 * h + st;
        ra0 = dzt[r * 64 + c];
        ra1 = dzt[r * 64 + 32 + c];
#pragma unroll
        for (int i = 0; i < NT; ++i) rb[i] = xt[i * (kD2Rows * 32) + r * 32 + c];
    };
    if (NT == 0) return;
    rd_step(0, a0n, a1n, bn);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float a0 = a0n, a1 = a1n;
        float b[NB];
#pragma unroll
        for (int i = 0; i < NT; ++i) b[i] = bn[i];
        if (s < 7) rd_step(s + 1, a0n, a1n, bn);
        prim::sched_fence();
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            acc[i][0] = prim::mfma32(a0, b[i], acc[i][0]);
            acc[i][1] = prim::mfma32(a1, b[i], acc[i][1]);
        }
    }
}

// The same 16-row tile on the bf16 matrix pipe (opt-in, option bit 256; see mlp_fwd4_kernel for the arithmetic): the
// contraction index is the ROW, and v_mfma_f32_32x32x16_bf16 takes all 16 rows of the tile in one step -- lane (c, g) holds
// rows 8 g .. 8 g + 7 of its column (dz1: feature c / 32 + c; x: column c of each k tile), read from the row-major LDS tiles
// with the same strided 4-byte reads as above, split into three bf16 planes in the wave, six terms per float32 product,
// smallest first: 36 MFMAs of 8 passes per tile (NT = 3) instead of 48 of 16 passes, + ~180 split instructions.
template <int NT>
__device__ __forceinline__ void dw1_tile_steps6(const float* xt, const float* dzt, int c, int h, f32x16 (*acc)[2]) {
    if (NT == 0) return;
    float a0[8], a1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a0[e] = dzt[(8 * h + e) * 64 + c];
        a1[e] = dzt[(8 * h + e) * 64 + 32 + c];
    }
    float b[NT > 0 ? NT : 1][8];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) b[i][e] = xt[i * (kD2Rows * 32) + (8 * h + e) * 32 + c];
    bf8 A[2][3];
    split3(a0, A[0][0], A[0][1], A[0][2]);
    split3(a1, A[1][0], A[1][1], A[1][2]);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        bf8 B[3];
        split3(b[i], B[0], B[1], B[2]);
        acc[i][0] = prim::mfma_bf16(A[0][0], B[2], acc[i][0]);
        acc[i][1] = prim::mfma_bf16(A[1][0], B[2], acc[i][1]);
        acc[i][0] = prim::mfma_bf16(A[0][2], B[0], acc[i][0]);
        acc[i][1] = prim::mfma_bf16(A[1][2], B[0], acc[i][1]);
        acc[i][0] = prim::mfma_bf16(A[0][1], B[1], acc[i][0]);
        acc[i][1] = prim::mfma_bf16(A[1][1], B[1], acc[i][1]);
        acc[i][0] = prim::mfma_bf16(A[0][0], B[1], acc[i][0]);
        acc[i][1] = prim::mfma_bf16(A[1][0], B[1], acc[i][1]);
        acc[i][0] = prim::mfma_bf16(A[0][1], B[0], acc[i][0]);
        acc[i][1] = prim::mfma_bf16(A[1][1], B[0], acc[i][1]);
        acc[i][0] = prim::mfma_bf16(A[0][0], B[0], acc[i][0]);
        acc[i][1] = prim::mfma_bf16(A[1][0], B[0], acc[i][1]);
    }
}

// the whole kernel for a wave that owns NT (0 .. MAXNT) k tiles: tiles wave, wave + 4, wave + 8 (, wave + 12) of the slab
template <int NT, int MAXNT, int SLOTS, bool SIX = false>
__device__ __forceinline__ void dw1_direct_body(const Dw1Args& a, float* lds, int wave, int k0) {
    // SLOTS = 4: one workgroup per CU, the loads of tile m + 3 in flight; SLOTS = 2: half the LDS, two workgroups per CU
    // (two waves per SIMD), the loads of tile m + 1 in flight -- the other workgroup's MFMA stream covers the wait
    constexpr int kD2Slots = SLOTS, kD2TabRing = 2 * SLOTS;
    constexpr int kD2XSlot = d2_xslot(MAXNT), kD2Lds = d2_lds_slots(MAXNT, SLOTS);
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int din = a.rs.din;
    const long long rows = a.rs.rows;
    const long long ntiles = (rows + kD2Rows - 1) / kD2Rows;
    const long long n_it = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    float* xs = lds + wave * kD2Slots * kD2XSlot;           // this wave's slots
    float* dzs = lds + 4 * kD2Slots * kD2XSlot;             // the workgroup's dz1 slots
    constexpr int G = 2 * NT + 2;                           // loads per group

    auto row0_of = [&](long long m) {       // first launch row of this workgroup's m-th tile; past the end: the last tile
        if (m >= n_it) m = n_it - 1;        // again (loaded into a slot nobody reads, keeps every group the same size)
        return (blockIdx.x + m * gridDim.x) * kD2Rows;
    };
    // lane roles in a load: x -- row 8 g + (lane >> 3) of the tile, 16-byte piece lane & 7 of a k tile;
    // dz1 -- row 4 wave + (lane >> 4), piece lane & 15
    // The sampler's row table takes the same road, kD2Slots - 1 tiles further ahead: 16 entries per tile into a ring of
    // 256-byte LDS slots (lanes >= 16 repeat entry 15), read back with two ds_read_b32 per lane when the tile is issued.
    const int q8 = lane >> 3;
    int* tabs = reinterpret_cast<int*>(lds + kD2Lds) + wave * kD2TabRing * 64;
    auto issue_table = [&](long long t) {
        prim::load_lds4(a.rs.srow + row0_of(t) + (lane < 15 ? lane : 15), tabs + (int)(t % kD2TabRing) * 64);
    };
    auto issue = [&](long long m) {         // loads of tile m into slot m % kD2Slots + the table of tile m + kD2Slots - 1
        const int slot = (int)(m % kD2Slots);
        float* xslot = xs + slot * kD2XSlot;
        const int* tb = tabs + (int)(m % kD2TabRing) * 64;
        const int sr0 = tb[q8], sr1 = tb[8 + q8];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            int k = k0 + 32 * (wave + 4 * i) + 4 * (lane & 7);
            if (k > din - 4) k = din - 4;           // a piece past the row's end: its columns are never written out
            prim::load_lds16s(a.rs.src + (long long)sr0 * din + k, xslot + i * (kD2Rows * 32));
            prim::load_lds16s(a.rs.src + (long long)sr1 * din + k, xslot + i * (kD2Rows * 32) + 256);
        }
        long long r = row0_of(m) + 4 * wave + (lane >> 4);
        if (r >= rows) r = rows - 1;                // (zeroed in LDS before the product)
        prim::load_lds16s(a.dz1 + r * 64 + 4 * (lane & 15), dzs + slot * kD2DzSlot + wave * 256);
        issue_table(m + kD2Slots - 1);
    };
    constexpr int NA = NT > 0 ? NT : 1;
    f32x16 acc[NA][2];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][t][v] = 0.f;
    if (n_it == 0) return;
    for (int t = 0; t < kD2Slots - 1; ++t) issue_table(t);
    prim::wait_lds_loads<0>();
    for (int m = 0; m < kD2Slots - 1; ++m) issue(m);
    const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
    for (long long m = 0; m < n_it; ++m) {
        if (stamp && m < 40) a.dbg[512 + 4 * m] = prim::clock();
        // this wave's loads of tile m (and the table of tile m + kD2Slots - 1) have landed: all but the youngest groups
        prim::wait_lds_loads<(kD2Slots - 2) * G>();
        if (stamp && m < 40) a.dbg[512 + 4 * m + 1] = prim::clock();
        const long long live = rows - (blockIdx.x + m * gridDim.x) * kD2Rows;
        if (live < kD2Rows && 4 * wave + (lane >> 4) >= live)      // last tile: rows past the end of the launch count as zero
            *reinterpret_cast<v4*>(dzs + (int)(m % kD2Slots) * kD2DzSlot + wave * 256 + 4 * lane) = v4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();            // ... and everybody else's; all waves are done with tile m - 1's slot
        if (stamp && m < 40) a.dbg[512 + 4 * m + 2] = prim::clock();
        issue(m + kD2Slots - 1);
        if (stamp && m < 40) a.dbg[512 + 4 * m + 3] = prim::clock();
        const float* xt = xs + (int)(m % kD2Slots) * kD2XSlot;
        const float* dzt = dzs + (int)(m % kD2Slots) * kD2DzSlot;
        if (SIX) dw1_tile_steps6<NT>(xt, dzt, c, h, acc);
        else dw1_tile_steps<NT>(xt, dzt, c, h, acc);
    }
    prim::wait_lds_loads<0>();      // (the groups issued past the end)
    float* prow = a.partials + (long long)blockIdx.x * 64 * din;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int k = k0 + 32 * (wave + 4 * i) + c;
        if (k < din) {
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

template <int MAXNT, int SLOTS, bool SIX = false>
__global__ void __launch_bounds__(kThreads) mlp_dw1_direct_kernel(Dw1Args a) {
    float* lds = prim::lds();
    const int wave = prim::uniform(threadIdx.x >> 6);
    const int din = a.rs.din;
    constexpr int slab = 128 * MAXNT;
    const int k0 = blockIdx.y * slab;
    const int kw = (din - k0 < slab ? ((din - k0 + 31) / 32) * 32 : slab);
    const int ntk = kw / 32;
    const int n_own = (wave < ntk) + (wave + 4 < ntk) + (wave + 8 < ntk) + (MAXNT > 3 && wave + 12 < ntk);
    if (MAXNT > 3 && n_own == 4) dw1_direct_body<MAXNT, MAXNT, SLOTS, SIX>(a, lds, wave, k0);
    else if (n_own == 3) dw1_direct_body<3, MAXNT, SLOTS, SIX>(a, lds, wave, k0);
    else if (n_own == 2) dw1_direct_body<2, MAXNT, SLOTS, SIX>(a, lds, wave, k0);
    else if (n_own == 1) dw1_direct_body<1, MAXNT, SLOTS, SIX>(a, lds, wave, k0);
    else dw1_direct_body<0, MAXNT, SLOTS, SIX>(a, lds, wave, k0);
}

// ---- narrow inputs (din <= 192, din % 4 == 0): up to six k tiles do not split evenly among four waves (five tiles: one wave
// with two, three with one, a barrier per tile -- config 2's 152-wide critic input ran at 0.36 of the peak), so every wave
// owns ALL k tiles and its own 16-row tiles (strided over the launch wave by wave): the dz1 tile is private too, and the
// loop has no barrier at all.  Same load path and slot ring as above (SLOTS = 4 up to two k tiles, 2 beyond: 14-16 KB per
// slot); the four waves' accumulators are added through LDS once, at the end.
template <int NT, int SLOTS>
__global__ void __launch_bounds__(kThreads) mlp_dw1_rows_kernel(Dw1Args a) {
    float* lds = prim::lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = prim::uniform(tid >> 6), c = lane & 31, h = lane >> 5;
    const int din = a.rs.din;
    const long long rows = a.rs.rows;
    const long long ntiles = (rows + kD2Rows - 1) / kD2Rows;
    const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
    const long long n_it = gw < ntiles ? (ntiles - gw + nw - 1) / nw : 0;
    constexpr int SL = NT * (kD2Rows * 32) + kD2DzSlot;     // floats per slot: [NT][16][32] x | [16][64] dz1
    constexpr int G = 2 * NT + 4 + 1;                       // loads per group
    constexpr int RING = 2 * SLOTS;                         // row-table ring, 256-byte slots per wave
    float* ws = lds + wave * SLOTS * SL;
    int* tabs = reinterpret_cast<int*>(lds + 4 * SLOTS * SL) + wave * RING * 64;
    auto row0_of = [&](long long m) {
        if (m >= n_it) m = n_it - 1;
        return (gw + m * nw) * kD2Rows;
    };
    const int q8 = lane >> 3;
    auto issue_table = [&](long long t) {
        prim::load_lds4(a.rs.srow + row0_of(t) + (lane < 15 ? lane : 15), tabs + (int)(t % RING) * 64);
    };
    auto issue = [&](long long m) {
        float* slot = ws + (int)(m % SLOTS) * SL;
        const int* tb = tabs + (int)(m % RING) * 64;
        const int sr0 = tb[q8], sr1 = tb[8 + q8];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            int k = 32 * i + 4 * (lane & 7);
            if (k > din - 4) k = din - 4;
            prim::load_lds16s(a.rs.src + (long long)sr0 * din + k, slot + i * (kD2Rows * 32));
            prim::load_lds16s(a.rs.src + (long long)sr1 * din + k, slot + i * (kD2Rows * 32) + 256);
        }
        const long long r0 = row0_of(m);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            long long r = r0 + 4 * g + (lane >> 4);
            if (r >= rows) r = rows - 1;
            prim::load_lds16s(a.dz1 + r * 64 + 4 * (lane & 15), slot + NT * (kD2Rows * 32) + g * 256);
        }
        issue_table(m + SLOTS - 1);
    };
    f32x16 acc[NT][2];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][t][v] = 0.f;
    if (n_it > 0) {
        for (int t = 0; t < SLOTS - 1; ++t) issue_table(t);
        prim::wait_lds_loads<0>();
        for (int m = 0; m < SLOTS - 1; ++m) issue(m);
        for (long long m = 0; m < n_it; ++m) {
            prim::wait_lds_loads<(SLOTS - 2) * G>();
            float* slot = ws + (int)(m % SLOTS) * SL;
            const long long live = rows - (gw + m * nw) * kD2Rows;
            if (live < kD2Rows) {       // last tile: rows past the end of the launch count as zero
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (4 * g + (lane >> 4) >= live)
                        *reinterpret_cast<v4*>(slot + NT * (kD2Rows * 32) + g * 256 + 4 * lane) = v4{0.f, 0.f, 0.f, 0.f};
                prim::wave_sync();      // (a wave's LDS operations execute in order: only the compiler / the emulator care)
            }
            issue(m + SLOTS - 1);
            dw1_tile_steps<NT>(slot, slot + NT * (kD2Rows * 32), c, h, acc);
        }
        prim::wait_lds_loads<0>();
    }
    // ---- add the four waves' tiles pairwise through LDS ([set][i][t][v][lane], 2 NT 2048 floats <= the slots' room):
    // (w0 + w2) + (w1 + w3), wave 0 writes the workgroup's partial row
    float* red = lds;
#pragma unroll
    for (int half = 2; half >= 1; half >>= 1) {
        __syncthreads();
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) red[(wave - half) * (NT * 2048) + ((i * 2 + t) * 16 + v) * 64 + lane] = acc[i][t][v];
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[i][t][v] += red[wave * (NT * 2048) + ((i * 2 + t) * 16 + v) * 64 + lane];
        }
    }
    if (wave == 0) {
        float* prow = a.partials + (long long)blockIdx.x * 64 * din;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int k = 32 * i + c;
            if (k < din) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) prow[(long long)(32 * t + (v & 3) + 8 * (v >> 2) + 4 * h) * din + k] = acc[i][t][v];
            }
        }
    }
}

// out[e] = sum over n partial rows (row stride `stride`); fixed order.  A block handles 32 consecutive elements with 8
// row groups (group g sums rows g, g + 8, ...), combined through LDS.
__device__ __forceinline__ void reduce_chunk(const float* partials, long long n, long long stride, long long count,
                                             float* out, long long chunk, float* sh /* [8][32] */) {
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long long e = chunk * 32 + el;
    // four independent chains: the loop is bound by load latency, not bandwidth (a single chain ran at ~0.4 TB/s)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < count) {
        const float* p = partials + e;
        long long r = g;
        for (; r + 24 < n; r += 32) {
            s0 += p[r * stride];
            s1 += p[(r + 8) * stride];
            s2 += p[(r + 16) * stride];
            s3 += p[(r + 24) * stride];
        }
        for (; r < n; r += 8) s0 += p[r * stride];
    }
    sh[g * 32 + el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && e < count) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += sh[q * 32 + el];
        out[e] = t;
    }
}
// The same sums for 16-byte aligned partial rows (count, stride multiples of 4): a block owns 128 consecutive elements,
// every load moves 16 bytes, four rows in flight per thread.  sh: [8][128].
__device__ __forceinline__ void reduce_chunk4(const float* partials, long long n, long long stride, long long count,
                                              float* out, long long chunk, float* sh) {
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long long e = chunk * 128 + 4 * el;
    v4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (e < count) {
        const float* p = partials + e;
        long long r = g;
        for (; r + 24 < n; r += 32) {
            s0 += *reinterpret_cast<const v4*>(p + r * stride);
            s1 += *reinterpret_cast<const v4*>(p + (r + 8) * stride);
            s2 += *reinterpret_cast<const v4*>(p + (r + 16) * stride);
            s3 += *reinterpret_cast<const v4*>(p + (r + 24) * stride);
        }
        for (; r < n; r += 8) s0 += *reinterpret_cast<const v4*>(p + r * stride);
    }
    *reinterpret_cast<v4*>(sh + g * 128 + 4 * el) = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && e < count) {
        v4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 8; ++q) t += *reinterpret_cast<const v4*>(sh + q * 128 + 4 * el);
        *reinterpret_cast<v4*>(out + e) = t;
    }
}
__global__ void __launch_bounds__(kThreads) mlp_reduce_kernel(const float* partials, long long n, long long stride,
                                                              long long count, float* out) {
    reduce_chunk(partials, n, stride, count, out, blockIdx.x, prim::lds());
}

// ================================================================== input LayerNorm, parameter-free half ====
// dst[r, :] = (src[r, :] - mean_r) / sqrt(var_r + eps) (population variance, as nn.LayerNorm: mlp.py:47-48).  The
// observation fields of the rollout buffer do not change during the ppo epochs, so this runs once per train() and the
// trunk kernels read standardised rows with no per-row constants in their inner loops (the LayerNorm's affine half is
// folded into the first Linear by the caller).  16 lanes per row; the row is re-read from cache for the second moment
// and for the output.
// NV = 16-byte pieces per lane kept in registers (rows up to 64 * NV floats are read from HBM exactly once); NV = 0:
// any width, the row is re-read from cache for the second moment and for the output.
// Rows of dst are `ld` >= D floats apart; columns D .. ld - 1 (padding to a 16-byte multiple, so that the trunk kernels
// take their aligned paths for odd observation widths) are written as zeros.
template <int NV>
__global__ void __launch_bounds__(kThreads) standardize_rows_kernel(const float* src, long long rows, int D, float eps,
                                                                    float* dst, int ld) {
    const int sub = threadIdx.x & 15;
    const long long groups = ((long long)gridDim.x * kThreads) >> 4;
    const long long first = ((long long)blockIdx.x * kThreads + threadIdx.x) >> 4;
    const long long trips = (rows + groups - 1) / groups;       // the same for every lane: sum16 is a wave collective
    for (long long it = 0; it < trips; ++it) {
        const long long r = first + it * groups;
        const bool ok = r < rows;
        const float* p = src + (ok ? r : rows - 1) * D;
        v4 keep[NV > 0 ? NV : 1];
        float s = 0.f;
        if (NV > 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = 4 * sub + 64 * i;
                keep[i] = load4_guard(p + k, D - k);
                s += (keep[i][0] + keep[i][1]) + (keep[i][2] + keep[i][3]);
            }
        } else {
            for (int k = 4 * sub; k < D; k += 64) {
                const v4 x = load4_guard(p + k, D - k);
                s += (x[0] + x[1]) + (x[2] + x[3]);
            }
        }
        s = prim::sum16(s);
        const float mean = s / (float)D;
        float q = 0.f;
        if (NV > 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * sub + 64 * i + e < D) {
                        const float d = keep[i][e] - mean;
                        q += d * d;
                    }
        } else {
            for (int k = 4 * sub; k < D; k += 64) {
                const v4 x = load4_guard(p + k, D - k);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e < D) {
                        const float d = x[e] - mean;
                        q += d * d;
                    }
            }
        }
        q = prim::sum16(q);
        const float rstd = 1.f / sqrtf(q / (float)D + eps);
        if (ok) {
            float* o = dst + r * ld;
            if (sub == 0)
                for (int k = D; k < ld; ++k) o[k] = 0.f;
            if (NV > 0) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int k = 4 * sub + 64 * i;
                    v4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = (keep[i][e] - mean) * rstd;
                    if (k + 3 < D) {
                        *reinterpret_cast<v4u*>(o + k) = y;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (k + e < D) o[k + e] = y[e];
                    }
                }
            } else {
                for (int k = 4 * sub; k < D; k += 64) {
                    const v4 x = load4_guard(p + k, D - k);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < D) o[k + e] = (x[e] - mean) * rstd;
                }
            }
        }
    }
}

// ================================================================== host side ====
inline bool net_ok(const mappo_mlp_t* m) {
    if (m->n_layers < 1 || m->n_layers > MAPPO_MLP_MAX_LAYERS || m->din < 4 || m->out < 0 || m->out > 64) return false;
    if (m->act < 0 || m->act > 2) return false;
    return true;
}

inline int fill(const mappo_mlp_t* m, RowSrc& rs, Net& n) {
    if (!m || !m->src || !m->w1 || !m->row_tab) return MAPPO_E_NULL;
    if (!net_ok(m) || m->rows <= 0) return MAPPO_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(m->row_tab) & 15) != 0) return MAPPO_E_ALIGN;
    rs.src = m->src;
    rs.srow = m->row_tab;
    rs.rows = m->rows;
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

// tuning / test hook (mappo_mlp_set_grid_cap): upper bound on the workgroups of the persistent kernels; 0 = one or
// two per CU as the kernels were sized.  Small caps make every workgroup loop over many tiles (tests).
inline int& grid_cap_override() {
    static int cap = 0;
    return cap;
}
// option bits of mappo_mlp_set_flags that exist (tuning / tests only; arithmetic is the per-call `arith` field)
constexpr int kTuningBits = 1 | 2 | 4 | 8 | 32 | 128 | 256;
inline int& tuning_flags_ref() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MAPPO_MLP_FLAGS");
        v = e ? atoi(e) : 0;
        if (v & ~kTuningBits) {
            fprintf(stderr, "libmappo_hip: MAPPO_MLP_FLAGS=%d names option bits that do not exist (arithmetic is chosen per call "
                            "through the `arith` field); ignoring them\n", v);
            v &= kTuningBits;
        }
    }
    return v;
}
inline int tuning_flags() { return tuning_flags_ref(); }
inline int set_tuning_flags(int flags) {
    if (flags < 0 || (flags & ~kTuningBits)) return -1;
    const int old = tuning_flags_ref();
    tuning_flags_ref() = flags;
    return old;
}
inline bool arith_ok(int arith) { return arith == MAPPO_ARITH_SIX_TERM || arith == MAPPO_ARITH_F32_MFMA; }
inline long long*& debug_buffer() {
    static long long* p = nullptr;
    return p;
}
inline long long capped(long long grid, int default_cap) {
    const int cap = grid_cap_override() > 0 && grid_cap_override() < default_cap ? grid_cap_override() : default_cap;
    return grid > cap ? cap : grid;
}

inline int forward(const mappo_mlp_t* m, hipStream_t stream) {
    FwdArgs a;
    int code = fill(m, a.rs, a.net);
    if (code) return code;
    if (!m->y) return MAPPO_E_NULL;
    a.y = m->y;
    a.dbg = debug_buffer();
    a.flags = tuning_flags();
    for (int l = 0; l < 3; ++l) {
        a.z[l] = l < m->n_layers ? m->z[l] : nullptr;
        a.st[l] = l < m->n_layers ? m->ln_stats[l] : nullptr;
        if (a.z[l] != nullptr && a.st[l] == nullptr) return MAPPO_E_NULL;
    }
    const bool al = m->din % 4 == 0;
    if (!arith_ok(m->arith)) return MAPPO_E_FLAGS;
    const bool six_term = m->arith == MAPPO_ARITH_SIX_TERM;
    if (six_term && fwd4_takes(m->din, m->n_layers, m->out, (tuning_flags() & 128) != 0)) {
        // version 4: the first layer (and the hidden layer) as six bf16 x bf16 terms per float32 product on the bf16 matrix pipe
        const int nsc = (m->din + 63) / 64;
        const Fwd4Lds o4 = fwd4_lds(nsc);
        // k = 16 steps of a row's last chunk that hold real columns (3 runs the full chunk against zero weights: its
        // shortened instances spilled 70 bytes per lane)
        int nsl = (m->din - 64 * (nsc - 1) + 15) / 16;
        if (nsl == 3) nsl = 4;
        const long long grid4 = capped(ceil_div(rows128(m->rows) / 32, 4), kF3GridCap);
#define MAPPO_FWD4_NSL(AA, SS)                                                                                       \
    if (m->act == AA && nsl == SS) {                                                                                \
        MAPPO_LAUNCH((mlp_fwd4_kernel<AA, SS, true>), (unsigned)grid4, 64 * 4, (size_t)o4.total * 4, stream, a);    \
    }
#define MAPPO_FWD4_CASE(AA) MAPPO_FWD4_NSL(AA, 1) MAPPO_FWD4_NSL(AA, 2) MAPPO_FWD4_NSL(AA, 4)
        MAPPO_FWD4_CASE(0) MAPPO_FWD4_CASE(1) MAPPO_FWD4_CASE(2)
#undef MAPPO_FWD4_CASE
#undef MAPPO_FWD4_NSL
        return MAPPO_LAUNCH_ERROR();
    }
    if (fwd3_takes(m->din, m->n_layers, m->out) && !(tuning_flags() & 4)) {
        // version 3: operands straight from global memory, resident first-layer weights, two waves per SIMD.
        // Option bit 4 (mappo_mlp_set_flags / MAPPO_MLP_FLAGS) keeps the loader / compute kernel below
        const int nch = (m->din + 31) / 32;
        const bool hid6 = six_term && m->n_layers == 2;        // the hidden layer of two-layer trunks in six-term form
        const Fwd3Lds o3 = fwd3_lds(m->n_layers, m->out, nch, hid6);
        // groups of 8 real columns in a row's last chunk: 1 and 2 have shortened instances (3, and 2 with three layers, run
        // the full chunk: their shortened forms needed a few bytes of scratch for 8 MFMAs saved)
        int nql = (m->din - 32 * (nch - 1) + 7) / 8;
        if (nql == 3 || (nql == 2 && m->n_layers == 3)) nql = 4;
        const long long grid3 = capped(ceil_div(rows128(m->rows) / 32, 8), kF3GridCap);
#define MAPPO_FWD3_NQL(LL, AA, QQ)                                                                                   \
    if (nql == QQ) {                                                                                                \
        if (LL == 2 && hid6) {                                                                                      \
            MAPPO_LAUNCH((mlp_fwd3_kernel<2, AA, QQ, true>), (unsigned)grid3, 64 * 8, (size_t)o3.total * 4, stream, a);   \
        } else {                                                                                                    \
            MAPPO_LAUNCH((mlp_fwd3_kernel<LL, AA, QQ>), (unsigned)grid3, 64 * 8, (size_t)o3.total * 4, stream, a);  \
        }                                                                                                           \
    }
#define MAPPO_FWD3_CASE(LL, AA)                                                                                      \
    if (m->n_layers == LL && m->act == AA) {                                                                        \
        MAPPO_FWD3_NQL(LL, AA, 1) MAPPO_FWD3_NQL(LL, AA, 4)                                                         \
        if (LL == 2) MAPPO_FWD3_NQL(2, AA, 2)                                                                       \
    }
        MAPPO_FWD3_CASE(2, 0) MAPPO_FWD3_CASE(2, 1) MAPPO_FWD3_CASE(2, 2)
        MAPPO_FWD3_CASE(3, 0) MAPPO_FWD3_CASE(3, 1) MAPPO_FWD3_CASE(3, 2)
#undef MAPPO_FWD3_CASE
#undef MAPPO_FWD3_NQL
        return MAPPO_LAUNCH_ERROR();
    }
    const FwdLds o = fwd_lds(m->n_layers, m->out);
    const long long grid = capped(ceil_div(m->rows, kTR), kFwdGridCap);
#define MAPPO_FWD_CASE(AA, AL)                                                                                 \
    if (m->act == AA && al == AL) {                                                                            \
        MAPPO_LAUNCH((mlp_fwd_kernel<AA, AL>), (unsigned)grid, kFwdThreads, (size_t)o.total * 4, stream, a);   \
    }
    MAPPO_FWD_CASE(0, false) MAPPO_FWD_CASE(1, false) MAPPO_FWD_CASE(2, false)
    MAPPO_FWD_CASE(0, true) MAPPO_FWD_CASE(1, true) MAPPO_FWD_CASE(2, true)
#undef MAPPO_FWD_CASE
    return MAPPO_LAUNCH_ERROR();
}

inline long long chain_floats(int n_layers, int out) {
    return ((long long)(kBwdGridCap + 1) * r_total(n_layers, out) + 3) & ~3LL;
}
inline long long workspace_floats(int din, int n_layers, int out) {
    // chain partials (one row per workgroup) | reduced raw sums | (16-byte boundary) first-layer partials | ticket word
    // (2 x kD2GridCap partial rows: the two-workgroups-per-CU form of the direct weight-gradient kernel)
    return chain_floats(n_layers, out) + 2LL * kD2GridCap * 64LL * din + 4;
}

inline int backward(const mappo_mlp_t* m, hipStream_t stream) {
    BwdArgs b;
    int code = fill(m, b.rs, b.net);
    if (code) return code;
    if (!m->dy || !m->dz1 || !m->workspace || !m->grads) return MAPPO_E_NULL;
    for (int l = 0; l < 3; ++l) {
        b.z[l] = l < m->n_layers ? m->z[l] : nullptr;
        b.st[l] = l < m->n_layers ? m->ln_stats[l] : nullptr;
        if (l < m->n_layers && (!m->z[l] || !m->ln_stats[l])) return MAPPO_E_NULL;
    }
    const int L = m->n_layers, out = m->out, din = m->din;
    if (!arith_ok(m->arith)) return MAPPO_E_FLAGS;
    const bool six_term = m->arith == MAPPO_ARITH_SIX_TERM;
    b.dbg = debug_buffer();
    b.dy = m->dy;
    b.dz1 = m->dz1;
    b.partials = m->workspace;
    b.ticket = reinterpret_cast<unsigned*>(m->workspace + workspace_floats(m->din, m->n_layers, m->out) - 4);
    long long grid;
    bool fused1 = false;
    {
        // head sums in registers for the value head (HR = 1)
        const int hr = (out == 1 && L <= 2) ? 1 : 0;
        const bool six = six_term && L == 2;       // the tile's 64 x 64 products of two-layer trunks in six-term bf16 form
        // ... and for inputs of at most 64 aligned columns the first-layer weight gradient in the same launch (option bit 256
        // keeps the separate mlp_dw1_rows_kernel: A/B and tests)
        fused1 = six && din % 4 == 0 && din <= 64 && !(tuning_flags() & 256);
        const Bwd2Lds o = bwd2_lds<kB2Waves>(L, out, six, fused1);
        grid = capped(ceil_div(m->rows, 32 * kB2Waves), kBwdGridCap);
        b.p1 = m->workspace + chain_floats(L, out);
#define MAPPO_BWD_SIX(AA, HH)                                                                                          \
    if (six && m->act == AA && hr == HH) {                                                                            \
        if (fused1) {                                                                                                 \
            MAPPO_LAUNCH((mlp_bwd_kernel<2, AA, HH, true, true>), (unsigned)grid, 64 * kB2Waves, (size_t)o.total * 4, stream, b); \
        } else {                                                                                                      \
            MAPPO_LAUNCH((mlp_bwd_kernel<2, AA, HH, true>), (unsigned)grid, 64 * kB2Waves, (size_t)o.total * 4, stream, b); \
        }                                                                                                             \
    }
        MAPPO_BWD_SIX(0, 0) MAPPO_BWD_SIX(0, 1) MAPPO_BWD_SIX(1, 0) MAPPO_BWD_SIX(1, 1) MAPPO_BWD_SIX(2, 0) MAPPO_BWD_SIX(2, 1)
#undef MAPPO_BWD_SIX
#define MAPPO_BWD_CASE(LL, AA, HH)                                                                                    \
    if (!six && L == LL && m->act == AA && hr == HH) {                                                                \
        MAPPO_LAUNCH((mlp_bwd_kernel<LL, AA, HH>), (unsigned)grid, 64 * kB2Waves, (size_t)o.total * 4, stream, b);   \
    }
#define MAPPO_BWD_CASES(LL, AA) MAPPO_BWD_CASE(LL, AA, 0) MAPPO_BWD_CASE(LL, AA, 1)
        MAPPO_BWD_CASES(1, 0) MAPPO_BWD_CASES(1, 1) MAPPO_BWD_CASES(1, 2)
        MAPPO_BWD_CASES(2, 0) MAPPO_BWD_CASES(2, 1) MAPPO_BWD_CASES(2, 2)
        MAPPO_BWD_CASE(3, 0, 0) MAPPO_BWD_CASE(3, 1, 0) MAPPO_BWD_CASE(3, 2, 0)
#undef MAPPO_BWD_CASES
#undef MAPPO_BWD_CASE
    }
    code = MAPPO_LAUNCH_ERROR();
    if (code) return code;

    Dw1Args d;
    d.dbg = debug_buffer();
    d.rs = b.rs;
    d.dz1 = m->dz1;
    const long long rt = r_total(L, out);
    float* raw = m->workspace + (long long)kBwdGridCap * rt;        // reduced raw sums of the version-2 chain
    d.partials = m->workspace + chain_floats(L, out);
    const bool direct = din % 4 == 0 && din > kD2RowsMaxWidth;
    const int maxnt = d2_maxnt(din);
    const int gy = (int)ceil_div(din, direct ? 128 * maxnt : kDw1Slab);
    long long gx;
    if (fused1) {
        gx = grid;          // the chain's workgroups wrote the first-layer partial rows
    } else if (din % 4 == 0 && din <= kD2RowsMaxWidth) {
        gx = capped(ceil_div(m->rows, 4 * kD2Rows), kD2GridCap);
#define MAPPO_DW1_ROWS(NT, SLOTS)                                                                                       \
    {                                                                                                                   \
        constexpr int SL = NT * (kD2Rows * 32) + kD2DzSlot;                                                             \
        MAPPO_LAUNCH((mlp_dw1_rows_kernel<NT, SLOTS>), (unsigned)gx, kThreads,                                         \
                     (size_t)(4 * SLOTS * SL + 4 * 2 * SLOTS * 64) * 4, stream, d);                                     \
    }
        if (din <= 32) MAPPO_DW1_ROWS(1, 4)
        else if (din <= 64) MAPPO_DW1_ROWS(2, 4)
        else if (din <= 96) MAPPO_DW1_ROWS(3, 2)
        else if (din <= 128) MAPPO_DW1_ROWS(4, 2)
        else if (din <= 160) MAPPO_DW1_ROWS(5, 2)
        else MAPPO_DW1_ROWS(6, 2)
#undef MAPPO_DW1_ROWS
    } else if (direct) {
        if (six_term) {
            // the tile products as six bf16 x bf16 terms per float32 product on the bf16 matrix pipe.  Two slots per wave and
            // TWO workgroups per CU (61 / 77 KB of LDS each): a tile's operand reads and splits are a serial prefix of its
            // MFMA stream, and the second wave of the SIMD runs its MFMAs under it (round 5, alternating on one box: north
            // star 215.4 -> 210.4 ms per step).  Option bit 32 keeps the four-slot form, one workgroup per CU.
            const bool one_wg = (tuning_flags() & 32) != 0;
            const int cap = (one_wg ? 1 : 2) * kD2GridCap;
            gx = capped(ceil_div(m->rows, kD2Rows), cap / gy > 0 ? cap / gy : 1);
            if (maxnt == 4 && one_wg) {
                MAPPO_LAUNCH((mlp_dw1_direct_kernel<4, 4, true>), dim3((unsigned)gx, (unsigned)gy), kThreads,
                             (size_t)(d2_lds(4) + 4 * kD2TabRing * 64) * 4, stream, d);
            } else if (maxnt == 4) {
                MAPPO_LAUNCH((mlp_dw1_direct_kernel<4, 2, true>), dim3((unsigned)gx, (unsigned)gy), kThreads,
                             (size_t)(d2_lds_slots(4, 2) + 4 * 4 * 64) * 4, stream, d);
            } else if (one_wg) {
                MAPPO_LAUNCH((mlp_dw1_direct_kernel<3, 4, true>), dim3((unsigned)gx, (unsigned)gy), kThreads,
                             (size_t)(d2_lds(3) + 4 * kD2TabRing * 64) * 4, stream, d);
            } else {
                MAPPO_LAUNCH((mlp_dw1_direct_kernel<3, 2, true>), dim3((unsigned)gx, (unsigned)gy), kThreads,
                             (size_t)(d2_lds_slots(3, 2) + 4 * 4 * 64) * 4, stream, d);
            }
        } else if (maxnt == 4) {
            gx = capped(ceil_div(m->rows, kD2Rows), kD2GridCap / gy > 0 ? kD2GridCap / gy : 1);
            MAPPO_LAUNCH((mlp_dw1_direct_kernel<4, 4>), dim3((unsigned)gx, (unsigned)gy), kThreads,
                         (size_t)(d2_lds(4) + 4 * kD2TabRing * 64) * 4, stream, d);
        } else if (tuning_flags() & 32) {
            // tuning: two slots per wave, two workgroups per CU (61 KB of LDS each)
            gx = capped(ceil_div(m->rows, kD2Rows), 2 * kD2GridCap / gy > 0 ? 2 * kD2GridCap / gy : 1);
            MAPPO_LAUNCH((mlp_dw1_direct_kernel<3, 2>), dim3((unsigned)gx, (unsigned)gy), kThreads,
                         (size_t)(d2_lds_slots(3, 2) + 4 * 4 * 64) * 4, stream, d);
        } else {
            gx = capped(ceil_div(m->rows, kD2Rows), kD2GridCap / gy > 0 ? kD2GridCap / gy : 1);
            MAPPO_LAUNCH((mlp_dw1_direct_kernel<3, 4>), dim3((unsigned)gx, (unsigned)gy), kThreads,
                         (size_t)(d2_lds(3) + 4 * kD2TabRing * 64) * 4, stream, d);
        }
    } else {
        gx = capped(ceil_div(m->rows, kDw1Rows), kDw1GridCap / gy > 0 ? kDw1GridCap / gy : 1);
        const size_t dw1_lds = (size_t)2 * kDw1Stage * 4;
        MAPPO_LAUNCH(mlp_dw1_kernel, dim3((unsigned)gx, (unsigned)gy), kPipeThreads, dw1_lds, stream, d);
    }
    code = MAPPO_LAUNCH_ERROR();
    if (code) return code;

    // every slab's workgroups write disjoint k columns of their partial row; rows of unused workgroups do not exist
    const long long p1 = 64LL * din;
    TailArgs t;
    t.fin.net = b.net;
    t.fin.R = raw;
    t.fin.grads = m->grads;
    t.p1 = d.partials;
    t.n1 = gx;
    t.c1 = p1;
    t.p2 = b.partials;
    t.n2 = grid;
    t.c2 = rt;
    t.raw = raw;
    t.ticket = b.ticket;
    t.wide1 = ((reinterpret_cast<uintptr_t>(t.p1) | reinterpret_cast<uintptr_t>(m->grads)) & 15) == 0 && p1 % 4 == 0;
    t.nb1 = (int)ceil_div(p1, t.wide1 ? 128 : 32);
    t.wide2 = ((reinterpret_cast<uintptr_t>(t.p2) | reinterpret_cast<uintptr_t>(raw)) & 15) == 0 && rt % 4 == 0;
    t.nb2 = (int)ceil_div(rt, t.wide2 ? 128 : 32);
    MAPPO_LAUNCH(mlp_tail_kernel, (unsigned)(t.nb1 + t.nb2), kThreads, 4 * 1025, stream, t);
    return MAPPO_LAUNCH_ERROR();
}

inline int row_table(const long long* idx, long long rows, long long mb, int chunk_len, int T, int N, int A, int* tab,
                     hipStream_t stream) {
    if (!tab) return MAPPO_E_NULL;
    if (rows <= 0) return MAPPO_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(tab) & 15) != 0) return MAPPO_E_ALIGN;
    if (chunk_len > 0 && (!idx || mb <= 0 || T <= 0 || N <= 0 || A <= 0 || rows != mb * (long long)chunk_len))
        return MAPPO_E_SHAPE;
    RowMapArgs m;
    m.idx = idx;
    m.rows = rows;
    m.mb = mb;
    m.chunk_len = chunk_len;
    m.T = T;
    m.N = N;
    m.A = A;
    m.srow = tab;
    long long grid = ceil_div(rows128(rows), kThreads);
    if (grid > 256 * 8) grid = 256 * 8;
    MAPPO_LAUNCH(rowtab_kernel, (unsigned)grid, kThreads, 0, stream, m);
    return MAPPO_LAUNCH_ERROR();
}

inline int standardize_rows(const float* src, long long rows, int D, float eps, float* dst, int ld, hipStream_t stream) {
    if (!src || !dst) return MAPPO_E_NULL;
    if (rows <= 0 || D <= 0 || ld < D) return MAPPO_E_SHAPE;
    long long grid = ceil_div(rows * 16, kThreads);
    if (grid > 256 * 8) grid = 256 * 8;
    if (D <= 64) {
        MAPPO_LAUNCH(standardize_rows_kernel<1>, (unsigned)grid, kThreads, 0, stream, src, rows, D, eps, dst, ld);
    } else if (D <= 256) {
        MAPPO_LAUNCH(standardize_rows_kernel<4>, (unsigned)grid, kThreads, 0, stream, src, rows, D, eps, dst, ld);
    } else if (D <= 512) {
        MAPPO_LAUNCH(standardize_rows_kernel<8>, (unsigned)grid, kThreads, 0, stream, src, rows, D, eps, dst, ld);
    } else {
        MAPPO_LAUNCH(standardize_rows_kernel<0>, (unsigned)grid, kThreads, 0, stream, src, rows, D, eps, dst, ld);
    }
    return MAPPO_LAUNCH_ERROR();
}

}  // namespace mlp
#endif
