// K12: the GRU of the recurrent policies over a whole chunk in one launch per direction (hidden width 64).
// Reference: onpolicy/algorithms/utils/rnn.py:7-80 -- RNNLayer = nn.GRU (gates stacked r | z | n, PyTorch cell)
//     r = sigmoid(W_ir x + b_ir + W_hr hm + b_hr),  z = sigmoid(W_iz x + b_iz + W_hz hm + b_hz),
//     n = tanh(W_in x + b_in + r * (W_hn hm + b_hn)),  h' = n + z * (hm - n),   hm = h * mask   (rnn.py:43-77: the
//     state is reset where an episode ended; multiplying by the mask at every step is the same function)
// followed by LayerNorm(h') (rnn.py:22, :79), driven over the [L * mb, 64] chunk rows of recurrent_generator
// (shared_buffer.py:499-608: row l * mb + j = step l of chunk j).
//
// Why: round 2 walked the L = 10 steps of a chunk with one gate kernel + one hidden GEMM per step and direction, plus
// library GEMMs for the input projection / its gradient, bias reductions and a separate LayerNorm pair -- ~55 launches
// per network and minibatch, 65 % of the ns_rnn step, the [rows, 192] projections and gate gradients making round trips
// through HBM between them.  Here a wave owns 32 chunks (lane = chunk, like a row of K9) and walks their L steps with the
// state in registers: both projections of a step run on the f32 MFMA against weights held in LDS (2 x 52 KB, staged once
// per workgroup in the order the accumulator registers feed the next MFMA: mappo_mlp_impl.h, "orientation"), the gates,
// the mask reset and the output LayerNorm are evaluated on the accumulators.  The backward walks the steps in reverse
// (truncated BPTT inside the launch): LayerNorm and gate backward in registers, d x and d hm on the MFMA against the
// transposed weights; it writes the gate gradients once ([rows, 192] + the n third of the hidden side) for the two
// weight-gradient GEMMs, which stay split-K library GEMMs over all L * mb rows.
//
// Written against the primitives of mappo_mlp_impl.h (included before this header); tests/simt runs the same source on
// the host SIMT emulator.
#ifndef MAPPO_GRU_IMPL_H
#define MAPPO_GRU_IMPL_H

namespace gru {

using mlp::feat_of;
using mlp::kTS;
using mlp::kWS;
using mlp::f2;

constexpr int kWaves = 4;
constexpr int kThreads = 64 * kWaves;
constexpr int kGridCap = 256;        // one workgroup per CU (104 KB of weights in LDS)
constexpr int kSaved = 5;            // fragments saved per row and step: r, z, n, q = W_hn hm + b_hn, nhat

struct Args {
    const float* x;         // [L * mb, 64] layer input, row l * mb + j
    const float* h0;        // [mb, 64]
    const float* masks;     // [L * mb]
    const float* w_ih;      // [192, 64]
    const float* w_hh;      // [192, 64]
    const float* b_ih;      // [192]
    const float* b_hh;      // [192]
    const float* ln_g;      // [64]
    const float* ln_b;      // [64]
    float eps;
    long long mb;
    int L;
    float* y;               // [L * mb, 64] LayerNorm(h_l)
    float* h_last;          // [mb, 64] or NULL
    float* gates;           // [L][tiles][kSaved][2048] fragment order (NULL: nothing is saved)
    float* hm;              // [L * mb, 64] masked previous state of every step (NULL with gates)
    float* stats;           // [L * tiles * 32, 2] {mean, rstd} of the output LayerNorm
    // backward only
    const float* dy;        // [L * mb, 64]
    float* dx;              // [L * mb, 64]
    float* dgi;             // [L * mb, 192] gradient at W_ih x + b_ih (r | z | n)
    float* dq;              // [L * mb, 64]  gradient at W_hn hm + b_hn (the r and z thirds of the hidden side equal dgi's)
    float* dh0;             // [mb, 64] or NULL
    const float* dh_last;   // [mb, 64] gradient at h_last, or NULL (none)
    // optional output Linear on y (the Categorical head's Linear / v_out): logits = y head_w^T + head_b
    const float* head_w;    // [head_out, 64] or NULL
    const float* head_b;    // [head_out]
    int head_out;
    float* logits;          // [L * mb, head_out]
    const float* dlogits;   // backward: [L * mb, head_out]; dy = dlogits head_w is formed in the launch (a.dy is not read)
    int head_sums;          // backward: also leave the head's gradient sums (head_out <= kHeadSumOutputs)
    float* partials;        // [gridDim.x][kSums]: LayerNorm weight | bias gradient sums, then the column sums of the gate
                            // gradients (r | z | n of dgi, then dq) per workgroup
};
constexpr int kHeadSumOutputs = 6;                       // heads up to this width: their weight / bias gradient sums too
constexpr int kSums = 128 + 256 + kHeadSumOutputs * 64 + 32;     // ... | sum dlogits[o] n^[f] [6][64] | sum dlogits[o] [6] (+ pad)

__host__ __device__ __forceinline__ long long tiles_of(long long mb) { return (mb + 31) / 32; }

// acc[t] += sum over (h, s) of Wp[t][lane & 31][h * 32 + s] * reg[s] for the two 32-feature tiles at wp, wp + 32 kWS
__device__ __forceinline__ void dense64_acc(const float* wp, int c, int h, const float* reg, f32x16* acc) {
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
            acc[0] = prim::mfma32(a0[e], reg[4 * q + e], acc[0]);
            acc[1] = prim::mfma32(a1[e], reg[4 * q + e], acc[1]);
        }
    }
}

// ---- the same 64 -> 64 product in six-term bf16 arithmetic (opt-in, option bit 1024 of mappo_mlp_set_flags; the
// arithmetic: mappo_mlp_impl.h, mlp_fwd4_kernel): the weights sit in LDS as three bf16 planes, split once per workgroup --
// per 64 x 64 block [plane][feature tile][k = 16 step][lane] x 16 bytes (24 KB) -- and the lane's 32 operand values are
// split once per step and reused by every block they meet (three gates), so a step costs 4 x 12 MFMAs of 8 passes per
// block instead of 64 of 16 passes + two splits of ~145 vector instructions.
constexpr int kSixBlock = 3 * 2 * 4 * 256;      // floats
struct Split32 {
    bf8 p[4][3];                                // [k = 16 step j: slots 8 j .. 8 j + 7][plane]
};
__device__ __forceinline__ void split32(const float* reg, Split32& o) {
#pragma unroll
    for (int j = 0; j < 4; ++j) mlp::split3(reg + 8 * j, o.p[j][0], o.p[j][1], o.p[j][2]);
}
// planes of block `blk` of matrix W (rows 64 blk .. 64 blk + 63, [rows][64] row-major) -> LDS at dst
// (piece (plane, t, j, lane (cc, hh)) = W[64 blk + 32 t + cc][f(hh, 8 j ..)])
__device__ __forceinline__ void stage_six_block(const float* W, int blk, float* dst, int tid, int nthreads, bool transposed) {
    for (int e = tid; e < 512; e += nthreads) {
        const int t = e >> 8, cc = (e >> 3) & 31, hh = (e >> 2) & 1, j = e & 3;
        float w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = feat_of(hh, 8 * j + i);
            // forward: A[output feature][input k]; backward (transposed): A[input feature ki = 32 t + cc][gate output k]
            w[i] = transposed ? W[(64 * blk + k) * 64 + 32 * t + cc] : W[(64 * blk + 32 * t + cc) * 64 + k];
        }
        bf8 p1, p2, p3;
        mlp::split3(w, p1, p2, p3);
        float* base = dst + (t * 4 + j) * 256 + (32 * hh + cc) * 4;
        *reinterpret_cast<bf8*>(base) = p1;
        *reinterpret_cast<bf8*>(base + 2048) = p2;
        *reinterpret_cast<bf8*>(base + 4096) = p3;
    }
}
__device__ __forceinline__ void dense64_six(const float* blk /* block base in LDS */, int lane, const Split32& b, f32x16* acc) {
    const float* wt = blk + lane * 4;
    bf8 wn[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t) wn[p][t] = *reinterpret_cast<const bf8*>(wt + p * 2048 + (t * 4) * 256);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bf8 w[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t) w[p][t] = wn[p][t];
        if (j < 3) {        // next step's operands in flight behind this step's MFMAs (and nothing further ahead)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    wn[p][t] = *reinterpret_cast<const bf8*>(wt + p * 2048 + (t * 4 + (j < 3 ? j + 1 : 0)) * 256);
        }
        prim::sched_fence();
        acc[0] = prim::mfma_bf16(w[0][0], b.p[j][2], acc[0]);
        acc[1] = prim::mfma_bf16(w[0][1], b.p[j][2], acc[1]);
        acc[0] = prim::mfma_bf16(w[2][0], b.p[j][0], acc[0]);
        acc[1] = prim::mfma_bf16(w[2][1], b.p[j][0], acc[1]);
        acc[0] = prim::mfma_bf16(w[1][0], b.p[j][1], acc[0]);
        acc[1] = prim::mfma_bf16(w[1][1], b.p[j][1], acc[1]);
        acc[0] = prim::mfma_bf16(w[0][0], b.p[j][1], acc[0]);
        acc[1] = prim::mfma_bf16(w[0][1], b.p[j][1], acc[1]);
        acc[0] = prim::mfma_bf16(w[1][0], b.p[j][0], acc[0]);
        acc[1] = prim::mfma_bf16(w[1][1], b.p[j][0], acc[1]);
        acc[0] = prim::mfma_bf16(w[0][0], b.p[j][0], acc[0]);
        acc[1] = prim::mfma_bf16(w[0][1], b.p[j][0], acc[1]);
    }
}

__device__ __forceinline__ void zero2(f32x16* acc) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
}

// slot-order registers <-> one 2048-float fragment tile (see mlp::load_frag64)
__device__ __forceinline__ void store_frag64(float* tile, int lane, const float* reg) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        v4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = reg[4 * b + e];
        *reinterpret_cast<v4*>(tile + 4 * lane + 256 * b) = o;
    }
}

// a per-feature vector [64] in LDS -> this lane's 32 slots
__device__ __forceinline__ void slots_of(const float* vec, int h, float* out) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4 b = *reinterpret_cast<const v4*>(vec + 32 * t + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[16 * t + 4 * q + e] = b[e];
        }
}

// Column sums over the wave's 32 rows of a slot-order array: lane (h, c) returns the sum over the 32 lanes of its
// half-wave of slot c, i.e. of feature f(h, c) -- a reduce-scatter butterfly (31 exchange-and-add steps, each halving the
// values a lane still carries), all in registers.
__device__ __forceinline__ float colsum32(const float* v) {
    float w[16], x[8], y[4], z[2];
    prim::rs16_8(v, v + 16, w);
    prim::rs16_8(v + 8, v + 24, w + 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = prim::rs_step<8, 3>(w[i], w[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = prim::rs_step<7, 2>(x[i], x[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) z[i] = prim::rs_step<2, 1>(y[i], y[i + 2]);
    return prim::rs_step<1, 0>(z[0], z[1]);
}

__device__ __forceinline__ float sigmoid_fast(float x) {
    return prim::rcp_fast(1.f + prim::exp2_fast(x * -1.4426950408889634f));
}
// two elements per instruction where the hardware has a packed form (see mlp::act_fn2)
__device__ __forceinline__ f2 sigmoid_fast2(f2 x) {
    const f2 m = x * -1.4426950408889634f;
    const f2 d = f2{prim::exp2_fast(m[0]), prim::exp2_fast(m[1])} + 1.f;
    return f2{prim::rcp_fast(d[0]), prim::rcp_fast(d[1])};
}

// LDS (floats): forward  wih [6][32][kWS] | whh [6][32][kWS] | vec [6][64] = b_ir + b_hr, b_iz + b_hz, b_in, b_hn, gamma, beta
//               backward wihT [3][2][32][kWS] | whhT [3][2][32][kWS] | gamma [64] | per wave T [64][kTS]
constexpr int kW = 6 * 32 * kWS;
constexpr int kFwdLds = 2 * kW + 6 * 64;                 // + head_out * 65 with a head: whp [out][64] permuted | bias [out]
constexpr int kMaxHeadSteps = 9;                         // backward with a head: head_out <= 18 (k steps of 2 on the MFMA)
constexpr int kHeadA = 2 * kMaxHeadSteps * 64;           // whA [2 feature tiles][NJ][64 lanes]
constexpr int kSumsPerWave = 1280;                       // >= kSums + scratch [6][64] for the lanes' sums of dlogits
constexpr int kBwdLds = 2 * kW + 64 + kHeadA + kWaves * kSumsPerWave;

constexpr int kWSix = 3 * kSixBlock;                     // one matrix as three blocks of bf16 planes
constexpr int kFwdLdsSix = 2 * kWSix + 6 * 64;
template <bool SIX>
__global__ void __launch_bounds__(kThreads, 1) gru_seq_fwd_kernel(Args a) {
    float* lds = prim::lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = prim::uniform(tid >> 6), c = lane & 31, h = lane >> 5;
    constexpr int kWm = SIX ? kWSix : kW;       // floats per staged matrix
    if (SIX) {
        for (int g = 0; g < 3; ++g) {
            stage_six_block(a.w_ih, g, lds + g * kSixBlock, tid, kThreads, false);
            stage_six_block(a.w_hh, g, lds + kWSix + g * kSixBlock, tid, kThreads, false);
        }
    } else {
        // w[T][i][h * 32 + s] = W[32 T + i][f(h, s)]: A operand (lane = output feature) of the step that consumes slot s
        for (int e = tid; e < 192 * 64; e += kThreads) {
            const int fo = e >> 6, hs = e & 63, k = feat_of(hs >> 5, hs & 31);
            lds[fo * kWS + hs] = a.w_ih[fo * 64 + k];
            lds[kW + fo * kWS + hs] = a.w_hh[fo * 64 + k];
        }
    }
    float* vec = lds + 2 * kWm;
    for (int e = tid; e < 64; e += kThreads) {
        vec[e] = a.b_ih[e] + a.b_hh[e];
        vec[64 + e] = a.b_ih[64 + e] + a.b_hh[64 + e];
        vec[128 + e] = a.b_ih[128 + e];
        vec[192 + e] = a.b_hh[128 + e];
        vec[256 + e] = a.ln_g[e];
        vec[320 + e] = a.ln_b[e];
    }
    // head weights in slot order: whp[o][h * 32 + s] = head_w[o][f(h, s)]
    const int hout = a.head_out;
    float* whp = vec + 384;
    for (int e = tid; e < hout * 64; e += kThreads) whp[e] = a.head_w[(e >> 6) * 64 + feat_of((e & 63) >> 5, e & 31)];
    for (int e = tid; e < hout; e += kThreads) whp[hout * 64 + e] = a.head_b[e];
    __syncthreads();
    const long long mb = a.mb, ntiles = tiles_of(mb);
    const int L = a.L;
    for (long long tile = (long long)blockIdx.x * kWaves + wave; tile < ntiles; tile += (long long)gridDim.x * kWaves) {
        const long long j = tile * 32 + c;
        const bool ok = j < mb;
        const long long jj = ok ? j : mb - 1;       // rows past the end repeat the last chunk (never stored row-major)
        float hcur[32], xn[32];
        mlp::load_row64(a.h0 + jj * 64, hcur, h);
        mlp::load_row64(a.x + jj * 64, xn, h);
        float mk = a.masks[jj];
        for (int l = 0; l < L; ++l) {
            const long long row = (long long)l * mb + jj;
            float x[32], hm[32];
#pragma unroll
            for (int s = 0; s < 32; s += 2) {
                x[s] = xn[s];
                x[s + 1] = xn[s + 1];
                const f2 m2 = f2{hcur[s], hcur[s + 1]} * mk;
                hm[s] = m2[0];
                hm[s + 1] = m2[1];
            }
            if (l + 1 < L) {                        // the next step's input and mask, in flight behind this step's MFMAs
                mlp::load_row64(a.x + (row + mb) * 64, xn, h);
                mk = a.masks[row + mb];
            }
            if (a.hm != nullptr && ok) mlp::store_row64(a.hm + row * 64, hm, h);
            f32x16 ar[2], az[2], ai[2], ah[2];
            zero2(ar);
            zero2(az);
            zero2(ai);
            zero2(ah);
            if (SIX) {
                Split32 xs, hs6;
                split32(x, xs);
                split32(hm, hs6);
                dense64_six(lds + 0 * kSixBlock, lane, xs, ar);
                dense64_six(lds + kWSix + 0 * kSixBlock, lane, hs6, ar);
                dense64_six(lds + 1 * kSixBlock, lane, xs, az);
                dense64_six(lds + kWSix + 1 * kSixBlock, lane, hs6, az);
                dense64_six(lds + 2 * kSixBlock, lane, xs, ai);
                dense64_six(lds + kWSix + 2 * kSixBlock, lane, hs6, ah);
            } else {
                dense64_acc(lds + 0 * 64 * kWS, c, h, x, ar);
                dense64_acc(lds + kW + 0 * 64 * kWS, c, h, hm, ar);
                dense64_acc(lds + 1 * 64 * kWS, c, h, x, az);
                dense64_acc(lds + kW + 1 * 64 * kWS, c, h, hm, az);
                dense64_acc(lds + 2 * 64 * kWS, c, h, x, ai);
                dense64_acc(lds + kW + 2 * 64 * kWS, c, h, hm, ah);
            }
            float r[32], z[32], n[32], q[32];
            {
                float b0[32], b1[32];
                slots_of(vec, h, b0);
                slots_of(vec + 64, h, b1);
#pragma unroll
                for (int s = 0; s < 32; s += 2) {
                    const int t = s >> 4, v = s & 15;
                    const f2 rr = sigmoid_fast2(f2{ar[t][v], ar[t][v + 1]} + f2{b0[s], b0[s + 1]});
                    const f2 zz = sigmoid_fast2(f2{az[t][v], az[t][v + 1]} + f2{b1[s], b1[s + 1]});
                    r[s] = rr[0];
                    r[s + 1] = rr[1];
                    z[s] = zz[0];
                    z[s + 1] = zz[1];
                }
                slots_of(vec + 128, h, b0);
                slots_of(vec + 192, h, b1);
#pragma unroll
                for (int s = 0; s < 32; s += 2) {
                    const int t = s >> 4, v = s & 15;
                    const f2 qq = f2{ah[t][v], ah[t][v + 1]} + f2{b1[s], b1[s + 1]};
                    const f2 nn = mlp::act_fn2<1>(f2{ai[t][v], ai[t][v + 1]} + f2{b0[s], b0[s + 1]} + f2{r[s], r[s + 1]} * qq);
                    const f2 hh = nn + f2{z[s], z[s + 1]} * (f2{hm[s], hm[s + 1]} - nn);
                    q[s] = qq[0];
                    q[s + 1] = qq[1];
                    n[s] = nn[0];
                    n[s + 1] = nn[1];
                    hcur[s] = hh[0];
                    hcur[s + 1] = hh[1];
                }
            }
            // output LayerNorm on this lane's row (rnn.py:79)
            float nh[32], mean, rstd;
#pragma unroll
            for (int s = 0; s < 32; ++s) nh[s] = hcur[s];
            mlp::ln_stats32(nh, a.eps, mean, rstd);
            {
                float g[32], be[32];
                slots_of(vec + 256, h, g);
                slots_of(vec + 320, h, be);
                float yv[32];
#pragma unroll
                for (int s = 0; s < 32; s += 2) {
                    const f2 nn = f2{nh[s], nh[s + 1]} * rstd;
                    const f2 yy = nn * f2{g[s], g[s + 1]} + f2{be[s], be[s + 1]};
                    nh[s] = nn[0];
                    nh[s + 1] = nn[1];
                    yv[s] = yy[0];
                    yv[s + 1] = yy[1];
                }
                if (ok && a.y != nullptr) mlp::store_row64(a.y + row * 64, yv, h);
                // the output Linear on this lane's row: 32 in-lane terms per output + the other half-wave's
                for (int o = 0; o < hout; ++o) {
                    const float* wp = whp + o * 64 + 32 * h;
                    float p = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const v4 w = *reinterpret_cast<const v4*>(wp + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) p += w[e] * yv[4 * q + e];
                    }
                    p += prim::xhalf(p);
                    if (ok) a.logits[row * hout + o] = p + whp[hout * 64 + o];
                }
            }
            if (a.gates != nullptr) {
                float* gt = a.gates + ((long long)l * ntiles + tile) * (kSaved * 2048);
                store_frag64(gt, lane, r);
                store_frag64(gt + 2048, lane, z);
                store_frag64(gt + 2 * 2048, lane, n);
                store_frag64(gt + 3 * 2048, lane, q);
                store_frag64(gt + 4 * 2048, lane, nh);
                *reinterpret_cast<f2*>(a.stats + 2 * (((long long)l * ntiles + tile) * 32 + c)) = f2{mean, rstd};
            }
        }
        if (a.h_last != nullptr && ok) mlp::store_row64(a.h_last + j * 64, hcur, h);
    }
}

// NJ > 0: the gradient at y comes from an output Linear, dy = dlogits head_w, as NJ k steps of 2 outputs on the MFMA
// (head_out <= 2 NJ; padded outputs have zero weights)
// HO > 0: also leave the gradient sums of the head's first HO outputs (head_out <= HO <= kHeadSumOutputs)
// SIX (opt-in, option bit 1024 like the forward): the r and z blocks of both transposed matrices as bf16 planes (4 x 24 KB;
// all six would need 173 KB next to the per-wave sums), d a_r and d a_z split once and used by both matrices; the two n
// blocks (operands d a_n and d q) stay float32.  LDS: [W_ih^T r | W_ih^T z | W_hh^T r | W_hh^T z planes][W_ih^T n | W_hh^T n f32]
constexpr int kBwdWSix = 4 * kSixBlock + 2 * 64 * kWS;
constexpr int kBwdLdsSix = kBwdWSix + 64 + kHeadA + kWaves * kSumsPerWave;
// ALL6 (option bit 8192 with 1024; emulator-green at the end of round 4, device A / B pending): all six blocks as planes --
// the per-wave sums, which are written only after the last step, then take the planes' LDS over behind a barrier (152 KB).
constexpr int kBwdWAll6 = 6 * kSixBlock;
constexpr int kBwdLdsAll6 = kBwdWAll6 + 64 + kHeadA;
static_assert(kWaves * kSumsPerWave <= kBwdWAll6, "the sums alias the planes");
template <int NJ, int HO, bool SIX = false, bool ALL6 = false>
__global__ void __launch_bounds__(kThreads, 1) gru_seq_bwd_kernel(Args a) {
    static_assert(!ALL6 || SIX, "ALL6 is a form of SIX");
    float* lds = prim::lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = prim::uniform(tid >> 6), c = lane & 31, h = lane >> 5;
    if (ALL6) {
        for (int g = 0; g < 3; ++g) {
            stage_six_block(a.w_ih, g, lds + g * kSixBlock, tid, kThreads, true);
            stage_six_block(a.w_hh, g, lds + (3 + g) * kSixBlock, tid, kThreads, true);
        }
    } else if (SIX) {
        for (int g = 0; g < 2; ++g) {
            stage_six_block(a.w_ih, g, lds + g * kSixBlock, tid, kThreads, true);
            stage_six_block(a.w_hh, g, lds + (2 + g) * kSixBlock, tid, kThreads, true);
        }
        for (int e = tid; e < 64 * 64; e += kThreads) {
            const int ki = (e >> 6) & 63, hs = e & 63, fo = 128 + feat_of(hs >> 5, hs & 31);
            lds[4 * kSixBlock + ki * kWS + hs] = a.w_ih[fo * 64 + ki];
            lds[4 * kSixBlock + 64 * kWS + ki * kWS + hs] = a.w_hh[fo * 64 + ki];
        }
    } else {
    // wT[g][t][i][h * 32 + s] = W[64 g + f(h, s)][32 t + i]: A operand (lane = input feature) of d input = W_g^T d gate_g
    for (int e = tid; e < 3 * 64 * 64; e += kThreads) {
        const int g = e >> 12, ki = (e >> 6) & 63, hs = e & 63, fo = 64 * g + feat_of(hs >> 5, hs & 31);
        lds[g * 64 * kWS + ki * kWS + hs] = a.w_ih[fo * 64 + ki];
        lds[kW + g * 64 * kWS + ki * kWS + hs] = a.w_hh[fo * 64 + ki];
    }
    }
    float* gam = lds + (ALL6 ? kBwdWAll6 : (SIX ? kBwdWSix : 2 * kW));
    for (int e = tid; e < 64; e += kThreads) gam[e] = a.ln_g[e];
    // whA[t][j][lane (i, hh)] = head_w[2 j + hh][32 t + i]: A operand (lane = feature 32 t + i) of k step j
    float* whA = gam + 64;
    constexpr int NJ1 = NJ > 0 ? NJ : 1;
    const int hout = a.head_out;
    for (int e = tid; e < 2 * NJ * 64; e += kThreads) {
        const int t = e / (NJ1 * 64), jq = (e >> 6) % NJ1, ln = e & 63, o = 2 * jq + (ln >> 5);
        whA[e] = o < hout ? a.head_w[o * 64 + 32 * t + (ln & 31)] : 0.f;
    }
    float* T = ALL6 ? lds + wave * kSumsPerWave : gam + 64 + kHeadA + wave * kSumsPerWave;
    __syncthreads();
    // Parameter gradients that are column sums over every row and step -- the LayerNorm weight / bias gradients and the
    // bias gradients (= column sums of the gate gradients) -- are folded over the wave's rows step by step (colsum32), so
    // a lane carries ONE running sum per vector: that of feature f(h, c).  (Row-layout accumulators for the LayerNorm
    // pair alone were 64 registers of a kernel that spills; as separate passes over the [rows, 192] + [rows, 64] arrays
    // this kernel writes, the bias sums were 6 % of the recurrent north-star step.)
    float sgam = 0.f, sbet = 0.f;
    float sbr = 0.f, sbz = 0.f, sbn = 0.f, sbq = 0.f;
    // a narrow head's own gradients, dW_h = gamma (.) GH + beta (x) db_h with GH[o][f] = sum dlogits[o] n^[f] (y = n^ gamma +
    // beta is never written then): one column-sum butterfly per output and step; db_h as per-lane sums, added at the end
    constexpr bool kHeadSums = HO > 0 && NJ > 0 && 2 * NJ <= kHeadSumOutputs;
    constexpr int NO = kHeadSums ? 2 * NJ : 1;
    float gh[NO], dbh[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) gh[o] = dbh[o] = 0.f;
    const long long mb = a.mb, ntiles = tiles_of(mb);
    const int L = a.L;
    for (long long tile = (long long)blockIdx.x * kWaves + wave; tile < ntiles; tile += (long long)gridDim.x * kWaves) {
        const long long j = tile * 32 + c;
        const bool ok = j < mb;
        const long long jj = ok ? j : mb - 1;
        float carry[32];            // d loss / d h_l arriving from step l + 1 (from the caller's use of h_last at l = L - 1)
#pragma unroll
        for (int s = 0; s < 32; ++s) carry[s] = 0.f;
        if (a.dh_last != nullptr) {
            mlp::load_row64(a.dh_last + jj * 64, carry, h);
            if (!ok) {
#pragma unroll
                for (int s = 0; s < 32; ++s) carry[s] = 0.f;
            }
        }
        // What a step reads first (the normalised output, the gradient from above, the masked previous state: the last
        // two row-major, i.e. the slow ones) is fetched one step ahead into the registers the previous step has finished
        // with, so that the loads fly under the step's 384 MFMAs; the four gate fragments are loaded at the top of their
        // step, behind the LayerNorm backward (all seven streams a step ahead need more registers than a wave has)
        float r[32], z[32], n[32], q[32], nh[32], hm[32], g[32];
        float dl[NJ1];              // with a head: this lane's B operands, dlogits[row][2 j + h]
        f2 st;
        float mk;
        auto fetch = [&](int l) {
            const long long row = (long long)l * mb + jj;
            const float* gt = a.gates + ((long long)l * ntiles + tile) * (kSaved * 2048);
            mlp::load_frag64(gt + 4 * 2048, lane, nh);
            st = *reinterpret_cast<const f2*>(a.stats + 2 * (((long long)l * ntiles + tile) * 32 + c));
            mlp::load_row64(a.hm + row * 64, hm, h);
            if (NJ == 0) {
                mlp::load_row64(a.dy + row * 64, g, h);
            } else {
                // (always a valid address and no select: outputs past head_out meet zero weights)
#pragma unroll
                for (int jq = 0; jq < NJ; ++jq) {
                    const int o = 2 * jq + h;
                    dl[jq] = a.dlogits[row * hout + (o < hout ? o : hout - 1)];
                }
            }
            mk = a.masks[row];
        };
        fetch(L - 1);
        for (int l = L - 1; l >= 0; --l) {
            const long long row = (long long)l * mb + jj;
            const float mkl = mk;
            {
                const float* gt = a.gates + ((long long)l * ntiles + tile) * (kSaved * 2048);
                mlp::load_frag64(gt + 2048, lane, z);
                mlp::load_frag64(gt + 2 * 2048, lane, n);
                mlp::load_frag64(gt, lane, r);
                mlp::load_frag64(gt + 3 * 2048, lane, q);
            }
            if (NJ > 0) {       // dy = dlogits head_w on the MFMA: D[feature][row] += whA[feature][o] dlogits[o][row]
                f32x16 ag[2];
                zero2(ag);
#pragma unroll
                for (int jq = 0; jq < NJ; ++jq) {
                    ag[0] = prim::mfma32(whA[jq * 64 + lane], dl[jq], ag[0]);
                    ag[1] = prim::mfma32(whA[(NJ + jq) * 64 + lane], dl[jq], ag[1]);
                }
#pragma unroll
                for (int s = 0; s < 32; ++s) g[s] = ag[s >> 4][s & 15];
                if (kHeadSums) {
#pragma unroll
                    for (int jq = 0; jq < NJ; ++jq) {
                        const float mine = ok ? dl[jq] : 0.f, other = prim::xhalf(mine);
                        const float d0 = h ? other : mine, d1 = h ? mine : other;      // dlogits of outputs 2 jq, 2 jq + 1
                        float v[32];
#pragma unroll
                        for (int s = 0; s < 32; ++s) v[s] = d0 * nh[s];
                        gh[2 * jq] += colsum32(v);
                        dbh[2 * jq] += d0;
                        if (2 * jq + 1 < HO) {
#pragma unroll
                            for (int s = 0; s < 32; ++s) v[s] = d1 * nh[s];
                            gh[2 * jq + 1] += colsum32(v);
                            dbh[2 * jq + 1] += d1;
                        }
                    }
                }
            }
            // output LayerNorm backward (rows past the end: dy = 0 -> every gradient below is 0)
            float m1 = 0.f, m2 = 0.f;
            {
                float gslot[32], gn[32];
                slots_of(gam, h, gslot);
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    if (!ok) g[s] = 0.f;
                    gn[s] = g[s] * nh[s];
                }
                sbet += colsum32(g);
                sgam += colsum32(gn);
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    g[s] *= gslot[s];
                    m1 += g[s];
                    m2 += gn[s] * gslot[s];
                }
            }
            m1 += prim::xhalf(m1);
            m2 += prim::xhalf(m2);
            m1 *= (1.f / 64.f);
            m2 *= (1.f / 64.f);
            float dar[32], daz[32], dan[32], dqq[32], gz[32];
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const float gg = st[1] * ((g[s] - m1) - nh[s] * m2) + carry[s];        // d loss / d h_l
                gz[s] = gg * z[s];
                dan[s] = gg * (1.f - z[s]) * (1.f - n[s] * n[s]);
                daz[s] = gg * (hm[s] - n[s]) * z[s] * (1.f - z[s]);
                dqq[s] = dan[s] * r[s];
                dar[s] = dan[s] * q[s] * r[s] * (1.f - r[s]);
            }
            if (l > 0) fetch(l - 1);
            if (ok) {
                mlp::store_row64(a.dgi + row * 192, dar, h);
                mlp::store_row64(a.dgi + row * 192 + 64, daz, h);
                mlp::store_row64(a.dgi + row * 192 + 128, dan, h);
                mlp::store_row64(a.dq + row * 64, dqq, h);
            }
            sbr += colsum32(dar);       // (rows past the end carry zeros)
            sbz += colsum32(daz);
            sbn += colsum32(dan);
            sbq += colsum32(dqq);
            f32x16 acc[2];
            zero2(acc);
            Split32 srs, szs;
            if (ALL6) {
                Split32 sns;
                split32(dar, srs);
                split32(daz, szs);
                split32(dan, sns);
                dense64_six(lds + 0 * kSixBlock, lane, srs, acc);
                dense64_six(lds + 1 * kSixBlock, lane, szs, acc);
                dense64_six(lds + 2 * kSixBlock, lane, sns, acc);
            } else if (SIX) {
                split32(dar, srs);
                split32(daz, szs);
                dense64_six(lds + 0 * kSixBlock, lane, srs, acc);
                dense64_six(lds + 1 * kSixBlock, lane, szs, acc);
                dense64_acc(lds + 4 * kSixBlock, c, h, dan, acc);
            } else {
            dense64_acc(lds + 0 * 64 * kWS, c, h, dar, acc);
            dense64_acc(lds + 1 * 64 * kWS, c, h, daz, acc);
            dense64_acc(lds + 2 * 64 * kWS, c, h, dan, acc);
            }
            {
                float dxv[32];
#pragma unroll
                for (int s = 0; s < 32; ++s) dxv[s] = acc[s >> 4][s & 15];
                if (ok) mlp::store_row64(a.dx + row * 64, dxv, h);
            }
            zero2(acc);
            if (ALL6) {
                Split32 sqs;
                split32(dqq, sqs);
                dense64_six(lds + 3 * kSixBlock, lane, srs, acc);
                dense64_six(lds + 4 * kSixBlock, lane, szs, acc);
                dense64_six(lds + 5 * kSixBlock, lane, sqs, acc);
            } else if (SIX) {
                dense64_six(lds + 2 * kSixBlock, lane, srs, acc);
                dense64_six(lds + 3 * kSixBlock, lane, szs, acc);
                dense64_acc(lds + 4 * kSixBlock + 64 * kWS, c, h, dqq, acc);
            } else {
            dense64_acc(lds + kW + 0 * 64 * kWS, c, h, dar, acc);
            dense64_acc(lds + kW + 1 * 64 * kWS, c, h, daz, acc);
            dense64_acc(lds + kW + 2 * 64 * kWS, c, h, dqq, acc);
            }
#pragma unroll
            for (int s = 0; s < 32; ++s) carry[s] = (acc[s >> 4][s & 15] + gz[s]) * mkl;     // d loss / d h_{l-1}
        }
        if (a.dh0 != nullptr && ok) mlp::store_row64(a.dh0 + j * 64, carry, h);
    }
    // ---- the waves' sums added through LDS
    if (ALL6) __syncthreads();      // every wave has read its last weight plane: the sums take that LDS over
    prim::wave_sync();
    {
        const int f = feat_of(h, c);
        T[f] = sgam;
        T[64 + f] = sbet;
        T[128 + f] = sbr;
        T[192 + f] = sbz;
        T[256 + f] = sbn;
        T[320 + f] = sbq;
#pragma unroll
        for (int o = 0; o < kHeadSumOutputs; ++o) {
            T[384 + 64 * o + f] = (kHeadSums && o < NO) ? gh[o < NO ? o : 0] : 0.f;
            T[kSums + 64 * o + lane] = (kHeadSums && o < NO) ? dbh[o < NO ? o : 0] : 0.f;
        }
    }
    prim::wave_sync();
    if (lane < 32) {        // sum over this wave's rows of dlogits[o]: the h = 0 lanes' values (both half-waves hold a row's)
        float s = 0.f;
        if (lane < kHeadSumOutputs)
            for (int cc = 0; cc < 32; ++cc) s += T[kSums + 64 * lane + cc];
        T[384 + 64 * kHeadSumOutputs + lane] = s;
    }
    __syncthreads();
    for (int e = tid; e < kSums; e += kThreads) {
        float s = 0.f;
        for (int w = 0; w < kWaves; ++w) s += ALL6 ? lds[w * kSumsPerWave + e] : gam[64 + kHeadA + w * kSumsPerWave + e];
        a.partials[(long long)blockIdx.x * kSums + e] = s;
    }
}

inline long long grid_of(long long mb) {
    long long g = (tiles_of(mb) + kWaves - 1) / kWaves;
    const int cap = mlp::grid_cap_override() > 0 && mlp::grid_cap_override() < kGridCap ? mlp::grid_cap_override() : kGridCap;
    return g > cap ? cap : g;
}

inline int check(const mappo_gru_seq_t* m, bool backward) {
    if (!m || !m->x || !m->h0 || !m->masks || !m->w_ih || !m->w_hh || !m->b_ih || !m->b_hh || !m->ln_g || !m->ln_b)
        return MAPPO_E_NULL;
    if (m->mb <= 0 || m->L <= 0 || m->H != 64) return MAPPO_E_SHAPE;
    if ((m->gates != nullptr) != (m->hm != nullptr) || (m->gates != nullptr) != (m->stats != nullptr)) return MAPPO_E_NULL;
    if (m->head_out < 0 || m->head_out > 64 || (backward && m->head_out > 2 * kMaxHeadSteps)) return MAPPO_E_SHAPE;
    if (backward && m->head_sums && (m->head_out <= 0 || m->head_out > kHeadSumOutputs)) return MAPPO_E_SHAPE;
    if (m->head_out > 0 && (!m->head_w || !m->head_b || (backward ? !m->dlogits : !m->logits))) return MAPPO_E_NULL;
    if (!backward && !m->y && m->head_out == 0) return MAPPO_E_NULL;
    if (backward && (!m->gates || (!m->dy && m->head_out == 0) || !m->dx || !m->dgi || !m->dq || !m->ln_grads || !m->workspace))
        return MAPPO_E_NULL;
    const void* al[] = {m->x, m->h0, m->y, m->h_last, m->gates, m->hm, m->stats, m->dy, m->dx, m->dgi, m->dq, m->dh0, m->dh_last};
    for (const void* p : al)
        if (p && (reinterpret_cast<uintptr_t>(p) & 15) != 0) return MAPPO_E_ALIGN;
    return 0;
}

inline void fill(const mappo_gru_seq_t* m, Args& a) {
    a.x = m->x;
    a.h0 = m->h0;
    a.masks = m->masks;
    a.w_ih = m->w_ih;
    a.w_hh = m->w_hh;
    a.b_ih = m->b_ih;
    a.b_hh = m->b_hh;
    a.ln_g = m->ln_g;
    a.ln_b = m->ln_b;
    a.eps = m->ln_eps;
    a.mb = m->mb;
    a.L = m->L;
    a.y = m->y;
    a.h_last = m->h_last;
    a.gates = m->gates;
    a.hm = m->hm;
    a.stats = m->stats;
    a.dy = m->dy;
    a.dx = m->dx;
    a.dgi = m->dgi;
    a.dq = m->dq;
    a.dh0 = m->dh0;
    a.dh_last = m->dh_last;
    a.partials = m->workspace;
    a.head_w = m->head_w;
    a.head_b = m->head_b;
    a.head_out = m->head_out;
    a.logits = m->logits;
    a.dlogits = m->dlogits;
    a.head_sums = m->head_sums;
}

inline int forward(const mappo_gru_seq_t* m, hipStream_t stream) {
    int code = check(m, false);
    if (code) return code;
    Args a;
    fill(m, a);
    if (!mlp::arith_ok(m->arith)) return MAPPO_E_FLAGS;
    if (m->arith == MAPPO_ARITH_SIX_TERM) {
        // both projections of a step in six-term bf16 arithmetic
        MAPPO_LAUNCH(gru_seq_fwd_kernel<true>, (unsigned)grid_of(m->mb), kThreads, (size_t)(kFwdLdsSix + 65 * m->head_out) * 4, stream, a);
    } else {
        MAPPO_LAUNCH(gru_seq_fwd_kernel<false>, (unsigned)grid_of(m->mb), kThreads, (size_t)(kFwdLds + 65 * m->head_out) * 4, stream, a);
    }
    return MAPPO_LAUNCH_ERROR();
}

// six-term form: all six blocks of the transposed products as bf16 planes wherever that instance is built, else four
template <int NJ, int HO>
inline void launch_bwd(bool six, long long grid, hipStream_t stream, const Args& a) {
    constexpr bool kAll6Built = !(NJ == 3 && HO == 6);
    if (six) {
        if constexpr (kAll6Built) {
            MAPPO_LAUNCH((gru_seq_bwd_kernel<NJ, HO, true, true>), (unsigned)grid, kThreads, (size_t)kBwdLdsAll6 * 4, stream, a);
        } else {
            MAPPO_LAUNCH((gru_seq_bwd_kernel<NJ, HO, true>), (unsigned)grid, kThreads, (size_t)kBwdLdsSix * 4, stream, a);
        }
    } else {
        MAPPO_LAUNCH((gru_seq_bwd_kernel<NJ, HO, false>), (unsigned)grid, kThreads, (size_t)kBwdLds * 4, stream, a);
    }
}

inline int backward(const mappo_gru_seq_t* m, hipStream_t stream) {
    int code = check(m, true);
    if (code) return code;
    Args a;
    fill(m, a);
    const long long grid = grid_of(m->mb);
    const int ho = m->head_out;
    const bool hs = m->head_sums != 0;
    if (!mlp::arith_ok(m->arith)) return MAPPO_E_FLAGS;
    const bool six = m->arith == MAPPO_ARITH_SIX_TERM;
    // (the widest head-sum instance keeps the four-block form: with all six blocks as planes it spilled 28 bytes per lane)
#define MAPPO_GRU_BWD(NJ, HO) launch_bwd<NJ, HO>(six, grid, stream, a)
    if (ho == 0) {
        MAPPO_GRU_BWD(0, 0);
    } else if (ho <= 2) {
        if (hs && ho == 1) MAPPO_GRU_BWD(1, 1);
        else if (hs) MAPPO_GRU_BWD(1, 2);
        else MAPPO_GRU_BWD(1, 0);
    } else if (ho <= 6) {
        if (hs && ho <= 4) MAPPO_GRU_BWD(3, 4);
        else if (hs && ho == 5) MAPPO_GRU_BWD(3, 5);
        else if (hs) MAPPO_GRU_BWD(3, 6);
        else MAPPO_GRU_BWD(3, 0);
    } else {
        MAPPO_GRU_BWD(9, 0);
    }
#undef MAPPO_GRU_BWD
    MAPPO_LAUNCH(mlp::mlp_reduce_kernel, (unsigned)(kSums / 32), mlp::kThreads, 1024, stream, (const float*)m->workspace, grid,
                 (long long)kSums, (long long)kSums, m->ln_grads);
    return MAPPO_LAUNCH_ERROR();
}

// ================================================================== K12's weight gradients in six-term arithmetic (round 5) ====
//   dW_ih = dgi^T x            [192, 64]      (dgi [rows, 192]: gate gradients of the input side, x [rows, 64])
//   dW_hh = [dgi_r | dgi_z | dq]^T hm          (the hidden side's n gradient is dq [rows, 64]; hm = mask * h_{t-1} [rows, 64])
// Until round 5 three split-K library GEMMs + their slab sums (5.75 ms per call at the recurrent north star, 111 TFLOP/s).  Here
// ONE launch reads each of the four matrices once (1.5 KB per row): a workgroup walks 16-row tiles -- 24 KB of LDS per tile,
// contiguous in every matrix, so the direct-to-LDS loads need no address arithmetic beyond the tile's first row -- and keeps
// all 24 output tiles of 32 x 32 in accumulators: waves 0 / 1 own dW_ih (gate rows 0 .. 95 / 96 .. 191), waves 2 / 3 dW_hh;
// a wave reads and splits its 3 gate tiles and its 2 feature tiles (5 three-way splits) for 6 output tiles x 6 terms = 36
// MFMAs per tile: the shape and the structure of mlp_dw1_direct_kernel<3, 2, true> -- three slots, two workgroups per CU, the
// partner's MFMAs running under a tile's operand reads and splits.  Per-workgroup sums -> mlp_reduce_kernel.
struct WgradArgs {
    const float *dgi, *dq, *x, *hm;
    long long rows;
    float* partials;        // [gridDim.x][2][192][64]
};
constexpr int kWgRows = 16;
constexpr int kWgSlot = kWgRows * 384;      // floats: dgi [16][192] | dq [16][64] | x [16][64] | hm [16][64]
constexpr int kWgSlots = 3;
constexpr int kWgGridCap = 512;
constexpr int kWgOut = 2 * 192 * 64;
constexpr int kWgLoads = 6;                 // direct-to-LDS loads per wave and tile (24 KB / 4 waves / 1 KB)

__global__ void __launch_bounds__(kThreads, 2) gru_wgrad_kernel(WgradArgs a) {
    float* lds = prim::lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = prim::uniform(tid >> 6), c = lane & 31, h = lane >> 5;
    const long long rows = a.rows;
    const long long ntiles = (rows + kWgRows - 1) / kWgRows;
    const long long n_it = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    // loads of this workgroup's tile mi into slot mi % kWgSlots (past the end: the last tile again, into a slot nobody reads, so
    // that every group has the same size); instruction j = wave + 4 u moves floats [256 j, 256 j + 256) of the slot
    auto issue = [&](long long mi) {
        float* slot = lds + (int)(mi % kWgSlots) * kWgSlot;
        const long long m = mi < n_it ? mi : n_it - 1;
        const long long row0 = (blockIdx.x + m * gridDim.x) * kWgRows;
#pragma unroll
        for (int u = 0; u < kWgLoads; ++u) {
            const int j = wave + 4 * u;
            const float* src;
            int W, e;
            if (u < 3) {                            // j < 12: dgi
                src = a.dgi;
                W = 192;
                e = 256 * j + 4 * lane;
            } else {
                src = u == 3 ? a.dq : u == 4 ? a.x : a.hm;      // j = 12 + w, 16 + w, 20 + w
                W = 64;
                e = 256 * wave + 4 * lane;
            }
            const int r = e / W, col = e - r * W;
            long long gr = row0 + r;
            if (gr >= rows) gr = rows - 1;          // past the last row: that row again (its x / hm entries are zeroed in LDS)
            prim::load_lds16(src + gr * W + col, slot + 256 * j);
        }
    };
    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][t][v] = 0.f;
    if (n_it == 0) return;
    // this wave's operands inside a slot: three 32-column gate tiles (A) and the two feature tiles of x or hm (B)
    const int side = wave >> 1;                     // 0: dW_ih, 1: dW_hh
    int aoff[3], ald[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int mt = 3 * (wave & 1) + i;          // gate tile 0 .. 5
        const bool from_dq = side == 1 && mt >= 4;  // the hidden side's n block
        aoff[i] = from_dq ? 3072 + 32 * (mt - 4) : 32 * mt;
        ald[i] = from_dq ? 64 : 192;
    }
    const int boff = 3072 + 1024 + 1024 * side;
    for (int t = 0; t < kWgSlots - 1; ++t) issue(t);
    for (long long m = 0; m < n_it; ++m) {
        prim::wait_lds_loads<(kWgSlots - 2) * kWgLoads>();      // this wave's loads of tile m: all but the youngest group
        float* slot = lds + (int)(m % kWgSlots) * kWgSlot;
        const long long live = rows - (blockIdx.x + m * gridDim.x) * kWgRows;
        if (live < kWgRows && 4 * wave + (lane >> 4) >= live) {  // last tile: rows past the end count as zero (this wave's own loads)
            *reinterpret_cast<v4*>(slot + 4096 + 256 * wave + 4 * lane) = v4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<v4*>(slot + 5120 + 256 * wave + 4 * lane) = v4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();            // ... and everybody else's; all waves are done with tile m - 1's slot
        issue(m + kWgSlots - 1);
        float av[3][8], bv[2][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int i = 0; i < 3; ++i) av[i][e] = slot[aoff[i] + (8 * h + e) * ald[i] + c];
#pragma unroll
            for (int t = 0; t < 2; ++t) bv[t][e] = slot[boff + (8 * h + e) * 64 + 32 * t + c];
        }
        bf8 B[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) mlp::split3(bv[t], B[t][0], B[t][1], B[t][2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bf8 A[3];
            mlp::split3(av[i], A[0], A[1], A[2]);
            acc[i][0] = prim::mfma_bf16(A[0], B[0][2], acc[i][0]);
            acc[i][1] = prim::mfma_bf16(A[0], B[1][2], acc[i][1]);
            acc[i][0] = prim::mfma_bf16(A[2], B[0][0], acc[i][0]);
            acc[i][1] = prim::mfma_bf16(A[2], B[1][0], acc[i][1]);
            acc[i][0] = prim::mfma_bf16(A[1], B[0][1], acc[i][0]);
            acc[i][1] = prim::mfma_bf16(A[1], B[1][1], acc[i][1]);
            acc[i][0] = prim::mfma_bf16(A[0], B[0][1], acc[i][0]);
            acc[i][1] = prim::mfma_bf16(A[0], B[1][1], acc[i][1]);
            acc[i][0] = prim::mfma_bf16(A[1], B[0][0], acc[i][0]);
            acc[i][1] = prim::mfma_bf16(A[1], B[1][0], acc[i][1]);
            acc[i][0] = prim::mfma_bf16(A[0], B[0][0], acc[i][0]);
            acc[i][1] = prim::mfma_bf16(A[0], B[1][0], acc[i][1]);
        }
    }
    prim::wait_lds_loads<0>();      // (the groups issued past the end)
    float* prow = a.partials + (long long)blockIdx.x * kWgOut + side * (192 * 64);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int mt = 3 * (wave & 1) + i;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int g = 32 * mt + (v & 3) + 8 * (v >> 2) + 4 * h;
                prow[g * 64 + 32 * t + c] = acc[i][t][v];
            }
    }
}

inline long long wgrad_grid(long long rows) {
    // (at least four tiles per workgroup: its 96 KB row of partial sums has to be worth writing)
    long long g = ((rows + kWgRows - 1) / kWgRows + 3) / 4;
    const int cap = mlp::grid_cap_override() > 0 && mlp::grid_cap_override() < kWgGridCap ? mlp::grid_cap_override() : kWgGridCap;
    return g > cap ? cap : (g < 1 ? 1 : g);
}
inline long long wgrad_workspace_floats() { return (long long)kWgGridCap * kWgOut; }

// dw = [dW_ih [192, 64] | dW_hh [192, 64]] (the layout of nn.GRU's weight_ih_l0 / weight_hh_l0 gradients)
inline int weight_grads(const float* dgi, const float* dq, const float* x, const float* hm, long long rows, float* dw,
                        float* workspace, hipStream_t stream) {
    if (!dgi || !dq || !x || !hm || !dw || !workspace) return MAPPO_E_NULL;
    if (rows <= 0) return MAPPO_E_SHAPE;
    const void* al[] = {dgi, dq, x, hm, dw, workspace};
    for (const void* p : al)
        if ((reinterpret_cast<uintptr_t>(p) & 15) != 0) return MAPPO_E_ALIGN;
    WgradArgs a;
    a.dgi = dgi;
    a.dq = dq;
    a.x = x;
    a.hm = hm;
    a.rows = rows;
    a.partials = workspace;
    const long long grid = wgrad_grid(rows);
    MAPPO_LAUNCH(gru_wgrad_kernel, (unsigned)grid, kThreads, (size_t)kWgSlots * kWgSlot * 4, stream, a);
    int code = MAPPO_LAUNCH_ERROR();
    if (code) return code;
    MAPPO_LAUNCH(mlp::mlp_reduce_kernel, (unsigned)(kWgOut / 32), mlp::kThreads, 1024, stream, (const float*)workspace, grid,
                 (long long)kWgOut, (long long)kWgOut, dw);
    return MAPPO_LAUNCH_ERROR();
}

}  // namespace gru
#endif
