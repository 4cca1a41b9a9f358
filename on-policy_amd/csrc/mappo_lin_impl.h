// K15: tall float32 GEMMs with 512 output features in six-term bf16 arithmetic -- the Linear layers of the hidden-512 trunks
// (BASELINE configs[4], Hanabi: reference onpolicy/algorithms/utils/mlp.py:6-58 at --hidden_size 512 --layer_N 2,
// scripts/train_hanabi_forward.sh:15-17; 85 % of that update's time was library float32 GEMMs):
//
//   lin_fwd_kernel    Y [rows, 512] = X [rows, K] W^T (+ bias)     forward of a Linear, and -- with the planes of W^T -- the
//                                                                   input gradient dX = dY W of a 512 -> 512 Linear
//   lin_wgrad_kernel  dW [512, K]   = dY^T [512, rows] X [rows, K]  weight gradient, the contraction runs over the rows
//
// Arithmetic: every float32 product from six bf16 x bf16 terms of the operands' exact three-way splits on
// v_mfma_f32_32x32x16_bf16, accumulated in float32 (include/mappo_hip.h MAPPO_ARITH_SIX_TERM; mlp::split3).  A split costs
// ~4.5 vector instructions per element; here an element of X meets 512 outputs (forward) and an element of dY / X meets 128 /
// 512 (weight gradient), so -- unlike in the 64-wide kernels of K9 -- the split disappears behind the matrix instructions and
// the kernels run at the bf16 pipe's pace: 6 x 2 x 512 x K FLOPs per row on a 2.5 PFLOP/s pipe against the library's float32
// GEMM on the 157 TFLOP/s float32 instruction.
//
// Both kernels move every operand with direct-to-LDS loads issued from inline assembly and counted s_waitcnt (the idiom of
// mlp_dw1_direct_kernel: the compiler sees no vector-memory load, so it inserts no wait of its own), one workgroup of four
// waves per CU, one wave per SIMD with its 256 accumulators in the AGPR half of the register file.
//
//   forward:  a workgroup owns 128 rows (32 per wave) x all 512 features.  Per k = 16 step the 48 KB block of W's three bf16
//             planes (pre-split once per call by lin_planes_kernel, stored in the order the MFMA's A operand reads it) streams
//             from L2 into one of two LDS buffers, each wave's [32 rows, 16 columns] piece of X into its own three-slot ring;
//             a lane splits its 8 values of X once and issues 16 feature tiles x 6 terms = 96 MFMAs against 48 16-byte LDS
//             reads.  One barrier per step.
//   wgrad:    a workgroup owns a [512 features, 128 columns] slab of dW and a range of rows; per 16-row step the [16, 512] tile
//             of dY and the [16, 128] tile of X land in one of three LDS slots, a wave (128 features x 128 columns: 4 x 4 MFMA
//             tiles) reads its operands transposed (8 rows of one column per lane), splits them and issues 96 MFMAs.  Partial
//             slabs of the row ranges are summed by lin_reduce_kernel in a fixed order.
#ifndef MAPPO_LIN_IMPL_H
#define MAPPO_LIN_IMPL_H

#include "mappo_mlp_impl.h"

namespace lin {

constexpr int kN = 512;                 // output features
constexpr int kThreads = 256;           // 4 waves
constexpr int kWBlock = 3 * 16 * 256;   // floats of one k = 16 block of planes: [plane 3][tile 16][lane 64] x 16 bytes = 48 KB
constexpr int kXSlot = 512;             // floats of a wave's [32 rows, 16 columns] piece of X
constexpr int kFwdLds = 3 * kWBlock + 4 * 2 * kXSlot;           // 144 KB + 16 KB = all of the CU's LDS
constexpr int kGridCap = 256;           // one workgroup per CU

struct FwdArgs {
    const float* x;         // [rows, ldx]
    long long rows;
    int K, ldx, nkb;        // nkb = ceil(K / 16) blocks of planes
    const float* planes;    // [nkb][kWBlock]
    const float* bias;      // [512] or nullptr
    float* y;               // [rows, 512]
    // NORM (the block's ReLU + LayerNorm in the epilogue): y receives the pre-activation x W^T + bias, yn the block's output
    const float* gamma;     // [512] LayerNorm weight
    const float* beta;      // [512] LayerNorm bias
    float eps;
    float* yn;              // [rows, 512] = gamma * (relu(y) - mean) * rstd + beta
    float* mean;            // [rows] statistics of relu(y) over the 512 features
    float* rstd;            // [rows] 1 / sqrt(biased variance + eps)
};

inline long long planes_floats(int K) { return (long long)((K + 15) / 16) * kWBlock; }

// planes of W [512, K] (row stride ldw) -- or, transposed = 1, of W^T for W [K = 512 rows of the contraction ..., i.e. element
// (feature f, k) is w[k * ldw + f] -- in MFMA A-operand order: block kb, plane p, tile t, lane (c, g): the 8 bf16
// W[32 t + c][16 kb + 8 g .. + 7] (zero beyond K)
__global__ void __launch_bounds__(kThreads) lin_planes_kernel(const float* w, int K, int ldw, int transposed, float* planes,
                                                              int nkb) {
    const long long total = (long long)nkb * 16 * 64;
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long long)gridDim.x * kThreads) {
        const int lane = (int)(e & 63), t = (int)((e >> 6) & 15);
        const long long kb = e >> 10;
        const int c = lane & 31, g = lane >> 5, f = 32 * t + c;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long k = 16 * kb + 8 * g + i;
            v[i] = k < K ? (transposed ? w[k * ldw + f] : w[(long long)f * ldw + k]) : 0.f;
        }
        bf8 p1, p2, p3;
        mlp::split3(v, p1, p2, p3);
        float* base = planes + kb * kWBlock + t * 256 + lane * 4;
        *reinterpret_cast<bf8*>(base) = p1;
        *reinterpret_cast<bf8*>(base + 4096) = p2;
        *reinterpret_cast<bf8*>(base + 8192) = p3;
    }
}

// X16: the rows of x start on 16-byte boundaries (ldx a multiple of 4 floats): two 16-byte pieces per lane and step; otherwise
// (Hanabi's 1285 / 1385-wide gathered minibatches) the same LDS image is filled with 4-byte pieces, eight per lane and step.
//
// Pipeline (round 5, second version; profiles/r05_lin512_*.json hold the measurements of all three).  With ONE wave per SIMD
// nothing overlaps what precedes a step's MFMAs in program order, and the first version (wait, barrier, 14 DMA issues, operand
// read, split, then 96 MFMAs) kept the matrix pipe 50 % busy.  Now a step's MFMAs run on operands prepared DURING the previous
// step:
//   W(j): block of planes of step j, in buffer j % 3, issued two steps ahead;
//   X(j): this wave's piece of step j, in its slot j % 2, read and split during step j - 1, issued three steps ahead into the
//         slot X(j - 2) was just read from.
// Step s: wait (all but the youngest W and X group) -> barrier -> [4 groups of 24 MFMAs on b(s), interleaved with: 12 DMA
// issues of W(s + 2), the read and the split of X(s + 1) -> b(s + 1), the DMA issue of X(s + 3)].  Every position the loads
// need (block of planes, row tile, column, buffer, slot) is a counter that advances with the step: no division in the loop.
// (A third version with [256 rows, 256 features] per workgroup -- half the plane traffic, X read twice -- was slower.)
//
// NORM (round 6): the block's ReLU and LayerNorm (reference onpolicy/algorithms/utils/mlp.py:17-22, Sequential(Linear, ReLU,
// LayerNorm)) in the tile's epilogue.  A lane holds 256 of the 512 features of ONE row (its partner in the other half-wave the
// other 256), so the row statistics are in-lane sums + one prim::xhalf exchange each.  Three passes that only READ the
// accumulators, value by value through prim::acc_get (ordinary uses of acc[t][v] in vector arithmetic -- in place or not, with
// or without scheduling fences -- made the register allocator move all 256 values into vector registers behind the loop and
// spill 116 bytes per lane); the ReLU is re-applied in each pass (one instruction): (1) store the pre-activation, sum relu; (2) squared deviations from the mean (the
// two-pass variance of K6's forward); (3) normalise, scale, shift, store.  ~2 300 vector instructions per lane and tile against >= 3 072 MFMA issue slots of ONE
// k-step: +3 % on a K = 512 launch, and the LayerNorm launch (one read of z, one write) is gone.
template <bool X16, bool BIAS, int GT, bool NORM = false>
__global__ void __launch_bounds__(kThreads, 1) lin_fwd_kernel(FwdArgs a) {
    constexpr int NG = 16 / GT;                     // groups of GT feature tiles per step (GT = 2: default; 4: tuning bit 2)
    static_assert(GT == 2 || GT == 4, "feature tiles per group");
    float* lds = prim::lds();
    float* wbuf = lds;                              // [3][kWBlock]
    float* xring = lds + 3 * kWBlock;               // [wave 4][slot 2][kXSlot]
    const int tid = threadIdx.x, lane = tid & 63, wave = prim::uniform(tid >> 6), c = lane & 31, g = lane >> 5;
    const int nkb = a.nkb;
    const long long ntiles = (a.rows + 127) / 128;
    const long long my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_tiles == 0) return;
    // ---- the W stream: step position (block kb, buffer); past the last step it stays on the last block (a buffer nobody reads)
    struct WPos {
        int kb, buf;
        long long left;     // steps that remain, this one included
    } wp = {0, 0, my_tiles * (long long)nkb};
    // (this wave's twelve 1 KB units of a block are contiguous in global memory and in LDS: three runs of four, each under one M0
    // set-up -- prim::load_lds16x4)
    auto issue_w4 = [&](int run) {
        const float* src = a.planes + (long long)wp.kb * kWBlock + (12 * wave + 4 * run) * 256 + 4 * lane;
        prim::load_lds16x4(src, wbuf + wp.buf * kWBlock + (12 * wave + 4 * run) * 256);
    };
    auto issue_w = [&]() {
        issue_w4(0);
        issue_w4(1);
        issue_w4(2);
    };
    auto next_w = [&]() {
        wp.buf = wp.buf == 2 ? 0 : wp.buf + 1;
        if (wp.left > 1) {
            --wp.left;
            wp.kb = wp.kb + 1 == nkb ? 0 : wp.kb + 1;
        }
    };
    // ---- the X stream: this wave's 32 rows x 16 columns of a step as two units: lane l of unit j loads the 16-byte piece
    // l >> 4 of row 16 j + (l & 15) -- the LDS image a half-wave then reads without bank conflicts (16 consecutive rows at one
    // piece = 16 consecutive 16-byte slots)
    struct XPos {
        int kb, slot;
        long long tile, left;
    } xp = {0, 0, (long long)blockIdx.x, my_tiles * (long long)nkb};
    auto issue_x = [&]() {
        float* dst = xring + (wave * 2 + xp.slot) * kXSlot;
        if (X16) {
            int k = 16 * xp.kb + 4 * (lane >> 4);
            if (k > a.ldx - 4) k = a.ldx - 4;       // a piece past the row's end: finite data against zero weights
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                long long r = xp.tile * 128 + 32 * wave + 16 * j + (lane & 15);
                if (r >= a.rows) r = a.rows - 1;
                prim::load_lds16(a.x + r * a.ldx + k, dst + j * 256);
            }
        } else {
            // unit u = (j, piece): lane l fills float e = l & 3 of the slot of row 16 j + (l >> 2)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                long long r = xp.tile * 128 + 32 * wave + 16 * (u >> 2) + (lane >> 2);
                if (r >= a.rows) r = a.rows - 1;
                int k = 16 * xp.kb + 4 * (u & 3) + (lane & 3);
                if (k > a.ldx - 1) k = a.ldx - 1;
                prim::load_lds4(reinterpret_cast<const int*>(a.x + r * a.ldx + k), reinterpret_cast<int*>(dst + u * 64));
            }
        }
        xp.slot ^= 1;
        if (xp.left > 1) {                          // (past the end: the last step again, into a slot nobody reads any more)
            --xp.left;
            if (++xp.kb == nkb) {
                xp.kb = 0;
                xp.tile += gridDim.x;
            }
        }
    };
    constexpr int kXGroup = X16 ? 2 : 8;            // loads of one issue_x
    // the three bf16 planes of this lane's 8 values of X in `slot`: row c, columns 8 g .. 8 g + 7 of the step
    auto read_x = [&](int slot, bf8& b1, bf8& b2, bf8& b3) {
        const float* xs = xring + (wave * 2 + slot) * kXSlot + (c >> 4) * 256 + ((2 * g) * 16 + (c & 15)) * 4;
        const v4 lo = *reinterpret_cast<const v4*>(xs), hi = *reinterpret_cast<const v4*>(xs + 64);
        float r[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r[i] = lo[i];
            r[4 + i] = hi[i];
        }
        mlp::split3(r, b1, b2, b3);
    };
    // prologue: the queue the steady state expects -- ..., W(s), X(s + 1), W(s + 1), X(s + 2)
    issue_x();                                      // X(0) -> slot 0
    issue_w();                                      // W(0) -> buffer 0
    next_w();
    issue_x();                                      // X(1) -> slot 1
    issue_w();                                      // W(1) -> buffer 1
    next_w();
    prim::wait_lds_loads<12 + 12 + kXGroup>();      // X(0)
    bf8 b1, b2, b3;
    read_x(0, b1, b2, b3);
    prim::wave_sync();                              // (X(2) goes into the slot X(0) was just read from -- by every lane)
    issue_x();                                      // X(2) -> slot 0
    int rbuf = 0, rslot = 1;                        // buffer of W(s), slot of X(s + 1)
    for (long long m = 0; m < my_tiles; ++m) {
        // the accumulators start from the bias (features 32 t + 8 q + 4 g + e of this lane; 64 16-byte loads per 128-row tile, L2
        // hits): the tile's epilogue is then 64 stores straight from the accumulator registers.  (Round 5 added the bias in the
        // epilogue: the compiler moved all 256 accumulators into vector registers first and spilled 36-38 of them, 148 / 156
        // bytes of scratch per lane.  The K9 kernels start their first layer from the bias the same way.)
        f32x16 acc[16];
        if (BIAS) {             // (a template parameter: a run-time choice made the compiler merge two sets of 256 start values)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4 b = *reinterpret_cast<const v4*>(a.bias + 32 * t + 8 * q + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][4 * q + e] = b[e];
                }
                prim::sched_fence();        // (tile by tile: no 64 bias loads in flight at once)
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
        }
        for (int kb = 0; kb < nkb; ++kb) {
            // this wave's part of W(s) and its X(s + 1) have landed: everything but the youngest W and X group
            prim::wait_lds_loads<12 + kXGroup>();
            __syncthreads();    // ... and everybody else's part of W(s); all waves are done with step s - 1's buffer
            const float* wb = wbuf + rbuf * kWBlock + lane * 4;
            bf8 nb1, nb2, nb3;
            // NG groups of GT feature tiles; the next group's 3 GT A operands are read while this group's 6 GT MFMAs issue (the
            // scheduling fences keep the compiler from hoisting all 48 reads to the top).  Two tiles per group (the default since
            // round 6; round 5 ran four) are enough to keep an MFMA from waiting for the one issued just before it and halve the
            // registers held by A operands.  The loads of the coming steps and the next step's operand ride along, a few
            // instructions per group.
            bf8 an[3][GT];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int t = 0; t < GT; ++t) an[p][t] = *reinterpret_cast<const bf8*>(wb + p * 4096 + t * 256);
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                bf8 af[3][GT];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int t = 0; t < GT; ++t) af[p][t] = an[p][t];
                if (q < NG - 1) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int t = 0; t < GT; ++t)
                            an[p][t] = *reinterpret_cast<const bf8*>(wb + p * 4096 + (GT * (q + 1) + t) * 256);
                }
                if (q == NG / 2 - 1) read_x(rslot, nb1, nb2, nb3);
                prim::sched_fence();
                // (in issue order: W(s + 2)'s twelve units in three runs, then X(s + 3))
                if (q % (NG / 4) == 0 && q / (NG / 4) < 3) issue_w4(q / (NG / 4));
                if (q == NG - 1) {                  // ... into the slot X(s + 1) was read from half a step ago (by every lane)
                    next_w();
                    prim::wave_sync();
                    issue_x();
                }
                // smallest terms first; term by term over the group's tiles, so that an MFMA never waits for the result of
                // the one issued just before it
#pragma unroll
                for (int t = 0; t < GT; ++t) acc[GT * q + t] = prim::mfma_bf16(af[0][t], b3, acc[GT * q + t]);
#pragma unroll
                for (int t = 0; t < GT; ++t) acc[GT * q + t] = prim::mfma_bf16(af[2][t], b1, acc[GT * q + t]);
#pragma unroll
                for (int t = 0; t < GT; ++t) acc[GT * q + t] = prim::mfma_bf16(af[1][t], b2, acc[GT * q + t]);
#pragma unroll
                for (int t = 0; t < GT; ++t) acc[GT * q + t] = prim::mfma_bf16(af[0][t], b2, acc[GT * q + t]);
#pragma unroll
                for (int t = 0; t < GT; ++t) acc[GT * q + t] = prim::mfma_bf16(af[1][t], b1, acc[GT * q + t]);
#pragma unroll
                for (int t = 0; t < GT; ++t) acc[GT * q + t] = prim::mfma_bf16(af[0][t], b1, acc[GT * q + t]);
            }
            b1 = nb1;
            b2 = nb2;
            b3 = nb3;
            rbuf = rbuf == 2 ? 0 : rbuf + 1;
            rslot ^= 1;
        }
        // the tile is complete: row c of this wave, features 32 t + 8 q + 4 g + e
        const long long tile = blockIdx.x + m * gridDim.x;
        const long long row = tile * 128 + 32 * wave + c;
        if (!NORM) {
            if (row < a.rows) {
                float* yr = a.y + row * kN + 4 * g;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = acc[t][4 * q + e];
                        *reinterpret_cast<v4*>(yr + 32 * t + 8 * q) = o;
                    }
                }
            }
        } else {
            // (rows past the end hold the LAST row's values -- their X was clamped to it, and a matrix instruction's result does
            // not depend on the lane that holds it -- and store them to the last row's place once more: no branch per store)
            const long long srow = row < a.rows ? row : a.rows - 1;
            float* yr = a.y + srow * kN + 4 * g;
            prim::mfma_drain();
            mlp::f2 s2 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 16; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = prim::acc_get(acc[t], 4 * q + e);
                    *reinterpret_cast<v4*>(yr + 32 * t + 8 * q) = o;
                    s2 += __builtin_elementwise_max(mlp::f2{o[0], o[1]}, mlp::f2{0.f, 0.f});
                    s2 += __builtin_elementwise_max(mlp::f2{o[2], o[3]}, mlp::f2{0.f, 0.f});
                    prim::sched_fence();
                }
            }
            float sum = s2[0] + s2[1];
            sum += prim::xhalf(sum);
            const float mu = sum * (1.f / (float)kN);
            mlp::f2 q2 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 16; ++t) {
#pragma unroll
                for (int v = 0; v < 16; v += 2) {
                    const mlp::f2 av = {prim::acc_get(acc[t], v), prim::acc_get(acc[t], v + 1)};
                    const mlp::f2 d = __builtin_elementwise_max(av, mlp::f2{0.f, 0.f}) - mu;
                    q2 += d * d;
                    if (v % 4 == 2) prim::sched_fence();
                }
            }
            float var = q2[0] + q2[1];
            var += prim::xhalf(var);
            const float r = 1.0f / sqrtf(var * (1.f / (float)kN) + a.eps);
            float* nr = a.yn + srow * kN + 4 * g;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4 gm = *reinterpret_cast<const v4*>(a.gamma + 32 * t + 8 * q + 4 * g);
                    const v4 bt = *reinterpret_cast<const v4*>(a.beta + 32 * t + 8 * q + 4 * g);
                    v4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float av = prim::acc_get(acc[t], 4 * q + e);
                        o[e] = ((av > 0.f ? av : 0.f) - mu) * r * gm[e] + bt[e];
                    }
                    *reinterpret_cast<v4*>(nr + 32 * t + 8 * q) = o;
                }
                prim::sched_fence();
            }
            a.mean[srow] = mu;
            a.rstd[srow] = r;
        }
    }
    prim::wait_lds_loads<0>();      // (the groups issued past the end)
}

// ------------------------------------------------------------------------------------------------ weight gradient
constexpr int kWgCols = 128;                        // columns of X per slab
constexpr int kWgDz = 16 * kN;                      // floats of a [16, 512] tile of dY
constexpr int kWgX = 16 * kWgCols;                  // floats of a [16, 128] tile of X
constexpr int kWgSlot = kWgDz + kWgX;               // 40 KB
constexpr int kWgSlots = 3;
constexpr int kWgLds = kWgSlots * kWgSlot;          // 120 KB
constexpr int kWgPlanes = 3 * 4 * 256;              // floats of the shared B planes of one X tile (SHB): 12 KB

struct WgArgs {
    const float* dy;        // [rows, 512]
    const float* x;         // [rows, ldx]
    long long rows;
    int K, ldx;
    float* partials;        // [gridDim.x][512][kp], kp = gridDim.y * 128
};

// Pipeline (second version, like lin_fwd_kernel): the operands of step m + 1 (8 rows of one column per lane: 64 4-byte LDS
// reads and eight 3-way splits) are read and split while the 96 MFMAs of step m issue; tiles are issued three steps ahead into
// a ring of three slots (tile j is read during step j - 1, so at step m the slots of tiles <= m are free).
//
// SHB (round 6, the default; tuning bit 8 keeps the other form): the four waves of a workgroup need the SAME four column tiles of
// X as their B operands (each wave owns 128 of the 512 features and all 128 columns), and until round 6 every wave read and split
// all of them: 32 of the 64 strided LDS reads and 4 of the 8 three-way splits of a step were done four times over.  Now wave w
// reads and splits column tile w of tile m + 1 at the start of step m, writes its three bf16 planes to a 12 KB LDS area in MFMA
// operand order, and after a second barrier (behind the step's first 24 MFMAs) every wave fetches the four tiles' planes with
// twelve 16-byte reads: 55 LDS instructions + 5 splits per step instead of 64 + 8 -- the non-MFMA stream that one wave per SIMD
// has to fit into the MFMAs' shadow shrinks by a third.
template <bool X16, bool SHB>
__global__ void __launch_bounds__(kThreads, 1) lin_wgrad_kernel(WgArgs a) {
    float* lds = prim::lds();
    float* bpl = lds + kWgLds;                      // SHB: [plane 3][column tile 4][lane 64] x 16 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = prim::uniform(tid >> 6), c = lane & 31, g = lane >> 5;
    const int col0 = blockIdx.y * kWgCols;
    const long long ntiles = (a.rows + 15) / 16;
    // contiguous ranges of 16-row tiles
    const long long per = (ntiles + gridDim.x - 1) / gridDim.x;
    const long long t0 = (long long)blockIdx.x * per;
    long long n_it = ntiles - t0;
    if (n_it > per) n_it = per;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    const int kp = gridDim.y * kWgCols;
    if (n_it > 0) {
        // loads of a step: dY -- 32 units of 1 KB (half a row each), 8 per wave; X -- 8 units (two rows of 128 columns each), 2 per wave
        auto issue = [&](long long m) {
            float* slot = lds + (int)(m % kWgSlots) * kWgSlot;
            if (m >= n_it) m = n_it - 1;            // (past the end: the last tile again, into a slot nobody reads any more)
            const long long row0 = (t0 + m) * 16;
            if (row0 + 16 <= a.rows) {
                // this wave's eight units are 8 KB that are contiguous in global memory and in LDS: two runs of four loads, each
                // under one M0 set-up
                const float* src = a.dy + (row0 + 4 * wave) * kN + 4 * lane;
                prim::load_lds16x4(src, slot + (8 * wave) * 256);
                prim::load_lds16x4(src + 1024, slot + (8 * wave + 4) * 256);
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int unit = 8 * wave + u;
                    long long r = row0 + (unit >> 1);
                    if (r >= a.rows) r = a.rows - 1;    // (zeroed in LDS before the product)
                    prim::load_lds16(a.dy + r * kN + 256 * (unit & 1) + 4 * lane, slot + unit * 256);
                }
            }
            if (X16) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int unit = 2 * wave + u;
                    long long r = row0 + 2 * unit + (lane >> 5);
                    if (r >= a.rows) r = a.rows - 1;
                    int k = col0 + 4 * (lane & 31);
                    if (k > a.ldx - 4) k = a.ldx - 4;       // a piece past the row's end: its columns are never written out
                    prim::load_lds16(a.x + r * a.ldx + k, slot + kWgDz + unit * 256);
                }
            } else {
                // 4-byte pieces: unit u of this wave = half a row (64 columns) of the tile's rows 4 wave .. 4 wave + 3
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = 4 * wave + (u >> 1);
                    long long r = row0 + row;
                    if (r >= a.rows) r = a.rows - 1;
                    int k = col0 + 64 * (u & 1) + lane;
                    if (k > a.ldx - 1) k = a.ldx - 1;
                    prim::load_lds4(reinterpret_cast<const int*>(a.x + r * a.ldx + k),
                                    reinterpret_cast<int*>(slot + kWgDz + row * kWgCols + 64 * (u & 1)));
                }
            }
        };
        constexpr int kGroup = X16 ? 10 : 16;       // loads of one issue
        // rows past the end of the matrix count as zero: this wave's units of the tile's dY rows >= live
        auto zero_tail = [&](long long m) {
            const long long live = a.rows - (t0 + m) * 16;
            if (m < n_it && live < 16) {
                float* slot = lds + (int)(m % kWgSlots) * kWgSlot;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int unit = 8 * wave + u;
                    if ((unit >> 1) >= live) *reinterpret_cast<v4*>(slot + unit * 256 + 4 * lane) = v4{0.f, 0.f, 0.f, 0.f};
                }
            }
        };
        // operands of tile m: A[i] = dY[rows 8 g .. 8 g + 7][feature 128 wave + 32 i + c], B[j] = X[rows 8 g ..][column 32 j + c]
        auto operand_a = [&](long long m, int i, bf8* A3) {
            const float* dz = lds + (int)(m % kWgSlots) * kWgSlot + 128 * wave + c;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = dz[(8 * g + e) * kN + 32 * i];
            mlp::split3(v, A3[0], A3[1], A3[2]);
        };
        auto operand_b = [&](long long m, int j, bf8* B3) {
            const float* xt = lds + (int)(m % kWgSlots) * kWgSlot + kWgDz + c;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xt[(8 * g + e) * kWgCols + 32 * j];
            mlp::split3(v, B3[0], B3[1], B3[2]);
        };
        // SHB: this wave's column tile (j = wave) of tile m -> the shared planes; the planes of column tile j -> registers
        auto write_b = [&](long long m) {
            bf8 q[3];
            operand_b(m, wave, q);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<bf8*>(bpl + (p * 4 + wave) * 256 + lane * 4) = q[p];
        };
        auto read_b = [&](int j, bf8* B3) {
#pragma unroll
            for (int p = 0; p < 3; ++p) B3[p] = *reinterpret_cast<const bf8*>(bpl + (p * 4 + j) * 256 + lane * 4);
        };
        issue(0);
        issue(1);
        issue(2);
        prim::wait_lds_loads<2 * kGroup>();         // tile 0
        zero_tail(0);
        __syncthreads();
        bf8 A[4][3], B[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) operand_a(0, i, A[i]);
        if (SHB) {
            write_b(0);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) read_b(j, B[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) operand_b(0, j, B[j]);
        }
        for (long long m = 0; m < n_it; ++m) {
            prim::wait_lds_loads<kGroup>();         // this wave's loads of tile m + 1: all but the youngest group (tile m + 2)
            zero_tail(m + 1);
            __syncthreads();                        // ... and everybody else's; all waves have read tile m (during step m - 1)
                                                    // (SHB: ... and the shared planes of tile m, during step m - 1)
            issue(m + 3);                           // -> the slot of tile m
            bf8 nA[4][3], nB[4][3];
            // smallest terms first, term by term over a feature tile's four column tiles (an MFMA never waits for the one before
            // it); the next step's operands are read and split alongside, one feature tile and one column tile per group
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                operand_a(m + 1, i, nA[i]);
                if (SHB) {
                    if (i == 0) write_b(m + 1);
                    if (i == 1) {                   // (behind the barrier that follows group 0)
                        read_b(0, nB[0]);
                        read_b(1, nB[1]);
                    }
                    if (i == 2) {
                        read_b(2, nB[2]);
                        read_b(3, nB[3]);
                    }
                } else {
                    operand_b(m + 1, i, nB[i]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = prim::mfma_bf16(A[i][0], B[j][2], acc[i][j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = prim::mfma_bf16(A[i][2], B[j][0], acc[i][j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = prim::mfma_bf16(A[i][1], B[j][1], acc[i][j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = prim::mfma_bf16(A[i][0], B[j][1], acc[i][j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = prim::mfma_bf16(A[i][1], B[j][0], acc[i][j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = prim::mfma_bf16(A[i][0], B[j][0], acc[i][j]);
                if (SHB && i == 0) __syncthreads();     // every wave's planes of tile m + 1 are written
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    A[i][p] = nA[i][p];
                    B[i][p] = nB[i][p];
                }
        }
        prim::wait_lds_loads<0>();
    }
    // partial slab of this row range (zeros for a range without rows): D row = feature within the tile, D column = lane & 31
    // (the lane's position is taken afresh from the hardware lane counter: kept alive across the loop -- where every vector
    // register is in use -- it was the one value the SHB form spilled)
    const int le = prim::lane_again();
    const int ce = le & 31, ge = le >> 5;
    float* prow = a.partials + (long long)blockIdx.x * kN * kp;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int f = 128 * wave + 32 * i + (v & 3) + 8 * (v >> 2) + 4 * ge;
                prow[(long long)f * kp + col0 + 32 * j + ce] = acc[i][j][v];
            }
}

// dw[f][k] = sum over the row ranges of partials[b][f][k], k < K (fixed order: deterministic run to run)
__global__ void __launch_bounds__(kThreads) lin_reduce_kernel(const float* partials, int n, int kp, int K, float* dw) {
    const long long total = (long long)kN * K;
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long long)gridDim.x * kThreads) {
        const long long f = e / K, k = e - f * K;
        float s = 0.f;
        for (int b = 0; b < n; ++b) s += partials[((long long)b * kN + f) * kp + k];
        dw[e] = s;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
inline int prepare(const float* w, int K, int ldw, int transposed, float* planes, hipStream_t stream) {
    if (!w || !planes) return MAPPO_E_NULL;
    if (K <= 0 || ldw <= 0 || (!transposed && ldw < K) || (transposed && ldw < kN)) return MAPPO_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(planes) & 15) != 0) return MAPPO_E_ALIGN;
    const int nkb = (K + 15) / 16;
    long long grid = ((long long)nkb * 1024 + kThreads - 1) / kThreads;
    if (grid > 2048) grid = 2048;
    MAPPO_LAUNCH(lin_planes_kernel, (unsigned)grid, kThreads, 0, stream, w, K, ldw, transposed, planes, nkb);
    return MAPPO_LAUNCH_ERROR();
}

inline int forward(const float* x, long long rows, int K, int ldx, const float* planes, const float* bias, float* y,
                   hipStream_t stream) {
    if (!x || !planes || !y) return MAPPO_E_NULL;
    if (rows <= 0 || K <= 0 || ldx < K || ldx < 4) return MAPPO_E_SHAPE;
    if (((reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(y)) & 15) != 0) return MAPPO_E_ALIGN;
    if ((reinterpret_cast<uintptr_t>(x) & 3) != 0) return MAPPO_E_ALIGN;
    if (bias && (reinterpret_cast<uintptr_t>(bias) & 15) != 0) return MAPPO_E_ALIGN;      // read as 16-byte vectors
    const bool x16 = ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    FwdArgs a;
    a.x = x;
    a.rows = rows;
    a.K = K;
    a.ldx = ldx;
    a.nkb = (K + 15) / 16;
    a.planes = planes;
    a.bias = bias;
    a.y = y;
    a.gamma = a.beta = nullptr;
    a.eps = 0.f;
    a.yn = a.mean = a.rstd = nullptr;
    const long long grid = mlp::capped((rows + 127) / 128, kGridCap);
    const bool four = (mlp::tuning_flags() & 2) != 0;      // tuning bit 2: four feature tiles per MFMA group (round 5's form)
#define MAPPO_LIN_FWD(X, B)                                                                                        \
    do {                                                                                                           \
        if (four) MAPPO_LAUNCH((lin_fwd_kernel<X, B, 4>), (unsigned)grid, kThreads, (size_t)kFwdLds * 4, stream, a); \
        else MAPPO_LAUNCH((lin_fwd_kernel<X, B, 2>), (unsigned)grid, kThreads, (size_t)kFwdLds * 4, stream, a);      \
    } while (0)
    if (x16 && bias) MAPPO_LIN_FWD(true, true);
    else if (x16) MAPPO_LIN_FWD(true, false);
    else if (bias) MAPPO_LIN_FWD(false, true);
    else MAPPO_LIN_FWD(false, false);
#undef MAPPO_LIN_FWD
    return MAPPO_LAUNCH_ERROR();
}

// the block form: y = x W^T + bias (the pre-activation K6's backward wants), yn = LayerNorm(relu(y)), mean / rstd per row
inline int forward_norm(const float* x, long long rows, int K, int ldx, const float* planes, const float* bias,
                        const float* gamma, const float* beta, float eps, int act, float* y, float* yn, float* mean,
                        float* rstd, hipStream_t stream) {
    if (!x || !planes || !bias || !gamma || !beta || !y || !yn || !mean || !rstd) return MAPPO_E_NULL;
    if (rows <= 0 || K <= 0 || ldx < K || ldx < 4) return MAPPO_E_SHAPE;
    if (act != 2) return MAPPO_E_FLAGS;             // 2 = relu (the reference's default --use_ReLU)
    if (((reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(yn) |
          reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) != 0)
        return MAPPO_E_ALIGN;
    if ((reinterpret_cast<uintptr_t>(x) & 3) != 0) return MAPPO_E_ALIGN;
    const bool x16 = ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    FwdArgs a;
    a.x = x;
    a.rows = rows;
    a.K = K;
    a.ldx = ldx;
    a.nkb = (K + 15) / 16;
    a.planes = planes;
    a.bias = bias;
    a.y = y;
    a.gamma = gamma;
    a.beta = beta;
    a.eps = eps;
    a.yn = yn;
    a.mean = mean;
    a.rstd = rstd;
    const long long grid = mlp::capped((rows + 127) / 128, kGridCap);
    if (x16) MAPPO_LAUNCH((lin_fwd_kernel<true, true, 2, true>), (unsigned)grid, kThreads, (size_t)kFwdLds * 4, stream, a);
    else MAPPO_LAUNCH((lin_fwd_kernel<false, true, 2, true>), (unsigned)grid, kThreads, (size_t)kFwdLds * 4, stream, a);
    return MAPPO_LAUNCH_ERROR();
}

inline int wgrad_slabs(int K) { return (K + kWgCols - 1) / kWgCols; }
inline int wgrad_ranges(int K) {
    const int r = kGridCap / wgrad_slabs(K);
    return r > 0 ? r : 1;
}
inline long long wgrad_workspace_floats(int K) { return (long long)wgrad_ranges(K) * kN * wgrad_slabs(K) * kWgCols; }

inline int wgrad(const float* dy, const float* x, long long rows, int K, int ldx, float* dw, float* workspace,
                 hipStream_t stream) {
    if (!dy || !x || !dw || !workspace) return MAPPO_E_NULL;
    if (rows <= 0 || K <= 0 || ldx < K || ldx < 4) return MAPPO_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(dy) & 15) != 0 || (reinterpret_cast<uintptr_t>(x) & 3) != 0) return MAPPO_E_ALIGN;
    const bool x16 = ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    WgArgs a;
    a.dy = dy;
    a.x = x;
    a.rows = rows;
    a.K = K;
    a.ldx = ldx;
    a.partials = workspace;
    const int gy = wgrad_slabs(K);
    int gx = wgrad_ranges(K);
    const int cap = mlp::grid_cap_override();       // (tests: few row ranges, many steps each)
    if (cap > 0 && cap < gx) gx = cap;
    const bool own_b = (mlp::tuning_flags() & 8) != 0;     // tuning bit 8: every wave splits all four column tiles itself (round 5)
    const size_t lds_shb = (size_t)(kWgLds + kWgPlanes) * 4;
    if (x16 && !own_b) MAPPO_LAUNCH((lin_wgrad_kernel<true, true>), dim3((unsigned)gx, (unsigned)gy), kThreads, lds_shb, stream, a);
    else if (!own_b) MAPPO_LAUNCH((lin_wgrad_kernel<false, true>), dim3((unsigned)gx, (unsigned)gy), kThreads, lds_shb, stream, a);
    else if (x16) MAPPO_LAUNCH((lin_wgrad_kernel<true, false>), dim3((unsigned)gx, (unsigned)gy), kThreads, (size_t)kWgLds * 4, stream, a);
    else MAPPO_LAUNCH((lin_wgrad_kernel<false, false>), dim3((unsigned)gx, (unsigned)gy), kThreads, (size_t)kWgLds * 4, stream, a);
    int code = MAPPO_LAUNCH_ERROR();
    if (code) return code;
    long long grid = ((long long)kN * K + kThreads - 1) / kThreads;
    if (grid > 2048) grid = 2048;
    MAPPO_LAUNCH(lin_reduce_kernel, (unsigned)grid, kThreads, 0, stream, (const float*)workspace, gx, gy * kWgCols, K, dw);
    return MAPPO_LAUNCH_ERROR();
}

}  // namespace lin
#endif
