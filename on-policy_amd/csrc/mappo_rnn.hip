// K8: GRU cell gate arithmetic for the step-by-step recurrence of RNNLayer (reference
// onpolicy/algorithms/utils/rnn.py:7-80 drives nn.GRU; the cell is PyTorch's, gates stacked r|z|n):
//     r = sigmoid(gi_r + b_ir + gh_r + b_hr),  z = sigmoid(gi_z + b_iz + gh_z + b_hz),
//     n = tanh(gi_n + b_in + r * (gh_n + b_hn)),  h' = n + z * (h - n)
// with gi = x W_ih^T for all steps at once and gh = h W_hh^T per step (library GEMMs).
//
// These kernels do everything of a step that is not a GEMM, writing straight into the per-sequence
// buffers so that autograd needs no stack / unbind / per-step bias reductions:
//   forward  (step t): h_t into out[t]; the masked state of the NEXT step, h_t * mask[t+1], into hm[t+1]
//                      (the episode-boundary reset of rnn.py:43-77 applied at every step); r, z, n and
//                      gh_n + b_hn into the workspace ws[t] for the backward.
//   backward (step t): g = dout[t] + carry * mask[t+1] (carry = d loss / d hm[t+1]); gate gradients into
//                      dgi[t] and dgh[t]; the direct part of d loss / d hm[t] (g * z) into dhx.
// One thread handles 4 adjacent hidden units of one row (16-byte accesses); pure streaming, HBM-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mappo_hip.h"
#include "mappo_internal.h"

typedef float v4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <int VEC> struct V;
template <> struct V<4> { typedef v4 type; };
template <> struct V<1> { typedef float type; };
template <int VEC> __device__ __forceinline__ float get(const typename V<VEC>::type& u, int k) {
    if constexpr (VEC == 1) return u; else return u[k];
}
template <int VEC> __device__ __forceinline__ void put(typename V<VEC>::type& u, int k, float v) {
    if constexpr (VEC == 1) u = v; else u[k] = v;
}

template <int VEC>
__global__ void __launch_bounds__(256) gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                      const float* __restrict__ hm, const float* __restrict__ b_ih,
                                                      const float* __restrict__ b_hh,
                                                      const float* __restrict__ mask_next, float* __restrict__ h_out,
                                                      float* __restrict__ hm_next, float* __restrict__ ws,
                                                      long long B, int H) {
    using U = typename V<VEC>::type;
    const int upr = H / VEC;  // units per row
    const long long total = B * upr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / upr;
        const int u = (int)(i - row * upr);
        const U* gir = reinterpret_cast<const U*>(gi + row * 3 * H);
        const U* ghr = reinterpret_cast<const U*>(gh + row * 3 * H);
        const U ir = gir[u], iz = gir[upr + u], in = gir[2 * upr + u];
        const U hr = ghr[u], hz = ghr[upr + u], hn = ghr[2 * upr + u];
        const U hx = reinterpret_cast<const U*>(hm + row * H)[u];
        const U bir = reinterpret_cast<const U*>(b_ih)[u], biz = reinterpret_cast<const U*>(b_ih)[upr + u],
                bin = reinterpret_cast<const U*>(b_ih)[2 * upr + u];
        const U bhr = reinterpret_cast<const U*>(b_hh)[u], bhz = reinterpret_cast<const U*>(b_hh)[upr + u],
                bhn = reinterpret_cast<const U*>(b_hh)[2 * upr + u];
        const float mk = mask_next != nullptr ? mask_next[row] : 1.f;
        U r, z, n, q, hy, hmn;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float rr = sigmoidf_(get<VEC>(ir, k) + get<VEC>(bir, k) + get<VEC>(hr, k) + get<VEC>(bhr, k));
            float zz = sigmoidf_(get<VEC>(iz, k) + get<VEC>(biz, k) + get<VEC>(hz, k) + get<VEC>(bhz, k));
            float qq = get<VEC>(hn, k) + get<VEC>(bhn, k);
            float nn = tanhf(get<VEC>(in, k) + get<VEC>(bin, k) + rr * qq);
            float hh = nn + zz * (get<VEC>(hx, k) - nn);
            put<VEC>(r, k, rr);
            put<VEC>(z, k, zz);
            put<VEC>(n, k, nn);
            put<VEC>(q, k, qq);
            put<VEC>(hy, k, hh);
            put<VEC>(hmn, k, hh * mk);
        }
        reinterpret_cast<U*>(h_out + row * H)[u] = hy;
        if (hm_next != nullptr) reinterpret_cast<U*>(hm_next + row * H)[u] = hmn;
        if (ws != nullptr) {
            U* w = reinterpret_cast<U*>(ws + row * 4 * H);
            w[u] = r;
            w[upr + u] = z;
            w[2 * upr + u] = n;
            w[3 * upr + u] = q;
        }
    }
}

template <int VEC>
__global__ void __launch_bounds__(256) gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ carry,
                                                      const float* __restrict__ mask_next,
                                                      const float* __restrict__ ws, const float* __restrict__ hm,
                                                      float* __restrict__ dgi, float* __restrict__ dgh,
                                                      float* __restrict__ dhx, long long B, int H) {
    using U = typename V<VEC>::type;
    const int upr = H / VEC;
    const long long total = B * upr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / upr;
        const int u = (int)(i - row * upr);
        const U* w = reinterpret_cast<const U*>(ws + row * 4 * H);
        const U r = w[u], z = w[upr + u], n = w[2 * upr + u], q = w[3 * upr + u];
        const U hx = reinterpret_cast<const U*>(hm + row * H)[u];
        U g = U(0.f);
        if (dout != nullptr) g = reinterpret_cast<const U*>(dout + row * H)[u];
        if (carry != nullptr) {
            const float mk = mask_next != nullptr ? mask_next[row] : 1.f;
            const U c = reinterpret_cast<const U*>(carry + row * H)[u];
#pragma unroll
            for (int k = 0; k < VEC; ++k) put<VEC>(g, k, get<VEC>(g, k) + get<VEC>(c, k) * mk);
        }
        U gr, gz, gn, hn, dx;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float gg = get<VEC>(g, k), rr = get<VEC>(r, k), zz = get<VEC>(z, k), nn = get<VEC>(n, k);
            const float dn = gg * (1.f - zz) * (1.f - nn * nn);              // through h' = n + z (h - n) and tanh
            const float dz = gg * (get<VEC>(hx, k) - nn) * zz * (1.f - zz);  // through the update gate
            const float dr = dn * get<VEC>(q, k) * rr * (1.f - rr);          // through r * (gh_n + b_hn)
            put<VEC>(gr, k, dr);
            put<VEC>(gz, k, dz);
            put<VEC>(gn, k, dn);
            put<VEC>(hn, k, dn * rr);
            put<VEC>(dx, k, gg * zz);
        }
        U* a = reinterpret_cast<U*>(dgi + row * 3 * H);
        U* b = reinterpret_cast<U*>(dgh + row * 3 * H);
        a[u] = gr;
        a[upr + u] = gz;
        a[2 * upr + u] = gn;
        b[u] = gr;
        b[upr + u] = gz;
        b[2 * upr + u] = hn;
        reinterpret_cast<U*>(dhx + row * H)[u] = dx;
    }
}

int grid_for(long long total) {
    long long blocks = (total + 255) / 256;
    long long cap = (long long)mappo::kCUs * 16;
    return (int)(blocks < cap ? blocks : cap);
}

bool all16(std::initializer_list<const void*> ps) {
    for (const void* p : ps)
        if (p && !mappo::aligned_to(p, 16)) return false;
    return true;
}

}  // namespace

extern "C" int mappo_gru_cell_fwd(const float* gi, const float* gh, const float* hm, const float* b_ih,
                                  const float* b_hh, const float* mask_next, float* h_out, float* hm_next,
                                  float* ws, int64_t B, int H, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!gi || !gh || !hm || !b_ih || !b_hh || !h_out) return MAPPO_E_NULL;
    if (B <= 0 || H <= 0) return MAPPO_E_SHAPE;
    const bool vec = H % 4 == 0 && all16({gi, gh, hm, b_ih, b_hh, h_out, hm_next, ws});
    if (vec)
        hipLaunchKernelGGL((gru_fwd_kernel<4>), dim3(grid_for(B * (H / 4))), dim3(256), 0, stream, gi, gh, hm, b_ih,
                           b_hh, mask_next, h_out, hm_next, ws, (long long)B, H);
    else
        hipLaunchKernelGGL((gru_fwd_kernel<1>), dim3(grid_for(B * H)), dim3(256), 0, stream, gi, gh, hm, b_ih, b_hh,
                           mask_next, h_out, hm_next, ws, (long long)B, H);
    return (int)hipGetLastError();
}

extern "C" int mappo_gru_cell_bwd(const float* dout, const float* carry, const float* mask_next, const float* ws,
                                  const float* hm, float* dgi, float* dgh, float* dhx, int64_t B, int H,
                                  mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!ws || !hm || !dgi || !dgh || !dhx) return MAPPO_E_NULL;
    if (!dout && !carry) return MAPPO_E_NULL;
    if (B <= 0 || H <= 0) return MAPPO_E_SHAPE;
    const bool vec = H % 4 == 0 && all16({dout, carry, ws, hm, dgi, dgh, dhx});
    if (vec)
        hipLaunchKernelGGL((gru_bwd_kernel<4>), dim3(grid_for(B * (H / 4))), dim3(256), 0, stream, dout, carry,
                           mask_next, ws, hm, dgi, dgh, dhx, (long long)B, H);
    else
        hipLaunchKernelGGL((gru_bwd_kernel<1>), dim3(grid_for(B * H)), dim3(256), 0, stream, dout, carry, mask_next,
                           ws, hm, dgi, dgh, dhx, (long long)B, H);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- fused step (H = 64) ----
// The hidden projection of a forward step is a [B, 64] x [64, 192] GEMM whose output is consumed immediately
// by the gate arithmetic above.  As a library call this skinny GEMM runs far from any roofline (159 us for
// 268 MB of traffic at B = 262 144) and the [B, 192] intermediate makes a round trip through HBM.  For H = 64
// the whole forward step is one kernel: W_hh (48 KB) sits in LDS, a wave owns 32 rows, the products run on
// the f32 MFMA (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain) and the gates are evaluated on the
// accumulators.  (The mirror-image backward kernel was tried and was slower than cell kernel + library
// GEMM: its operands are row-per-lane, so every global access touched 64 different lines; it is not kept.)
//
// MFMA operand maps (cdna_hip_programming.md): A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31],
// C/D[row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)][col = lane & 31].  The two k slots of an instruction
// are fed with k = kk and k = 32 + kk (half-wave h takes the h-th half of the reduction range), so a lane keeps
// 32 CONTIGUOUS floats of its row as A operands.
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kH = 64, kG = 192;

__global__ void __launch_bounds__(256) gru_step_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ hm,
                                                           const float* __restrict__ w_hh,
                                                           const float* __restrict__ b_ih,
                                                           const float* __restrict__ b_hh,
                                                           const float* __restrict__ mask_next,
                                                           float* __restrict__ h_out, float* __restrict__ hm_next,
                                                           float* __restrict__ ws, long long B) {
    __shared__ float Wl[kH * kG];  // Wl[k][j] = W_hh[j][k]
    for (int e = threadIdx.x; e < kH * kG; e += 256) {
        int j = e / kH, k = e - j * kH;
        Wl[k * kG + j] = w_hh[e];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    // biases of this lane's two hidden units (c and 32 + c)
    float bir[2], biz[2], bin[2], bhr[2], bhz[2], bhn[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int u = 32 * s + c;
        bir[s] = b_ih[u];
        biz[s] = b_ih[kH + u];
        bin[s] = b_ih[2 * kH + u];
        bhr[s] = b_hh[u];
        bhz[s] = b_hh[kH + u];
        bhn[s] = b_hh[2 * kH + u];
    }
    const long long ntiles = (B + 31) / 32;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        const long long row0 = tile * 32;
        long long rowA = row0 + c;
        if (rowA >= B) rowA = B - 1;
        float a[32];
        const v4* ap = reinterpret_cast<const v4*>(hm + rowA * kH + 32 * h);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v4 t = ap[q];
            a[4 * q + 0] = t[0];
            a[4 * q + 1] = t[1];
            a[4 * q + 2] = t[2];
            a[4 * q + 3] = t[3];
        }
        f32x16 acc[6];
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const float* wrow = Wl + (32 * h + kk) * kG + c;
#pragma unroll
            for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], wrow[32 * t], acc[t], 0, 0, 0);
        }
        // gates on the accumulators: lane holds columns c and 32 + c of r | z | n for 16 rows.  All loads are
        // issued unconditionally on clamped rows (a branch per row would serialise 16 memory round trips);
        // only the stores are predicated.
        float gr[16][2], gz[16][2], gn[16][2], hx[16][2], mk[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            long long row = row0 + (v & 3) + 8 * (v >> 2) + 4 * h;
            if (row >= B) row = B - 1;
            mk[v] = mask_next != nullptr ? mask_next[row] : 1.f;
            const float* gir = gi + row * kG;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int u = 32 * s + c;
                gr[v][s] = gir[u];
                gz[v][s] = gir[kH + u];
                gn[v][s] = gir[2 * kH + u];
                hx[v][s] = hm[row * kH + u];
            }
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const long long row = row0 + (v & 3) + 8 * (v >> 2) + 4 * h;
            const bool ok = row < B;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int u = 32 * s + c;
                const float rr = sigmoidf_(gr[v][s] + bir[s] + acc[s][v] + bhr[s]);
                const float zz = sigmoidf_(gz[v][s] + biz[s] + acc[2 + s][v] + bhz[s]);
                const float qq = acc[4 + s][v] + bhn[s];
                const float nn = tanhf(gn[v][s] + bin[s] + rr * qq);
                const float hh = nn + zz * (hx[v][s] - nn);
                if (ok) {
                    h_out[row * kH + u] = hh;
                    if (hm_next != nullptr) hm_next[row * kH + u] = hh * mk[v];
                    if (ws != nullptr) {
                        float* w = ws + row * 4 * kH;
                        w[u] = rr;
                        w[kH + u] = zz;
                        w[2 * kH + u] = nn;
                        w[3 * kH + u] = qq;
                    }
                }
            }
        }
    }
}

extern "C" int mappo_gru_step_fwd(const float* gi, const float* hm, const float* w_hh, const float* b_ih,
                                  const float* b_hh, const float* mask_next, float* h_out, float* hm_next,
                                  float* ws, int64_t B, int H, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!gi || !hm || !w_hh || !b_ih || !b_hh || !h_out) return MAPPO_E_NULL;
    if (B <= 0 || H != kH) return MAPPO_E_SHAPE;
    if (!all16({gi, hm, w_hh, h_out, hm_next, ws})) return MAPPO_E_ALIGN;
    long long blocks = ((B + 31) / 32 + 3) / 4;
    if (blocks > mappo::kCUs * 2) blocks = mappo::kCUs * 2;
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, gi, hm, w_hh, b_ih, b_hh,
                       mask_next, h_out, hm_next, ws, (long long)B);
    return (int)hipGetLastError();
}
