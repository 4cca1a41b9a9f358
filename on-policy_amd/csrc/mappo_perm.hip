// K10: sort-free minibatch index lists for the samplers.
//
// The reference draws torch.randperm(B) and cuts it into num_mini_batch slices (onpolicy/utils/shared_buffer.py:360-361,
// :415-416, :511-512).  On the device that permutation cost 8 radix-sort passes + duplicate handling per epoch (5 % of
// the north-star step) and produced index lists in random memory order, i.e. every gathered row was a random DRAM access.
// What a minibatch needs is only WHICH samples it holds: the loss is a mean over the minibatch, the order of its rows is
// irrelevant.  So the device sampler
//   1. assigns sample r to slice perm(r) / mb, where perm is a keyed bijection of [0, B) evaluated on the fly: a 6-round
//      balanced Feistel network on the next even power-of-two domain with cycle walking (re-encrypt until the value
//      falls back into [0, B)) -- a permutation for every key, no sort, no B-sized temporary;
//   2. emits every slice's members in ASCENDING memory order (two-level counting: per-block counts, scan, ordered
//      scatter), so the gathers and the fused trunk kernels walk the buffer monotonically.
// Slices are disjoint, have exactly mb members, and together hold the samples with perm(r) < n_mb * mb -- the same
// distribution over partitions as slicing a uniformly random permutation, to the quality of the Feistel mixing.
// Host-RNG mode (--sampler_rng host) keeps the reference's CPU randperm and order bit for bit.
#include <hip/hip_runtime.h>

#include "../../include/mappo_hip.h"
#include "mappo_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPer = 8;                     // consecutive samples per thread
constexpr int kBlockRows = kThreads * kPer; // 2048 samples per workgroup
constexpr int kMaxMB = MAPPO_PERM_MAX_MINIBATCHES;

struct Perm {
    unsigned long long n;
    unsigned half, mask;
    unsigned key[6];
};

__device__ __forceinline__ unsigned mix(unsigned r, unsigned k) {
    unsigned h = r * 0x9E3779B1u + k;
    h ^= h >> 15;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    return h;
}

// keyed bijection of [0, n)
__device__ __forceinline__ unsigned long long permute(const Perm& p, unsigned long long x) {
    do {
        unsigned l = (unsigned)(x >> p.half), r = (unsigned)x & p.mask;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const unsigned t = l ^ (mix(r, p.key[i]) & p.mask);
            l = r;
            r = t;
        }
        x = ((unsigned long long)l << p.half) | r;
    } while (x >= p.n);
    return x;
}

struct Args {
    Perm perm;
    long long mb;
    int n_mb;
    long long n_blocks;
    int* counts;        // [n_mb][n_blocks] members of slice m in block b, then their exclusive scan over b
    long long* idx;     // [n_mb * mb]
};

// slice of sample r (n_mb = dropped)
__device__ __forceinline__ int slice_of(const Args& a, long long r) {
    if (r >= (long long)a.perm.n) return a.n_mb;
    const unsigned long long p = permute(a.perm, (unsigned long long)r);
    const long long m = (long long)(p / (unsigned long long)a.mb);
    return m < a.n_mb ? (int)m : a.n_mb;
}

__global__ void __launch_bounds__(kThreads) count_kernel(Args a) {
    __shared__ int cnt[kMaxMB];
    for (int m = threadIdx.x; m < a.n_mb; m += kThreads) cnt[m] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * kBlockRows + (long long)threadIdx.x * kPer;
#pragma unroll
    for (int e = 0; e < kPer; ++e) {
        const int m = slice_of(a, base + e);
        if (m < a.n_mb) atomicAdd(&cnt[m], 1);
    }
    __syncthreads();
    for (int m = threadIdx.x; m < a.n_mb; m += kThreads) a.counts[(long long)m * a.n_blocks + blockIdx.x] = cnt[m];
}

// exclusive scan of counts[m][:] over the blocks, one workgroup per slice
__global__ void __launch_bounds__(kThreads) scan_kernel(Args a) {
    __shared__ int part[kThreads];
    int* c = a.counts + (long long)blockIdx.x * a.n_blocks;
    const long long per = (a.n_blocks + kThreads - 1) / kThreads;
    const long long lo = threadIdx.x * per, hi = lo + per < a.n_blocks ? lo + per : a.n_blocks;
    int s = 0;
    for (long long b = lo; b < hi; ++b) s += c[b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < kThreads; ++t) {
            const int v = part[t];
            part[t] = run;
            run += v;
        }
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (long long b = lo; b < hi; ++b) {
        const int v = c[b];
        c[b] = run;
        run += v;
    }
}

__global__ void __launch_bounds__(kThreads) scatter_kernel(Args a) {
    __shared__ int part[kThreads];
    const long long base = (long long)blockIdx.x * kBlockRows + (long long)threadIdx.x * kPer;
    int sl[kPer];
#pragma unroll
    for (int e = 0; e < kPer; ++e) sl[e] = slice_of(a, base + e);
    for (int m = 0; m < a.n_mb; ++m) {
        int mine = 0;
#pragma unroll
        for (int e = 0; e < kPer; ++e) mine += sl[e] == m;
        // exclusive scan of `mine` over the workgroup's threads (thread order = memory order)
        part[threadIdx.x] = mine;
        __syncthreads();
        for (int d = 1; d < kThreads; d <<= 1) {
            const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        long long pos = (long long)m * a.mb + a.counts[(long long)m * a.n_blocks + blockIdx.x] + part[threadIdx.x] - mine;
#pragma unroll
        for (int e = 0; e < kPer; ++e)
            if (sl[e] == m) a.idx[pos++] = base + e;
        __syncthreads();
    }
}

}  // namespace

extern "C" int64_t mappo_minibatch_workspace_ints(int64_t n, int n_mb) {
    const long long n_blocks = (n + kBlockRows - 1) / kBlockRows;
    return n_blocks * (n_mb > 0 ? n_mb : 1);
}

extern "C" int mappo_minibatch_indices(int64_t n, int64_t mb, int n_mb, const uint32_t* keys, int64_t* idx,
                                       int32_t* workspace, mappo_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!keys || !idx || !workspace) return MAPPO_E_NULL;
    if (n <= 0 || mb <= 0 || n_mb <= 0 || (long long)mb * n_mb > n) return MAPPO_E_SHAPE;
    if (n_mb > kMaxMB) return MAPPO_E_TOO_MANY;
    Args a;
    a.perm.n = (unsigned long long)n;
    unsigned bits = 2;
    while ((1ull << bits) < (unsigned long long)n) bits += 2;      // even number of bits: balanced halves
    if (bits > 62) return MAPPO_E_SHAPE;
    a.perm.half = bits / 2;
    a.perm.mask = (unsigned)((1ull << a.perm.half) - 1ull);
    for (int i = 0; i < 6; ++i) a.perm.key[i] = keys[i];
    a.mb = mb;
    a.n_mb = n_mb;
    a.n_blocks = (n + kBlockRows - 1) / kBlockRows;
    a.counts = workspace;
    a.idx = reinterpret_cast<long long*>(idx);
    hipLaunchKernelGGL(count_kernel, dim3((unsigned)a.n_blocks), dim3(kThreads), 0, stream, a);
    hipLaunchKernelGGL(scan_kernel, dim3((unsigned)n_mb), dim3(kThreads), 0, stream, a);
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)a.n_blocks), dim3(kThreads), 0, stream, a);
    return (int)hipGetLastError();
}
