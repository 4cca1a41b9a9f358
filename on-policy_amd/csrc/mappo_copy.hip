// K3 / K4 -- fused multi-field minibatch gathers, and K2 -- fused slab copies, for gfx950.
//
// K3 replaces the 12 fancy-index gathers per minibatch of
//   SharedReplayBuffer.feed_forward_generator (reference onpolicy/utils/shared_buffer.py:379-396);
// K4 replaces recurrent_generator's transposing copies + Python chunk loop + np.stack
//   (:514-604) and naive_recurrent_generator (:417-493) by reading L-row chunks straight out
//   of the time-major buffer;
// K2 replaces the per-field slab assignments of insert / after_update (:107-121,:162-170).
//
// All of it is pure HBM traffic (no arithmetic except the optional advantage
// normalisation), so the design goals are: one launch for all fields, 16-byte accesses
// wherever the row width allows, output writes fully coalesced, many independent loads in
// flight per lane, and a persistent grid (a few workgroups per CU walking a tile list)
// instead of one tiny workgroup per row.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mappo_hip.h"
#include "mappo_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxUnroll = 8;                    // independent loads per lane per pass (slab copy)
constexpr unsigned kTileUnits = kThreads * kMaxUnroll;

struct GField {
    const float* src;
    float* dst;
    unsigned width;          // floats per row
    unsigned vec;            // floats per unit (1, 2, 4)
    unsigned upr;            // units per row
    unsigned rows_per_tile;
    unsigned inv_upr;        // ceil(2^32 / upr): i / upr == mulhi(i, inv_upr) for i * upr < 2^32
    unsigned rows_out;       // output rows of this field
    unsigned tile_begin;     // first global tile id of this field
    unsigned first_only;
    unsigned normalize;
};

struct GatherArgs {
    GField f[MAPPO_MAX_FIELDS];
    int nf;
    unsigned total_tiles;
    mappo::RowMap map;
    const float* stats;
};

// native clang vectors (the non-temporal builtins do not take HIP's float4 struct)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int VEC> struct VecT;
template <> struct VecT<4> { using type = f32x4; };
template <> struct VecT<2> { using type = f32x2; };
template <> struct VecT<1> { using type = float; };

__device__ __forceinline__ float norm1(float x, float mean, float den) { return (x - mean) / den; }
__device__ __forceinline__ void normv(float& v, float m, float d) { v = norm1(v, m, d); }
__device__ __forceinline__ void normv(f32x2& v, float m, float d) {
    v.x = norm1(v.x, m, d);
    v.y = norm1(v.y, m, d);
}
__device__ __forceinline__ void normv(f32x4& v, float m, float d) {
    v.x = norm1(v.x, m, d);
    v.y = norm1(v.y, m, d);
    v.z = norm1(v.z, m, d);
    v.w = norm1(v.w, m, d);
}

__device__ __forceinline__ unsigned source_row(const GatherArgs& a, const GField& fd, unsigned r) {
    return mappo::source_row(a.map, fd.first_only, r);
}

template <bool NT, typename V>
__device__ __forceinline__ V ld(const V* p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT, typename V>
__device__ __forceinline__ void st(V* p, V v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <int VEC, int UNROLL, bool NT>
__device__ __forceinline__ void copy_tile(const GatherArgs& a, const GField& fd, unsigned tile_local) {
    using V = typename VecT<VEC>::type;
    constexpr unsigned kPass = kThreads * UNROLL;
    const unsigned row0 = tile_local * fd.rows_per_tile;
    unsigned nrows = fd.rows_out - row0;
    if (nrows > fd.rows_per_tile) nrows = fd.rows_per_tile;
    const unsigned nunits = nrows * fd.upr;
    float mean = 0.f, den = 1.f;
    if (fd.normalize) {
        mean = a.stats[0];
        den = a.stats[1] + 1e-5f;  // r_mappo.py:187
    }
    for (unsigned i0 = 0; i0 < nunits; i0 += kPass) {
        V val[UNROLL];
        long long doff[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            unsigned i = i0 + u * kThreads + threadIdx.x;
            doff[u] = -1;
            val[u] = V(0.f);
            if (i < nunits) {
                unsigned r = (fd.upr == 1) ? i : __umulhi(i, fd.inv_upr);
                unsigned w = i - r * fd.upr;
                unsigned orow = row0 + r;
                unsigned srow = source_row(a, fd, orow);
                val[u] = ld<NT>(reinterpret_cast<const V*>(fd.src + (long long)srow * fd.width + w * VEC));
                doff[u] = (long long)orow * fd.width + w * VEC;
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (doff[u] >= 0) {
                V v = val[u];
                if (fd.normalize) normv(v, mean, den);
                st<NT>(reinterpret_cast<V*>(fd.dst + doff[u]), v);
            }
        }
    }
}

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(kThreads) gather_kernel(GatherArgs a) {
    for (unsigned tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        int f = 0;
#pragma unroll 1
        for (int k = 1; k < a.nf; ++k)
            if (tile >= a.f[k].tile_begin) f = k;
        const GField& fd = a.f[f];
        const unsigned tl = tile - fd.tile_begin;
        if (fd.vec == 4) copy_tile<4, UNROLL, NT>(a, fd, tl);
        else if (fd.vec == 2) copy_tile<2, UNROLL, NT>(a, fd, tl);
        else copy_tile<1, UNROLL, NT>(a, fd, tl);
    }
}

int g_gather_variant = 5;  // default: 2 loads in flight per lane, non-temporal.  bits 0-1: log2(unroll), bit 2: nontemporal, bits 4+: blocks per CU (0 = occupancy)

template <int UNROLL, bool NT>
hipError_t launch_gather(const GatherArgs& a, unsigned tile_units, int blocks_per_cu, hipStream_t stream) {
    static int occ = 0;  // resident workgroups per CU for this instantiation
    if (occ == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gather_kernel<UNROLL, NT>, kThreads, 0) != hipSuccess || n <= 0)
            n = 4;
        occ = n;
    }
    // a persistent grid: exactly the resident workgroups (a second, partial wave of workgroups
    // would run at reduced occupancy), each walking the tile list with stride gridDim.x
    int per_cu = blocks_per_cu > 0 ? blocks_per_cu : occ;
    unsigned grid = (unsigned)(mappo::kCUs * per_cu);
    if (a.total_tiles < grid) grid = a.total_tiles;
    hipLaunchKernelGGL((gather_kernel<UNROLL, NT>), dim3(grid), dim3(kThreads), 0, stream, a);
    return hipGetLastError();
}

int build_and_launch(const mappo_field_t* fields, int n_fields, const int64_t* idx, int64_t mb,
                     int chunked, int L, int T, int64_t N, int A, const float* stats,
                     hipStream_t stream) {
    if (!fields || !idx) return MAPPO_E_NULL;
    if (n_fields <= 0 || mb <= 0) return MAPPO_E_SHAPE;
    if (n_fields > MAPPO_MAX_FIELDS) return MAPPO_E_TOO_MANY;
    if (mb >= (1ll << 31)) return MAPPO_E_SHAPE;
    if (chunked) {
        if (L <= 0 || T <= 0 || N <= 0 || A <= 0) return MAPPO_E_SHAPE;
        if ((long long)T * N * A >= (1ll << 31) || mb * (long long)L >= (1ll << 31)) return MAPPO_E_SHAPE;
    }
    const int variant = g_gather_variant;
    const int unroll = 1 << (variant & 3);
    const bool nt = (variant & 4) != 0;
    const int blocks_per_cu = variant >> 4;
    const unsigned tile_units = (unsigned)(kThreads * unroll);
    GatherArgs a;
    a.map.idx = reinterpret_cast<const long long*>(idx);
    a.map.mb = (unsigned)mb;
    a.map.chunked = chunked;
    a.map.L = (unsigned)L;
    a.map.T = (unsigned)T;
    a.map.N = (unsigned)N;
    a.map.A = (unsigned)A;
    a.stats = stats;
    unsigned long long tiles = 0;
    int nf = 0;
    for (int k = 0; k < n_fields; ++k) {
        const mappo_field_t& s = fields[k];
        if (!s.src || !s.dst) return MAPPO_E_NULL;
        if (s.width <= 0) return MAPPO_E_SHAPE;
        if (!mappo::aligned_to(s.src, 4) || !mappo::aligned_to(s.dst, 4)) return MAPPO_E_ALIGN;
        if (s.normalize && !stats) return MAPPO_E_NULL;
        if (s.standardize) {
            // row-standardised fields go through the row-owning kernel (it needs whole-row statistics)
            const unsigned fo = (chunked && s.first_only) ? 1u : 0u;
            const long long rows = (chunked && !fo) ? mb * (long long)L : mb;
            int rc = mappo::gather_standardize(s.src, s.dst, s.width, rows, a.map, fo, 1e-5f, stream);
            if (rc != 0) return rc;
            continue;
        }
        GField& g = a.f[nf++];
        g.src = s.src;
        g.dst = s.dst;
        g.width = (unsigned)s.width;
        g.vec = (unsigned)mappo::row_vec(s.src, s.dst, s.width);
        g.upr = g.width / g.vec;
        g.rows_per_tile = g.upr >= tile_units ? 1u : tile_units / g.upr;
        g.inv_upr = (unsigned)(((1ull << 32) + g.upr - 1) / g.upr);
        g.first_only = (chunked && s.first_only) ? 1u : 0u;
        g.normalize = s.normalize ? 1u : 0u;
        long long rows_out = (chunked && !g.first_only) ? mb * (long long)L : mb;
        g.rows_out = (unsigned)rows_out;
        g.tile_begin = (unsigned)tiles;
        tiles += (unsigned long long)((rows_out + g.rows_per_tile - 1) / g.rows_per_tile);
        if (tiles >= (1ull << 32)) return MAPPO_E_SHAPE;
    }
    a.nf = nf;
    a.total_tiles = (unsigned)tiles;
    if (nf == 0) return 0;
    hipError_t e;
    switch ((variant & 3) | (nt ? 4 : 0)) {
        case 0: e = launch_gather<1, false>(a, tile_units, blocks_per_cu, stream); break;
        case 1: e = launch_gather<2, false>(a, tile_units, blocks_per_cu, stream); break;
        case 2: e = launch_gather<4, false>(a, tile_units, blocks_per_cu, stream); break;
        case 3: e = launch_gather<8, false>(a, tile_units, blocks_per_cu, stream); break;
        case 4: e = launch_gather<1, true>(a, tile_units, blocks_per_cu, stream); break;
        case 5: e = launch_gather<2, true>(a, tile_units, blocks_per_cu, stream); break;
        case 6: e = launch_gather<4, true>(a, tile_units, blocks_per_cu, stream); break;
        default: e = launch_gather<8, true>(a, tile_units, blocks_per_cu, stream); break;
    }
    return (int)e;
}

// ------------------------------------------------ packed records of the narrow fields ----
// A 4-byte gather costs a whole 64-byte HBM sector, so the seven per-sample scalars (action,
// value_pred, return, mask, active_mask, log-prob, advantage) and the small available-actions
// row would cost 8 sectors per sample when gathered field by field.  Once per epoch they are
// packed into one record per (t, n, a) row ([rows, RW] floats, RW a multiple of 4); the sampler
// then reads ONE record per sample and scatters its components to the per-field outputs.
struct RecField {
    const float* src;   // pack: field base, rows of `width` floats
    float* dst;         // unpack: output base, rows of `width` floats
    unsigned width;
    unsigned offset;    // first component inside the record
    unsigned normalize;
};
struct RecArgs {
    RecField f[MAPPO_MAX_FIELDS];
    int nf;
    unsigned rw;        // record width in floats
    float* records;
    unsigned long long rows;
    mappo::RowMap map;
    const float* stats;
};

constexpr int kRecMaxWidth = 32;

__global__ void __launch_bounds__(kThreads) pack_records_kernel(RecArgs a) {
    __shared__ float tile[kThreads * kRecMaxWidth];
    const unsigned rw = a.rw;
    for (unsigned long long row0 = (unsigned long long)blockIdx.x * kThreads; row0 < a.rows;
         row0 += (unsigned long long)gridDim.x * kThreads) {
        const unsigned long long row = row0 + threadIdx.x;
        const bool ok = row < a.rows;
        for (unsigned c = threadIdx.x; c < kThreads * rw; c += kThreads) tile[c] = 0.f;
        __syncthreads();
        for (int k = 0; k < a.nf; ++k) {
            const float* src = a.f[k].src;
            const unsigned w = a.f[k].width, off = a.f[k].offset;
            // coalesced read of the [256, w] block of this field, scattered into the LDS records
            for (unsigned e = threadIdx.x; e < kThreads * w; e += kThreads) {
                unsigned r = e / w, c = e - r * w;
                if (row0 + r < a.rows) tile[r * rw + off + c] = src[(row0 + r) * w + c];
            }
        }
        __syncthreads();
        const unsigned long long base = row0 * rw;
        const unsigned long long limit = a.rows * rw;
        for (unsigned e = threadIdx.x; e < kThreads * rw; e += kThreads)
            if (base + e < limit) a.records[base + e] = tile[e];
        __syncthreads();
        (void)ok;
    }
}

// Records are read by q4 = rw / 4 adjacent lanes, one 16-byte piece each, so that one wave instruction fetches
// whole records (a lane walking its record with q4 successive loads touched the record's cache line q4 times:
// 4.5 GB of fetches for 0.63 GB of records at the north star, PMC FETCH_SIZE).  Rounds of kThreads / q4 records
// are issued back to back until the LDS tile is full, so every lane has several independent loads in flight.
__global__ void __launch_bounds__(kThreads) gather_records_kernel(RecArgs a, unsigned long long rows_out) {
    __shared__ float tile[kThreads * kRecMaxWidth];
    const unsigned rw = a.rw, q4 = rw / 4;
    const unsigned per_round = kThreads / q4;                     // records fetched by one round of loads
    const unsigned rounds = (kThreads * kRecMaxWidth / rw) / per_round;
    const unsigned pass_rows = per_round * rounds;                // rows per pass: what the LDS tile holds
    const unsigned sub = threadIdx.x / q4, piece = threadIdx.x - sub * q4;
    float mean = 0.f, den = 1.f;
    if (a.stats) {
        mean = a.stats[0];
        den = a.stats[1] + 1e-5f;
    }
    for (unsigned long long row0 = (unsigned long long)blockIdx.x * pass_rows; row0 < rows_out;
         row0 += (unsigned long long)gridDim.x * pass_rows) {
        if (sub < per_round) {
            for (unsigned r = 0; r < rounds; ++r) {
                const unsigned lr = r * per_round + sub;            // local row inside the pass
                const unsigned long long j = row0 + lr;
                if (j < rows_out) {
                    const unsigned srow = mappo::source_row(a.map, 0u, (unsigned)j);
                    const f32x4* rec = reinterpret_cast<const f32x4*>(a.records + (unsigned long long)srow * rw);
                    reinterpret_cast<f32x4*>(tile + lr * rw)[piece] = __builtin_nontemporal_load(rec + piece);
                }
            }
        }
        __syncthreads();
        const unsigned nrows = rows_out - row0 < (unsigned long long)pass_rows ? (unsigned)(rows_out - row0) : pass_rows;
        for (int k = 0; k < a.nf; ++k) {
            float* dst = a.f[k].dst;
            const unsigned w = a.f[k].width, off = a.f[k].offset;
            const bool norm = a.f[k].normalize != 0;
            // coalesced write of the [nrows, w] output block
            for (unsigned e = threadIdx.x; e < nrows * w; e += kThreads) {
                unsigned r = e / w, c = e - r * w;
                float v = tile[r * rw + off + c];
                if (norm) v = (v - mean) / den;
                dst[row0 * w + e] = v;
            }
        }
        __syncthreads();
    }
}

int fill_rec_args(RecArgs& a, const mappo_record_field_t* fields, int n_fields, int record_width, bool packing) {
    if (!fields) return MAPPO_E_NULL;
    if (n_fields <= 0 || record_width <= 0 || record_width % 4 != 0 || record_width > kRecMaxWidth) return MAPPO_E_SHAPE;
    if (n_fields > MAPPO_MAX_FIELDS) return MAPPO_E_TOO_MANY;
    a.nf = n_fields;
    a.rw = (unsigned)record_width;
    for (int k = 0; k < n_fields; ++k) {
        const mappo_record_field_t& s = fields[k];
        if (packing ? !s.src : !s.dst) return MAPPO_E_NULL;
        if (s.width <= 0 || s.offset < 0 || s.offset + s.width > record_width) return MAPPO_E_SHAPE;
        a.f[k].src = s.src;
        a.f[k].dst = s.dst;
        a.f[k].width = (unsigned)s.width;
        a.f[k].offset = (unsigned)s.offset;
        a.f[k].normalize = s.normalize ? 1u : 0u;
    }
    return 0;
}

// ------------------------------------------------------------------ K2: slabs ----
struct Slab {
    const float* src;
    float* dst;
    unsigned long long units;  // float4 (vec) or float units
    unsigned vec;
    unsigned tile_begin;
};
struct SlabArgs {
    Slab s[MAPPO_MAX_FIELDS];
    int n;
    unsigned total_tiles;
};

__global__ void __launch_bounds__(kThreads) slab_kernel(SlabArgs a) {
    for (unsigned tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        int k = 0;
#pragma unroll 1
        for (int q = 1; q < a.n; ++q)
            if (tile >= a.s[q].tile_begin) k = q;
        // copy the descriptor into registers (indexing the by-value argument with a runtime k
        // otherwise sends it through scratch memory)
        const float* s_src = a.s[k].src;
        float* s_dst = a.s[k].dst;
        const unsigned long long units = a.s[k].units;
        const unsigned vec = a.s[k].vec;
        unsigned long long base = (unsigned long long)(tile - a.s[k].tile_begin) * kTileUnits;
        if (vec == 4) {
            const float4* src = reinterpret_cast<const float4*>(s_src);
            float4* dst = reinterpret_cast<float4*>(s_dst);
            float4 v[kMaxUnroll];
#pragma unroll
            for (int u = 0; u < kMaxUnroll; ++u) {
                unsigned long long i = base + u * kThreads + threadIdx.x;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < units) v[u] = src[i];
            }
#pragma unroll
            for (int u = 0; u < kMaxUnroll; ++u) {
                unsigned long long i = base + u * kThreads + threadIdx.x;
                if (i < units) dst[i] = v[u];
            }
        } else {
            const float* src = s_src;
            float* dst = s_dst;
            float v[kMaxUnroll];
#pragma unroll
            for (int u = 0; u < kMaxUnroll; ++u) {
                unsigned long long i = base + u * kThreads + threadIdx.x;
                v[u] = 0.f;
                if (i < units) v[u] = src[i];
            }
#pragma unroll
            for (int u = 0; u < kMaxUnroll; ++u) {
                unsigned long long i = base + u * kThreads + threadIdx.x;
                if (i < units) dst[i] = v[u];
            }
        }
    }
}

}  // namespace

extern "C" int mappo_gather_rows(const mappo_field_t* fields, int n_fields, const int64_t* idx,
                                 int64_t mb, const float* stats, mappo_stream_t stream) {
    return build_and_launch(fields, n_fields, idx, mb, 0, 1, 1, 1, 1, stats,
                            static_cast<hipStream_t>(stream));
}

extern "C" int mappo_gather_chunks(const mappo_field_t* fields, int n_fields, const int64_t* idx,
                                   int64_t mb, int L, int T, int64_t N, int A, const float* stats,
                                   mappo_stream_t stream) {
    return build_and_launch(fields, n_fields, idx, mb, 1, L, T, N, A, stats,
                            static_cast<hipStream_t>(stream));
}

// ---- K2 with the standardised copies of the observation slabs kept current in the SAME launch (round 6) ----
// insert / chooseinsert / after_update write one slab per field; networks with an input LayerNorm read the observation fields
// through a resident row-standardised copy (utils/shared_buffer.py: _obs_rows), whose slab has to follow.  As launches of their
// own the two standardisations were two more launches per rollout step on a path that is launch-bound (config 3 end to end:
// 0.140 -> 0.154 ms per env step).  Here the workgroups behind the copy tiles standardise the rows of the incoming VALUE (not of
// the buffer slab: no dependence on the copy tiles of the same launch): 16 lanes per row, the row read three times from cache
// (sum, centred second moment, output).  The FULL pass over a field (first train(), or after an in-place torch write) goes
// through the same code (n_slabs = 0), so a slab comes out exactly as the full pass would write it
// (tests/test_gpu_standardize_at_insert.py) -- by construction, not by keeping two kernels in step: the instances of
// mlp::standardize_rows_kernel (mappo_mlp_impl.h, built with fp contract(fast): the rollout's network inputs) round their
// second moment in whatever mixture of fused and unfused operations the compiler picked per instance.  Here every operation
// is written out: IEEE add / sub / mul / div / sqrt, and one explicit fused multiply-add per term of the second moment.
struct StdSlab {
    const float* src;   // [rows, D] the value being inserted
    float* dst;         // [rows, ld] the slab of the standardised copy
    long long rows;
    int D, ld;
    float eps;
    unsigned block_begin, blocks;
};
struct StdArgs {
    StdSlab s[MAPPO_MAX_STD_SLABS];
    int n;
    unsigned copy_blocks;
};

__device__ __forceinline__ float sum16_xor(float v) {      // = prim::sum16 (mappo_mlp.hip)
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}
// 4 consecutive floats of a row from column k on, zero beyond the row's end (mlp::load4_guard)
__device__ __forceinline__ void load4_guard(const float* p, int remaining, float* x) {
    x[0] = remaining > 0 ? p[0] : 0.f;
    x[1] = remaining > 1 ? p[1] : 0.f;
    x[2] = remaining > 2 ? p[2] : 0.f;
    x[3] = remaining > 3 ? p[3] : 0.f;
}

__device__ void standardize_slab(const StdSlab& d, unsigned block, unsigned nblocks) {
    const int sub = threadIdx.x & 15;
    const long long groups = ((long long)nblocks * kThreads) >> 4;
    const long long first = ((long long)block * kThreads + threadIdx.x) >> 4;
    const long long trips = (d.rows + groups - 1) / groups;     // the same for every lane: sum16 is a wave collective
    const int D = d.D;
    for (long long it = 0; it < trips; ++it) {
        const long long r = first + it * groups;
        const bool ok = r < d.rows;
        const float* p = d.src + (ok ? r : d.rows - 1) * D;
        float s = 0.f;
        for (int k = 4 * sub; k < D; k += 64) {
            float x[4];
            load4_guard(p + k, D - k, x);
            s += (x[0] + x[1]) + (x[2] + x[3]);
        }
        s = sum16_xor(s);
        const float mean = s / (float)D;
        float q = 0.f;
        for (int k = 4 * sub; k < D; k += 64) {
            float x[4];
            load4_guard(p + k, D - k, x);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k + e < D) {
                    const float c = x[e] - mean;
                    q = __builtin_fmaf(c, c, q);
                }
        }
        q = sum16_xor(q);
        const float rstd = 1.f / sqrtf(q / (float)D + d.eps);
        if (ok) {
            float* o = d.dst + r * d.ld;
            if (sub == 0)
                for (int k = D; k < d.ld; ++k) o[k] = 0.f;
            for (int k = 4 * sub; k < D; k += 64) {
                float x[4];
                load4_guard(p + k, D - k, x);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e < D) o[k + e] = (x[e] - mean) * rstd;
            }
        }
    }
}

__device__ __forceinline__ void copy_tiles(const SlabArgs& a, unsigned block, unsigned nblocks) {
    for (unsigned tile = block; tile < a.total_tiles; tile += nblocks) {
        int k = 0;
#pragma unroll 1
        for (int q = 1; q < a.n; ++q)
            if (tile >= a.s[q].tile_begin) k = q;
        const float* s_src = a.s[k].src;
        float* s_dst = a.s[k].dst;
        const unsigned long long units = a.s[k].units;
        const unsigned vec = a.s[k].vec;
        unsigned long long base = (unsigned long long)(tile - a.s[k].tile_begin) * kTileUnits;
        if (vec == 4) {
            const float4* src = reinterpret_cast<const float4*>(s_src);
            float4* dst = reinterpret_cast<float4*>(s_dst);
            float4 v[kMaxUnroll];
#pragma unroll
            for (int u = 0; u < kMaxUnroll; ++u) {
                unsigned long long i = base + u * kThreads + threadIdx.x;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < units) v[u] = src[i];
            }
#pragma unroll
            for (int u = 0; u < kMaxUnroll; ++u) {
                unsigned long long i = base + u * kThreads + threadIdx.x;
                if (i < units) dst[i] = v[u];
            }
        } else {
            float v[kMaxUnroll];
#pragma unroll
            for (int u = 0; u < kMaxUnroll; ++u) {
                unsigned long long i = base + u * kThreads + threadIdx.x;
                v[u] = 0.f;
                if (i < units) v[u] = s_src[i];
            }
#pragma unroll
            for (int u = 0; u < kMaxUnroll; ++u) {
                unsigned long long i = base + u * kThreads + threadIdx.x;
                if (i < units) s_dst[i] = v[u];
            }
        }
    }
}

__global__ void __launch_bounds__(kThreads) slab_std_kernel(SlabArgs a, StdArgs st) {
    if (blockIdx.x < st.copy_blocks) {
        copy_tiles(a, blockIdx.x, st.copy_blocks);
        return;
    }
    const unsigned b = blockIdx.x - st.copy_blocks;
    int k = 0;
#pragma unroll 1
    for (int q = 1; q < st.n; ++q)
        if (b >= st.s[q].block_begin) k = q;
    // (the descriptor into registers: indexing the by-value argument with a runtime k otherwise goes through scratch memory)
    StdSlab d;
    d.src = st.s[k].src;
    d.dst = st.s[k].dst;
    d.rows = st.s[k].rows;
    d.D = st.s[k].D;
    d.ld = st.s[k].ld;
    d.eps = st.s[k].eps;
    standardize_slab(d, b - st.s[k].block_begin, st.s[k].blocks);
}

namespace {
int fill_slabs(const mappo_slab_t* slabs, int n_slabs, SlabArgs& a) {
    if (!slabs) return MAPPO_E_NULL;
    if (n_slabs <= 0) return MAPPO_E_SHAPE;
    if (n_slabs > MAPPO_MAX_FIELDS) return MAPPO_E_TOO_MANY;
    a.n = n_slabs;
    unsigned long long tiles = 0;
    for (int k = 0; k < n_slabs; ++k) {
        const mappo_slab_t& s = slabs[k];
        if (!s.src || !s.dst) return MAPPO_E_NULL;
        if (s.count <= 0) return MAPPO_E_SHAPE;
        if (!mappo::aligned_to(s.src, 4) || !mappo::aligned_to(s.dst, 4)) return MAPPO_E_ALIGN;
        Slab& d = a.s[k];
        d.src = s.src;
        d.dst = s.dst;
        d.vec = (unsigned)(mappo::row_vec(s.src, s.dst, s.count) == 4 ? 4 : 1);
        d.units = (unsigned long long)s.count / d.vec;
        d.tile_begin = (unsigned)tiles;
        tiles += (d.units + kTileUnits - 1) / kTileUnits;
        if (tiles >= (1ull << 32)) return MAPPO_E_SHAPE;
    }
    a.total_tiles = (unsigned)tiles;
    return 0;
}
}  // namespace

extern "C" int mappo_slab_copy(const mappo_slab_t* slabs, int n_slabs, mappo_stream_t stream) {
    SlabArgs a;
    int code = fill_slabs(slabs, n_slabs, a);
    if (code) return code;
    unsigned grid = a.total_tiles < (unsigned)(mappo::kCUs * 8) ? a.total_tiles
                                                                  : (unsigned)(mappo::kCUs * 8);
    hipLaunchKernelGGL(slab_kernel, dim3(grid), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int mappo_slab_copy_std(const mappo_slab_t* slabs, int n_slabs, const mappo_std_slab_t* std, int n_std,
                                   mappo_stream_t stream) {
    if (n_std == 0) return mappo_slab_copy(slabs, n_slabs, stream);
    if (!std) return MAPPO_E_NULL;
    if (n_std < 0 || n_std > MAPPO_MAX_STD_SLABS) return MAPPO_E_TOO_MANY;
    SlabArgs a;
    a.n = 0;
    a.total_tiles = 0;
    if (n_slabs != 0) {         // (n_slabs = 0: only the standardisation -- the full pass over a field)
        int code = fill_slabs(slabs, n_slabs, a);
        if (code) return code;
    }
    StdArgs st;
    st.n = n_std;
    st.copy_blocks = a.total_tiles < (unsigned)(mappo::kCUs * 8) ? a.total_tiles : (unsigned)(mappo::kCUs * 8);
    unsigned blocks = 0;
    for (int k = 0; k < n_std; ++k) {
        const mappo_std_slab_t& s = std[k];
        if (!s.src || !s.dst) return MAPPO_E_NULL;
        if (s.rows <= 0 || s.D <= 0 || s.ld < s.D) return MAPPO_E_SHAPE;
        if (!mappo::aligned_to(s.src, 4) || !mappo::aligned_to(s.dst, 4)) return MAPPO_E_ALIGN;
        StdSlab& d = st.s[k];
        d.src = s.src;
        d.dst = s.dst;
        d.rows = s.rows;
        d.D = s.D;
        d.ld = s.ld;
        d.eps = s.eps;
        long long nb = (s.rows * 16 + kThreads - 1) / kThreads;
        if (nb > mappo::kCUs * 4) nb = mappo::kCUs * 4;
        d.block_begin = blocks;
        d.blocks = (unsigned)nb;
        blocks += (unsigned)nb;
    }
    hipLaunchKernelGGL(slab_std_kernel, dim3(st.copy_blocks + blocks), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), a, st);
    return (int)hipGetLastError();
}

extern "C" int mappo_pack_records(const mappo_record_field_t* fields, int n_fields, float* records,
                                  int record_width, int64_t rows, mappo_stream_t stream) {
    if (!records) return MAPPO_E_NULL;
    if (rows <= 0) return MAPPO_E_SHAPE;
    if (!mappo::aligned_to(records, 16)) return MAPPO_E_ALIGN;
    RecArgs a;
    int rc = fill_rec_args(a, fields, n_fields, record_width, true);
    if (rc != 0) return rc;
    a.records = records;
    a.rows = (unsigned long long)rows;
    a.stats = nullptr;
    long long blocks = (rows + kThreads - 1) / kThreads;
    if (blocks > mappo::kCUs * 8) blocks = mappo::kCUs * 8;
    hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)blocks), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int mappo_gather_records(const float* records, int record_width, const mappo_record_field_t* fields,
                                    int n_fields, const int64_t* idx, int64_t mb, int L, int T, int64_t N,
                                    int A, const float* stats, mappo_stream_t stream) {
    if (!records || !idx) return MAPPO_E_NULL;
    if (mb <= 0 || mb >= (1ll << 31)) return MAPPO_E_SHAPE;
    if (!mappo::aligned_to(records, 16)) return MAPPO_E_ALIGN;
    const int chunked = L > 0 ? 1 : 0;
    if (chunked && (T <= 0 || N <= 0 || A <= 0 || (long long)T * N * A >= (1ll << 31) ||
                    mb * (long long)L >= (1ll << 31)))
        return MAPPO_E_SHAPE;
    RecArgs a;
    int rc = fill_rec_args(a, fields, n_fields, record_width, false);
    if (rc != 0) return rc;
    bool any_norm = false;
    for (int k = 0; k < n_fields; ++k) any_norm = any_norm || fields[k].normalize;
    if (any_norm && !stats) return MAPPO_E_NULL;
    a.records = const_cast<float*>(records);
    a.rows = 0;
    a.stats = any_norm ? stats : nullptr;
    a.map.idx = reinterpret_cast<const long long*>(idx);
    a.map.mb = (unsigned)mb;
    a.map.chunked = chunked;
    a.map.L = (unsigned)(chunked ? L : 1);
    a.map.T = (unsigned)T;
    a.map.N = (unsigned)N;
    a.map.A = (unsigned)A;
    const unsigned long long rows_out = chunked ? (unsigned long long)mb * L : (unsigned long long)mb;
    long long blocks = (long long)((rows_out + kThreads - 1) / kThreads);
    if (blocks > mappo::kCUs * 5) blocks = mappo::kCUs * 5;     // 32 KB of LDS per workgroup: 5 per CU
    hipLaunchKernelGGL(gather_records_kernel, dim3((unsigned)blocks), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), a, rows_out);
    return (int)hipGetLastError();
}

extern "C" int mappo_gather_set_variant(int variant) {
    int old = g_gather_variant;
    g_gather_variant = variant;
    return old;
}

extern "C" int mappo_abi_version(void) { return MAPPO_ABI_VERSION; }

extern "C" const char* mappo_build_info(void) {
    return "libmappo_hip gfx950 (CDNA4) fp-contract=off (fused trunk: fast) " __DATE__;
}

extern "C" const char* mappo_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case MAPPO_E_NULL: return "a required pointer is NULL";
        case MAPPO_E_SHAPE: return "a size is <= 0, inconsistent or too large";
        case MAPPO_E_FLAGS: return "unsupported flag combination";
        case MAPPO_E_TOO_MANY: return "too many fields / slabs (MAPPO_MAX_FIELDS)";
        case MAPPO_E_ALIGN: return "a pointer is not 4-byte aligned";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown error";
    }
}
