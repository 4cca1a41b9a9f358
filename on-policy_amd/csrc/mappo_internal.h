// Shared helpers of libmappo_hip.so (not part of the ABI).
#ifndef MAPPO_INTERNAL_H
#define MAPPO_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mappo {

constexpr int kCUs = 256;  // MI355X

inline bool aligned_to(const void* p, uintptr_t a) {
    return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0;
}

// Largest vector width (in floats: 4, 2 or 1) usable for rows of `width` floats that
// start at `src` / `dst` (row r begins at base + r*width floats).
inline int row_vec(const void* src, const void* dst, long long width) {
    if (width % 4 == 0 && aligned_to(src, 16) && aligned_to(dst, 16)) return 4;
    if (width % 2 == 0 && aligned_to(src, 8) && aligned_to(dst, 8)) return 2;
    return 1;
}

// Source-row mapping of the samplers (shared by the tile gather and the standardising gather).
struct RowMap {
    const long long* idx;
    unsigned mb;
    int chunked;      // 0: rows mode (shared_buffer.py:379-396), 1: chunk mode (:554-604)
    unsigned L, T, N, A;
};

// Source row (in the time-major [T, N, A] row space) of output row `r`.
__device__ __forceinline__ unsigned source_row(const RowMap& m, unsigned first_only, unsigned r) {
    if (!m.chunked) return (unsigned)m.idx[r];
    unsigned l = 0, j = r;
    if (!first_only) {
        l = r / m.mb;
        j = r - l * m.mb;
    }
    unsigned f = (unsigned)m.idx[j] * m.L + l;
    unsigned at = m.A * m.T;
    unsigned n = f / at;
    unsigned rem = f - n * at;
    unsigned ag = rem / m.T;
    unsigned t = rem - ag * m.T;
    return (t * m.N + n) * m.A + ag;
}

// dst[r, :] = standardise(src[source_row(r), :]) = (x - mean) / sqrt(var + eps), row statistics over
// `width` elements (mappo_norm.hip).  Returns a MAPPO_E_* / hipError_t code.
int gather_standardize(const float* src, float* dst, int width, long long rows_out, const RowMap& map,
                       unsigned first_only, float eps, hipStream_t stream);

}  // namespace mappo
#endif
