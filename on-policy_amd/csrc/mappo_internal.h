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

}  // namespace mappo
#endif
