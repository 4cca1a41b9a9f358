// Diagnostic kernels (not part of the product library): bandwidth ceilings for the GAE scan's
// traffic mix (3 reads + 1 write of equal size) under two access patterns.
//   probe_flat : flat grid-stride float4 streaming  out = a + b * c
//   probe_strip: the GAE strip pattern -- a workgroup owns W=64 columns and walks T rows in tiles
//                of TC rows (256-byte segments at a row stride of C*4 bytes), no recurrence/LDS
#include <hip/hip_runtime.h>

__global__ void __launch_bounds__(256) flat_kernel(const float4* a, const float4* b, const float4* c,
                                                    float4* out, long long n4) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long step = (long long)gridDim.x * 256;
    for (; i < n4; i += step) {
        float4 x = a[i], y = b[i], z = c[i], o;
        o.x = x.x + y.x * z.x; o.y = x.y + y.y * z.y; o.z = x.z + y.z * z.z; o.w = x.w + y.w * z.w;
        out[i] = o;
    }
}

template <int NWAVES, int TC, int W = 64>
__global__ void __launch_bounds__(NWAVES * 64) strip_kernel(const float* a, const float* b, const float* c,
                                                             float* out, int T, long long C) {
    constexpr int NT = NWAVES * 64, V = W / 4, NVEC = TC * V, PER = NVEC / NT;
    const long long col0 = (long long)blockIdx.x * W;
    for (int tb = T - TC; tb > -TC; tb -= TC) {
        float4 x[PER], y[PER], z[PER];
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            int i = threadIdx.x + p * NT, row = i / V, c4 = i % V, t = tb + row;
            long long o = (long long)(t < 0 ? 0 : t) * C + col0 + c4 * 4;
            x[p] = *reinterpret_cast<const float4*>(a + o);
            y[p] = *reinterpret_cast<const float4*>(b + o);
            z[p] = *reinterpret_cast<const float4*>(c + o);
        }
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            int i = threadIdx.x + p * NT, row = i / V, c4 = i % V, t = tb + row;
            if (t >= 0) {
                float4 o;
                o.x = x[p].x + y[p].x * z[p].x; o.y = x[p].y + y[p].y * z[p].y;
                o.z = x[p].z + y[p].z * z[p].z; o.w = x[p].w + y[p].w * z[p].w;
                *reinterpret_cast<float4*>(out + (long long)t * C + col0 + c4 * 4) = o;
            }
        }
    }
}

extern "C" int probe_flat(const float* a, const float* b, const float* c, float* out, long long n,
                          int blocks, void* stream) {
    hipLaunchKernelGGL(flat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)a, (const float4*)b, (const float4*)c, (float4*)out, n / 4);
    return (int)hipGetLastError();
}

extern "C" int probe_strip(const float* a, const float* b, const float* c, float* out, int T, long long C,
                           int variant, void* stream) {
    dim3 grid((unsigned)(C / 64));
    if (variant == 0) hipLaunchKernelGGL((strip_kernel<4, 32>), grid, dim3(256), 0, (hipStream_t)stream, a, b, c, out, T, C);
    else if (variant == 1) hipLaunchKernelGGL((strip_kernel<4, 64>), grid, dim3(256), 0, (hipStream_t)stream, a, b, c, out, T, C);
    else if (variant == 2) hipLaunchKernelGGL((strip_kernel<8, 64>), grid, dim3(512), 0, (hipStream_t)stream, a, b, c, out, T, C);
    else if (variant == 3) hipLaunchKernelGGL((strip_kernel<1, 32>), grid, dim3(64), 0, (hipStream_t)stream, a, b, c, out, T, C);
    else if (variant == 4) hipLaunchKernelGGL((strip_kernel<4, 32, 128>), dim3((unsigned)(C / 128)), dim3(256), 0, (hipStream_t)stream, a, b, c, out, T, C);
    else if (variant == 5) hipLaunchKernelGGL((strip_kernel<8, 16, 256>), dim3((unsigned)(C / 256)), dim3(512), 0, (hipStream_t)stream, a, b, c, out, T, C);
    else if (variant == 6) hipLaunchKernelGGL((strip_kernel<4, 16, 256>), dim3((unsigned)(C / 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, out, T, C);
    else if (variant == 7) hipLaunchKernelGGL((strip_kernel<8, 16, 128>), dim3((unsigned)(C / 128)), dim3(512), 0, (hipStream_t)stream, a, b, c, out, T, C);
    else hipLaunchKernelGGL((strip_kernel<4, 16>), grid, dim3(256), 0, (hipStream_t)stream, a, b, c, out, T, C);
    return (int)hipGetLastError();
}
