// Microbenchmark (tools/probes): cycles per v_mfma_f32_32x32x2_f32 on one SIMD in the patterns the fused trunk kernels use.
//   hipcc --offload-arch=gfx950 -O3 probe_mfma.hip -o probe_mfma && ./probe_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters, int waves_active) {
    const int wave = threadIdx.x >> 6;
    __shared__ float sh[8192];
    if (wave >= 4) {
        // partner wave on the same SIMD: MODE 2/3 = busy with VALU + LDS stores
        if (MODE >= 2 && wave - 4 < waves_active) {
            float v = threadIdx.x;
            for (int i = 0; i < iters * 16; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v = v * 1.0001f + 0.5f;
                sh[(threadIdx.x * 4 + i) & 8191] = v;
            }
            out[blockIdx.x * 512 + threadIdx.x] = v;
        }
        return;
    }
    if (MODE == 3) __builtin_amdgcn_s_setprio(3);
    f32x16 a0, a1;
    for (int v = 0; v < 16; ++v) a0[v] = a1[v] = 0.f;
    float b = threadIdx.x * 0.001f, x = 1.f, mu = 0.25f, rs = 0.5f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float bb = b;
            if (MODE >= 1) {
                bb = (b - mu) * rs;
                b += 1.f;
            }
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, bb, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x + 1.f, bb, a1, 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int v = 0; v < 16; ++v) s += a0[v] + a1[v];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    const char* names[] = {"pure MFMA pairs", "+2 VALU per pair", "+busy partner wave (VALU + LDS stores)",
                           "+busy partner, compute at s_setprio 3"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, cyc, iters, 4);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, cyc, iters, 4);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, cyc, iters, 4);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, cyc, iters, 4);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            double n = (double)iters * 32;
            if (rep == 1)
                printf("%-45s %.1f ticks / MFMA, kernel %.3f ms, %.1f TFLOP/s chip-wide (tick rate %.2f GHz)\n", names[mode],
                       c / n, ms, 256.0 * 4 * n * 4096 / (ms * 1e-3) / 1e12, c / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
