// Microbenchmark (tools/probes): how many INDEPENDENT VALU instructions of the same wave fit under a v_mfma_f32_32x32x2_f32
// (64 cycles in the matrix pipe) for free?  One wave per SIMD, dependent MFMA pairs, K v_fma_f32 per MFMA on registers the
// MFMAs do not touch.
//   hipcc --offload-arch=gfx950 -O3 probe_mfma_valu.hip -o probe_mfma_valu && ./probe_mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int KIND>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int iters) {
    f32x16 a0, a1;
    for (int v = 0; v < 16; ++v) a0[v] = a1[v] = 0.f;
    float x = threadIdx.x * 0.001f, y = 1.f;
    float r[16];
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x + i;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (KIND == 0) r[j % 16] = __builtin_fmaf(r[j % 16], 1.0001f, 0.5f);                // v_fma_f32
                else if (KIND == 1) r[j % 16] = __builtin_amdgcn_exp2f(r[j % 16]) * 0.5f;            // transcendental + mul
            }
            __builtin_amdgcn_sched_barrier(0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int v = 0; v < 16; ++v) s += a0[v] + a1[v];
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int K, int KIND>
void run(float* out, long long* cyc) {
    const int iters = 1000;
    long long c = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<K, KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    }
    printf("%s x %2d per MFMA pair: %.1f ticks per MFMA\n", KIND == 0 ? "v_fma_f32        " : "v_exp_f32 + v_mul", K,
           (double)c / (iters * 32.0));
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 8);
    run<0, 0>(out, cyc);
    run<4, 0>(out, cyc);
    run<8, 0>(out, cyc);
    run<12, 0>(out, cyc);
    run<16, 0>(out, cyc);
    run<24, 0>(out, cyc);
    run<32, 0>(out, cyc);
    run<2, 1>(out, cyc);
    run<4, 1>(out, cyc);
    run<8, 1>(out, cyc);
    return 0;
}
