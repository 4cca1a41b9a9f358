// Probe (tools/probes): is a float32 product on the bf16 matrix pipe an option for the fused trunk (K9)?
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16 x the FLOP rate of v_mfma_f32_32x32x2_f32.  A float32 value splits exactly
// into three bf16 terms (x = x1 + x2 + x3, 8 + 8 + 8 mantissa bits), so a float32 product is the sum of bf16 x bf16
// products accumulated in float32 by the matrix core:
//   3 terms (x1 y1 + x1 y2 + x2 y1): relative error ~2^-16 per product     -- NOT float32 arithmetic
//   6 terms (+ x2 y2 + x1 y3 + x3 y1): what is dropped is ~2^-24 per product -- the size of float32's own rounding
//   9 terms: everything
// Part A measures the error of each against a float64 sum on data shaped like K9's operands, next to the error of the
// float32 MFMA the kernels use today.  Part B measures what a k = 16 step costs a SIMD: 16 float32 MFMAs (today), 12 bf16
// MFMAs (6 terms x 2 feature tiles) with operands that are already split, and the same with the 3-way split of the
// activation operand done by the wave itself (VALU work that, per the round-4 counters, adds to the MFMA time).
//   hipcc --offload-arch=gfx950 -O3 probe_bf16_split.hip -o probe_bf16_split && ./probe_bf16_split
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float* x, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        p1[e] = (__bf16)x[e];
        const float r = x[e] - (float)p1[e];
        p2[e] = (__bf16)r;
        p3[e] = (__bf16)(r - (float)p2[e]);
    }
}

// ---- part A: one wave, D[i][j] = sum_k W[i][k] XT[j][k]
template <int TERMS>
__global__ void __launch_bounds__(64) gemm_bf16(const float* W, const float* XT, int K, float* D) {
    const int lane = threadIdx.x, c = lane & 31, g = lane >> 5;
    f32x16 acc;
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float a[8], b[8];
        for (int e = 0; e < 8; ++e) {
            a[e] = W[(long long)c * K + k0 + 8 * g + e];
            b[e] = XT[(long long)c * K + k0 + 8 * g + e];
        }
        bf16x8 a1, a2, a3, b1, b2, b3;
        split3(a, a1, a2, a3);
        split3(b, b1, b2, b3);
        // small terms first
        if (TERMS >= 9) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b3, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b3, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b2, acc, 0, 0, 0);
        }
        if (TERMS >= 6) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
        }
        if (TERMS >= 3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
    }
    for (int v = 0; v < 16; ++v) D[((v & 3) + 8 * (v >> 2) + 4 * g) * 32 + c] = acc[v];
}

__global__ void __launch_bounds__(64) gemm_f32(const float* W, const float* XT, int K, float* D) {
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    for (int k = 0; k < K; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W[(long long)c * K + k + h], XT[(long long)c * K + k + h], acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) D[((v & 3) + 8 * (v >> 2) + 4 * h) * 32 + c] = acc[v];
}

static double frand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double gauss() { return sqrt(-2.0 * log(frand())) * cos(6.283185307179586 * frand()); }

static void accuracy(const char* what, int K, double wscale, double xscale, double xshift) {
    std::vector<float> W(32 * (size_t)K), XT(32 * (size_t)K);
    for (auto& v : W) v = (float)(gauss() * wscale);
    for (auto& v : XT) v = (float)(gauss() * xscale + xshift);
    std::vector<double> ref(1024);
    double scale = 0.0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += (double)W[(size_t)i * K + k] * (double)XT[(size_t)j * K + k];
            ref[i * 32 + j] = s;
            scale += s * s;
        }
    scale = sqrt(scale / 1024.0);
    // the plain float32 loop a CPU reference runs (sequential fused-free multiply-add in float32)
    double cpu_max = 0.0, cpu_rms = 0.0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) {
                volatile float p = W[(size_t)i * K + k] * XT[(size_t)j * K + k];
                s = s + p;
            }
            const double e = fabs((double)s - ref[i * 32 + j]) / scale;
            cpu_max = e > cpu_max ? e : cpu_max;
            cpu_rms += e * e;
        }
    float *dW, *dX, *dD;
    hipMalloc(&dW, W.size() * 4);
    hipMalloc(&dX, XT.size() * 4);
    hipMalloc(&dD, 4096);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dX, XT.data(), XT.size() * 4, hipMemcpyHostToDevice);
    printf("{\"part\": \"accuracy\", \"case\": \"%s\", \"K\": %d, \"rms_of_result\": %.4g, \"cpu_f32_loop\": {\"max\": %.3g, \"rms\": %.3g}",
           what, K, scale, cpu_max, sqrt(cpu_rms / 1024.0));
    const char* names[] = {"mfma_f32", "bf16x1", "bf16x3", "bf16x6", "bf16x9"};
    for (int m = 0; m < 5; ++m) {
        hipMemset(dD, 0, 4096);
        if (m == 0) hipLaunchKernelGGL(gemm_f32, dim3(1), dim3(64), 0, 0, dW, dX, K, dD);
        if (m == 1) hipLaunchKernelGGL(gemm_bf16<1>, dim3(1), dim3(64), 0, 0, dW, dX, K, dD);
        if (m == 2) hipLaunchKernelGGL(gemm_bf16<3>, dim3(1), dim3(64), 0, 0, dW, dX, K, dD);
        if (m == 3) hipLaunchKernelGGL(gemm_bf16<6>, dim3(1), dim3(64), 0, 0, dW, dX, K, dD);
        if (m == 4) hipLaunchKernelGGL(gemm_bf16<9>, dim3(1), dim3(64), 0, 0, dW, dX, K, dD);
        float D[1024];
        hipMemcpy(D, dD, 4096, hipMemcpyDeviceToHost);
        double mx = 0.0, rms = 0.0;
        for (int e = 0; e < 1024; ++e) {
            const double err = fabs((double)D[e] - ref[e]) / scale;
            mx = err > mx ? err : mx;
            rms += err * err;
        }
        printf(", \"%s\": {\"max\": %.3g, \"rms\": %.3g}", names[m], mx, sqrt(rms / 1024.0));
    }
    printf("}\n");
    hipFree(dW);
    hipFree(dX);
    hipFree(dD);
}

// ---- part B: cost of a k = 16 step (two 32-feature tiles x 32 rows) on one SIMD
// MODE 0: 16 v_mfma_f32_32x32x2_f32; 1: 12 v_mfma_f32_32x32x16_bf16, operands already split; 2: + the wave splits its
// activation operand (8 float32 -> 3 x 8 bf16) every step; 3: like 2 with 9 terms (18 MFMAs); 4: like 2 with 3 terms (6)
template <int MODE>
__global__ void __launch_bounds__(512) step_kernel(float* out, int iters) {
    f32x16 acc0, acc1;
    for (int v = 0; v < 16; ++v) acc0[v] = acc1[v] = 0.f;
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = threadIdx.x * 0.001f + e;
    bf16x8 w1, w2, w3, b1, b2, b3;
    split3(x, w1, w2, w3);
    split3(x, b1, b2, b3);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[e], x[7 - e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[e], x[7 - e], acc1, 0, 0, 0);
            }
        } else {
            if (MODE >= 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += 0.37f;       // (fresh values every step: 4 packed adds)
                split3(x, b1, b2, b3);
            }
            const int T = MODE == 3 ? 9 : (MODE == 4 ? 3 : 6);
            if (T >= 9) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, b3, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, b3, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b3, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b3, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, b2, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, b2, acc1, 0, 0, 0);
            }
            if (T >= 6) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b3, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b3, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, b1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, b1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b2, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b2, acc1, 0, 0, 0);
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b2, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b1, acc1, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int v = 0; v < 16; ++v) s += acc0[v] + acc1[v];
    out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void time_steps(const char* what, int waves_per_simd, float* out) {
    const int iters = 20000, threads = 256 * waves_per_simd;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((step_kernel<MODE>), dim3(256), dim3(threads), 0, 0, out, iters);     // one workgroup per CU
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    // a step = 2 x 32 x 32 x 16 multiply-adds per wave
    const double steps = (double)iters * waves_per_simd;       // per SIMD
    const double tflops = 256.0 * 4 * steps * 2.0 * 2 * 32 * 32 * 16 / (best * 1e-3) / 1e12;
    printf("{\"part\": \"step_cost\", \"mode\": \"%s\", \"waves_per_simd\": %d, \"ns_per_step_per_simd\": %.2f, "
           "\"f32_equivalent_tflops\": %.1f}\n", what, waves_per_simd, best * 1e6 / steps, tflops);
}

int main() {
    srand(12345);
    // forward, first layer: standardised observations (unit variance) x orthogonal-scale weights, K = 384
    accuracy("fwd_layer1", 384, 0.05, 1.0, 0.0);
    // hidden layer: normalised activations, K = 64
    accuracy("fwd_hidden", 64, 0.15, 1.0, 0.0);
    // first-layer weight gradient: K = rows a wave folds before the fixed-order reduction, small gradients x observations
    accuracy("dw1_rows", 49152, 1e-4, 1.0, 0.0);
    // same with a non-zero mean on one operand (sums that grow linearly)
    accuracy("dw1_rows_biased", 49152, 1e-4, 1.0, 0.5);
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    for (int w = 1; w <= 2; ++w) {
        time_steps<0>("f32_mfma_16", w, out);
        time_steps<1>("bf16x6_presplit_12", w, out);
        time_steps<2>("bf16x6_split_in_wave", w, out);
        time_steps<3>("bf16x9_split_in_wave", w, out);
        time_steps<4>("bf16x3_split_in_wave", w, out);
    }
    hipFree(out);
    return 0;
}
