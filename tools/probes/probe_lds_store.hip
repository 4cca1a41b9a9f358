// Microbenchmark (tools/probes): do LDS stores from a loader wave make progress while the compute wave on the same SIMD
// streams v_mfma_f32_32x32x2_f32?  8 waves per workgroup, 1 workgroup per CU: waves 0-3 run dependent MFMA pairs (with
// or without ds_read_b128 operand fetches), waves 4-7 run ds_write_b128 bursts (6 per s_waitcnt, as the forward kernel's
// loaders) or direct-to-LDS global loads.  Each side reports its own elapsed shader-clock ticks alone and together.
//   hipcc --offload-arch=gfx950 -O3 probe_lds_store.hip -o probe_lds_store && ./probe_lds_store
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4 __attribute__((ext_vector_type(4)));

// WHO: bit 0 = MFMA waves work, bit 1 = store waves work.  READS: MFMA waves fetch operands with ds_read_b128.
// KIND: 0 = ds_write_b128 from registers, 1 = global_load_lds_dwordx4 (no registers, no ds_write)
template <int WHO, bool READS, int KIND, int PRIO = 0, int SELF = 0>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, const float* src, int it_m, int it_s) {
    extern __shared__ __attribute__((aligned(16))) float sh[];      // [2][128 * 36 + 64 * 36]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) {
        if (!(WHO & 2)) return;
        const int lt = threadIdx.x - 256;
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);       // PRIO 1: store waves above the MFMA waves; 2: below
        v4 d[6];
        for (int p = 0; p < 6; ++p) d[p] = v4{(float)lt, (float)p, 1.f, 2.f};
        long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < it_s; ++i) {
            float* st = sh + (i & 1) * (192 * 36);
            if (KIND == 0) {
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    *reinterpret_cast<v4*>(st + ((lt >> 3) * 6 + p) * 36 + 4 * (lt & 7)) = d[p];
                    d[p][0] += 1.f;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0)
            } else {
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    // wave-uniform LDS base (M0) + lane * 16; every lane fetches its own 16 bytes
                    __builtin_amdgcn_global_load_lds(src + ((long long)(i * 6 + p) * 256 + lt) * 4 % (1 << 22),
                                                     (__attribute__((address_space(3))) void*)(st + (wave - 4) * 6 * 256 + p * 256), 16, 0, 0);
                }
                __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
            }
        }
        long long t1 = __builtin_readcyclecounter();
        if (lt == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
        if (d[0][0] == -1.f) out[threadIdx.x] = d[1][0];
        return;
    }
    if (!(WHO & 1)) return;
    if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
    f32x16 a0, a1;
    for (int v = 0; v < 16; ++v) a0[v] = a1[v] = 0.f;
    const int c = lane & 31, h = lane >> 5;
    // SELF 1: this wave also stores 6 x 16 B per lane to the other stage and loads 7 x 16 B from global memory per 32
    // MFMAs (registers); SELF 2: 7 direct-to-LDS loads instead (no registers, no ds_write)
    v4 d[6], g[7];
    for (int p = 0; p < 6; ++p) d[p] = v4{(float)lane, (float)p, 1.f, 2.f};
    for (int p = 0; p < 7; ++p) g[p] = v4{0.f, 0.f, 0.f, 0.f};
    const int lt = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < it_m; ++i) {
        float* so = sh + ((i + 1) & 1) * (192 * 36);
        const float* st = sh + (i & 1) * (192 * 36);
        const float* xa = st + (32 * wave + c) * 36 + 16 * h;
        const float* w0 = st + 128 * 36 + c * 36 + 16 * h;
        const float* w1 = w0 + 32 * 36;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v4 b = {1.f, 2.f, 3.f, 4.f}, x0 = b, x1 = b;
            if (READS) {
                b = *reinterpret_cast<const v4*>(xa + 4 * q);
                x0 = *reinterpret_cast<const v4*>(w0 + 4 * q);
                x1 = *reinterpret_cast<const v4*>(w1 + 4 * q);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[e], b[e], a0, 0, 0, 0);
                const int n = 4 * q + e;        // 16 slots per chunk
                if (SELF == 1) {
                    if (n < 6) {
                        v4 w = d[n];
                        w[0] += g[n][0];
                        *reinterpret_cast<v4*>(so + ((lt >> 3) * 6 + n) * 36 + 4 * (lt & 7)) = w;
                    } else if (n < 13) {
                        g[n - 6] = *reinterpret_cast<const v4*>(src + ((long long)(i * 7 + n - 6) * 256 + lt) * 4 % (1 << 22));
                    }
                } else if (SELF == 2 && n < 7) {
                    __builtin_amdgcn_global_load_lds(src + ((long long)(i * 7 + n) * 256 + lt) * 4 % (1 << 22),
                                                     (__attribute__((address_space(3))) void*)(so + wave * 7 * 256 + n * 256), 16, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[e], b[e], a1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (SELF == 1 && g[0][0] == -1.f) out[0] = g[1][1] + g[2][0] + g[3][0] + g[4][0] + g[5][0] + g[6][0];
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int v = 0; v < 16; ++v) s += a0[v] + a1[v];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int WHO, bool READS, int KIND, int PRIO = 0, int SELF = 0>
void run(const char* name, float* out, long long* cyc, const float* src, int it_m, int it_s) {
    const size_t lds = 2 * 192 * 36 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<WHO, READS, KIND, PRIO, SELF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float ms = 0.f;
    long long c[2] = {0, 0};
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, 16);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<WHO, READS, KIND, PRIO, SELF>), dim3(256), dim3(512), lds, 0, out, cyc, src, it_m, it_s);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    }
    printf("%-58s MFMA side %7.1f ticks / 32 MFMAs | store side %7.1f ticks / burst of 6 | kernel %.3f ms\n", name,
           (WHO & 1) ? (double)c[0] / it_m : 0.0, (WHO & 2) ? (double)c[1] / it_s : 0.0, ms);
}

int main() {
    float *out, *src;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 16);
    hipMalloc(&src, (1 << 22) * 4 + 4096);
    hipMemset(src, 0, (1 << 22) * 4 + 4096);
    const int it_m = 2000, it_s = 2000;
    run<1, false, 0>("MFMA alone", out, cyc, src, it_m, it_s);
    run<1, true, 0>("MFMA + ds_read_b128 operands alone", out, cyc, src, it_m, it_s);
    run<2, false, 0>("ds_write_b128 bursts alone", out, cyc, src, it_m, it_s);
    run<3, false, 0>("MFMA | ds_write_b128 bursts", out, cyc, src, it_m, it_s);
    run<3, true, 0>("MFMA + ds_read_b128 | ds_write_b128 bursts", out, cyc, src, it_m, it_s);
    run<2, false, 1>("global_load_lds bursts alone", out, cyc, src, it_m, it_s);
    run<3, false, 1>("MFMA | global_load_lds bursts", out, cyc, src, it_m, it_s);
    run<3, true, 1>("MFMA + ds_read_b128 | global_load_lds bursts", out, cyc, src, it_m, it_s);
    run<3, true, 0, 1>("MFMA + ds_read | ds_write bursts, store waves prio 3", out, cyc, src, it_m, it_s);
    run<3, true, 0, 2>("MFMA + ds_read | ds_write bursts, MFMA waves prio 3", out, cyc, src, it_m, it_s);
    run<3, true, 1, 1>("MFMA + ds_read | global_load_lds, store waves prio 3", out, cyc, src, it_m, it_s);
    run<1, true, 0, 0, 1>("MFMA + ds_read + own 6 ds_write + 7 global loads", out, cyc, src, it_m, it_s);
    run<1, true, 0, 0, 2>("MFMA + ds_read + own 7 global_load_lds", out, cyc, src, it_m, it_s);
    return 0;
}
