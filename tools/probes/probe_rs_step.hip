// Checks the lane pairing of the reduce-scatter steps of mappo_mlp.hip (prim::rs_step) on the device:
// prints, per step, which lane each lane received from and whether the a / b selection follows the test bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#define MAPPO_DPP(v, ctrl) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    const float a = (float)lane, b = 1000.f + lane;
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
        out[lane] = __builtin_bit_cast(float, r[0]);
        out[64 + lane] = __builtin_bit_cast(float, r[1]);
        float x = a, y = b;
        asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(x), "+v"(y));
        out[384 + lane] = x;
        out[448 + lane] = y;
    }
    out[128 + lane] = MAPPO_DPP(a, 0x128);
    out[192 + lane] = MAPPO_DPP(a, 0x141);
    out[256 + lane] = MAPPO_DPP(a, 0x4E);
    out[320 + lane] = MAPPO_DPP(a, 0xB1);
}
int main() {
    float* d;
    hipMalloc(&d, 512 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[512];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"swap16.r0", "swap16.r1", "row_ror:8", "half_mirror", "quad 2301", "quad 1032", "asm swap.0", "asm swap.1"};
    for (int s = 0; s < 8; ++s) {
        printf("%-12s", names[s]);
        for (int l = 0; l < 64; ++l) printf(" %g", h[64 * s + l]);
        printf("\n");
    }
    return 0;
}
