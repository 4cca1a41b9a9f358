// K9: fused hidden-64 trunk (forward, backward chain, first-layer weight gradient) -- gfx950 binding of
// mappo_mlp_impl.h, which holds the kernels and their launchers.
#include <hip/hip_runtime.h>

#include "mappo_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4 __attribute__((ext_vector_type(4)));
typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));

namespace prim {
// v_mfma_f32_32x32x2_f32: A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31],
// D[row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5)][col = lane & 31]
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32); }
__device__ __forceinline__ float sum16(float v) {
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}
__device__ __forceinline__ float exp2_fast(float v) { return __builtin_amdgcn_exp2f(v); }   // v_exp_f32
__device__ __forceinline__ float rcp_fast(float v) { return __builtin_amdgcn_rcpf(v); }     // v_rcp_f32
// scheduling barrier: the compiler may not move instructions across it (used to keep operand prefetches early)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void set_priority_high() { __builtin_amdgcn_s_setprio(3); }
// wave-uniform value -> scalar register
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// global_load_lds_dwordx4: lane l's 16 bytes at g go to LDS address (wave-uniform) base + 16 l, not through VGPRs.  Issued
// from inline asm, so the compiler neither counts nor drains these loads: wait_lds_loads() before the data is read.
__device__ __forceinline__ void load_lds16(const float* g, float* lds_wave_base) {
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)lds_wave_base);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(dst)
        : "memory");
}
// ... until at most N of this wave's vector memory loads are outstanding (they complete in order)
template <int N>
__device__ __forceinline__ void wait_lds_loads() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// the 4-byte form (global_load_lds_dword): lane l's int at g goes to base + 4 l
__device__ __forceinline__ void load_lds4(const int* g, int* lds_wave_base) {
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)lds_wave_base);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(dst)
        : "memory");
}
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }
// device-wide release of this thread's earlier writes / a ticket from a counter in device memory (mlp_tail_kernel)
__device__ __forceinline__ void fence() { __threadfence(); }
__device__ __forceinline__ unsigned ticket(unsigned* counter) { return atomicAdd(counter, 1u); }
// the value is produced here, in program order: the compiler may neither sink its load into a later branch nor merge it
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void wait_loads_14() { asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); }   // (stamps only)
__device__ __forceinline__ long long clock() { return (long long)__builtin_readcyclecounter(); }
__device__ __forceinline__ int f2i(float v) { return __float_as_int(v); }
__device__ __forceinline__ float i2f(int v) { return __int_as_float(v); }
__device__ __forceinline__ float* lds() {
    extern __shared__ __attribute__((aligned(16))) float mappo_dyn_lds[];
    return mappo_dyn_lds;
}
}  // namespace prim

namespace prim {
// one ticket counter per process and device context: the K9 backward calls of a process are ordered on one stream
__device__ unsigned g_ticket_counter = 0;
inline unsigned* ticket_counter() {
    unsigned* p = nullptr;
    (void)hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(g_ticket_counter));
    return p;
}
}  // namespace prim

namespace {
int g_launch_error = 0;
// dynamic LDS above 64 KB has to be granted per kernel function before the launch
template <class K>
void grant_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) g_launch_error = (int)e;
    }
}
}  // namespace

#define MAPPO_LAUNCH(kernel, grid, block, lds_bytes, stream, ...)                                      \
    do {                                                                                               \
        grant_lds(kernel, (lds_bytes));                                                                \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (lds_bytes), stream, __VA_ARGS__);         \
    } while (0)
#define MAPPO_LAUNCH_ERROR() (g_launch_error ? g_launch_error : (int)hipGetLastError())

#include "mappo_mlp_impl.h"
#include "mappo_gru_impl.h"

extern "C" int mappo_mlp_forward(const mappo_mlp_t* net, mappo_stream_t stream) {
    g_launch_error = 0;
    return mlp::forward(net, static_cast<hipStream_t>(stream));
}
extern "C" int mappo_mlp_backward(const mappo_mlp_t* net, mappo_stream_t stream) {
    g_launch_error = 0;
    return mlp::backward(net, static_cast<hipStream_t>(stream));
}
extern "C" int mappo_mlp_set_debug(long long* buf) {
    mlp::debug_buffer() = buf;
    return 0;
}
extern "C" int mappo_mlp_set_grid_cap(int cap) {
    mlp::grid_cap_override() = cap;
    return 0;
}
extern "C" int64_t mappo_mlp_row_table_ints(int64_t rows) { return mlp::rows128(rows); }
extern "C" int mappo_mlp_row_table(const int64_t* idx, int64_t rows, int64_t mb, int chunk_len, int T, int N, int A,
                                   int32_t* row_tab, mappo_stream_t stream) {
    g_launch_error = 0;
    return mlp::row_table(reinterpret_cast<const long long*>(idx), rows, mb, chunk_len, T, N, A, row_tab, static_cast<hipStream_t>(stream));
}
extern "C" int64_t mappo_mlp_grad_floats(int din, int n_layers, int out) { return mlp::g_total(din, n_layers, out); }
extern "C" int64_t mappo_mlp_workspace_floats(int din, int n_layers, int out) {
    return mlp::workspace_floats(din, n_layers, out);
}
extern "C" int mappo_standardize_rows(const float* src, int64_t rows, int D, float eps, float* dst, mappo_stream_t stream) {
    g_launch_error = 0;
    return mlp::standardize_rows(src, rows, D, eps, dst, D, static_cast<hipStream_t>(stream));
}
extern "C" int mappo_standardize_rows_ld(const float* src, int64_t rows, int D, float eps, float* dst, int ld,
                                         mappo_stream_t stream) {
    g_launch_error = 0;
    return mlp::standardize_rows(src, rows, D, eps, dst, ld, static_cast<hipStream_t>(stream));
}

// K12: the GRU over a whole chunk (mappo_gru_impl.h, same primitives)
extern "C" int64_t mappo_gru_seq_gates_floats(int L, int64_t mb) { return (int64_t)L * gru::tiles_of(mb) * gru::kSaved * 2048; }
extern "C" int64_t mappo_gru_seq_stats_floats(int L, int64_t mb) { return (int64_t)L * gru::tiles_of(mb) * 64; }
extern "C" int64_t mappo_gru_seq_workspace_floats(void) { return (int64_t)gru::kGridCap * 128; }
extern "C" int mappo_gru_seq_forward(const mappo_gru_seq_t* seq, mappo_stream_t stream) {
    g_launch_error = 0;
    return gru::forward(seq, static_cast<hipStream_t>(stream));
}
extern "C" int mappo_gru_seq_backward(const mappo_gru_seq_t* seq, mappo_stream_t stream) {
    g_launch_error = 0;
    return gru::backward(seq, static_cast<hipStream_t>(stream));
}
