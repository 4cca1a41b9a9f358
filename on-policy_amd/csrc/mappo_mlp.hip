// K9: fused hidden-64 trunk (forward, backward chain, first-layer weight gradient) -- gfx950 binding of
// mappo_mlp_impl.h, which holds the kernels and their launchers.
#include <hip/hip_runtime.h>
#include <mutex>
#include <map>
#include <unordered_map>
#include <utility>

#include "mappo_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4 __attribute__((ext_vector_type(4)));
typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

namespace prim {
// v_mfma_f32_32x32x16_bf16: A[i = lane & 31][k = 8 (lane >> 5) + e], B[k = 8 (lane >> 5) + e][j = lane & 31], e < 8; D as below
__device__ __forceinline__ f32x16 mfma_bf16(bf8 a, bf8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
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
// One step of a reduce-scatter over lanes: lanes whose bit BIT is clear return a + (a of lane ^ XOR), the others
// b + (b of lane ^ XOR).  The exchanges are register-to-register (v_permlane16_swap between the 16-lane rows, DPP inside a
// row: no LDS crossbar like __shfl_xor's ds_bpermute), so 31 of them fold a lane's 32 values over the 32 lanes of its
// half-wave in ~90 VALU instructions.
template <int XOR, int BIT>
__device__ __forceinline__ float rs_step(float a, float b);
#define MAPPO_DPP(v, ctrl) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
// The step between the 16-lane rows, eight pairs at a time: out[i] = rs_step<16, 4>(a[i], b[i]).  v_permlane16_swap_b32
// exchanges rows 1 / 3 of its first operand with rows 0 / 2 of the second, after which every lane holds its own and
// its partner's copy of the value it keeps.  Inline assembly: this compiler's __builtin_amdgcn_permlane16_swap returns its
// FIRST result in both halves of the pair (tools/probes/probe_rs_step.hip); the leading s_nop covers the two wait states
// the instruction needs after a VALU write of its operands (the compiler does not look inside the block).
__device__ __forceinline__ void rs16_8(const float* a, const float* b, float* out) {
    float x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3], x4 = a[4], x5 = a[5], x6 = a[6], x7 = a[7];
    float y0 = b[0], y1 = b[1], y2 = b[2], y3 = b[3], y4 = b[4], y5 = b[5], y6 = b[6], y7 = b[7];
    asm("s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %8\n\t"
        "v_permlane16_swap_b32 %1, %9\n\t"
        "v_permlane16_swap_b32 %2, %10\n\t"
        "v_permlane16_swap_b32 %3, %11\n\t"
        "v_permlane16_swap_b32 %4, %12\n\t"
        "v_permlane16_swap_b32 %5, %13\n\t"
        "v_permlane16_swap_b32 %6, %14\n\t"
        "v_permlane16_swap_b32 %7, %15\n\t"
        "s_nop 0"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(y0), "+v"(y1), "+v"(y2),
          "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7));
    out[0] = x0 + y0;
    out[1] = x1 + y1;
    out[2] = x2 + y2;
    out[3] = x3 + y3;
    out[4] = x4 + y4;
    out[5] = x5 + y5;
    out[6] = x6 + y6;
    out[7] = x7 + y7;
}
template <>
__device__ __forceinline__ float rs_step<8, 3>(float a, float b) {        // row_ror:8
    const float t = a + MAPPO_DPP(a, 0x128), u = b + MAPPO_DPP(b, 0x128);
    return (threadIdx.x & 8) ? u : t;
}
template <>
__device__ __forceinline__ float rs_step<7, 2>(float a, float b) {        // row_half_mirror: lane ^ 7
    const float t = a + MAPPO_DPP(a, 0x141), u = b + MAPPO_DPP(b, 0x141);
    return (threadIdx.x & 4) ? u : t;
}
template <>
__device__ __forceinline__ float rs_step<2, 1>(float a, float b) {        // quad_perm [2, 3, 0, 1]
    const float t = a + MAPPO_DPP(a, 0x4E), u = b + MAPPO_DPP(b, 0x4E);
    return (threadIdx.x & 2) ? u : t;
}
template <>
__device__ __forceinline__ float rs_step<1, 0>(float a, float b) {        // quad_perm [1, 0, 3, 2]
    const float t = a + MAPPO_DPP(a, 0xB1), u = b + MAPPO_DPP(b, 0xB1);
    return (threadIdx.x & 1) ? u : t;
}
#undef MAPPO_DPP
__device__ __forceinline__ float exp2_fast(float v) { return __builtin_amdgcn_exp2f(v); }   // v_exp_f32
__device__ __forceinline__ float rcp_fast(float v) { return __builtin_amdgcn_rcpf(v); }     // v_rcp_f32
__device__ __forceinline__ float rsq_fast(float v) { return __builtin_amdgcn_rsqf(v); }     // v_rsq_f32 (1 ulp)
// scheduling barrier: the compiler may not move instructions across it (used to keep operand prefetches early)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void set_priority_high() { __builtin_amdgcn_s_setprio(3); }
// element i of an accumulator tile that must STAY in the accumulator half of the register file (one wave per SIMD: 256 + 256
// registers): a plain a[i] in vector arithmetic makes the register allocator move the whole tile -- in K15's forward all 16
// tiles -- into vector registers at once (and spill); this reads one value when it is needed.  Volatile: a second read is a
// second instruction, not a value kept alive.  mfma_drain() first: the hazard recogniser does not look inside inline asm, and
// a matrix instruction's result may not be read for up to 18 passes after its issue.
__device__ __forceinline__ float acc_get(const f32x16& a, int i) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a[i]));
    return v;
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }
// wave-uniform value -> scalar register
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// this lane's index in its wave, read afresh from the hardware lane counter (two instructions): for an epilogue that would
// otherwise keep a function of threadIdx alive -- or spilled -- across a loop that uses every vector register
__device__ __forceinline__ int lane_again() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
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
// Four of them under ONE M0 set-up: units 0 .. 3 of a run that is contiguous in global memory AND in LDS (1 KB apart in both) --
// the instruction's offset field moves the global address and the LDS address together.
__device__ __forceinline__ void load_lds16x4(const float* g, float* lds_wave_base) {
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)lds_wave_base);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\t"
        "global_load_lds_dwordx4 %1, off offset:2048\n\t"
        "global_load_lds_dwordx4 %1, off offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(dst)
        : "memory");
}
#ifndef MAPPO_K9_NT
#define MAPPO_K9_NT 30      // (the default of mappo_mlp_impl.h, needed before its include)
#endif
// the same with the non-temporal bit when the build streams the weight-gradient kernels' operands (MAPPO_K9_NT & 4)
__device__ __forceinline__ void load_lds16s(const float* g, float* lds_wave_base) {
#if (MAPPO_K9_NT) & 4
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)lds_wave_base);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(dst)
        : "memory");
#else
    load_lds16(g, lds_wave_base);
#endif
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
__device__ __forceinline__ void pin(int& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void wait_loads_14() { asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); }   // (stamps only)
__device__ __forceinline__ long long clock() { return (long long)__builtin_readcyclecounter(); }
__device__ __forceinline__ int f2i(float v) { return __float_as_int(v); }
__device__ __forceinline__ float i2f(int v) { return __int_as_float(v); }
__device__ __forceinline__ float* lds() {
    extern __shared__ __attribute__((aligned(16))) float mappo_dyn_lds[];
    return mappo_dyn_lds;
}
}  // namespace prim

namespace {
int g_launch_error = 0;
// dynamic LDS above 64 KB has to be granted per kernel function before the launch
// (granted once per kernel and size: the attribute call is kept out of the launch path -- in particular out of launches that
// are being captured into a HIP graph from the autograd thread, onpolicy/algorithms/r_mappo/update_graph.py)
template <class K>
void grant_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024) {
        // the attribute belongs to (device, function): a process that drives two GPUs must be granted on both
        static std::mutex mu;
        static std::map<std::pair<int, const void*>, size_t> granted;
        const void* fn = reinterpret_cast<const void*>(kernel);
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess) device = -1;
        std::lock_guard<std::mutex> lock(mu);
        size_t& have = granted[std::make_pair(device, fn)];
        if (bytes > have) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e != hipSuccess) g_launch_error = (int)e;
            else if (device >= 0) have = bytes;
        }
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
#include "mappo_lin_impl.h"

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
extern "C" int mappo_mlp_set_flags(int flags) { return mlp::set_tuning_flags(flags); }
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
extern "C" int64_t mappo_gru_seq_workspace_floats(void) { return (int64_t)gru::kGridCap * gru::kSums; }
extern "C" int mappo_gru_seq_forward(const mappo_gru_seq_t* seq, mappo_stream_t stream) {
    g_launch_error = 0;
    return gru::forward(seq, static_cast<hipStream_t>(stream));
}
extern "C" int mappo_gru_seq_backward(const mappo_gru_seq_t* seq, mappo_stream_t stream) {
    g_launch_error = 0;
    return gru::backward(seq, static_cast<hipStream_t>(stream));
}
extern "C" int64_t mappo_gru_weight_grads_workspace_floats(void) { return gru::wgrad_workspace_floats(); }
extern "C" int mappo_gru_weight_grads(const float* dgi, const float* dq, const float* x, const float* hm, int64_t rows, float* dw,
                                      float* workspace, mappo_stream_t stream) {
    g_launch_error = 0;
    return gru::weight_grads(dgi, dq, x, hm, rows, dw, workspace, static_cast<hipStream_t>(stream));
}

// ---- K15: tall Linear layers with 512 outputs in six-term bf16 arithmetic (mappo_lin_impl.h)
extern "C" int64_t mappo_linear512_planes_floats(int K) { return lin::planes_floats(K); }
extern "C" int mappo_linear512_prepare(const float* w, int K, int ldw, int transposed, float* planes, mappo_stream_t stream) {
    g_launch_error = 0;
    return lin::prepare(w, K, ldw, transposed, planes, static_cast<hipStream_t>(stream));
}
extern "C" int mappo_linear512_forward(const float* x, int64_t rows, int K, int ldx, const float* planes, const float* bias,
                                       float* y, mappo_stream_t stream) {
    g_launch_error = 0;
    return lin::forward(x, rows, K, ldx, planes, bias, y, static_cast<hipStream_t>(stream));
}
extern "C" int mappo_linear512_forward_norm(const float* x, int64_t rows, int K, int ldx, const float* planes, const float* bias,
                                            const float* gamma, const float* beta, float eps, int act, float* y, float* yn,
                                            float* mean, float* rstd, mappo_stream_t stream) {
    g_launch_error = 0;
    return lin::forward_norm(x, rows, K, ldx, planes, bias, gamma, beta, eps, act, y, yn, mean, rstd,
                             static_cast<hipStream_t>(stream));
}
extern "C" int64_t mappo_linear512_wgrad_workspace_floats(int K) { return lin::wgrad_workspace_floats(K); }
extern "C" int mappo_linear512_wgrad(const float* dy, const float* x, int64_t rows, int K, int ldx, float* dw,
                                     float* workspace, mappo_stream_t stream) {
    g_launch_error = 0;
    return lin::wgrad(dy, x, rows, K, ldx, dw, workspace, static_cast<hipStream_t>(stream));
}
