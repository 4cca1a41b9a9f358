// TEST INFRASTRUCTURE: the kernels of csrc/mappo_mlp_impl.h compiled for the host against the SIMT emulator
// (simt_emu.h), exported under the product's C ABI names with HOST pointers.  Built by tests/simt/build.py into
// tests/simt/_build/libmlp_emu.so; used only by tests/test_mlp_kernels_emulated.py.
#include "simt_emu.h"

simt_dim3 simt::g_threadIdx, simt::g_blockIdx, simt::g_blockDim, simt::g_gridDim;

namespace prim {
inline float exp2_fast(float v) { return exp2f(v); }
inline float rcp_fast(float v) { return 1.f / v; }
inline float rsq_fast(float v) { return 1.f / sqrtf(v); }
inline void sched_fence() {}
inline void set_priority_high() {}
inline long long clock() { return 0; }
inline void wait_loads_14() {}
inline void wave_sync() { simt::wait(my_wave().bar); }
inline void pin(float&) {}
inline void pin(int&) {}
inline void fence() {}
inline unsigned ticket(unsigned* counter) { return (*counter)++; }       // (blocks run one after the other)
inline int f2i(float v) { int i; memcpy(&i, &v, 4); return i; }
inline float i2f(int v) { float f; memcpy(&f, &v, 4); return f; }
inline int uniform(int v) { return v; }
inline float acc_get(const f32x16& a, int i) { return a[i]; }
inline void mfma_drain() {}
// direct-to-LDS load: lane l's 16 bytes land at base + 16 l.  The copy happens at once; wait_lds_loads() is a wave
// barrier, so that no lane reads a slot before every lane of its wave has issued its part.
inline void load_lds16(const float* g, float* lds_wave_base) {
    memcpy(lds_wave_base + 4 * (simt::st().cur->tid & 63), g, 16);
}
inline void load_lds16s(const float* g, float* lds_wave_base) { load_lds16(g, lds_wave_base); }
inline void load_lds16x4(const float* g, float* lds_wave_base) {
    for (int u = 0; u < 4; ++u) load_lds16(g + 256 * u, lds_wave_base + 256 * u);
}
template <int N>
inline void wait_lds_loads() { simt::wait(my_wave().bar); }
inline void load_lds4(const int* g, int* lds_wave_base) { lds_wave_base[simt::st().cur->tid & 63] = *g; }
// sum over the 16-lane group of the calling lane (a wave collective)
inline float sum16(float v) {
    simt::Wave& w = my_wave();
    const unsigned lane = simt::st().cur->tid & 63;
    w.x[lane] = v;
    simt::wait(w.bar);
    float s = 0.f;
    // butterfly order of the device version: ((v + v^8) + (..^4)) ... evaluated pairwise
    float t[16];
    for (int i = 0; i < 16; ++i) t[i] = w.x[(lane & ~15u) + i];
    for (int m = 8; m >= 1; m >>= 1) {
        float u[16];
        for (int i = 0; i < 16; ++i) u[i] = t[i] + t[i ^ m];
        for (int i = 0; i < 16; ++i) t[i] = u[i];
    }
    s = t[lane & 15];
    simt::wait(w.bar);
    return s;
}
// one step of a reduce-scatter over lanes: lanes whose bit BIT is clear return a + (a of lane ^ XOR), the others
// b + (b of lane ^ XOR) (XOR always flips bit BIT)
template <int XOR, int BIT>
inline float rs_step(float a, float b) {
    static_assert((XOR >> BIT) & 1, "the partner must sit on the other side");
    simt::Wave& w = my_wave();
    const unsigned lane = simt::st().cur->tid & 63;
    w.a[lane] = a;
    w.b[lane] = b;
    simt::wait(w.bar);
    const unsigned p = lane ^ XOR;
    const float r = ((lane >> BIT) & 1) ? b + w.b[p] : a + w.a[p];
    simt::wait(w.bar);
    return r;
}
inline void rs16_8(const float* a, const float* b, float* out) {
    for (int i = 0; i < 8; ++i) out[i] = rs_step<16, 4>(a[i], b[i]);
}
}  // namespace prim

#include "../../on-policy_amd/csrc/mappo_mlp_impl.h"
#include "../../on-policy_amd/csrc/mappo_gru_impl.h"
#include "../../on-policy_amd/csrc/mappo_lin_impl.h"

extern "C" int mappo_mlp_forward(const mappo_mlp_t* net, mappo_stream_t stream) { return mlp::forward(net, stream); }
extern "C" int mappo_mlp_backward(const mappo_mlp_t* net, mappo_stream_t stream) { return mlp::backward(net, stream); }
extern "C" int mappo_mlp_set_debug(long long* buf) {
    mlp::debug_buffer() = buf;
    return 0;
}
extern "C" int mappo_mlp_set_grid_cap(int cap) {
    mlp::grid_cap_override() = cap;
    return 0;
}
extern "C" int mappo_mlp_set_flags(int flags) { return mlp::set_tuning_flags(flags); }
extern "C" int64_t mappo_mlp_row_table_ints(int64_t rows) { return mlp::rows128(rows); }
extern "C" int mappo_mlp_row_table(const int64_t* idx, int64_t rows, int64_t mb, int chunk_len, int T, int N, int A,
                                   int32_t* row_tab, mappo_stream_t stream) {
    return mlp::row_table(reinterpret_cast<const long long*>(idx), rows, mb, chunk_len, T, N, A, row_tab, stream);
}
extern "C" int64_t mappo_mlp_grad_floats(int din, int n_layers, int out) { return mlp::g_total(din, n_layers, out); }
extern "C" int64_t mappo_mlp_workspace_floats(int din, int n_layers, int out) {
    return mlp::workspace_floats(din, n_layers, out);
}
extern "C" int mappo_standardize_rows(const float* src, int64_t rows, int D, float eps, float* dst, mappo_stream_t stream) {
    return mlp::standardize_rows(src, rows, D, eps, dst, D, stream);
}
extern "C" int mappo_standardize_rows_ld(const float* src, int64_t rows, int D, float eps, float* dst, int ld,
                                         mappo_stream_t stream) {
    return mlp::standardize_rows(src, rows, D, eps, dst, ld, stream);
}
extern "C" unsigned long long simt_mfma_count() { return simt::st().n_mfma; }
// the three-way bf16 split of the six-term kernels on n (a multiple of 8) values: parts widened back to float32
extern "C" void simt_split3(const float* x, long long n, float* p1, float* p2, float* p3) {
    for (long long i = 0; i + 8 <= n; i += 8) {
        bf8 a, b, c;
        mlp::split3(x + i, a, b, c);
        for (int e = 0; e < 8; ++e) {
            p1[i + e] = (float)a[e];
            p2[i + e] = (float)b[e];
            p3[i + e] = (float)c[e];
        }
    }
}

extern "C" int64_t mappo_gru_seq_gates_floats(int L, int64_t mb) { return (int64_t)L * gru::tiles_of(mb) * gru::kSaved * 2048; }
extern "C" int64_t mappo_gru_seq_stats_floats(int L, int64_t mb) { return (int64_t)L * gru::tiles_of(mb) * 64; }
extern "C" int64_t mappo_gru_seq_workspace_floats(void) { return (int64_t)gru::kGridCap * gru::kSums; }
extern "C" int mappo_gru_seq_forward(const mappo_gru_seq_t* seq, mappo_stream_t stream) { return gru::forward(seq, stream); }
extern "C" int mappo_gru_seq_backward(const mappo_gru_seq_t* seq, mappo_stream_t stream) { return gru::backward(seq, stream); }
extern "C" int64_t mappo_gru_weight_grads_workspace_floats(void) { return gru::wgrad_workspace_floats(); }
extern "C" int mappo_gru_weight_grads(const float* dgi, const float* dq, const float* x, const float* hm, int64_t rows, float* dw,
                                      float* workspace, mappo_stream_t stream) {
    return gru::weight_grads(dgi, dq, x, hm, rows, dw, workspace, stream);
}

// ---- K15: tall Linear layers with 512 outputs in six-term bf16 arithmetic (mappo_lin_impl.h)
extern "C" int64_t mappo_linear512_planes_floats(int K) { return lin::planes_floats(K); }
extern "C" int mappo_linear512_prepare(const float* w, int K, int ldw, int transposed, float* planes, mappo_stream_t stream) {
    return lin::prepare(w, K, ldw, transposed, planes, stream);
}
extern "C" int mappo_linear512_forward(const float* x, int64_t rows, int K, int ldx, const float* planes, const float* bias,
                                       float* y, mappo_stream_t stream) {
    return lin::forward(x, rows, K, ldx, planes, bias, y, stream);
}
extern "C" int mappo_linear512_forward_norm(const float* x, int64_t rows, int K, int ldx, const float* planes, const float* bias,
                                            const float* gamma, const float* beta, float eps, int act, float* y, float* yn,
                                            float* mean, float* rstd, mappo_stream_t stream) {
    return lin::forward_norm(x, rows, K, ldx, planes, bias, gamma, beta, eps, act, y, yn, mean, rstd, stream);
}
extern "C" int64_t mappo_linear512_wgrad_workspace_floats(int K) { return lin::wgrad_workspace_floats(K); }
extern "C" int mappo_linear512_wgrad(const float* dy, const float* x, int64_t rows, int K, int ldx, float* dw,
                                     float* workspace, mappo_stream_t stream) {
    return lin::wgrad(dy, x, rows, K, ldx, dw, workspace, stream);
}
