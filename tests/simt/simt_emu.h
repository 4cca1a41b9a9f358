// TEST INFRASTRUCTURE: a minimal single-threaded SIMT emulator for the hand-written gfx950 kernels whose logic is
// non-trivial across lanes (MFMA fragment layouts, half-wave exchanges, LDS staging between barriers).  There is no GPU
// in the build container, so the kernel *templates* (csrc/mappo_mlp_impl.h) are compiled a second time for the host
// against this header and run block by block: every "thread" of a workgroup is a ucontext fiber, a collective
// (__syncthreads, the 32x32x2 f32 MFMA, the half-wave exchange) parks the fiber until all participants arrived.
// The MFMA follows the operand maps of the ISA (A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31],
// D[row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5)][col = lane & 31], k-ordered fma chain) -- the maps the GPU-verified
// GRU step kernel (csrc/mappo_rnn.hip) relies on.  Never linked into the product library.
#ifndef SIMT_EMU_H
#define SIMT_EMU_H
#include <ucontext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <functional>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4 __attribute__((ext_vector_type(4)));
typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

struct simt_dim3 {
    unsigned x, y, z;
    simt_dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef simt_dim3 dim3;
typedef void* hipStream_t;

namespace simt {

struct Barrier {
    unsigned n = 0, count = 0, gen = 0;
};

struct Wave {
    Barrier bar;
    float a[64], b[64], x[64];
    f32x16 c[64];
    bf8 abf[64], bbf[64];
};

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    unsigned tid = 0;
};

struct State {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    Barrier block;
    Fiber* cur = nullptr;
    std::function<void()> body;
    float* lds = nullptr;
    unsigned long long n_mfma = 0;
};

inline State& st() {
    static State s;
    return s;
}

extern simt_dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

inline void yield() {
    State& s = st();
    Fiber* me = s.cur;
    swapcontext(&me->ctx, &s.sched);
}

inline void wait(Barrier& b) {
    unsigned gen = b.gen;
    if (++b.count == b.n) {
        b.count = 0;
        ++b.gen;
        return;
    }
    while (b.gen == gen) yield();
}

inline void trampoline() {
    State& s = st();
    s.body();
    s.cur->done = true;
    swapcontext(&s.cur->ctx, &s.sched);
}

constexpr size_t kStack = 1 << 20;
constexpr size_t kLdsBytes = 160 * 1024;

// Runs kernel body `f` for every block of the grid, one block after the other.
inline void launch(simt_dim3 grid, simt_dim3 block, size_t lds_bytes, const std::function<void()>& f) {
    State& s = st();
    if (lds_bytes > kLdsBytes) {
        fprintf(stderr, "simt: %zu bytes of LDS requested (limit %zu)\n", lds_bytes, kLdsBytes);
        abort();
    }
    if (!s.lds) s.lds = static_cast<float*>(aligned_alloc(64, kLdsBytes));
    const unsigned nt = block.x;
    if (nt % 64 != 0 || block.y != 1 || block.z != 1) abort();
    s.body = f;
    g_blockDim = block;
    g_gridDim = grid;
    if (s.fibers.size() < nt) {
        size_t old = s.fibers.size();
        s.fibers.resize(nt);
        for (size_t i = old; i < nt; ++i) s.fibers[i].stack = static_cast<char*>(malloc(kStack));
    }
    s.waves.resize(nt / 64);
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
            // LDS contents are undefined at workgroup start on the hardware: poison them so that a kernel relying
            // on zero-initialised LDS fails here too
            for (size_t i = 0; i < kLdsBytes / 4; ++i) s.lds[i] = NAN;
            s.block = Barrier();
            s.block.n = nt;
            for (auto& w : s.waves) {
                w.bar = Barrier();
                w.bar.n = 64;
            }
            for (unsigned t = 0; t < nt; ++t) {
                Fiber& fb = s.fibers[t];
                fb.done = false;
                fb.tid = t;
                getcontext(&fb.ctx);
                fb.ctx.uc_stack.ss_sp = fb.stack;
                fb.ctx.uc_stack.ss_size = kStack;
                fb.ctx.uc_link = nullptr;
                makecontext(&fb.ctx, (void (*)())trampoline, 0);
            }
            g_blockIdx = simt_dim3(bx, by, 0);
            unsigned live = nt;
            while (live) {
                unsigned progressed = 0;
                for (unsigned t = 0; t < nt; ++t) {
                    Fiber& fb = s.fibers[t];
                    if (fb.done) continue;
                    s.cur = &fb;
                    g_threadIdx = simt_dim3(t, 0, 0);
                    swapcontext(&s.sched, &fb.ctx);
                    ++progressed;
                    if (fb.done) --live;
                }
                if (!progressed) abort();
            }
        }
}

}  // namespace simt

#define threadIdx simt::g_threadIdx
#define blockIdx simt::g_blockIdx
#define blockDim simt::g_blockDim
#define gridDim simt::g_gridDim

inline void __syncthreads() { simt::wait(simt::st().block); }

namespace prim {

inline simt::Wave& my_wave() { return simt::st().waves[simt::st().cur->tid >> 6]; }

// v_mfma_f32_32x32x2_f32
inline f32x16 mfma32(float a, float b, f32x16 c) {
    simt::Wave& w = my_wave();
    const unsigned lane = simt::st().cur->tid & 63;
    w.a[lane] = a;
    w.b[lane] = b;
    w.c[lane] = c;
    simt::wait(w.bar);
    f32x16 d;
    const unsigned col = lane & 31, hh = lane >> 5;
    for (int v = 0; v < 16; ++v) {
        const unsigned row = (v & 3) + 8 * (v >> 2) + 4 * hh;
        float acc = w.c[lane][v];
        for (int k = 0; k < 2; ++k) acc = fmaf(w.a[row + 32 * k], w.b[col + 32 * k], acc);
        d[v] = acc;
    }
    if (lane == 0) ++simt::st().n_mfma;
    simt::wait(w.bar);
    return d;
}

// v_mfma_f32_32x32x16_bf16: A[i = lane & 31][k = 8 (lane >> 5) + e], B[k = 8 (lane >> 5) + e][j = lane & 31]; every product
// is exact in float32, the sum is carried in float32 (k-ordered chain here; the hardware's internal order is its own)
inline f32x16 mfma_bf16(bf8 a, bf8 b, f32x16 c) {
    simt::Wave& w = my_wave();
    const unsigned lane = simt::st().cur->tid & 63;
    w.abf[lane] = a;
    w.bbf[lane] = b;
    w.c[lane] = c;
    simt::wait(w.bar);
    f32x16 d;
    const unsigned col = lane & 31, hh = lane >> 5;
    for (int v = 0; v < 16; ++v) {
        const unsigned row = (v & 3) + 8 * (v >> 2) + 4 * hh;
        float acc = w.c[lane][v];
        for (int g = 0; g < 2; ++g)
            for (int e = 0; e < 8; ++e) acc = fmaf((float)w.abf[row + 32 * g][e], (float)w.bbf[col + 32 * g][e], acc);
        d[v] = acc;
    }
    if (lane == 0) ++simt::st().n_mfma;
    simt::wait(w.bar);
    return d;
}

// value held by the lane with the other half-wave index (lane ^ 32)
inline float xhalf(float v) {
    simt::Wave& w = my_wave();
    const unsigned lane = simt::st().cur->tid & 63;
    w.x[lane] = v;
    simt::wait(w.bar);
    float r = w.x[lane ^ 32];
    simt::wait(w.bar);
    return r;
}

inline float* lds() { return simt::st().lds; }
inline int lane_again() { return (int)(threadIdx.x & 63); }

}  // namespace prim

#define MAPPO_LAUNCH(kernel, grid, block, lds_bytes, stream, ...)                                   \
    simt::launch(dim3(grid), dim3(block), (lds_bytes), [&]() { kernel(__VA_ARGS__); })
#define MAPPO_LAUNCH_ERROR() 0

#endif
