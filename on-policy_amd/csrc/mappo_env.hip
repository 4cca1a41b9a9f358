// K11: one step of all simple_spread worlds as ONE launch (row f1 of the scope table: the device-resident rollout loop).
// The worlds of config 1 / config 3 (MPE cooperative navigation; reference onpolicy/envs/mpe/core.py:120-190 physics,
// environment.py:100-180 step / action decoding, scenarios/simple_spread.py:60-103 reward / observation) are a few
// dozen float64 operations per agent pair.  Written as array operations on device tensors a step is ~60 small launches
// (measured: 140 ms per step at 4096 worlds, ten times slower than the numpy env on the host); here a thread owns a
// world and keeps its agents in registers.  State is float64 like the reference's numpy physics, observations and rewards
// leave as float32.  Restarted worlds take their positions from `fresh_*`, uniform draws the caller makes every step
// (branch-free: drawn for all worlds, used where a world restarts), so the trajectories are those of the tensor
// implementation for the same generator.
#include <hip/hip_runtime.h>

#include "../../include/mappo_hip.h"
#include "mappo_internal.h"

namespace {

constexpr int kMax = MAPPO_ENV_MAX_ENTITIES;
constexpr double kDt = 0.1, kDamping = 0.25, kSens = 5.0, kForce = 1e2, kMargin = 1e-3, kSize = 0.15;

struct Args {
    double* pos;            // [N, A, 2]
    double* vel;            // [N, A, 2]
    double* land;           // [N, L, 2]
    long long* t;           // [N]
    const long long* act;   // [N, A] action indices 0..4
    const double* fresh_pos;    // [N, A, 2] uniform(-1, 1)
    const double* fresh_land;   // [N, L, 2]
    float* obs;             // [N, A, Do], Do = 4 + 2 L + 4 (A - 1)
    float* rew;             // [N, A, 1]
    unsigned char* done;    // [N, A] (bool)
    double* per_agent;      // [N, A]
    long long n;
    int A, L, world_length, auto_reset;
};

__global__ void __launch_bounds__(64) spread_step_kernel(Args a) {
    const long long w = (long long)blockIdx.x * 64 + threadIdx.x;
    if (w >= a.n) return;
    const int A = a.A, L = a.L;
    double px[kMax], py[kMax], vx[kMax], vy[kMax], lx[kMax], ly[kMax];
#pragma unroll
    for (int i = 0; i < kMax; ++i) {
        if (i < A) {
            px[i] = a.pos[(w * A + i) * 2];
            py[i] = a.pos[(w * A + i) * 2 + 1];
            vx[i] = a.vel[(w * A + i) * 2];
            vy[i] = a.vel[(w * A + i) * 2 + 1];
        }
        if (i < L) {
            lx[i] = a.land[(w * L + i) * 2];
            ly[i] = a.land[(w * L + i) * 2 + 1];
        }
    }
    // ---- forces: action (environment.py: u[0] += a[1] - a[2], u[1] += a[3] - a[4], x sensitivity) + soft contacts
    double fx[kMax], fy[kMax];
#pragma unroll
    for (int i = 0; i < kMax; ++i) {
        if (i >= A) continue;
        const long long ac = a.act[w * A + i];
        fx[i] = (ac == 1 ? 1.0 : ac == 2 ? -1.0 : 0.0) * kSens;
        fy[i] = (ac == 3 ? 1.0 : ac == 4 ? -1.0 : 0.0) * kSens;
    }
#pragma unroll
    for (int i = 0; i < kMax; ++i) {
        if (i >= A) continue;
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int j = 0; j < kMax; ++j) {
            if (j >= A || j == i) continue;
            const double dx = px[i] - px[j], dy = py[i] - py[j];
            const double dist = sqrt(dx * dx + dy * dy);
            const double x = -(dist - 2 * kSize) / kMargin;
            const double pen = (fmax(x, 0.0) + log1p(exp(-fabs(x)))) * kMargin;     // logaddexp(0, x) * margin
            double gx = kForce * dx / dist * pen, gy = kForce * dy / dist * pen;
            if (!isfinite(gx)) gx = 0.0;                                             // coincident agents: no force
            if (!isfinite(gy)) gy = 0.0;
            sx += gx;
            sy += gy;
        }
        fx[i] += sx;
        fy[i] += sy;
    }
    // ---- integrate (core.py:160-175: damping, then force * dt; no mass / max speed in this scenario)
#pragma unroll
    for (int i = 0; i < kMax; ++i) {
        if (i >= A) continue;
        vx[i] = vx[i] * (1 - kDamping) + fx[i] * kDt;
        vy[i] = vy[i] * (1 - kDamping) + fy[i] * kDt;
        px[i] = px[i] + vx[i] * kDt;
        py[i] = py[i] + vy[i] * kDt;
    }
    long long t = a.t[w] + 1;
    // ---- reward (simple_spread.py:60-84): -sum over landmarks of the closest agent's distance, -1 per contact
    double cover = 0.0;
#pragma unroll
    for (int l = 0; l < kMax; ++l) {
        if (l >= L) continue;
        double best = 1e300;
#pragma unroll
        for (int i = 0; i < kMax; ++i) {
            if (i >= A) continue;
            const double dx = px[i] - lx[l], dy = py[i] - ly[l];
            best = fmin(best, sqrt(dx * dx + dy * dy));
        }
        cover -= best;
    }
    double total = 0.0;
    double pa[kMax];
#pragma unroll
    for (int i = 0; i < kMax; ++i) {
        if (i >= A) continue;
        int hits = 0;
#pragma unroll
        for (int j = 0; j < kMax; ++j) {
            if (j >= A) continue;
            const double dx = px[i] - px[j], dy = py[i] - py[j];
            hits += sqrt(dx * dx + dy * dy) < 2 * kSize;        // the agent itself included, as in the reference
        }
        pa[i] = cover - hits;
        total += pa[i];
    }
    const bool done = t >= a.world_length;
#pragma unroll
    for (int i = 0; i < kMax; ++i) {
        if (i >= A) continue;
        a.per_agent[w * A + i] = pa[i];
        a.rew[w * A + i] = (float)total;
        a.done[w * A + i] = done ? 1 : 0;
    }
    if (done && a.auto_reset) {
        t = 0;
#pragma unroll
        for (int i = 0; i < kMax; ++i) {
            if (i < A) {
                px[i] = a.fresh_pos[(w * A + i) * 2];
                py[i] = a.fresh_pos[(w * A + i) * 2 + 1];
                vx[i] = 0.0;
                vy[i] = 0.0;
            }
            if (i < L) {
                lx[i] = a.fresh_land[(w * L + i) * 2];
                ly[i] = a.fresh_land[(w * L + i) * 2 + 1];
            }
        }
    }
    a.t[w] = t;
#pragma unroll
    for (int i = 0; i < kMax; ++i) {
        if (i < A) {
            a.pos[(w * A + i) * 2] = px[i];
            a.pos[(w * A + i) * 2 + 1] = py[i];
            a.vel[(w * A + i) * 2] = vx[i];
            a.vel[(w * A + i) * 2 + 1] = vy[i];
        }
        if (i < L && done && a.auto_reset) {
            a.land[(w * L + i) * 2] = lx[i];
            a.land[(w * L + i) * 2 + 1] = ly[i];
        }
    }
    // ---- observation of the (possibly restarted) world: vel, pos, landmarks and other agents relative to the agent,
    // (A - 1) * 2 zero communication channels (simple_spread.py:86-103)
    const int Do = 4 + 2 * L + 4 * (A - 1);
#pragma unroll
    for (int i = 0; i < kMax; ++i) {
        if (i >= A) continue;
        float* o = a.obs + (w * A + i) * Do;
        int k = 0;
        o[k++] = (float)vx[i];
        o[k++] = (float)vy[i];
        o[k++] = (float)px[i];
        o[k++] = (float)py[i];
#pragma unroll
        for (int l = 0; l < kMax; ++l) {
            if (l >= L) continue;
            o[k++] = (float)(lx[l] - px[i]);
            o[k++] = (float)(ly[l] - py[i]);
        }
#pragma unroll
        for (int j = 0; j < kMax; ++j) {
            if (j >= A || j == i) continue;
            o[k++] = (float)(px[j] - px[i]);
            o[k++] = (float)(py[j] - py[i]);
        }
        for (int z = 0; z < 2 * (A - 1); ++z) o[k++] = 0.f;
    }
}

}  // namespace

extern "C" int mappo_simple_spread_step(double* pos, double* vel, double* landmarks, int64_t* t, const int64_t* actions,
                                        const double* fresh_pos, const double* fresh_landmarks, float* obs,
                                        float* rewards, uint8_t* dones, double* per_agent, int64_t n_worlds,
                                        int num_agents, int num_landmarks, int world_length, int auto_reset,
                                        mappo_stream_t stream_) {
    if (!pos || !vel || !landmarks || !t || !actions || !obs || !rewards || !dones || !per_agent) return MAPPO_E_NULL;
    if (auto_reset && (!fresh_pos || !fresh_landmarks)) return MAPPO_E_NULL;
    if (n_worlds <= 0 || num_agents < 1 || num_landmarks < 1 || world_length < 1) return MAPPO_E_SHAPE;
    if (num_agents > kMax || num_landmarks > kMax) return MAPPO_E_TOO_MANY;
    Args a;
    a.pos = pos;
    a.vel = vel;
    a.land = landmarks;
    a.t = reinterpret_cast<long long*>(t);
    a.act = reinterpret_cast<const long long*>(actions);
    a.fresh_pos = fresh_pos;
    a.fresh_land = fresh_landmarks;
    a.obs = obs;
    a.rew = rewards;
    a.done = dones;
    a.per_agent = per_agent;
    a.n = n_worlds;
    a.A = num_agents;
    a.L = num_landmarks;
    a.world_length = world_length;
    a.auto_reset = auto_reset;
    hipLaunchKernelGGL(spread_step_kernel, dim3((unsigned)((n_worlds + 63) / 64)), dim3(64), 0,
                       static_cast<hipStream_t>(stream_), a);
    return (int)hipGetLastError();
}
