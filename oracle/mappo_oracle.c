/*
 * mappo_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded restatement of the reference's rollout-buffer hot path
 * (marlbenchmark/on-policy).  It exists only so that tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg can check / time the HIP kernels against something that
 * follows the reference line by line.  Nothing under on-policy_amd/ may import, link or
 * call it.
 *
 * Pinning: oracle/make_golden.py runs the reference itself (imported from
 * /root/reference in the build container) on seeded inputs and stores its outputs under
 * tests/golden/; tests/test_oracle_golden.py checks this file bit-for-bit against those
 * fixtures, including the five known-answer vectors of SURVEY.md section 8c.
 *
 * Every function cites the reference lines it restates (paths relative to the reference
 * root).  float32 arithmetic, one rounding per numpy operation: build with
 * -ffp-contract=off (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_USE_GAE 1u
#define ORC_PROPER_TIME_LIMITS 2u
#define ORC_DENORM 4u

/* onpolicy/utils/valuenorm.py:68-79 (== onpolicy/algorithms/utils/popart.py:88-98):
 * out = x * sqrt(var) + mean, evaluated as a float32 multiply followed by a float32 add. */
static float denorm(float x, float sigma, float mu, int on) {
    if (!on) return x;
    volatile float s = x * sigma;
    return s + mu;
}

/*
 * onpolicy/utils/shared_buffer.py:179-262, SharedReplayBuffer.compute_returns, all
 * non-MAT branches.  Arrays are [T or T+1, C] float32 (C = n_rollout_threads*num_agents).
 * gamma / lam are the Python floats; gamma*lam is formed in float64 and then rounded to
 * float32 when it meets the float32 array (shared_buffer.py:239), gamma alone is rounded
 * to float32 where it multiplies an array.
 */
void orc_compute_returns(const float* rewards, float* value_preds, const float* next_value,
                         const float* masks, const float* bad_masks, float* returns,
                         float sigma, float mu, int T, int64_t C, double gamma, double lam,
                         unsigned flags) {
    const int use_gae = (flags & ORC_USE_GAE) != 0;
    const int ptl = (flags & ORC_PROPER_TIME_LIMITS) != 0;
    const int dn = (flags & ORC_DENORM) != 0;
    const float g32 = (float)gamma;
    const float gl32 = (float)(gamma * lam);
    if (use_gae) {
        memcpy(value_preds + (int64_t)T * C, next_value, (size_t)C * sizeof(float)); /* :187,:218 */
    } else {
        memcpy(returns + (int64_t)T * C, next_value, (size_t)C * sizeof(float)); /* :205,:260 */
    }
    for (int64_t c = 0; c < C; ++c) {
        volatile float gae = 0.0f; /* :188,:219 */
        for (int t = T - 1; t >= 0; --t) { /* :189,:206,:220,:261 */
            const int64_t o = (int64_t)t * C + c;
            const float r = rewards[o];
            const float m1 = masks[o + C];
            const float b1 = ptl ? bad_masks[o + C] : 1.0f;
            if (use_gae) {
                const float dv1 = denorm(value_preds[o + C], sigma, mu, dn);
                const float dv0 = denorm(value_preds[o], sigma, mu, dn);
                /* delta = r + gamma*D(v[t+1])*m[t+1] - D(v[t])   (:192-194,:199-200,:236-238,:255-256) */
                volatile float x = g32 * dv1;
                volatile float y = x * m1;
                volatile float z = r + y;
                volatile float delta = z - dv0;
                volatile float carry;
                if (ptl && dn) { /* :195 gamma*lambda*gae*mask */
                    volatile float u = gl32 * gae;
                    carry = u * m1;
                } else { /* :201,:239,:257 gamma*lambda*mask*gae */
                    volatile float u = gl32 * m1;
                    carry = u * gae;
                }
                gae = delta + carry;
                if (ptl) gae = gae * b1; /* :196,:202 */
                returns[o] = gae + dv0;   /* :197,:203,:240,:258 */
            } else {
                /* returns[t] = returns[t+1]*gamma*m[t+1] + r[t]   (:262); with proper time
                 * limits (:208-215): (..)*bad[t+1] + (1-bad[t+1])*D(v[t]) */
                volatile float x = returns[o + C] * g32;
                volatile float y = x * m1;
                volatile float z = y + r;
                if (ptl) {
                    const float dv0 = denorm(value_preds[o], sigma, mu, dn);
                    volatile float zb = z * b1;
                    volatile float ob = 1.0f - b1;
                    volatile float w = ob * dv0;
                    returns[o] = zb + w;
                } else {
                    returns[o] = z;
                }
            }
        }
    }
}

/* numpy's float32 add.reduce over a contiguous axis (pairwise summation,
 * numpy/core/src/umath/loops_utils.h.src: <8 plain loop, <=128 eight partial sums, else halves). */
static float np_pairwise_sum_f32(const float* a, int64_t n) {
    if (n < 8) {
        volatile float res = 0.0f;
        for (int64_t i = 0; i < n; ++i) res = res + a[i];
        return res;
    }
    if (n <= 128) {
        volatile float r[8];
        int64_t i;
        for (i = 0; i < 8; ++i) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
        volatile float p01 = r[0] + r[1], p23 = r[2] + r[3], p45 = r[4] + r[5], p67 = r[6] + r[7];
        volatile float q0 = p01 + p23, q1 = p45 + p67;
        volatile float res = q0 + q1;
        for (; i < n; ++i) res = res + a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    volatile float lo = np_pairwise_sum_f32(a, n2), hi = np_pairwise_sum_f32(a + n2, n - n2);
    return lo + hi;
}

/*
 * The "mat" / "mat_dec" branches of SharedReplayBuffer.compute_returns
 * (onpolicy/utils/shared_buffer.py:222-232 with a value normaliser, :241-251 without); columns are
 * ordered (thread, agent) with A agents.  advantages[t] = gae.
 */
void orc_compute_returns_mat(const float* rewards, float* value_preds, const float* next_value,
                             const float* masks, float* returns, float* advantages, float sigma,
                             float mu, int T, int64_t C, int A, double gamma, double lam, int dn) {
    const float g32 = (float)gamma;
    const float gl32 = (float)(gamma * lam);
    memcpy(value_preds + (int64_t)T * C, next_value, (size_t)C * sizeof(float)); /* :218 */
    for (int64_t c = 0; c < C; ++c) {
        const int64_t g0 = (c / A) * A;
        volatile float gae = 0.0f;
        for (int t = T - 1; t >= 0; --t) {
            const int64_t o = (int64_t)t * C + c;
            float next_term, cur_term, base;
            if (dn) { /* :223-224 */
                base = denorm(value_preds[o], sigma, mu, 1);
                cur_term = base;
                next_term = denorm(value_preds[o + C], sigma, mu, 1);
            } else { /* :243-244: np.mean(value_preds[step], axis=-2) in float32 */
                volatile float s0 = np_pairwise_sum_f32(value_preds + (int64_t)t * C + g0, A);
                volatile float s1 = np_pairwise_sum_f32(value_preds + (int64_t)(t + 1) * C + g0, A);
                cur_term = s0 / (float)A;
                next_term = s1 / (float)A;
                base = value_preds[o];
            }
            /* delta = r + gamma * mask * next - cur   (:229, :245) */
            volatile float x = g32 * masks[o + C];
            volatile float y = x * next_term;
            volatile float z = rewards[o] + y;
            volatile float delta = z - cur_term;
            volatile float u = gl32 * masks[o + C]; /* :230, :249 */
            volatile float carry = u * gae;
            gae = delta + carry;
            advantages[o] = gae;      /* :231, :250 */
            returns[o] = gae + base;  /* :232, :251 */
        }
    }
}

/* onpolicy/algorithms/r_mappo/r_mappo.py:179-182: advantages = returns[:-1] - D(value_preds[:-1]) */
void orc_advantages(const float* returns, const float* value_preds, float* adv, float sigma,
                    float mu, int denorm_on, int64_t n) {
    for (int64_t i = 0; i < n; ++i) adv[i] = returns[i] - denorm(value_preds[i], sigma, mu, denorm_on);
}

/*
 * r_mappo.py:183-186: entries whose active mask is exactly 0 become NaN and are skipped by
 * np.nanmean / np.nanstd (population std).  Accumulated in float64 here (numpy's float32
 * pairwise sums differ from this in the last float32 bits; tests use a tolerance for these
 * two scalars and bit-exactness for everything else).  out = {mean, std, count}.
 */
void orc_adv_moments(const float* adv, const float* active, int64_t n, double* out) {
    double s1 = 0.0, cnt = 0.0;
    for (int64_t i = 0; i < n; ++i)
        if (!active || active[i] != 0.0f) { s1 += adv[i]; cnt += 1.0; }
    double mean = s1 / cnt, s2 = 0.0;
    for (int64_t i = 0; i < n; ++i)
        if (!active || active[i] != 0.0f) { double d = (double)adv[i] - mean; s2 += d * d; }
    out[0] = mean;
    out[1] = sqrt(s2 / cnt);
    out[2] = cnt;
}

/* r_mappo.py:187: (advantages - mean) / (std + 1e-5), float32 with float32 mean/std */
void orc_adv_normalize(const float* adv, float mean, float std, float* out, int64_t n) {
    volatile float den = std + 1e-5f;
    for (int64_t i = 0; i < n; ++i) {
        volatile float d = adv[i] - mean;
        out[i] = d / den;
    }
}

/* shared_buffer.py:363-396: out[j, :] = field.reshape(-1, D)[idx[j], :] */
void orc_gather_rows(const float* src, const int64_t* idx, int64_t mb, int D, float* dst) {
    for (int64_t j = 0; j < mb; ++j)
        memcpy(dst + j * D, src + idx[j] * (int64_t)D, (size_t)D * sizeof(float));
}

/*
 * shared_buffer.py:499-608 recurrent_generator for one field of row width D.
 *   _cast (:11-12): x[T,N,A,D].transpose(1,2,0,3).reshape(-1, D)  -> rows in (n, a, t) order
 *   chunk c = rows [c*L, (c+1)*L) of that array (:554-566)
 *   np.stack(axis=1) -> [L, mb, D], _flatten -> row l*mb + j (:574-604)
 * first_only: the RNN-state fields keep only the chunk's first row -> [mb, D] (:568-569,:588-589).
 * Implemented literally: materialise the cast copy, then slice.
 */
void orc_gather_chunks(const float* src, const int64_t* idx, int64_t mb, int L, int T, int64_t N,
                       int A, int D, int first_only, float* dst) {
    const int64_t rows = (int64_t)T * N * A;
    float* cast = (float*)malloc((size_t)rows * D * sizeof(float));
    for (int64_t n = 0; n < N; ++n)
        for (int a = 0; a < A; ++a)
            for (int t = 0; t < T; ++t)
                memcpy(cast + ((n * A + a) * T + t) * D, src + (((int64_t)t * N + n) * A + a) * D,
                       (size_t)D * sizeof(float));
    for (int64_t j = 0; j < mb; ++j) {
        const int64_t ind = idx[j] * L; /* :554 */
        if (first_only) {
            memcpy(dst + j * D, cast + ind * D, (size_t)D * sizeof(float));
        } else {
            for (int l = 0; l < L; ++l)
                memcpy(dst + ((int64_t)l * mb + j) * D, cast + (ind + l) * D, (size_t)D * sizeof(float));
        }
    }
    free(cast);
}
