/*
 * mappo_hip.h -- C ABI of libmappo_hip.so: the MI355X (gfx950) device side of the
 * MAPPO rollout-buffer hot path (GAE scan, advantage moments, minibatch gathers,
 * slab writes).
 *
 * The reference (marlbenchmark/on-policy) has no FFI for this path: it is numpy
 * code inside onpolicy/utils/shared_buffer.py and onpolicy/algorithms/r_mappo/r_mappo.py.
 * Each entry point below names the reference lines it replaces.  A maintainer binds
 * them with ctypes; no torch / C++ types cross this boundary.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - all payload is float32, C-contiguous, time-major: element (t, n, a, d) of a
 *     field with row width D lives at ((t*N + n)*A + a)*D + d; "C" below is the
 *     number of columns N*A of the [T(+1), C] scalar fields;
 *   - the library borrows pointers for the duration of the call and allocates
 *     nothing persistent; work is enqueued on `stream` (a hipStream_t) and the call
 *     returns without synchronising;
 *   - return value: 0 on success, a negative MAPPO_E_* for argument errors (nothing
 *     enqueued), or a positive hipError_t from the launch.
 */
#ifndef MAPPO_HIP_H
#define MAPPO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mappo_stream_t; /* hipStream_t; NULL = the null stream */

#define MAPPO_ABI_VERSION 2

/* argument errors */
#define MAPPO_E_NULL      (-1) /* a required pointer is NULL                */
#define MAPPO_E_SHAPE     (-2) /* a size is <= 0 or inconsistent            */
#define MAPPO_E_FLAGS     (-3) /* unsupported flag combination              */
#define MAPPO_E_TOO_MANY  (-4) /* more fields / slabs than MAPPO_MAX_FIELDS */
#define MAPPO_E_ALIGN     (-5) /* a pointer is not 4-byte aligned           */

/* Arithmetic of the matrix products of K9 / K12 (field `arith` of mappo_mlp_t / mappo_gru_seq_t; a PER-CALL choice, nothing
 * process-wide: two trainers in one process may differ).  Inputs, outputs, accumulation and every non-matrix operation are
 * float32 in both forms.
 *   MAPPO_ARITH_SIX_TERM (0, what a zero-initialised struct selects): wherever a kernel of that form exists -- the forward of
 *     two-layer trunks with aligned inputs, the backward chain of two-layer trunks, the direct first-layer weight-gradient
 *     kernel (aligned inputs wider than 192 floats), both directions of K12 -- every float32 product x y is formed on the
 *     bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16 x the rate of the float32 instruction) from the operands' EXACT
 *     three-way bf16 splits x = x1 + x2 + x3 (8 + 8 + 8 significand bits, round-to-nearest, residuals exact):
 *     x y ~ x1 y1 + (x1 y2 + x2 y1) + (x2 y2 + x1 y3 + x3 y1), six products that are exact in float32, accumulated in
 *     float32 smallest first; the three dropped terms are < 2^-23 |x y| together (profiles/r04_probe_bf16_split.json, r05
 *     adversarial tests: the error against float64 is of the float32 MFMA chain's order, bounded by a few 2^-23 sum|x||y|).
 *     Shapes without such a kernel (single- / three-layer trunks, unaligned or very narrow weight-gradient shapes) run the
 *     float32-MFMA kernels under either value.  NON-FINITE AND OUT-OF-RANGE OPERANDS: an operand that is +-inf, NaN or so
 *     large that its bf16 rounding overflows (|x| >= 3.3961e38) makes every output it reaches NaN (inf - inf in the split)
 *     -- a superset of where the float32 form is non-finite (which yields +-inf / NaN, or, behind a saturating Tanh on raw
 *     inputs, finite values); a non-finite loss / gradient norm poisons the update identically in both forms
 *     (mappo_clip_adam).  Float32 subnormal operands are split inexactly (bf16 keeps 7 of their bits per part): absolute
 *     error <= 2^-133 |y| per product.
 *   MAPPO_ARITH_F32_MFMA (1): v_mfma_f32_32x32x2_f32 everywhere (exact float32 fma chains). */
#define MAPPO_ARITH_SIX_TERM 0
#define MAPPO_ARITH_F32_MFMA 1

/* ---------------------------------------------------------------- K1: GAE ----
 * Replaces SharedReplayBuffer.compute_returns
 *   (onpolicy/utils/shared_buffer.py:179-262) and, when `advantages` is given, the
 *   first line of R_MAPPO.train (onpolicy/algorithms/r_mappo/r_mappo.py:179-182).
 *
 * flags select the reference branch:
 *   USE_GAE              args.use_gae                 (shared_buffer.py:186,217)
 *   PROPER_TIME_LIMITS   args.use_proper_time_limits  (shared_buffer.py:185)
 *   DENORM               use_popart or use_valuenorm: D(x) = x*sigma + mu with
 *                        (sigma, mu) = denorm[0..1]    (valuenorm.py:68-79,
 *                        popart.py:88-98); without it D is the identity.
 * gamma and gae_lambda are the Python floats (float64) of args.gamma /
 * args.gae_lambda: the reference multiplies gamma*gae_lambda in float64 and only then
 * rounds to float32 (shared_buffer.py:239), and so does this call.
 * Arithmetic is float32 in exactly the reference's operation order, with no FMA
 * contraction, so results are bit-identical to the numpy path given the same
 * (sigma, mu) -- with one exception: narrow buffers in the GAE modes (2048 <= C < 16384
 * columns, 64 <= T <= 416) run a TIME-PARALLEL scan (the recurrence is affine in the
 * accumulator, T is cut into 16 segments that are folded independently and stitched by
 * composing their affine maps; the serial walk otherwise leaves such buffers at < 10 % of
 * HBM bandwidth).  Inside a segment the operations are the reference's, the 15 segment
 * boundaries carry a few ulp of re-association error: results agree to ~1e-6 relative, the
 * "fp32 tolerance" of BASELINE.json's north star.  MAPPO_GAE_EXACT forces the bit-exact
 * kernels for every shape.
 *
 *   rewards      [T,   C]  read
 *   value_preds  [T+1, C]  read; row T is first overwritten with next_value in the
 *                          GAE modes (shared_buffer.py:187,218)
 *   next_value   [C]       read
 *   masks        [T+1, C]  read (rows 1..T)
 *   bad_masks    [T+1, C]  read (rows 1..T); required iff PROPER_TIME_LIMITS
 *   returns      [T+1, C]  rows 0..T-1 written; row T is written (= next_value)
 *                          only in the non-GAE modes (shared_buffer.py:205,260)
 *   denorm       [2]       {sigma, mu}; required iff DENORM
 * Optional fused epilogue (all three NULL to skip):
 *   advantages   [T, C]    advantages[t] = fl(returns[t] - D(value_preds[t]))
 *                          (r_mappo.py:179-182: computed from the ROUNDED returns)
 *   active_masks [T+1, C]  rows 0..T-1 select the entries that count towards the
 *                          moments (r_mappo.py:184: active_masks[:-1] == 0 -> NaN);
 *                          NULL = every entry counts
 *   adv_partials [mappo_gae_partial_rows(C), 3] float64; per-workgroup partial
 *                          {sum adv, sum adv^2, count} over counted entries; rows
 *                          that no workgroup owns are zero-filled by this call.
 */
#define MAPPO_GAE_USE_GAE            1u
#define MAPPO_GAE_PROPER_TIME_LIMITS 2u
#define MAPPO_GAE_EXACT              8u   /* never take the time-parallel scan (see below): bit-identical results */
#define MAPPO_GAE_DENORM             4u

int mappo_gae_f32(const float* rewards, float* value_preds, const float* next_value,
                  const float* masks, const float* bad_masks, float* returns,
                  const float* denorm, float* advantages, const float* active_masks,
                  double* adv_partials, int T, int64_t C, double gamma, double gae_lambda,
                  unsigned flags, mappo_stream_t stream);

/* Multi-agent-transformer branches of SharedReplayBuffer.compute_returns (shared_buffer.py:222-232
 * with a value normaliser, :241-251 without; taken when args.algorithm_name is "mat" / "mat_dec",
 * use_gae is set and use_proper_time_limits is not).  Same storage contract as mappo_gae_f32 with
 * C = n_rollout_threads * num_agents columns ordered (thread, agent); differences:
 *   advantages [T, C]  REQUIRED: advantages[t] = the GAE accumulator itself (:231,:250), which is
 *                      what MATTrainer.train normalises (mat_trainer.py:160-164);
 *   flags              MAPPO_GAE_DENORM: delta uses each agent's own D(v) (:223-229);
 *                      0: delta uses the float32 mean over the env's num_agents agents of the raw
 *                      value predictions, summed in numpy's pairwise order (:243-245), and
 *                      returns[t] = gae + value_preds[t] of the agent itself (:251);
 *                      num_agents <= 128 in this mode (MAPPO_E_TOO_MANY otherwise).
 * adv_partials / active_masks as in mappo_gae_f32 (moments of the advantages written). */
int mappo_gae_mat_f32(const float* rewards, float* value_preds, const float* next_value,
                      const float* masks, float* returns, const float* denorm, float* advantages,
                      const float* active_masks, double* adv_partials, int T, int64_t C,
                      int num_agents, double gamma, double gae_lambda, unsigned flags,
                      mappo_stream_t stream);

/* Number of [.,3] float64 rows mappo_gae_f32 / mappo_advantages_f32 need in adv_partials
 * for C columns. */
int64_t mappo_gae_partial_rows(int64_t C);

/* Standalone form of the fused epilogue above, for callers whose returns / value_preds /
 * normaliser changed after mappo_gae_f32 ran (r_mappo.py:179-186):
 *   advantages[i] = fl(returns[i] - D(value_preds[i])), i < T*C, plus the same partial
 *   moments. denorm NULL = identity, active_masks NULL = every entry counts. */
int mappo_advantages_f32(const float* returns, const float* value_preds, const float* denorm,
                         const float* active_masks, float* advantages, double* adv_partials,
                         int T, int64_t C, mappo_stream_t stream);

/* Tuning hook for benchmarks: selects the kernel variant used by mappo_gae_f32
 * (0 = automatic). All variants produce bit-identical results. Returns the previous
 * value. */
int mappo_gae_set_variant(int variant);

/* The variant the most recent mappo_gae_f32 call of this process launched (0 before the first call):
 * 70-72 = the time-parallel scan (tolerance mode), 99 = one lane per column, the others the bit-exact
 * strip kernels.  For tests that must know which arithmetic a shape took. */
int mappo_gae_last_variant(void);

/* Measurement hook for benchmarks: mappo_gae_time_next_launch arms one of 64 slots (returned, >= 0; they are reused in a ring)
 * and the NEXT strip / LDS-DMA launch of mappo_gae_f32 in this process (every GAE-mode call with aligned columns; not the
 * one-lane-per-column kernel, not the scan) is bracketed by that slot's HIP event pair at dispatch level
 * (hipExtLaunchKernelGGL: kernel begin / end timestamps, what rocprofv3 --kernel-trace reports), on the stream of that call.
 * mappo_gae_timed_launch_ms(slot) waits for that launch and returns its duration in milliseconds; MAPPO_E_FLAGS if the slot
 * was never armed or its armed call launched a kernel without the hook.  (A pair of hipEventRecord calls around a launch also
 * times two packets of the command processor: + 5-6 us on this 50 us kernel.)  Not thread-safe, like the other hooks. */
int mappo_gae_time_next_launch(void);
int mappo_gae_timed_launch_ms(int slot, float* ms);

/* --------------------------------------------- K5: advantage moments / stats ----
 * Replaces np.nanmean / np.nanstd over the masked advantages
 *   (onpolicy/algorithms/r_mappo/r_mappo.py:183-186).
 * mappo_adv_reduce:  sums[0..2] = column sums of partials[rows,3], fixed order
 *                    (deterministic). In a multi-GPU job the caller all-reduces
 *                    `sums` (3 float64) across ranks between the two calls.
 * mappo_adv_stats:   stats[0] = mean = S1/n, stats[1] = population std
 *                    = sqrt(max(S2/n - mean^2, 0)), both rounded to float32.
 * The normalisation (adv - mean) / (std + 1e-5) (r_mappo.py:187) is applied by the
 * gather kernels (field.normalize) so no standalone pass over [T,C] is needed.
 */
int mappo_adv_reduce(const double* partials, int64_t rows, double* sums, mappo_stream_t stream);
int mappo_adv_stats(const double* sums, float* stats, mappo_stream_t stream);

/* Standalone normalisation pass, for callers that want the normalised advantages
 * materialised (out may alias adv): out = (adv - stats[0]) / (stats[1] + 1e-5f). */
int mappo_adv_normalize(const float* adv, const float* stats, float* out, int64_t n,
                        mappo_stream_t stream);

/* ------------------------------------------------- K3 / K4: minibatch gathers ----
 * One launch copies a minibatch of rows of up to MAPPO_MAX_FIELDS buffer fields.
 *
 * mappo_gather_rows replaces the per-minibatch fancy indexing of
 *   SharedReplayBuffer.feed_forward_generator (shared_buffer.py:363-396):
 *   dst_f[j, :] = src_f[idx[j], :],  j in [0, mb), rows in (t, n, a) order.
 *
 * mappo_gather_chunks replaces recurrent_generator (shared_buffer.py:499-608,
 *   helpers _cast :11-12, _flatten :7-8) and, with L == T, naive_recurrent_generator
 *   (:402-497), reading straight from the time-major buffer (no transposed copy):
 *   output row (l, j) -> flat f = idx[j]*L + l of the (n, a, t)-ordered view,
 *   n = f / (A*T), a = (f / T) % A, t = f % T, source row (t*N + n)*A + a;
 *   dst_f[l*mb + j, :] = src_f[that row, :].  Fields with first_only != 0 (the RNN
 *   states, shared_buffer.py:568-569,588-589) take only l = 0: dst_f[j, :].
 *   A chunk may straddle two (n, a) trajectories when T % L != 0, exactly as the
 *   reference does.
 *
 * fields is a HOST array; idx is a DEVICE int64 array; stats (device, {mean, std})
 * is required iff some field has normalize != 0.
 */
#define MAPPO_MAX_FIELDS 16

typedef struct mappo_field {
    const float* src;   /* field base (row 0), rows of `width` floats          */
    float*       dst;   /* output base, contiguous rows of `width` floats      */
    int32_t      width; /* floats per row (> 0)                                */
    int32_t      first_only; /* chunk gather only: copy the l = 0 row per chunk */
    int32_t      normalize;  /* dst = (src - stats[0]) / (stats[1] + 1e-5f)     */
    int32_t      standardize; /* dst row = (row - mean(row)) / sqrt(var(row) + 1e-5): the parameter-free
                                 part of the input nn.LayerNorm (ref algorithms/utils/mlp.py:47-53),
                                 applied while the row is copied; widths as for mappo_layernorm_fwd */
} mappo_field_t;

int mappo_gather_rows(const mappo_field_t* fields, int n_fields, const int64_t* idx,
                      int64_t mb, const float* stats, mappo_stream_t stream);

int mappo_gather_chunks(const mappo_field_t* fields, int n_fields, const int64_t* idx,
                        int64_t mb, int L, int T, int64_t N, int A, const float* stats,
                        mappo_stream_t stream);

/* Packed records of the NARROW fields.  A 4-byte gather costs a whole HBM sector, so the per-sample
 * scalars of feed_forward_generator / recurrent_generator (actions, value_preds, returns, masks,
 * active_masks, action_log_probs, advantages, and small rows such as available_actions;
 * shared_buffer.py:383-396, 558-566) are first packed into one record per (t, n, a) row:
 *   mappo_pack_records:   records[row, offset_k .. offset_k + width_k) = src_k[row, :], rows in
 *                         (t, n, a) order; record_width a multiple of 4, <= 32; padding zeroed.
 *   mappo_gather_records: dst_k[j, :] = records[source_row(j), offset_k ..] (normalised like
 *                         mappo_field.normalize if asked) -- one record read per sample.
 *                         L = 0: rows mode (source_row(j) = idx[j]); L > 0: chunk mode with the row
 *                         mapping of mappo_gather_chunks (no first_only fields here).
 * Bit-identical to gathering the fields one by one.  `fields` is a HOST array. */
typedef struct mappo_record_field {
    const float* src;    /* pack: field base (rows of `width` floats); unused by gather          */
    float*       dst;    /* gather: output base (rows of `width` floats); unused by pack         */
    int32_t      width;
    int32_t      offset; /* first component of this field inside a record                        */
    int32_t      normalize;
    int32_t      reserved;
} mappo_record_field_t;

int mappo_pack_records(const mappo_record_field_t* fields, int n_fields, float* records,
                       int record_width, int64_t rows, mappo_stream_t stream);
int mappo_gather_records(const float* records, int record_width, const mappo_record_field_t* fields,
                         int n_fields, const int64_t* idx, int64_t mb, int L, int T, int64_t N, int A,
                         const float* stats, mappo_stream_t stream);

/* Tuning hook for benchmarks (results do not depend on it): bits 0-1 log2 of the loads in
 * flight per lane, bit 2 non-temporal accesses, bits 4.. workgroups per CU (0 = occupancy).
 * Returns the previous value. */
int mappo_gather_set_variant(int variant);

/* ------------------------------------------------------------ K2: slab writes ----
 * Replaces the per-field `buf[step] = x.copy()` statements of insert / chooseinsert /
 * after_update / chooseafter_update (shared_buffer.py:107-121,142-156,162-170,
 * 174-177): one launch copies up to MAPPO_MAX_FIELDS contiguous [N, A, D] slabs
 * (device to device).  slabs is a HOST array.
 */
typedef struct mappo_slab {
    const float* src;
    float*       dst;
    int64_t      count; /* floats */
} mappo_slab_t;

int mappo_slab_copy(const mappo_slab_t* slabs, int n_slabs, mappo_stream_t stream);

/* The same launch + the slabs of row-standardised observation copies that have to follow the slabs being written (round 6).
 * Networks with an input LayerNorm (reference onpolicy/algorithms/utils/mlp.py:47-53: feature_norm) read obs / share_obs
 * through a resident copy whose rows hold (x - mean) / sqrt(var + eps) (mappo_standardize_rows_ld; the LayerNorm's affine half
 * is folded into the first Linear); when insert / chooseinsert / after_update (shared_buffer.py:90-177) rewrite a slab of the
 * field, the same slab of the copy is recomputed from the VALUE being written -- by extra workgroups of this launch, in the
 * arithmetic of mappo_standardize_rows_ld (bit-identical rows).  src [rows, D] floats; dst [rows, ld >= D] floats, columns
 * D .. ld - 1 are written as zeros.  std is a HOST array of at most MAPPO_MAX_STD_SLABS entries; n_std = 0 is
 * mappo_slab_copy; n_slabs = 0 (slabs may be NULL) only standardises -- the full pass over a field goes through here too, so
 * that slab and full pass are one piece of code (every operation IEEE, one explicit fused multiply-add per term of the
 * second moment) and bit-identical by construction. */
#define MAPPO_MAX_STD_SLABS 4
typedef struct mappo_std_slab {
    const float* src;
    float*       dst;
    int64_t      rows;
    int          D, ld;
    float        eps;
} mappo_std_slab_t;

int mappo_slab_copy_std(const mappo_slab_t* slabs, int n_slabs, const mappo_std_slab_t* std, int n_std,
                        mappo_stream_t stream);

/* ------------------------------------------------------------ K6: row LayerNorm ----
 * The LayerNorm the reference applies to the observations and after every Linear / GRU of the
 * actor and critic trunks (onpolicy/algorithms/utils/mlp.py:17-22,47-53, rnn.py:22,79:
 * nn.LayerNorm over the last dimension, eps = 1e-5, biased variance), as one streaming kernel per
 * direction for [M, D] float32 rows with D up to 2048 (multiple of 4) / 1536 (any).
 *
 * mappo_layernorm_fwd: y = (x - mean) * rstd * weight + bias; also writes mean[M], rstd[M]
 *                      (saved for the backward).
 * mappo_layernorm_bwd: dx (NULL to skip: input does not require grad), dweight[D], dbias[D] from
 *                      dy, x and the saved statistics.  `partials` is a device workspace of
 *                      2 * mappo_layernorm_max_blocks() * D floats (per-workgroup column sums,
 *                      reduced in a fixed order: deterministic).
 * Returns MAPPO_E_SHAPE for an unsupported D (callers then use the framework's own LayerNorm).
 */
/* mappo_act_layernorm_fwd/_bwd: the same with an elementwise activation fused in front,
 * y = LayerNorm(act(x)), act: 0 = identity, 1 = tanh, 2 = relu -- one pass over the rows instead of
 * two for the `Linear -> act -> LayerNorm` blocks of mlp.py:17-22.  `x` is the PRE-activation in
 * both directions (the backward recomputes act(x)); dx is the gradient w.r.t. the pre-activation.
 */
/* mappo_bias_act_layernorm_fwd/_bwd: additionally the bias add of the preceding Linear,
 * y = LayerNorm(act(x + pre_bias)) with pre_bias [D] (NULL = 0): the GEMM runs without a bias and the
 * backward returns dpre_bias [D] = column sums of dx, i.e. the Linear's bias gradient, for free --
 * otherwise a separate pass over the [M, D] gradient (mlp.py:17-22 `nn.Linear` + `nn.LayerNorm`).
 * `partials` is 3 * mappo_layernorm_max_blocks() * D floats when dpre_bias is requested (D <= 1024,
 * dx required), 2 * ... otherwise. */
int mappo_layernorm_max_blocks(void);
int mappo_bias_act_layernorm_fwd(const float* x, const float* pre_bias, const float* weight, const float* bias,
                                 float* y, float* mean, float* rstd, int64_t M, int D, float eps, int act,
                                 mappo_stream_t stream);
int mappo_bias_act_layernorm_bwd(const float* dy, const float* x, const float* pre_bias, const float* mean,
                                 const float* rstd, const float* weight, float* dx, float* dweight, float* dbias,
                                 float* dpre_bias, float* partials, int64_t M, int D, int act,
                                 mappo_stream_t stream);
int mappo_act_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y, float* mean,
                            float* rstd, int64_t M, int D, float eps, int act, mappo_stream_t stream);
int mappo_act_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                            const float* weight, float* dx, float* dweight, float* dbias, float* partials,
                            int64_t M, int D, int act, mappo_stream_t stream);
int mappo_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y, float* mean,
                        float* rstd, int64_t M, int D, float eps, mappo_stream_t stream);
int mappo_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                        const float* weight, float* dx, float* dweight, float* dbias, float* partials,
                        int64_t M, int D, mappo_stream_t stream);

/* --------------------------------------------------------------- K7: fused PPO loss ----
 * The clipped-surrogate loss of R_MAPPO.ppo_update for a Discrete action head, value and gradient in
 * one pass over the rows of a minibatch (r_mappo.py:52-89 cal_value_loss, :119-153 policy loss and
 * entropy; act.py:115-170 and distributions.py FixedCategorical for log-prob / entropy of the masked
 * logits).  Per row i (all arrays float32, row-major):
 *   logits [rows, n_actions]  raw outputs of the action head; available [rows, n_actions] (NULL = all
 *   available; entries == 0 take logit -1e10 and receive zero gradient); actions [rows] (index as
 *   float); old_logp, adv, active (NULL = 1), factor (NULL = 1; HAPPO, happo_trainer.py:137-141),
 *   values, value_preds, returns [rows]; norm = {sigma, mu} of the value normaliser, the layout of
 *   mappo_gae_f32's denorm: target = (returns - mu) / sigma (NULL = raw returns); inv_denoms = {1 / D_policy, 1 / D_value} device scalars -- the denominators of the
 *   masked means (sum of active masks, or the row count) over the WHOLE minibatch (all spans, all
 *   data-parallel ranks), so that the gradients of successive spans simply accumulate.
 * Outputs: dlogits [rows, n_actions] = d(policy_loss - entropy_coef * entropy) / dlogits,
 *          dvalues [rows] = d(value_loss_coef * value_loss) / dvalues (either may be NULL),
 *          sums[4] += { sum w_p * (-surrogate), sum w_p * entropy, sum w_v * value_loss, sum ratio }
 *          (float64, atomically accumulated: multiply by inv_denoms for the logged means).
 * logits == NULL skips the actor half, values == NULL the critic half. */
#define MAPPO_LOSS_HUBER               1u  /* args.use_huber_loss (else mse) */
#define MAPPO_LOSS_CLIPPED_VALUE       2u  /* args.use_clipped_value_loss */
#define MAPPO_LOSS_POLICY_ACTIVE_MASKS 4u  /* args.use_policy_active_masks */
#define MAPPO_LOSS_VALUE_ACTIVE_MASKS  8u  /* args.use_value_active_masks */
typedef struct mappo_ppo_loss {
    const float* logits;
    const float* available;
    const float* actions;
    const float* old_logp;
    const float* adv;
    const float* active;
    const float* factor;
    const float* values;
    const float* value_preds;
    const float* returns;
    const float* norm;
    const float* inv_denoms;
    float* dlogits;
    float* dvalues;
    double* sums;
    int64_t rows;
    int n_actions;
    float clip;
    float huber_delta;
    float entropy_coef;
    float value_loss_coef;
    unsigned flags;
} mappo_ppo_loss_t;
int mappo_ppo_loss_f32(const mappo_ppo_loss_t* args, mappo_stream_t stream);

/* ------------------------------------------------------------------ K14: categorical sampling for the rollout ----
 * Replaces, for one Discrete head and one draw per row, Categorical.forward's availability masking
 * (onpolicy/algorithms/utils/distributions.py:64-67), FixedCategorical.sample (:15-16) and .log_probs (:18-25) as
 * called by ACTLayer.forward (onpolicy/algorithms/utils/act.py:55-60):
 *   x_i = available_i == 0 ? -1e10 : logits_i;  l = x - logsumexp(x);  action = argmax_i exp(l_i) / noise_i
 *   (torch.multinomial's rule for one sample: noise ~ Exponential(1), drawn by the caller);  log_probs = l[action].
 *   logits [rows, n_actions], available [rows, n_actions] or NULL, noise [rows, n_actions] (> 0),
 *   actions [rows] int64, log_probs [rows]; n_actions <= 64. */
int mappo_categorical_sample(const float* logits, const float* available, const float* noise, int64_t* actions,
                             float* log_probs, int64_t rows, int n_actions, mappo_stream_t stream);

/* --------------------------------------------------------------- K8: GRU cell gates ----
 * Everything of one GRU step that is not a GEMM (reference onpolicy/algorithms/utils/rnn.py:7-80 runs
 * nn.GRU; PyTorch cell, gates stacked r|z|n), reading / writing the per-sequence buffers in place:
 *   r = sigmoid(gi_r + b_ir + gh_r + b_hr), z = sigmoid(gi_z + b_iz + gh_z + b_hz),
 *   n = tanh(gi_n + b_in + r * (gh_n + b_hn)), h' = n + z * (hm - n)
 * mappo_gru_cell_fwd: gi, gh [B, 3H] (input / hidden projections without bias), hm [B, H] (previous state,
 *   already multiplied by this step's mask), b_ih, b_hh [3H]; writes h_out [B, H] = h', optionally
 *   hm_next [B, H] = h' * mask_next[row] (mask_next [B]: the next step's episode-boundary mask, rnn.py:43-77;
 *   NULL mask = 1) and ws [B, 4H] = {r, z, n, gh_n + b_hn} for the backward.
 * mappo_gru_cell_bwd: g = dout + carry * mask_next (dout [B, H] = d loss / d h' from the layers above, carry
 *   [B, H] = d loss / d hm of the next step, either may be NULL); writes dgi, dgh [B, 3H] (gradients of the
 *   two projections; their column sums are the bias gradients) and dhx [B, H] = g * z, the direct part of
 *   d loss / d hm (the caller adds dgh W_hh). */
int mappo_gru_cell_fwd(const float* gi, const float* gh, const float* hm, const float* b_ih, const float* b_hh,
                       const float* mask_next, float* h_out, float* hm_next, float* ws, int64_t B, int H,
                       mappo_stream_t stream);
int mappo_gru_cell_bwd(const float* dout, const float* carry, const float* mask_next, const float* ws,
                       const float* hm, float* dgi, float* dgh, float* dhx, int64_t B, int H,
                       mappo_stream_t stream);

/* mappo_gru_step_fwd: a whole forward GRU step for H = 64 (MAPPO_E_SHAPE otherwise) -- the hidden projection
 * hm W_hh^T runs on the f32 MFMA inside the kernel with W_hh [3H, H] held in LDS, so the [B, 3H] projection
 * never goes through HBM and no separate skinny GEMM is launched.  Same buffers and semantics as
 * mappo_gru_cell_fwd without `gh`. */
int mappo_gru_step_fwd(const float* gi, const float* hm, const float* w_hh, const float* b_ih, const float* b_hh,
                       const float* mask_next, float* h_out, float* hm_next, float* ws, int64_t B, int H,
                       mappo_stream_t stream);

/* --------------------------------------------------------------- K12: the GRU over a whole chunk ----
 * RNNLayer of the recurrent policies (reference onpolicy/algorithms/utils/rnn.py:7-80: nn.GRU, gates r | z | n, state
 * multiplied by the episode mask before every step, LayerNorm on the outputs) for H = 64, ONE launch per direction over
 * the [L * mb, 64] chunk rows of recurrent_generator (shared_buffer.py:499-608; row l * mb + j = step l of chunk j; the
 * rollout is the case L = 1).  A wave walks the L steps of 32 chunks with the state in registers; both projections of a
 * step run on the matrix cores against weights held in LDS (field `arith`: MAPPO_ARITH_SIX_TERM = bf16 planes of the weights in
 * LDS, the step's input and state split once and reused by the three gates, all six blocks of the backward's transposed products
 * as planes; MAPPO_ARITH_F32_MFMA = float32 operands); gates, mask reset and the output LayerNorm on the accumulators.
 * mappo_gru_seq_forward: x, h0 [mb, 64], masks [L * mb] -> y = LayerNorm(h_l) [L * mb, 64], h_last [mb, 64] (optional).
 *   For the backward (all three or none): gates [mappo_gru_seq_gates_floats(L, mb)] (r, z, n, W_hn hm + b_hn and the
 *   normalised output per row and step, opaque order), hm [L * mb, 64] (the masked previous state of every step),
 *   stats [mappo_gru_seq_stats_floats(L, mb)].
 * mappo_gru_seq_backward (truncated BPTT inside the launch): dy [L * mb, 64] (+ dh_last [mb, 64], optional) -> dx [L * mb, 64], dh0 [mb, 64] (optional),
 *   dgi [L * mb, 192] = gradient at W_ih x + b_ih, dq [L * mb, 64] = gradient at W_hn hm + b_hn (the r and z thirds of the
 *   hidden side's gradient are dgi's) -- the caller forms dW_ih = dgi^T x and dW_hh = [dgi_rz | dq]^T hm from them -- and
 *   ln_grads [800] = LayerNorm weight | bias gradients [128], then the column sums of dgi [192] (= db_ih; its first 128 are
 *   also the r and z thirds of db_hh) and of dq [64] (= the n third of db_hh), folded over the rows inside the launch
 *   (then, with a head of <= 6 outputs, its sums: see head_w below);
 *   workspace [mappo_gru_seq_workspace_floats()] scratch.  Deterministic run to run. */
typedef struct mappo_gru_seq {
    const float* x;
    const float* h0;
    const float* masks;
    const float* w_ih;      /* [192, 64] */
    const float* w_hh;      /* [192, 64] */
    const float* b_ih;      /* [192] */
    const float* b_hh;      /* [192] */
    const float* ln_g;      /* [64] */
    const float* ln_b;      /* [64] */
    float ln_eps;
    int32_t H;              /* 64 */
    int32_t L;
    int32_t arith;          /* MAPPO_ARITH_* (occupies what was padding in ABI version 1: the layout is unchanged) */
    int64_t mb;
    float* y;
    float* h_last;
    float* gates;
    float* hm;
    float* stats;
    const float* dy;
    float* dx;
    float* dgi;
    float* dq;
    float* dh0;
    const float* dh_last;   /* [mb, 64] gradient at h_last, or NULL */
    float* ln_grads;        /* [800], see above */
    float* workspace;
    /* Optional output Linear on y inside the launches (the Categorical head's Linear, distributions.py:55-68, or the
     * critic's v_out, r_actor_critic.py:147-175): head_out > 0 makes the forward also write logits [L * mb, head_out] =
     * y head_w^T + head_b (y may then be NULL), and the backward (head_out <= 18) form dy = dlogits head_w itself from
     * dlogits [L * mb, head_out] -- dy is not read.  The head's own gradients: with head_sums = 1 (head_out <= 6) the backward also leaves
     * ln_grads[384 + 64 o + f] = sum over rows of dlogits[o] n^[f] (n^ = the normalised output, y = n^ ln_g + ln_b) and
     * ln_grads[768 + o] = sum of dlogits[o], from which dW_h = ln_g (.) GH + ln_b (x) db_h, db_h = the second sums -- y is
     * never needed; wider heads: dlogits^T y and the column sums of dlogits, formed by the caller from y. */
    const float* head_w;    /* [head_out, 64] or NULL */
    const float* head_b;    /* [head_out] */
    int32_t head_out;       /* 0: no head */
    float* logits;
    const float* dlogits;
    int32_t head_sums;      /* backward: 1 = leave the head's gradient sums in ln_grads (head_out <= 6) */
} mappo_gru_seq_t;
int64_t mappo_gru_seq_gates_floats(int L, int64_t mb);
int64_t mappo_gru_seq_stats_floats(int L, int64_t mb);
int64_t mappo_gru_seq_workspace_floats(void);
int     mappo_gru_seq_forward(const mappo_gru_seq_t* seq, mappo_stream_t stream);
int     mappo_gru_seq_backward(const mappo_gru_seq_t* seq, mappo_stream_t stream);
/* The weight gradients of that GRU from what mappo_gru_seq_backward left (reference: autograd through nn.GRU,
 * onpolicy/algorithms/utils/rnn.py:24-80), ONE launch + a reduction, float32 products from six bf16 x bf16 terms
 * (MAPPO_ARITH_SIX_TERM; the float32-MFMA route forms them with library GEMMs on the host side):
 *   dw[0 .. 192 * 64)          = dW_ih = dgi^T x                      dgi [rows, 192], x [rows, 64] (the GRU's inputs)
 *   dw[192 * 64 .. 2 * 192 * 64) = dW_hh = [dgi_r | dgi_z | dq]^T hm    dq [rows, 64], hm [rows, 64] (mask * h_{t-1})
 * rows = L * mb; all pointers 16-byte aligned; workspace [mappo_gru_weight_grads_workspace_floats()] scratch.  Deterministic. */
int64_t mappo_gru_weight_grads_workspace_floats(void);
int     mappo_gru_weight_grads(const float* dgi, const float* dq, const float* x, const float* hm, int64_t rows, float* dw,
                               float* workspace, mappo_stream_t stream);

/* --------------------------------------------------------------- K13: gradient clipping + Adam of one network ----
 * What the reference does between backward() and the next minibatch for each network (r_mappo.py:146-167:
 * nn.utils.clip_grad_norm_(parameters, max_grad_norm) or get_gard_norm, then optimizer.step(); the optimiser is
 * torch.optim.Adam(lr, eps = opti_eps, weight_decay), rMAPPOPolicy.py:31-37) as two launches over all tensors of the
 * network: total L2 norm of the gradients -> grad_norm [1] (before clipping), gradients scaled in place by
 * min(1, max_grad_norm / (norm + 1e-6)) (max_grad_norm <= 0: no clipping), then Adam without amsgrad on the optimiser's
 * own state tensors: step[t] (float32 scalar, incremented), exp_avg[t], exp_avg_sq[t], param[t], all float32 and
 * contiguous.  workspace [mappo_adam_workspace_floats()]. */
#define MAPPO_ADAM_MAX_TENSORS 64
typedef struct mappo_adam {
    float*  param[MAPPO_ADAM_MAX_TENSORS];
    float*  grad[MAPPO_ADAM_MAX_TENSORS];
    float*  exp_avg[MAPPO_ADAM_MAX_TENSORS];
    float*  exp_avg_sq[MAPPO_ADAM_MAX_TENSORS];
    float*  step[MAPPO_ADAM_MAX_TENSORS];
    int64_t numel[MAPPO_ADAM_MAX_TENSORS];
    int32_t n;
    double  lr, beta1, beta2, eps, weight_decay, max_grad_norm;     /* doubles, as torch.optim.Adam holds them */
    float*  grad_norm;
    float*  workspace;
    const double* lr_device;    /* optional: the learning rate is read from this device double instead of `lr` -- a launch that
                                   was captured into a HIP graph then follows lr_decay (rMAPPOPolicy.py:39-46) between replays */
} mappo_adam_t;
int64_t mappo_adam_workspace_floats(void);
int     mappo_clip_adam(const mappo_adam_t* adam, mappo_stream_t stream);

/* ValueNorm.update + running_mean_var as two launches (reference onpolicy/utils/valuenorm.py:32-55): the mean and mean of
 * squares of x [n] (or batch_moments [2] = {mean, mean_sq} when the caller all-reduced them over the ranks; x is then
 * ignored) are folded into the scalar statistics in place, m <- weight * m + (1 - weight) * E[.], debiasing_term <-
 * weight * d + (1 - weight), and denorm [2] receives {sigma, mu} = {sqrt(max(m2 / max(d, eps) - mu^2, 1e-2)), m1 / max(d,
 * eps)} -- what the loss normalises returns with and the GAE scan de-normalises values with.
 * workspace [mappo_valuenorm_workspace_doubles()] float64. */
int64_t mappo_valuenorm_workspace_doubles(void);
int     mappo_valuenorm_update(const float* x, int64_t n, const float* batch_moments, double weight, float eps,
                               float* running_mean, float* running_mean_sq, float* debiasing_term, float* denorm,
                               double* workspace, mappo_stream_t stream);

/* What a minibatch needs before its loss is formed (reference onpolicy/algorithms/r_mappo/r_mappo.py:135-139 and :84-87: the
 * masked means divide by active_masks.sum(), the unmasked ones by the row count; :65 feeds ValueNorm / PopArt with the batch
 * moments of the returns), three launches instead of ~13 PyTorch ones (27 with the data-parallel bookkeeping).
 * mappo_minibatch_sums: active_masks [n], returns [n] -> sums [4] float64 = {sum active_masks, n, sum returns,
 *   sum returns^2} (fixed summation order).  A data-parallel caller all-reduces a copy of them over the ranks.
 * mappo_minibatch_scales: local / global sums (the same array for one rank) -> out [8] float32 =
 *   {1 / global policy denominator, 1 / global value denominator, 1 / local policy denominator (twice), 1 / local value
 *   denominator, 1 / local rows, mean and mean of squares of the returns over the global minibatch}; a denominator is the
 *   active-mask sum when the loss is masked (use_policy_active_masks / use_value_active_masks), else the row count.
 *   out[0..1] scale the fused loss (K7: `inv`), out[2..5] turn its four sums into the logged means, out[6..7] are
 *   mappo_valuenorm_update's batch_moments. */
int64_t mappo_minibatch_sums_workspace_doubles(void);
int     mappo_minibatch_sums(const float* active_masks, const float* returns, int64_t n, double* sums, double* workspace,
                             mappo_stream_t stream);
int     mappo_minibatch_scales(const double* local_sums, const double* global_sums, int policy_masked, int value_masked,
                               float* out, mappo_stream_t stream);

/* --------------------------------------------------------------- K10: sort-free minibatch index lists ----
 * Device-side replacement of `rand = torch.randperm(B); slices = [rand[i*mb:(i+1)*mb] for i in range(n_mb)]`
 * (reference onpolicy/utils/shared_buffer.py:360-361 feed-forward, :415-416 whole trajectories, :511-512 chunks).
 * Sample r belongs to slice perm(r) / mb, perm = a keyed bijection of [0, n) (6-round balanced Feistel network with
 * cycle walking, keys[0..5] = 32-bit round keys drawn by the caller); samples with perm(r) >= n_mb * mb are dropped,
 * as the reference drops the permutation's tail.  idx [n_mb * mb] int64 receives the slices back to back, every
 * slice in ASCENDING sample order (a minibatch is a set: its loss does not depend on row order, and ascending order turns
 * the gathers into monotonic walks over the buffer).  No sort, no n-sized temporary: workspace
 * [mappo_minibatch_workspace_ints(n, n_mb)] int32.  n_mb <= MAPPO_PERM_MAX_MINIBATCHES.  The integer-parity mode of the
 * samplers (--sampler_rng host) does not use this: it uploads the reference's CPU permutation unchanged. */
#define MAPPO_PERM_MAX_MINIBATCHES 64
int64_t mappo_minibatch_workspace_ints(int64_t n, int n_mb);
int     mappo_minibatch_indices(int64_t n, int64_t mb, int n_mb, const uint32_t* keys, int64_t* idx, int32_t* workspace,
                                mappo_stream_t stream);

/* --------------------------------------------------------------- K9: fused hidden-64 trunk ----
 * The actor / critic network of the update as three kernels instead of ~25 launches per minibatch span: what
 * MLPBase + the output Linear compute (reference onpolicy/algorithms/utils/mlp.py:6-58: [LayerNorm(obs)] ->
 * (Linear -> Tanh | ReLU -> LayerNorm) x (1 + layer_N); act.py / distributions.py:55-68 Categorical head and
 * r_actor_critic.py:147-175 v_out are plain Linears on the trunk's features), evaluated straight from the rollout
 * buffer: the rows are read through the sampler's index list (shared_buffer.py:379-396 rows mode, :554-604 chunk
 * mode), so the gathered [mb, obs_dim] minibatch of feed_forward_generator / recurrent_generator is never written.
 * hidden_size must be 64.  The matrix products run in the arithmetic the call's `arith` field names (MAPPO_ARITH_* above).
 *
 * Rows: the caller resolves the sampler's row map once per minibatch into a row table (mappo_mlp_row_table): the source
 * row (int32) of every launch row, mappo_mlp_row_table_ints(rows) = rows rounded up to 128 entries (padding entries repeat
 * the last row, so no kernel clamps a row index).  The kernels then issue coalesced table loads instead of a dependent
 * idx chain and do no index arithmetic.  Input LayerNorm: pass src = a standardised copy of the observation matrix
 * (mappo_standardize_rows, made once per train() -- the observations do not change during the ppo epochs) and fold the
 * LayerNorm's affine half into the first Linear on the caller's side (w1 = W * gamma, bias[0] = b + W beta); the kernels
 * themselves use the rows as they are (use_feature_normalization = False: src = the field itself).
 *
 *   src        [src_rows, din]   matrix the rows come from (e.g. buffer.share_obs[:-1] viewed [T*N*A, din])
 *   row_tab    see above (int32, 16-byte aligned); din >= 4
 *   w1 [64, din], bias[l] / ln_g[l] / ln_b[l] [64] for l < n_layers (n_layers = 1 + layer_N <= 3), w2[l-1] [64, 64]
 *   act        1 Tanh, 2 ReLU (0 identity)
 *   wh [out, 64], bh [out]   output Linear (out <= 64); out = 0: y receives the trunk's features [rows, 64]
 * mappo_mlp_forward writes y [rows, out] and, when z[l] != NULL, what the backward needs of every layer: z[l] = the
 *   LayerNorm's normalised input (act(.) - mean) * rstd, 64 floats for each of mappo_mlp_row_table_ints(rows) rows (rows
 *   rounded up to 128) in the kernels' own order -- opaque scratch between the two calls: element (row r, feature
 *   32 t + 8 q + 4 h + e) sits at float (r / 32) * 2048 + (4 t + q) * 256 + (32 h + r % 32) * 4 + e -- and ln_stats[l]
 *   = {mean, rstd} per row, room for the same padded row count (both NULL in rollouts).
 * mappo_mlp_backward reads dy [rows, out] (or [rows, 64] for out = 0), z[l] and ln_stats[l]; writes `grads`, the parameter gradients
 *   as one flat array [w1 64*din | per layer: bias 64, ln weight 64, ln bias 64 | per hidden layer: w 64*64 | wh out*64 |
 *   bh out] (mappo_mlp_grad_floats), using dz1 [mappo_mlp_row_table_ints(rows), 64] and workspace [mappo_mlp_workspace_floats] as scratch.
 *   Gradients are plain sums over the rows in a fixed order (deterministic run to run).
 * mappo_mlp_row_table: idx [mb] int64 sampler indices (NULL: launch row r = source row r);
 *   chunk_len 0: rows mode (rows = mb, row r <- idx[r]); L > 0: chunk mode, rows = L * mb, row l * mb + j <- element
 *   idx[j] * L + l of the (n, a, t)-ordered sequence (needs T, N, A). */
#define MAPPO_MLP_MAX_LAYERS 3
typedef struct mappo_mlp {
    const float* src;
    const int32_t* row_tab;
    int64_t rows;
    int32_t din, n_layers, act, out;
    float ln_eps;
    int32_t arith;      /* MAPPO_ARITH_* (occupies what was padding in ABI version 1: the layout is unchanged) */
    const float* w1;
    const float* bias[MAPPO_MLP_MAX_LAYERS];
    const float* ln_g[MAPPO_MLP_MAX_LAYERS];
    const float* ln_b[MAPPO_MLP_MAX_LAYERS];
    const float* w2[MAPPO_MLP_MAX_LAYERS - 1];
    const float* wh;
    const float* bh;
    float* y;
    float* z[MAPPO_MLP_MAX_LAYERS];
    float* ln_stats[MAPPO_MLP_MAX_LAYERS];
    const float* dy;
    float* dz1;
    float* workspace;
    float* grads;
} mappo_mlp_t;
int64_t mappo_mlp_row_table_ints(int64_t rows);
int     mappo_mlp_row_table(const int64_t* idx, int64_t rows, int64_t mb, int chunk_len, int T, int N, int A,
                            int32_t* row_tab, mappo_stream_t stream);
/* tuning / test hook: at most `cap` workgroups per K9 kernel (0 = default sizing); small caps make every workgroup loop
 * over many tiles */
int     mappo_mlp_set_grid_cap(int cap);
/* tuning hook: device buffer of >= 512 int64 that receives shader-clock stamps of workgroup 0's first pipeline
 * iterations of mappo_mlp_forward (tools/bench_mlp.py --stamps), NULL = off */
int     mappo_mlp_set_debug(long long* buf);
/* tuning / test hook: option bits of the K9 launchers (initial value: environment variable MAPPO_MLP_FLAGS, default 0);
 * returns the previous value, or -1 (nothing changed) for a bit that does not exist.  NONE of them selects arithmetic
 * (that is the per-call `arith` field).  1 = the forward's compute waves keep the default priority; 2 = K15's forward
 * (mappo_linear512_forward) issues its MFMAs in four groups of four feature tiles per step (round 5's form; default: eight
 * groups of two); 8 = K15's weight gradient (mappo_linear512_wgrad): every wave reads and splits all four column tiles of X
 * itself (round 5's form; default: one tile per wave, shared through LDS as bf16 planes); 4 = mappo_mlp_forward
 * keeps the loader / compute kernel (mlp_fwd_kernel) for shapes the version-3 kernel (operands straight from global memory,
 * resident first-layer weights; aligned rows up to 448 floats wide, two or three layers) would take; 32 = the OTHER form of
 * the direct-to-LDS first-layer weight-gradient kernel: under MAPPO_ARITH_F32_MFMA two slots per wave and two workgroups per
 * CU (default: four slots, one workgroup), under MAPPO_ARITH_SIX_TERM four slots and one workgroup (default: two and two);
 * 128 = under MAPPO_ARITH_SIX_TERM the version-4 forward (first layer in six-term form, one wave per SIMD) also for aligned inputs narrower than 128 floats,
 * which by default take version 3 with only the hidden layer in six-term form (tests: every chunk shape of version 4);
 * 256 = under MAPPO_ARITH_SIX_TERM two-layer trunks with aligned inputs of at most 64 columns keep the separate first-layer
 * weight-gradient kernel (mlp_dw1_rows_kernel) instead of accumulating that gradient inside the chain's launch (A/B, tests). */
int     mappo_mlp_set_flags(int flags);
int     mappo_mlp_forward(const mappo_mlp_t* net, mappo_stream_t stream);
int     mappo_mlp_backward(const mappo_mlp_t* net, mappo_stream_t stream);
int64_t mappo_mlp_grad_floats(int din, int n_layers, int out);
int64_t mappo_mlp_workspace_floats(int din, int n_layers, int out);
/* dst[r, :] = (src[r, :] - mean_r) / sqrt(var_r + eps) for every row of a [rows, D] matrix (population variance: the
 * parameter-free half of nn.LayerNorm, reference onpolicy/algorithms/utils/mlp.py:47-48, 56-57) */
int     mappo_standardize_rows(const float* src, int64_t rows, int D, float eps, float* dst, mappo_stream_t stream);
/* ... with the rows of dst `ld` >= D floats apart and zeros in columns D .. ld - 1: a copy padded to a multiple of 4
 * floats lets the trunk kernels take their 16-byte aligned paths for odd observation widths (pass din = ld and a
 * first-layer weight matrix padded with zero columns; the zero products change no bit of the result) */
int     mappo_standardize_rows_ld(const float* src, int64_t rows, int D, float eps, float* dst, int ld,
                                  mappo_stream_t stream);
/* The affine half of that LayerNorm folded into the Linear behind it (reference mlp.py:47-48 feature_norm, :20 fc1):
 * Linear(LayerNorm(x)) = x^ w_folded^T + b_folded with w_folded[f][k] = w[f][k] gamma[k] (rows `ld` >= din floats apart,
 * zero columns beyond din: the padded copy above), b_folded[f] = b[f] + sum_k w[f][k] beta[k].  One launch; the
 * backward turns the gradients at (w_folded [out_features, ld], b_folded) into those of w [out_features, din], gamma,
 * beta [din] (the gradient of b is db_folded itself), one launch. */
int     mappo_fold_input_norm_forward(const float* w, const float* b, const float* gamma, const float* beta, int out_features,
                                      int din, int ld, float* w_folded, float* b_folded, mappo_stream_t stream);
int     mappo_fold_input_norm_backward(const float* w, const float* gamma, const float* beta, const float* dw_folded,
                                       const float* db_folded, int out_features, int din, int ld, float* dw, float* dgamma,
                                       float* dbeta, mappo_stream_t stream);

/* --------------------------------------------------------------- K15: tall Linear layers with 512 outputs ----
 * The GEMMs of the hidden-512 trunks (BASELINE configs[4], Hanabi: reference onpolicy/algorithms/utils/mlp.py:6-58 at
 * --hidden_size 512 --layer_N 2, scripts/train_hanabi_forward.sh:15-17) in MAPPO_ARITH_SIX_TERM arithmetic: float32 in,
 * float32 out, every product from six bf16 x bf16 terms of the operands' exact three-way splits on the bf16 matrix cores,
 * float32 accumulation.  (Callers that want the float32 matrix instruction use the library GEMM: torch.nn.functional.linear.)
 *   mappo_linear512_prepare:  the three bf16 planes of W [512, K] (rows ldw floats apart) -- or, transposed = 1, of W^T for a
 *     W whose element (k, feature) is w[k * ldw + feature] (the input gradient of a 512 -> 512 Linear: dX = dY W) -- in the
 *     order the forward reads them; planes [mappo_linear512_planes_floats(K)] floats, 16-byte aligned.  Once per weight value.
 *   mappo_linear512_forward:  y [rows, 512] = x [rows, K] W^T (+ bias [512] unless NULL); x rows ldx >= K floats apart
 *     (columns K .. ldx - 1 must hold finite values: the zero padding of mappo_standardize_rows_ld); rows that start on 16-byte
 *     boundaries (ldx a multiple of 4) are loaded in 16-byte pieces, any other ldx (Hanabi's 1285 / 1385-wide gathered
 *     minibatches) in 4-byte pieces.  planes, y and bias must be 16-byte aligned (MAPPO_E_ALIGN otherwise).
 *   mappo_linear512_forward_norm:  a whole block Sequential(Linear, ReLU, LayerNorm) of the reference's MLPLayer (mlp.py:17-22)
 *     in the forward's launch: y [rows, 512] = x W^T + bias (the pre-activation, what mappo_bias_act_layernorm_bwd reads with
 *     a zero pre_bias), yn [rows, 512] = gamma * (relu(y) - mean) * rstd + beta, mean / rstd [rows] the statistics of
 *     relu(y) over the 512 features (biased variance, rstd = 1 / sqrt(var + eps): the arithmetic of
 *     mappo_bias_act_layernorm_fwd); act must be 2 = relu (the codes of mappo_bias_act_layernorm_fwd; MAPPO_E_FLAGS otherwise); bias, gamma, beta, planes, y,
 *     yn 16-byte aligned.  Replaces mappo_linear512_forward + mappo_bias_act_layernorm_fwd: the pre-activation is written
 *     once and not read back.
 *   mappo_linear512_wgrad:    dw [512, K] = dy^T [512, rows] x [rows, K] (the weight gradient of y = x W^T; the contraction
 *     runs over the rows, partial sums of row ranges are added in a fixed order: deterministic run to run);
 *     workspace [mappo_linear512_wgrad_workspace_floats(K)] floats. */
int64_t mappo_linear512_planes_floats(int K);
int     mappo_linear512_prepare(const float* w, int K, int ldw, int transposed, float* planes, mappo_stream_t stream);
int     mappo_linear512_forward(const float* x, int64_t rows, int K, int ldx, const float* planes, const float* bias,
                                float* y, mappo_stream_t stream);
int     mappo_linear512_forward_norm(const float* x, int64_t rows, int K, int ldx, const float* planes, const float* bias,
                                     const float* gamma, const float* beta, float eps, int act, float* y, float* yn,
                                     float* mean, float* rstd, mappo_stream_t stream);
int64_t mappo_linear512_wgrad_workspace_floats(int K);
int     mappo_linear512_wgrad(const float* dy, const float* x, int64_t rows, int K, int ldx, float* dw, float* workspace,
                              mappo_stream_t stream);

/* --------------------------------------------------------------- K11: simple_spread worlds on the device ----
 * One env step of `n_worlds` cooperative-navigation worlds (the env of BASELINE.json configs[0] / configs[2]; reference
 * onpolicy/envs/mpe/core.py:120-190, environment.py:100-180, scenarios/simple_spread.py:60-103) as one launch, so that
 * a rollout step with the policy, the env and the buffer on the GPU moves nothing over PCIe (scope row f1).  State is
 * float64 and updated in place: pos / vel [n, A, 2], landmarks [n, L, 2], t [n] int64.  actions [n, A] int64 in 0..4.
 * Worlds whose t reaches world_length report done and (auto_reset) restart from fresh_pos [n, A, 2] / fresh_landmarks
 * [n, L, 2] (uniform(-1, 1) draws for every world, used where needed).  Outputs: obs [n, A, 4 + 2 L + 4 (A - 1)]
 * float32 (of the restarted world where one restarted), rewards [n, A, 1] float32 (shared reward), dones [n, A] bytes,
 * per_agent [n, A] float64 (the individual_reward info).  A, L <= MAPPO_ENV_MAX_ENTITIES. */
#define MAPPO_ENV_MAX_ENTITIES 16
int mappo_simple_spread_step(double* pos, double* vel, double* landmarks, int64_t* t, const int64_t* actions,
                             const double* fresh_pos, const double* fresh_landmarks, float* obs, float* rewards,
                             uint8_t* dones, double* per_agent, int64_t n_worlds, int num_agents, int num_landmarks,
                             int world_length, int auto_reset, mappo_stream_t stream);

/* --------------------------------------------------------------------- misc ---- */
int         mappo_abi_version(void);
const char* mappo_build_info(void);        /* "gfx950 ..." static string */
const char* mappo_error_string(int code);  /* static string for a return code */

#ifdef __cplusplus
}
#endif
#endif /* MAPPO_HIP_H */
