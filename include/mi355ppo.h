/* libmi355ppo -- C ABI of the MI355X-native PPO hot path.
 *
 * The reference (tbuckworth/train-procgen-pytorch) has no FFI for this path: its boundary is the
 * duck-typed Python API of agents/ppo.py (PPO), common/storage.py (Storage) and common/policy.py
 * (CategoricalPolicy) -- SURVEY.md section 8(b).  This header is the C boundary a binding for that API
 * sits on (ours: train-procgen-pytorch_amd/mi355/engine.py via ctypes).  Each entry
 * point names the reference interface it replaces.
 *
 * Conventions: every function returns 0 on success and a negative code on error (mi_last_error()
 * gives the text).  The caller owns all host buffers; the context owns all device memory.  One
 * context per GPU per process.  All work is issued on the context's HIP stream; functions that
 * return data to host buffers synchronise that stream before returning, the others are asynchronous
 * (use mi_sync).  No torch types appear here.
 */
#ifndef MI355PPO_H
#define MI355PPO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_ctx mi_ctx;

enum { MI_ARCH_IMPALA = 0, MI_ARCH_MLP = 1 };

typedef struct mi_config {
    int32_t arch;          /* MI_ARCH_IMPALA: common/model.py:167-209 ; MI_ARCH_MLP: common/model.py:954-980 */
    int32_t n_steps;       /* T  (Storage num_steps, common/storage.py:9) */
    int32_t n_envs;        /* E  local envs on this rank (Storage num_envs) */
    int32_t n_actions;     /* A  (CategoricalPolicy action_size, common/policy.py:22) ; 1 <= A <= 16 */
    int32_t obs_dim;       /* MLP only: observation vector length (IMPALA is fixed 64x64x3 uint8 NHWC) */
    int32_t mlp_depth;     /* MLP only: MLPModel depth (>= 2) */
    int32_t mlp_width;     /* MLP only: MLPModel mid_weight */
    int32_t out_dim;       /* embedder.output_dim: 256 for IMPALA, latent_size for MLP */
    int32_t max_batch;     /* largest number of samples one mi_minibatch / mi_forward call may carry */
    int32_t device;        /* HIP device ordinal */
    int32_t precision;     /* IMPALA activations / activation gradients in HBM: 0 = fp32 (parity mode), 1 = bf16 storage +
                              bf16 matrix cores with fp32 accumulation for the 16/32-channel convs (BASELINE config 3);
                              parameters, gradients of parameters, Adam, losses and GAE are fp32 in both modes */
    int32_t reserved[5];
    void*   stream;        /* hipStream_t to issue on, or NULL for a stream owned by the context */
} mi_config;

/* PPO.__init__ hyper-parameters used inside the minibatch step (agents/ppo.py:10-70) */
typedef struct mi_hparams {
    float eps_clip, value_coef, entropy_coef, x_entropy_coef, entropy_multiplier, fs_coef;
} mi_hparams;

const char* mi_last_error(void);
int mi_create(const mi_config* cfg, mi_ctx** out);
int mi_destroy(mi_ctx* ctx);
int mi_sync(mi_ctx* ctx);
/* pinned (page-locked) host memory for the rollout staging buffers: mi_put_obs / mi_put_step from such a
 * buffer is a true asynchronous DMA (the reference copies pageable numpy arrays, agents/ppo.py:74-76) */
void* mi_host_alloc(size_t bytes);
void  mi_host_free(void* p);
/* page-lock caller-owned memory in place -- the buffer an env hands its frames out in (Procgen's rgb buffer keeps its address from
 * step to step) -- so that mi_rollout_submit / mi_put_obs DMA straight out of it: no staging copy on the host (the reference copies
 * every observation twice on the host before its H2D, common/storage.py:39-44 and agents/ppo.py:74).  Negative where the runtime
 * refuses the range; the caller then stages through mi_host_alloc memory as before. */
int   mi_host_register(void* p, size_t bytes);
int   mi_host_unregister(void* p);

/* ---- parameters / optimiser state: flat fp32 vectors in the REFERENCE's policy.parameters() order and
 *      tensor layouts (policy.state_dict(), train.py:257-263 / agents/ppo.py:271-276).  The library
 *      converts to / from its device layouts (NHWC filter banks, NHWC-flatten fc columns). */
int64_t mi_param_count(mi_ctx* ctx);
int mi_set_params(mi_ctx* ctx, const float* flat, int64_t n);
int mi_get_params(mi_ctx* ctx, float* flat, int64_t n);
/* dst's parameters := src's, device to device in stream order (same architecture, sizes and device): how the validation rollouts'
 * inference-only twin context follows the trained policy (agents/ppo.py:241-252 use one policy object for both env sets) */
int mi_copy_params(mi_ctx* dst, mi_ctx* src);
int mi_get_grads(mi_ctx* ctx, float* flat, int64_t n);                     /* accumulated, un-clipped */
int mi_set_adam_state(mi_ctx* ctx, const float* exp_avg, const float* exp_avg_sq, int64_t n);
int mi_get_adam_state(mi_ctx* ctx, float* exp_avg, float* exp_avg_sq, int64_t n);

/* ---- rollout storage (Storage.store / store_last, common/storage.py:39-54).  t in [0, T]. */
int mi_put_obs(mi_ctx* ctx, int32_t t, const void* obs, size_t bytes);     /* E frames uint8 NHWC, or E x obs_dim fp32 */
int mi_get_obs(mi_ctx* ctx, int32_t t, void* obs, size_t bytes);           /* Storage.obs_batch[t] read-back */
int mi_put_step(mi_ctx* ctx, int32_t t, const float* rew, const float* done);            /* rew/done (E,) */
/* teacher forcing / compat path: overwrite what the policy step stored.  Any pointer may be NULL. */
int mi_put_policy_outputs(mi_ctx* ctx, int32_t t, const int32_t* act, const float* logp, const float* value);
enum { MI_F_REW = 0, MI_F_DONE, MI_F_VALUE, MI_F_LOGP, MI_F_ADV, MI_F_RET, MI_F_ACT };
int mi_read_field(mi_ctx* ctx, int32_t field, float* out, int64_t n);      /* (T,E) fp32; VALUE is (T+1,E); ACT as float */
int mi_write_field(mi_ctx* ctx, int32_t field, const float* in, int64_t n);

/* ---- PPO.predict on the stored observation of step t (agents/ppo.py:72-81): forward, sample, log_prob.
 *      t < T: writes act/logp/value[t]; t == T: writes value[T] only (store_last).  u: optional E uniforms
 *      in [0,1) for the inverse-CDF sampler (NULL -> Philox4x32-10 keyed by seed and t*E+e).
 *      Host outputs may be NULL (then the call does not synchronise). */
int mi_policy_step(mi_ctx* ctx, int32_t t, uint64_t seed, const float* u,
                   int64_t* act_out, float* logp_out, float* value_out);

/* ---- one call per rollout step for the agent's own loop (agents/ppo.py:228-231 fused): stores the PREVIOUS step's
 *      reward / done (t >= 1; what Storage.store(t-1) receives after env.step), runs the policy step on slot t
 *      (forward, GRU if set -- its done mask is done_prev --, fused heads + sample) and returns act / logp / value
 *      through ONE packed read-back.  rew_prev / done_prev may be NULL (t == 0). */
int mi_rollout_step(mi_ctx* ctx, int32_t t, const float* rew_prev, const float* done_prev, uint64_t seed, const float* u,
                    int64_t* act_out, float* logp_out, float* value_out);

/* ---- pipelined rollout: the same step as mi_rollout_step, per ENV GROUP and split into submit / wait, so that the frame upload and
 *      forward pass of one group run beside the host's env.step of another (agents/ppo.py:225-231 is strictly serial; an env's next
 *      observation depends on its own action only, so G contiguous groups of E/G envs form G independent chains).
 *      mi_rollout_groups(G) (1 <= G <= 4, E % G == 0) creates a stream per group.  mi_rollout_submit(t, g, frames, ...) enqueues on
 *      group g's stream: the H2D copy of the group's E/G observations (frames: pinned host memory from mi_host_alloc, valid until the
 *      matching wait; NULL = ring slot t already holds them) into ring slot t, the forward + sample on them, and the store of the
 *      previous step's reward / done of the group (rew_prev / done_prev: E/G floats each, may be NULL at t == 0); it returns at once.
 *      mi_rollout_wait(g, ...) blocks until that step's results (E/G entries each) are on the host.  One step per group in flight.
 *      Sampling uses the Philox counters t*E + e of mi_rollout_step: grouped and ungrouped rollouts draw the same actions.
 *      Any other entry point first orders the context's main stream behind all group streams.
 *      The groups' uploads take turns on the PCIe link (each reserves bytes / rate from the end of the previous reservation; environment
 *      variable MI355_COPY_GBPS, default 40, 0 = off): issued together they share the link, finish together and keep the chains in
 *      lock-step.  Uploads of at most 512 KB from device-visible pinned memory are pulled by a kernel instead of the copy engine. */
int mi_rollout_groups(mi_ctx* ctx, int32_t n_groups);
int mi_rollout_submit(mi_ctx* ctx, int32_t t, int32_t group, const void* frames, size_t bytes, const float* rew_prev,
                      const float* done_prev, uint64_t seed, const float* u);
int mi_rollout_wait(mi_ctx* ctx, int32_t group, int64_t* act_out, float* logp_out, float* value_out);

/* ---- PPO.predict(obs, hidden, done) on caller data (agents/ppo.py:72-81) when the caller has not said which
 *      storage slot the observation belongs to: obs (E frames / rows) is staged on the device, forward + sample
 *      run on it, and mi_commit_staged(t) later moves the staged observation and policy outputs into ring slot t
 *      (what Storage.store / store_last do with the same arrays, common/storage.py:39-54) without a second PCIe trip. */
int mi_predict_staged(mi_ctx* ctx, const void* obs, size_t bytes, uint64_t seed, uint64_t counter, const float* u,
                      int64_t* act_out, float* logp_out, float* value_out);
/* PPO.predict_w_value_saliency (agents/ppo.py:83-94): mi_predict_staged + d value / d observation (value.backward()): grad_out is
 * [E][64][64][3] floats (NHWC, with respect to the k/255 float frames) for IMPALA, [E][obs_dim] for the MLP.  With a GRU set (mi_set_gru)
 * the step is the recurrent one -- it consumes the hidden state / done flags of mi_rec_state and advances the state like a policy step --
 * and the gradient runs back through the GRU cell's input path (common/model.py:219-225 under autograd).  Discards (zeroes) the
 * parameter-gradient buffer, so not inside an accumulating update. */
int mi_value_saliency(mi_ctx* ctx, const void* obs, size_t bytes, uint64_t seed, uint64_t counter, const float* u,
                      int64_t* act_out, float* logp_out, float* value_out, float* grad_out);
int mi_commit_staged(mi_ctx* ctx, int32_t t);

/* ---- recurrent policies (CategoricalPolicy(recurrent=True), common/policy.py:48-50,66-67; GRU.forward prediction
 *      branch, common/model.py:219-225): h' = GRU(feat, h * (1 - done)), heads on h'.  As in the reference's
 *      algo: ppo the GRU runs in predict only and is never trained: its weights are set once,
 *      in nn.GRU's layout (weight_ih_l0 (3H,H), weight_hh_l0 (3H,H), bias_ih_l0 (3H), bias_hh_l0 (3H); gates r,z,n).
 *      mi_rec_state sets the hidden state (NULL keeps the device copy) and the done flags (NULL = zeros) that the
 *      NEXT mi_policy_step / mi_predict_staged consumes; mi_get_hidden reads the state after it. */
int mi_set_gru(mi_ctx* ctx, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh);
int mi_rec_state(mi_ctx* ctx, const float* hidden /* E x H or NULL */, const float* done /* E or NULL */);
int mi_get_hidden(mi_ctx* ctx, float* hidden /* E x H */);
/* policy(obs, hx, masks) for a recurrent policy on E observations: consumes / advances the state like a policy step */
int mi_forward_rec(mi_ctx* ctx, const void* obs, float* logp_all, float* value, float* hidden_out);

/* ---- stateless forward on caller data (policy(obs, hx, masks) / hidden_to_output, common/policy.py:61-87).
 *      obs: n frames uint8 NHWC or n x obs_dim fp32.  logp_all: n x A normalised log-probs
 *      (Categorical.logits); value: n; feat: n x out_dim.  Outputs may be NULL. */
int mi_forward(mi_ctx* ctx, const void* obs, int32_t n, float* logp_all, float* value, float* feat);

/* ---- Storage.compute_estimates (common/storage.py:56-79) on the device-resident rollout */
int mi_compute_estimates(mi_ctx* ctx, float gamma, float lmbda, int32_t use_gae, int32_t normalize_adv);
/* multi-rank advantage normalisation: stats = {count, mean, M2} (fp64) of the local un-normalised advantages */
int mi_adv_stats(mi_ctx* ctx, double stats3[3]);
int mi_adv_apply(mi_ctx* ctx, const double stats3[3]);

/* ---- one minibatch of PPO.optimize (agents/ppo.py:119-170): gather by flat index i = t*E + e (local),
 *      forward, loss, backward; gradients ACCUMULATE into the flat gradient buffer (the reference's
 *      unscaled accumulation).  n_global = size of the global minibatch the means are taken over
 *      (= n_idx on one GPU).  Appends one 8-float record to the device loss log:
 *      {pi_loss, value_loss, entropy, x_ent, total, feature_sparsity, marginal_entropy, 0}. */
int mi_minibatch(mi_ctx* ctx, const int64_t* idx, int32_t n_idx, int32_t n_global, const mi_hparams* hp);
/* gradient accumulation (agents/ppo.py:155-177: the gradients of batch_size / mini_batch_size minibatches are SUMMED before one
 * optimizer step) in one pass: idx holds the gathered samples of n_seg minibatches back to back (seg_n[k] entries each, sum =
 * n_idx <= max_batch), every one a global minibatch of n_global samples.  Same gradients up to fp32 summation order, one log
 * record per segment; needs x_entropy_coef == 0 and fs_coef == 0 (no batch-level loss term) and multirank mode 0 or 2.  On R ranks
 * a rank's share of a global minibatch is ~1/R of it: this keeps its launches at single-GPU size. */
int mi_minibatch_multi(mi_ctx* ctx, const int64_t* idx, int32_t n_idx, const int32_t* seg_n, int32_t n_seg, int32_t n_global,
                       const mi_hparams* hp);
/* clip_grad_norm_ + Adam.step + zero_grad (agents/ppo.py:174-176); adam_step = 1-based step count */
int mi_optimizer_step(mi_ctx* ctx, float lr, float max_grad_norm, int32_t adam_step, float* grad_norm_out);
int mi_loss_log_read(mi_ctx* ctx, float* out, int32_t max_records, int32_t* n_records, int32_t reset);

/* ---- data-parallel collectives inside the boundary: RCCL over xGMI, one communicator per context (SURVEY 8(b) mi_allreduce_grads,
 *      8(e) C1-C3).  Rank 0 obtains a 128-byte id (mi_comm_unique_id) and hands it to every rank by any host channel
 *      (mi355/dist.py uses torch.distributed.broadcast_object_list); every rank then calls mi_comm_init(id, rank, world).
 *      C1: the flat gradient is summed over the ranks once per optimizer step.  mi_allreduce_arm() before the LAST accumulated
 *      mi_minibatch* of a step makes that backward pass hand its gradient regions to a side stream as they become final (embedder.fc +
 *      heads -- 84 % of the bytes -- right after the first launches of the backward pass, the conv layers after the slab reduction), so
 *      the exchange overlaps the backward pass; mi_allreduce_grads() sends whatever an armed pass has not sent (the whole buffer
 *      when nothing was armed).  mi_optimizer_step waits for the exchange on the device (no host sync).
 *      C2: mi_adv_normalize_global = advantage statistics {count, mean, M2} all-gathered in fp64, merged and applied on the device.
 *      C3 / logging: mi_allreduce_buffer sums the first n floats of MI_PTR_LOSS_STATS / MI_PTR_STATS_RING on the context's stream. */
enum { MI_PTR_GRADS = 0, MI_PTR_LOSS_STATS = 1, MI_PTR_PARAMS = 2, MI_PTR_STATS_RING = 3, MI_PTR_FS_KEYS = 4 };
int mi_comm_unique_id(void* out128, size_t bytes);
int mi_comm_init(mi_ctx* ctx, const void* id128, size_t bytes, int32_t rank, int32_t world);
int mi_comm_destroy(mi_ctx* ctx);
int mi_allreduce_arm(mi_ctx* ctx);
int mi_allreduce_grads(mi_ctx* ctx);
int mi_allreduce_buffer(mi_ctx* ctx, int32_t which, int64_t n_floats);
int mi_adv_normalize_global(mi_ctx* ctx);

/* ---- raw device pointers for collectives issued by the host side (the gloo path of the CPU tests; RCCL runs inside the library, above) */
int mi_device_ptr(mi_ctx* ctx, int32_t which, void** ptr, int64_t* n_floats);
/* two-phase loss finalisation for multi-rank runs (phase 2 after the cross-rank sum of the stats) */
int mi_set_multirank(mi_ctx* ctx, int32_t enabled);   /* 0 single rank; 1 stats all-reduced per minibatch (mi_minibatch_finish); 2 deferred */
int mi_minibatch_finish(mi_ctx* ctx);      /* multirank mode 1: phase 2 + backward after the stats all-reduce */
/* fs_coef != 0 on more than one rank (IMPALA, multirank mode 1; SURVEY 8(e) C3 -- reference agents/ppo.py:148-169, common/model.py:207 on the
 * GLOBAL minibatch): before mi_minibatch, hand over the global minibatch position of every local row (ascending); mi_minibatch then
 * leaves this rank's per-column candidates (value bits << 32 | 0xffffffff - position, 2048 int64) in MI_PTR_FS_KEYS; max-all-reduce them
 * (mi_allreduce_buffer(MI_PTR_FS_KEYS, 2048), or torch.distributed on the aliased buffer) before mi_minibatch_finish, which adds the
 * gradient on the rank that owns each column's winning row and logs the global metric. */
int mi_minibatch_positions(mi_ctx* ctx, const int32_t* global_positions, int32_t n);
/* multirank mode 2 (x_entropy_coef == 0 and fs_coef == 0: the backward pass needs no cross-rank statistic): mi_minibatch runs to
 * completion, this rank's partial loss sums of minibatch k go to ring slot k (MI_PTR_STATS_RING, 32 floats each); once per
 * optimize() the caller sums the first 32 * n_minibatches floats over the ranks and calls mi_loss_log_finalize. */
int mi_loss_log_finalize(mi_ctx* ctx);

/* ---- live kernel timing for bench.py's roofline leg: HIP events recorded on the context's stream around every
 *      conv / pool / GEMM launch.  Classes are reported separately for the rollout phase (phase 0, n = E) and the
 *      update phase (phase 1, n = minibatch).  mi_profile_read fills up to max_rows rows of
 *      {class id, phase, launches, total ms, total samples, total algorithmic bytes, total algorithmic flops}
 *      (layer-boundary model of SURVEY.md 8(d)); names via mi_profile_class_name. */
int mi_profile_enable(mi_ctx* ctx, int32_t enabled);   /* low byte: 0 off, 1 update phase only, 2 rollout + update; bits 8.. = P (0 -> 1):
                                                         * bracket every P-th minibatch of the update phase (sampling keeps the
                                                         * event records' own cost, ~7 us of stream time per launch, out of the run) */
int mi_profile_read(mi_ctx* ctx, double* rows7, int32_t max_rows, int32_t* n_rows, int32_t reset);
const char* mi_profile_class_name(int32_t class_id);

/* ---- op-level entry points for parity tests (host buffers in, host buffers out; NHWC fp32) */
int mi_op_conv3x3(mi_ctx* ctx, int32_t mode /*0 fwd,1 dgrad,2 wgrad; a block's first conv, bf16: 3 conv+pool fwd, 4 / 5 wgrad / dgrad from the pooled gradient,
                  6 / 7 block2.conv's fused dgrad+wgrad launch returning dW / dx*/, int32_t cin, int32_t cout, int32_t hw, int32_t n,
                  const void* in, int32_t in_is_u8, int32_t relu_in, const float* w_ref /*[cout][cin][3][3]*/,
                  const float* bias, const float* res, const float* mask, const float* dout,
                  float* out /*fwd/dgrad: activations; wgrad: [cout][cin][3][3]*/, float* dbias_out);
/* bf16 precision only: the fused residual-block kernels.  mode 0 forward (x, w1, b1, w2, b2 -> out_a = conv1 output,
 * out_y = block output); mode 1 data gradients (x = dy, a_fwd, x_fwd, w1, w2 -> out_a = d conv1-output, out_y = d block-input);
 * mode 3: res1 + res2 forward in one launch with the same (w1, b1, w2, b2) for both blocks (out_a = second conv1 output);
 * mode 2 (16 channels @32x32, 32 channels @16x16) the whole backward in one launch: out_y = d block-input, out_a[0 .. 2*(9*ch*ch+ch)) = {dW1, db1, dW2, db2}.
 * Replaces ResidualBlock.forward and its autograd (common/model.py:141-146). */
int mi_op_resblock(mi_ctx* ctx, int32_t mode, int32_t ch, int32_t hw, int32_t n, const float* x, const float* w1, const float* b1,
                   const float* w2, const float* b2, const float* a_fwd, const float* x_fwd, float* out_a, float* out_y);
int mi_op_maxpool(mi_ctx* ctx, int32_t mode /*0 fwd,1 bwd*/, int32_t n, int32_t hw, int32_t c,
                  const float* in, const float* dout, float* out);
int mi_op_gemm(mi_ctx* ctx, int32_t M, int32_t N, int32_t K, const float* A, int64_t sam, int64_t sak,
               const float* B, int64_t sbk, int64_t sbn, float* C);
int mi_selftest_mfma(mi_ctx* ctx, float* max_err);
/* what the last mi_minibatch left in the activation buffers (first n samples), fp32 NHWC: which = 8 * block + k, k = 0 pooled map,
 * 1 res1.conv1 out, 2 res1 out, 3 res2.conv1 out, 4 block out, 5 max-pool arg-max (window position ky*3+kx); which = 100: features.
 * Lets a parity test run the oracle's backward pass on the ENGINE's forward tensors (teacher forcing, tests/test_gpu_bf16.py). */
int mi_debug_read(mi_ctx* ctx, int32_t which, int32_t n, float* out);
/* The production sampler's generator (dist.sample() of agents/ppo.py:77 is torch's; here Philox4x32-10 keyed by (seed, t*E + e)):
 * n x {c0,c1,c2,c3,k0,k1} in; out4 = n x 4 output words of the ten-round bijection (checked against the Random123 known-answer
 * vectors), u_out = n uniforms exactly as the sample kernels draw them for seed = k0 | k1 << 32, counter = c0 | c1 << 32. */
int mi_debug_philox(mi_ctx* ctx, const uint32_t* ctr_key6, int32_t n, uint32_t* out4, float* u_out);
/* measurement hook: wall-clock microseconds per policy step of slot t (the step's launches + a stream wait, `iters` times), issued
 * eagerly (mode 0) or as one replay of a hipGraph captured from the same launches (mode 1) */
int mi_debug_step_latency(mi_ctx* ctx, int32_t t, int32_t iters, int32_t mode, float* us_out);
/* bit 0 set: rollout-sized bf16 inference passes use the separate block-2 / block-3 kernels instead of the fused launch
 * (rollout_bf16.hip) -- the A side of the bit-equality test of the two paths;
 * bit 2 set: a group's frames always go up by DMA copy, never pulled by a kernel (A/B timing);
 * bit 4 set: a minibatch pass keeps the logged statistics and embedder.fc's weight gradient on the main stream instead of
 * forking them onto the side stream -- the A side of the bit-equality test of the two orders */
int mi_debug_flags(mi_ctx* ctx, int32_t flags);

#ifdef __cplusplus
}
#endif
#endif
