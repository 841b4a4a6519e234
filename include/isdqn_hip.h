/*
 * isdqn_hip.h -- C ABI of the MI355X (gfx950) iS-DQN hot path.
 *
 * The reference (theovincent/iS-DQN, python package `slimdqn`) has no FFI layer:
 * its hot path is numpy + an XLA executable.  This header is the boundary a
 * maintainer would bind instead (ctypes stub: INTEGRATION.md).  Every entry
 * point names the reference code it replaces (paths relative to the reference
 * root).  Conventions:
 *   - plain pointers and sizes only; all data pointers are DEVICE pointers
 *     unless the parameter name ends in `_host`;
 *   - `stream` is a hipStream_t (0 = default stream); calls enqueue work and
 *     return without synchronising;
 *   - return value: ISDQN_OK or a negative ISDQN_ERR_* (host-detectable
 *     argument errors).  Data-dependent violations that the reference reports
 *     as exceptions (negative priority, query target outside [0, root)) are
 *     reported through a device status word (`dev_status`, OR-ed bits
 *     ISDQN_STATUS_*), so that the fused training step never synchronises;
 *   - nothing here allocates device memory: the caller owns every buffer
 *     (torch tensors in the python host layer).
 */
#ifndef ISDQN_HIP_H
#define ISDQN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------- */
#define ISDQN_OK 0
#define ISDQN_ERR_CAPACITY (-1)    /* sum_tree.py:12  "Capacity to sum tree must be positive."   -> AssertionError */
#define ISDQN_ERR_NEGATIVE (-2)    /* sum_tree.py:31  "Values must be positive."                 -> AssertionError */
#define ISDQN_ERR_SHAPE (-3)       /* sum_tree.py:30  indices/values shape mismatch              -> AssertionError */
#define ISDQN_ERR_EMPTY (-4)       /* replay_buffer.py:200 / samplers.py:41 empty buffer         -> AssertionError */
#define ISDQN_ERR_RANGE (-5)       /* sum_tree.py:73-74 target outside [0, root)                 -> ValueError     */
#define ISDQN_ERR_UNSUPPORTED (-6) /* configuration that is not built (BatchNorm, conv widths above 64 channels, ...) */
#define ISDQN_ERR_HIP (-7)         /* a HIP runtime call failed; see isdqn_last_error()           */
#define ISDQN_ERR_ARG (-8)         /* null pointer / bad size                                     */

/* bits of the device status word */
#define ISDQN_STATUS_NEGATIVE_VALUE 1u /* tree_set saw a value < 0: nothing was modified          */
#define ISDQN_STATUS_TARGET_RANGE 2u   /* tree_query saw a target outside [0, root)               */
#define ISDQN_STATUS_EMPTY_TREE 4u     /* tree_query with root == 0                               */

const char* isdqn_version(void);
const char* isdqn_last_error(void); /* thread-local text of the last ISDQN_ERR_HIP / _ARG */

/* ========================================================================== */
/* Sum tree  (slimdqn/sample_collection/sum_tree.py)                          */
/* ========================================================================== */

/* SumTree.__init__ sizing, sum_tree.py:11-18.  Host-only arithmetic. */
int isdqn_tree_layout(int64_t capacity, int32_t* depth, int64_t* first_leaf_offset, int64_t* n_nodes);

/* SumTree.set, sum_tree.py:20-47.  `nodes` is the float64 node array (n_nodes).
 * Batch of n <= ISDQN_TREE_MAX_BATCH (leaf index, value) pairs.  Bit-exact with the
 * reference: duplicates keep the FIRST occurrence (np.unique), deltas are added to every
 * ancestor sequentially in ascending leaf order (np.add.at).  `max_recorded_priority`
 * (device double, may be NULL) is raised to max(values) as sum_tree.py:32 does. */
#define ISDQN_TREE_MAX_BATCH 4096
int isdqn_tree_set(double* nodes, int32_t depth, const int32_t* indices, const double* values, int32_t n,
                   double* max_recorded_priority, uint32_t* dev_status, void* stream);

/* The two-leaf set of PrioritizedSamplingDistribution.remove, samplers.py:89-103:
 * set([index, last_index], [get(last_index), 0.0]) (or set(index, 0.0) when equal), with the
 * leaf read done on the device so that eviction never synchronises. */
int isdqn_tree_swap_remove(double* nodes, int32_t depth, int32_t index, int32_t last_index, uint32_t* dev_status,
                           void* stream);

/* SumTree.query, sum_tree.py:58-102.  If `targets_are_unit` != 0 the inputs are unit
 * draws u in [0,1) and the target is 0.0 + root*u, which is bit-identical to numpy's
 * Generator.uniform(0.0, root, n) of samplers.py:110 on the same PCG64 stream.
 * Writes n leaf indices (int32). */
int isdqn_tree_query(const double* nodes, int32_t depth, const double* targets, int32_t n, int32_t targets_are_unit,
                     int32_t* out_indices, uint32_t* dev_status, void* stream);

/* ========================================================================== */
/* Device-resident replay  (slimdqn/sample_collection/replay_buffer.py)        */
/* ========================================================================== */
/* Layout in HBM: single frames `frames[slot][h*w]` (uint8) + an element table indexed by
 * element slot (= key % capacity): frame slots of the state stack and of the next-state
 * stack (-1 = all-zero frame: the zero padding of replay_buffer.py:131-134), action,
 * n-step reward, terminal flag.  ReplayBuffer.sample (:198-213) becomes two kernels:
 * the row gather below and, only for callers that want the reference's batch layout,
 * the stack materialisation. */

/* ReplayBuffer.add, device side (replay_buffer.py:185-196: `self._memory[key] = ...`, the FIFO eviction and the sampler's
 * index bookkeeping).  The host accumulator stages what changed since the last flush in one buffer -- byte offsets into
 * `staged` below, every section 16-byte aligned -- and one launch scatters it:
 *   frames[frame_slots[i]] <- frame_data[i]                          (n_frames new observations of frame_bytes bytes)
 *   element row rows[r]    <- row_frames[r][stack2], row_action[r], row_reward[r], row_terminal[r]     (n_rows)
 *   index_to_slot[index_rows[i]] <- index_vals[i]                                                      (n_index)   */
typedef struct isdqn_staged_updates {
    int32_t n_frames, frame_bytes;
    int64_t off_frame_slots; /* int32 [n_frames]              */
    int64_t off_frame_data;  /* uint8 [n_frames][frame_bytes] */
    int32_t n_rows, stack2;
    int64_t off_rows;         /* int32 [n_rows]               */
    int64_t off_row_frames;   /* int32 [n_rows][stack2]       */
    int64_t off_row_action;   /* int32 [n_rows]               */
    int64_t off_row_reward;   /* float [n_rows]               */
    int64_t off_row_terminal; /* uint8 [n_rows]               */
    int32_t n_index, reserved;
    int64_t off_index_rows;   /* int32 [n_index]              */
    int64_t off_index_vals;   /* int32 [n_index]              */
} isdqn_staged_updates;
int isdqn_replay_apply_staged(const uint8_t* staged, const isdqn_staged_updates* updates, uint8_t* frames, int64_t frame_stride,
                              int32_t* elem_frames, int32_t* elem_action, float* elem_reward, uint8_t* elem_terminal,
                              int32_t* index_to_slot, void* stream);

/* Gather the element rows of `B` sampled elements: frame ids [B][2*stack], action, reward,
 * terminal (itemgetter + np.stack of the scalar fields, replay_buffer.py:206-212).  `slots`
 * are element slots, or -- when `index_to_slot` is not NULL -- the sampler's dense indices
 * (samplers.py:43, :111), mapped through that table (the device copy of `_index_to_key`,
 * samplers.py:45-49, reduced modulo the capacity). */
int isdqn_replay_gather_rows(const int32_t* elem_frames, const int32_t* elem_action, const float* elem_reward,
                             const uint8_t* elem_terminal, int32_t stack, const int32_t* index_to_slot,
                             const int32_t* slots, int32_t B, int32_t* out_frame_ids, int32_t* out_action,
                             float* out_reward, uint8_t* out_terminal, void* stream);

/* Materialise ReplayElement.state / .next_state, each (B, h, w, stack) uint8 in the
 * reference's channel-last layout (replay_buffer.py:131-147 + np.stack :212). */
int isdqn_replay_materialize(const uint8_t* frames, int64_t frame_stride, int32_t h, int32_t w, int32_t stack,
                             const int32_t* frame_ids, int32_t B, uint8_t* out_state, uint8_t* out_next_state,
                             void* stream);

/* Inverse: split channel-last stacks (B, h, w, stack) into 2*B*stack single frames
 * (planar) + the id table, for callers that hold reference-layout batches. */
int isdqn_replay_deinterleave(const uint8_t* state, const uint8_t* next_state, int32_t h, int32_t w, int32_t stack,
                              int32_t B, uint8_t* out_frames, int32_t* out_frame_ids, void* stream);

/* ========================================================================== */
/* Q-network + iS-DQN update  (slimdqn/networks/architectures/dqn.py, isdqn.py) */
/* ========================================================================== */
#define ISDQN_ARCH_CNN 0 /* dqn.py:48-74  three SAME convs (8s4, 4s2, 3s1) + dense stack */
#define ISDQN_ARCH_FC 1  /* dqn.py:89-103 dense stack only                               */
#define ISDQN_ARCH_IMPALA 2 /* dqn.py:7-36, 75-88  three Stacks (conv3x3, max-pool 3x3/2, two residual blocks) + dense stack */
#define ISDQN_MAX_FEATURES 8

#define ISDQN_PRECISION_BF16X3 0 /* split-bf16 MFMA (hi*hi + lo*hi + hi*lo), fp32 accumulate: ~2^-17 */
#define ISDQN_PRECISION_BF16 1   /* single-pass bf16 MFMA, fp32 accumulate: ~2^-9                    */

typedef struct isdqn_net_config {
    int32_t arch;                         /* ISDQN_ARCH_*                                              */
    int32_t obs_h, obs_w, obs_c;          /* cnn: frame h, w and stack size; fc: obs_c = obs dim, h=w=1 */
    int32_t n_features;                   /* len(features) (isdqn.py:19)                               */
    int32_t features[ISDQN_MAX_FEATURES]; /* cnn: 3 conv widths then dense widths; fc: dense widths     */
    int32_t n_actions;                    /* A                                                         */
    int32_t n_heads;                      /* 1 + n_bellman_iterations (isdqn.py:34-41); 1 = DQN / TF-DQN */
    int32_t layer_norm;                   /* 0/1 (dqn.py:56, 63, 70, 97)                               */
    int32_t batch_size;                   /* B: learn_on_batch runs the network on 2B rows (isdqn.py:95) */
    int32_t precision;                    /* ISDQN_PRECISION_*                                         */
    float gamma_n;                        /* gamma ** update_horizon (isdqn.py:107)                    */
    float learning_rate, adam_b1, adam_b2, adam_eps; /* optax.adam(lr, eps=adam_eps) (isdqn.py:46)    */
    float huber_delta;                    /* 0: squared TD error, the reference's loss (isdqn.py:102); > 0: Huber loss with
                                           * this delta (0.5 d^2 for |d| <= delta, delta (|d| - delta/2) beyond; the north
                                           * star's wording), gradient clip(d, -delta, delta)                */
    int32_t batch_norm;                   /* 0/1 (dqn.py:52-53, 59-60, 66-67, 73-74, 100-101): flax.linen.BatchNorm behind the
                                           * input scaling and behind every hidden layer's ReLU -- `axis=(1, 2)` on image
                                           * tensors (statistics per pixel position over batch AND channels), per feature on
                                           * flattened / dense activations; momentum 0.99, epsilon 1e-5.  learn / loss run it on
                                           * the batch statistics of concat(state, next_state) (isdqn.py:95), which couples the two
                                           * halves: the backward then runs over all 2B rows (csrc/batchnorm.h, generic engine,
                                           * one stream).  forward / best_action(s) use the running averages (isdqn.py:130).
                                           * impala: also behind the ReLU of every residual block (dqn.py:29-30, module
                                           * names "Stack_s/BatchNorm_b").  grad_on_batch runs with it (the analysis agents,
                                           * analysisdqn.py:162-219; with `target_params` the two halves are two forwards on
                                           * their own statistics and the backward covers the B state rows).  The *_target
                                           * learn / loss entry points (DQN) return ISDQN_ERR_UNSUPPORTED with it: the reference's
                                           * DQN cannot run with it either (dqn.py:86 applies the network without a mutable
                                           * batch_stats collection).                                                           */
} isdqn_net_config;

/* One parameter tensor inside the flat fp32 parameter buffer.  `name` is the Flax
 * module/leaf ("Conv_0/kernel", "LayerNorm_3/scale", "Dense_1/bias", ...); `flax_shape`
 * is the reference shape; `offset`/`size` locate the tensor in the INTERNAL layout:
 *   conv kernel  -> [out][kh*kw taps][in padded to 8]   (Conv_0: [out][in_plane][kh][kw])
 *   dense kernel -> [out][in]          (in = h*w*(c padded to 8) after the conv torso)
 *   vectors      -> as is
 * The python layer converts between the two (import/export of reference checkpoints). */
typedef struct isdqn_tensor_info {
    char name[48];
    int64_t offset; /* in floats */
    int64_t size;   /* in floats, internal (padded) */
    int32_t kind;   /* 0 conv kernel, 1 dense kernel, 2 bias, 3 ln scale, 4 ln bias; BatchNorm_i: 5 scale, 6 bias ("params"
                     * collection), 7 mean, 8 var ("batch_stats" collection: running averages, not touched by Adam).
                     * BatchNorm tensors: dims = [groups, P, C, C padded to 8]; spatial sites (flax_shape (H, W)) hold one
                     * value per pixel position, feature sites (flax_shape (P*C,)) one per internal column p*Cpad + c */
    int32_t layer;  /* index into the layer list */
    int32_t ndim;
    int32_t flax_shape[4];
    int32_t dims[4]; /* internal dims: conv [out, taps, in_pad, 0] ; dense [out, in_internal, 0, 0] */
} isdqn_tensor_info;

/* Parameter buffer size (floats) and tensor table.  `infos` may be NULL to query the count. */
int isdqn_net_param_layout(const isdqn_net_config* cfg, int64_t* n_param_floats, isdqn_tensor_info* infos,
                           int32_t max_infos, int32_t* n_infos);

/* Workspace bytes needed by forward / learn_on_batch for cfg->batch_size. */
int isdqn_net_workspace_bytes(const isdqn_net_config* cfg, int64_t* bytes);

/* Named workspace regions, so tests can read intermediates (activations, gradients). */
int isdqn_net_workspace_region(const isdqn_net_config* cfg, const char* name, int64_t* offset_bytes,
                               int64_t* size_bytes);

/* A batch of B transitions as the update consumes it (ReplayElement fields,
 * replay_buffer.py:26-34).  cnn: stacks are referenced as single frames. */
typedef struct isdqn_batch {
    int32_t B;
    const uint8_t* frames;    /* cnn: base of the frame store                                  */
    int64_t frame_stride;     /* cnn: bytes between consecutive frame slots (>= h*w)           */
    const int32_t* frame_ids; /* cnn: [B][2*stack] state planes then next_state planes, -1 = 0 */
    const float* state;       /* fc:  [B][obs]                                                 */
    const float* next_state;  /* fc:  [B][obs]                                                 */
    const int32_t* action;    /* [B]                                                           */
    const float* reward;      /* [B]  n-step discounted reward                                 */
    const uint8_t* terminal;  /* [B]                                                           */
    int32_t flags;            /* ISDQN_BATCH_* (0 when in doubt)                               */
    void* priorities_ready;   /* hipEvent_t or NULL: recorded on `stream` once q_values / targets / priorities / losses of
                               * this call are final (long before the call's last kernel), so that a caller's second stream
                               * can write the priorities back (R6) and draw the next batch (S4, R5) under the backward pass */
} isdqn_batch;

/* The workspace keeps a pre-split (bf16 hi + lo) mirror of the weights that the MFMA stages copy from.  Every entry point
 * rebuilds it from `params` first -- `params` is caller-owned memory -- unless the caller passes this flag, promising that
 * the previous call on this workspace was isdqn_net_learn_on_batch with the same `params` and that nothing has written
 * `params` since (learn_on_batch leaves the mirror current: Adam writes both forms).  The captured multi-step graphs of
 * slimdqn/_graph.py use it for every step of a replay (isdqn_net_refresh_mirror in front of a replay when needed). */
#define ISDQN_BATCH_MIRROR_CURRENT 1

/* DQNNet.apply on `n_rows` observations (dqn.py:47-103) -> q [n_rows][n_heads*n_actions].
 * cnn: image j reads frame_ids[j*stack .. j*stack+stack-1]; fc: obs [n_rows][obs_c]. */
int isdqn_net_forward(const isdqn_net_config* cfg, const float* params, const uint8_t* frames, int64_t frame_stride,
                      const int32_t* frame_ids, const float* obs, int32_t n_rows, float* q_out, void* workspace,
                      void* stream);

/* iSDQN.learn_on_batch (isdqn.py:82-109): forward on concat(state, next_state), iterated
 * Bellman targets from heads 0..K-1 of the next states, squared TD loss on heads 1..K,
 * backward, Adam (in place on params / adam_m / adam_v; `adam_count` is a device int32
 * step counter incremented by the call).  Outputs (device): losses[K] = td.mean(axis=0)
 * (isdqn.py:103); optionally losses_accum[K] += losses (the `cumulated_losses += losses` of
 * update_online_params, isdqn.py:62, kept on the device so the step never synchronises),
 * q_values[B][K], targets[B][K] and priorities[B] (float64, sqrt(mean_k td + 1e-10): the
 * TD-error writeback the north star asks for; the reference has no trainer wiring for it --
 * see DESIGN.md). */
int isdqn_net_learn_on_batch(const isdqn_net_config* cfg, float* params, float* adam_m, float* adam_v,
                             int32_t* adam_count, const isdqn_batch* batch, float* losses, float* losses_accum,
                             float* q_values, float* targets, double* priorities, void* workspace, void* stream);

/* Loss only, no update: iSDQN.loss_on_batch (isdqn.py:92-103). */
int isdqn_net_loss_on_batch(const isdqn_net_config* cfg, const float* params, const isdqn_batch* batch, float* losses,
                            float* q_values, float* targets, void* workspace, void* stream);

/* DQN.learn_on_batch / DQN.loss_on_batch (slimdqn/networks/dqn.py:59-83): the baselines on the same kernels.
 * cfg->n_heads == 1 (one head of n_actions outputs; K = 1): head 0 of the states is regressed on
 * r + (1 - terminal) * gamma^n * max_a head 0 of the next states.
 *   - isdqn_net_learn_on_batch / isdqn_net_loss_on_batch with n_heads == 1 is TF-DQN (tfdqn.py:55-80: the next
 *     states go through the SAME parameters, stop-gradient target);
 *   - the *_target forms take the next states through `target_params` (DQN: a copy refreshed every
 *     target_update_frequency steps by the caller, dqn.py:49-50).  losses[1] is the batch mean (dqn.py:69-70). */
int isdqn_net_learn_on_batch_target(const isdqn_net_config* cfg, float* params, const float* target_params, float* adam_m,
                                    float* adam_v, int32_t* adam_count, const isdqn_batch* batch, float* losses,
                                    float* losses_accum, float* q_values, float* targets, double* priorities,
                                    void* workspace, void* stream);
int isdqn_net_loss_on_batch_target(const isdqn_net_config* cfg, const float* params, const float* target_params,
                                   const isdqn_batch* batch, float* losses, float* q_values, float* targets,
                                   void* workspace, void* stream);

/* Rebuild the workspace's weight mirror from `params` now (what every entry point does at its head unless the caller passes
 * ISDQN_BATCH_MIRROR_CURRENT).  For callers that replay a captured graph whose steps all trust the mirror: one eager launch in front
 * of the replay when something wrote the parameters in between (slimdqn/_graph.py), instead of a second capture. */
int isdqn_net_refresh_mirror(const isdqn_net_config* cfg, const float* params, void* workspace, void* stream);

/* BatchNorm networks: params["batch_stats"] <- the batch_stats collection returned by the LAST training-mode forward that ran
 * in `workspace` (learn / loss / grad_on_batch; flax: apply(..., mutable=["batch_stats"])), i.e. running = 0.99 * running +
 * 0.01 * (that forward's batch statistics).  learn_on_batch does this itself for its own forward (isdqn.py:87-88); the analysis
 * agents store the collection of ANOTHER forward -- the evaluation batch's, analysisdqn.py:121-131, analysistfdqn.py:85-95 -- and
 * call this behind that loss pass.  ISDQN_ERR_ARG for a configuration without batch_norm. */
int isdqn_net_bn_commit_running(const isdqn_net_config* cfg, float* params, const void* workspace, void* stream);

/* Gradient of a TD loss, no update: the three gradients AnalysisDQN compares (slimdqn/networks/analysisdqn.py:156-219 --
 * jax.grad of compute_loss_is / compute_loss_tf / compute_loss_tb).  `grad_out` receives the gradient w.r.t. every parameter
 * in the internal layout of isdqn_net_param_layout (n_param_floats floats); parameters and optimizer state are untouched.
 *   n_pairs <= 0 : the configuration's own loss (iS-DQN: online head 1+k on target head k of the same parameters);
 *   n_pairs  > 0 : online heads online_head + k regressed on target heads target_head + k, k < n_pairs
 *                  (analysisdqn.py:162-176: head 1 on head 1);
 *   target_params != NULL : the next states go through `target_params` (compute_loss_tb).
 * losses[n_pairs or K], q_values / targets [B][n_pairs or K] (may be NULL). */
int isdqn_net_grad_on_batch(const isdqn_net_config* cfg, const float* params, const float* target_params,
                            const isdqn_batch* batch, int32_t online_head, int32_t target_head, int32_t n_pairs, float* grad_out,
                            float* losses, float* q_values, float* targets, void* workspace, void* stream);

/* iSDQN.shift_params (isdqn.py:111-125): head k <- head k+1 on the last Dense; moments untouched. */
int isdqn_net_shift_params(const isdqn_net_config* cfg, float* params, void* stream);

/* iSDQN.best_action (isdqn.py:127-135): forward one observation, argmax of head 1+idx_network. */
int isdqn_net_best_action(const isdqn_net_config* cfg, const float* params, const uint8_t* frames,
                          int64_t frame_stride, const int32_t* frame_ids, const float* obs, int32_t idx_network,
                          int32_t* out_action, void* workspace, void* stream);

/* best_action for `n_rows` observations at once (vectorised host environments: one forward, one argmax launch):
 * out_actions[i] = argmax_a of online head idx_networks[i] (device int32 arrays) of observation i.  n_rows <= 2 * batch_size.
 * flags: ISDQN_BATCH_MIRROR_CURRENT when the caller knows the workspace's weight mirror matches `params`. */
int isdqn_net_best_actions(const isdqn_net_config* cfg, const float* params, const uint8_t* frames, int64_t frame_stride,
                           const int32_t* frame_ids, const float* obs, int32_t n_rows, const int32_t* idx_networks,
                           int32_t* out_actions, int32_t flags, void* workspace, void* stream);

/* AnalysisNet.apply (slimdqn/utils/analysis_architecture.py:9-122) as eval_srank_and_dead_neurons uses it
 * (experiments/base/srank_and_dead_neurons.py:8-22): the network without its last layer on `n_rows` observations
 * (<= 2 * batch_size).  features_out [n_rows][width of the last hidden layer] = its post-ReLU activations (behind the last
 * BatchNorm when the network has them, as the reference returns them); scores_out = for every recorded layer in the reference's
 * order, the sum over the rows of its post-ReLU activations in the reference's feature order ((H, W, C) flattened for conv
 * layers) -- cnn: the three conv layers, impala: the two ReLU outputs of each residual block of each Stack and the flattened
 * torso output, then the hidden Dense layers; sizes from isdqn_net_analysis_layout (at most 32 entries).  BatchNorm networks
 * run on the batch statistics of these rows (the reference applies AnalysisNet with mutable batch_stats).  The srank (an SVD)
 * and the dead-neuron fraction are host arithmetic on these two arrays, as in the reference (utils/analysis.py:4-17). */
int isdqn_net_analysis_layout(const isdqn_net_config* cfg, int32_t* n_hidden, int64_t* sizes, int32_t max_sizes);
int isdqn_net_analysis(const isdqn_net_config* cfg, const float* params, const uint8_t* frames, int64_t frame_stride,
                       const int32_t* frame_ids, const float* obs, int32_t n_rows, float* features_out, float* scores_out,
                       void* workspace, void* stream);

/* Engine self-test: C[M][N] = A . B on the MFMA tile engine for every operand-layout
 * combination (a_tr/b_tr: 0 = operand stored [rows][K], 1 = stored [K][rows]).  Test hook. */
int isdqn_selftest_gemm(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int32_t a_tr,
                        int32_t b_tr, int32_t precision, int32_t split_k, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISDQN_HIP_H */
