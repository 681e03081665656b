/*
 * fdql.h — C ABI of the MI355X-native franQ hot path (replay ring -> windowed minibatch
 *          -> SAC/TQC update).  Plain pointers and sizes only; no torch types.
 *
 * The reference (llucid-97/FastDeepQLearning, "franQ") has NO FFI of its own: its
 * boundary is two duck-typed Python protocols.  Each entry point below cites the
 * reference interface it stands under (paths relative to the reference root);
 * fastdeepqlearning_amd/{Replay,Agent} keep those Python object shapes and call through
 * here with ctypes (INTEGRATION.md shows the binding a franQ maintainer would add).
 *
 * Conventions
 *   - every function returns 0 on success, a negative FDQL_E* code otherwise;
 *     fdql_last_error() returns a thread-local message for the last failure;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Nothing in this
 *     library synchronises the host unless its comment says so;
 *   - device pointers handed in stay owned by the caller (torch tensors); a ring owns its
 *     own HBM;
 *   - threads: every fdql_ring_* call on one ring handle is serialised by a mutex inside the
 *     handle (host bookkeeping + launch order; it never waits for the GPU unless the entry
 *     point's comment says it synchronises), so one writer thread per shard may add() while
 *     the trainer thread samples the same shard - the reference's pattern
 *     (franQ/Replay/async_replay_memory.py:55-70, franQ/Runner/runner.py:177-191), which
 *     there relies on the GIL.  On the device, calls on ONE stream are ordered by the
 *     stream; once a handle has seen two different streams, its writes wait for earlier
 *     reads issued elsewhere and its reads for earlier writes (events).  fdql_agent_* calls
 *     on one agent handle are serialised the same way (update / act / set_* / scalars).
 */
#ifndef FDQL_H
#define FDQL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDQL_OK 0
#define FDQL_EINVAL (-1)     /* bad argument / shape mismatch                                   */
#define FDQL_EHIP (-2)       /* a HIP runtime call failed                                       */
#define FDQL_EOVERSAMPLE (-3)/* franQ/Replay/replay_memory.py:6,50,57-58  OversampleError       */
#define FDQL_ESTATE (-4)     /* call order violated (e.g. update before bind)                   */
#define FDQL_ENOMEM (-5)

const char *fdql_last_error(void);
int fdql_version(void);
/* sizeof() of the structs below as this library was compiled, so a binding can verify its
 * own mirror: out[0]=fdql_agent_config_t, [1]=fdql_batch_t, [2]=fdql_agent_stats_t,
 * [3]=fdql_kernel_time_t, [4]=fdql_reward_fn_t, [5]=fdql_episode_spec_t. */
void fdql_abi_sizes(int32_t *out6);

/* ------------------------------------------------------------------------------------ */
/* Replay ring: structure-of-arrays ring in HBM                                          */
/* replaces franQ/Replay/replay_memory.py:18-73 (ReplayMemory) behind                    */
/*          franQ/Replay/async_replay_memory.py:9-70 and                                 */
/*          franQ/Replay/wrappers/torch_dataloader.py:11-50 (host->device + f32 cast)    */
/* ------------------------------------------------------------------------------------ */
typedef struct fdql_ring fdql_ring_t;

/* One f32 column block per key: ring[k] is [maxlen, dims[k]] float32 in HBM.  The f32 cast
 * the reference applies at read time (torch_dataloader.py:36) is applied at write time. */
int fdql_ring_create(fdql_ring_t **out, int64_t maxlen, int32_t n_keys, const int32_t *dims);
/* Same, with a storage type per key: 0 = float32, 1 = uint8.  The reference ring keeps every key in its
 * source dtype (replay_memory.py:26-35: a uint8 frame stack stays uint8, 28 224 B per 4x84x84 observation
 * instead of 113 KB) and TorchDataLoader casts to float32 at read time (torch_dataloader.py:36).  A uint8
 * key does exactly that: rows still enter and leave the C ABI as float32 (integral values 0..255), the
 * ring block holds one byte per element and the gather kernel widens on the fly.                        */
#define FDQL_F32 0
#define FDQL_U8 1
int fdql_ring_create_typed(fdql_ring_t **out, int64_t maxlen, int32_t n_keys, const int32_t *dims,
                           const int32_t *dtypes);
int fdql_ring_destroy(fdql_ring_t *ring);

/* replay_memory.py:38-46 add(): append `n` packed host rows (sum(dims) floats each, keys in
 * creation order).  Rows are staged in pinned memory and reach HBM at the next flush,
 * sample, or when the staging area fills.  Host-side only unless a flush is triggered. */
int fdql_ring_add(fdql_ring_t *ring, const float *host_rows, int64_t n, void *stream);
/* Same, rows already on the device (packed [n, sum(dims)] f32). */
int fdql_ring_add_device(fdql_ring_t *ring, const float *dev_rows, int64_t n, void *stream);
int fdql_ring_flush(fdql_ring_t *ring, void *stream);

/* replay_memory.py:45-46,72-73: __len__ including the max(top,len) quirk (caps at
 * maxlen-1 after the first wrap) and the write cursor. Counts staged rows too. */
/* Checkpoint of the ring contents (true resume; the reference keeps its ring in host memory and does not
 * save it - SURVEY 8f rank 3).  snapshot: slots [0, n_slots) in slot order as packed rows
 * [n_slots, row_floats] into host memory (synchronises `stream`).  restore: the inverse, then the
 * write position and length are set to (top, len) as fdql_ring_top / fdql_ring_len reported them.  */
int fdql_ring_snapshot(fdql_ring_t *ring, float *host_rows_out, int64_t n_slots, void *stream);
int fdql_ring_restore(fdql_ring_t *ring, const float *host_rows, int64_t n_slots, int64_t top, int64_t len,
                      void *stream);
int64_t fdql_ring_len(const fdql_ring_t *ring);
int64_t fdql_ring_top(const fdql_ring_t *ring);
int64_t fdql_ring_row_floats(const fdql_ring_t *ring);
/* Device base pointer of key k ([maxlen, dims[k]] f32) — for tests and bulk fills. */
int fdql_ring_key_ptr(fdql_ring_t *ring, int32_t key, float **dev_ptr);   /* float32 keys only */

int fdql_ring_key_ptr_u8(fdql_ring_t *ring, int32_t key, uint8_t **dev_ptr); /* uint8 keys only: [maxlen, dims[k]] bytes */
/* Reading a key's block IN PLACE instead of gathering it (the pixel encoder's first layer reads the uint8 frames of a window
 * straight from the ring: fdql_batch_t.obs_2d_u8 / obs_2d_slots): slots_out_dev[t*B + b] = (starts_dev[b] + t) % len - the
 * index matrix of replay_memory.py:63-65 as int32 [T, B] (device) - for the int64 [B] window starts a sample call returned
 * (starts_out_dev) or the caller supplied. */
int fdql_ring_window_slots(fdql_ring_t *ring, int32_t T, int32_t B, const int64_t *starts_dev, int32_t *slots_out_dev,
                           void *stream);
/* ... and the bracket around such a reader on `stream` (e.g. fdql_agent_update): begin != 0 flushes staged add()s and orders
 * the reader behind writes issued on other streams; begin == 0, after the reader was enqueued, makes later writes on other
 * streams wait for it - what a sample call does for its own gather (see "threads" above).  No-ops on one stream. */
int fdql_ring_external_read(fdql_ring_t *ring, int32_t begin, void *stream);

/* replay_memory.py:54-70 temporal_sample(): out[k] is [T, B, dims[k]] f32 (device, caller
 * owned), out[k][t,b,:] = ring[k][(start[b] + t) % len, :].
 *   starts_dev != NULL : int64 [B] window starts supplied by the caller (parity runs);
 *   starts_dev == NULL : start[b] = Philox4x32-10(seed, counter, b) mapped to [0, len-T)
 *                        (replay_memory.py:59 uses numpy's global MT19937 instead).
 *   starts_out_dev     : optional int64 [B] that receives the starts used.
 * Returns FDQL_EOVERSAMPLE when len < 2T or len < B (replay_memory.py:57-58). */
int fdql_ring_sample_windows(fdql_ring_t *ring, int32_t T, int32_t B, const int64_t *starts_dev,
                             uint64_t seed, uint64_t counter, float *const *out_dev_ptrs,
                             int64_t *starts_out_dev, void *stream);
/* Same gather with per-key sub-row selection, for the read side of HER-vmap
 * (franQ/Replay/wrappers/her_vmap.py:104-123 picks ONE of the K+1 stored virtual columns):
 *   out_dev_ptrs[k] == NULL      : key k is not gathered;
 *   sel_off / sel_dim (or NULL)  : out[k] is [T, B, sel_dim[k]] = ring[k][row, sel_off[k] : sel_off[k]+sel_dim[k]]
 *                                  (sel_dim[k] <= 0 means the whole row). */
int fdql_ring_sample_windows_sel(fdql_ring_t *ring, int32_t T, int32_t B, const int64_t *starts_dev,
                                 uint64_t seed, uint64_t counter, float *const *out_dev_ptrs,
                                 const int32_t *sel_off, const int32_t *sel_dim, int64_t *starts_out_dev,
                                 void *stream);
/* replay_memory.py:48-52 sample(): out[k] is [B, dims[k]]; idx as for starts (range [0,len)). */
int fdql_ring_sample_rows(fdql_ring_t *ring, int32_t B, const int64_t *idx_dev, uint64_t seed,
                          uint64_t counter, float *const *out_dev_ptrs, int64_t *idx_out_dev,
                          void *stream);

/* replay_memory.py:67-70 __getitem__(): out[k] is [n, dims[k]] = ring[k][idx[i], :] for explicit
 * slot indices (int64, device), each in [0, maxlen): slots are addressed up to maxlen whatever
 * len() is, like numpy indexing of the reference's [maxlen, ...] arrays, and there is no
 * OversampleError.  The caller normalises negative indices and rejects idx >= maxlen
 * (IndexError in the reference); the kernel itself reduces mod maxlen, so no index can read
 * outside the ring. */
int fdql_ring_gather_rows(fdql_ring_t *ring, int64_t n, const int64_t *idx_dev, float *const *out_dev_ptrs,
                          void *stream);

/* ------------------------------------------------------------------------------------ */
/* Write-time episode transforms, on device                                              */
/* replaces franQ/Replay/wrappers/nstep_return.py:60-72 (calculate_montecarlo_return)    */
/*          franQ/Replay/wrappers/her.py:55-95 (_hindsight_flush)                        */
/* ------------------------------------------------------------------------------------ */
/* ret[i] = r[i] + gamma * ret[i+1] over one episode given OLDEST-FIRST rewards [n] (device). */
int fdql_episode_mc_return(const float *reward_dev, float *ret_dev, int32_t n, float gamma, void *stream);

/* Sparse L2 goal reward (same shape as franQ/Env/bitflip.py:143-152,
 * franQ/Env/classic_control_goal/classic_goal.py:88-93):
 *   R(ag,g) = ||ag-g||_2 > thr ? miss_reward : 0 ;  done = (R == 0).                     */
typedef struct {
  int32_t kind;      /* 0 = sparse L2 threshold */
  float threshold;
  float miss_reward; /* -1 in the reference envs */
} fdql_reward_fn_t;

/* her.py:55-95 for one finished episode, oldest-first device arrays:
 *   reward[n], episode_step[n], achieved_goal[n,g], desired_goal[n,g], goal[g]
 * writes reward_out[n], task_done_out[n], episode_step_out[n] (f32).                    */
int fdql_episode_her_relabel(const float *reward, const float *episode_step, const float *achieved_goal,
                             const float *desired_goal, const float *goal, int32_t n, int32_t goal_dim,
                             const fdql_reward_fn_t *fn, float *reward_out, float *task_done_out,
                             float *episode_step_out, void *stream);

/* Write-path ingestion: one finished episode -> ring, in ONE call.
 * Replaces the per-record emit of HindsightNStepReplay -> NStepReturn -> ReplayMemory.add
 * (franQ/Replay/wrappers/her.py:28-95, nstep_return.py:24-57, replay_memory.py:37-46; driven one dict
 * at a time by Runner._replay_handler, Runner/runner.py:177-191): the episode's packed rows
 * [n, row_floats] (oldest first, host memory) are staged through pinned memory with one H2D copy, the
 * n-step scan and the hindsight relabel run on the device, and the resulting rows are scattered into
 * the SoA ring in exactly the order the reference's wrapper stack would have added them:
 *   [ _pop's duplicate of record 0 with the return over the first n_step rewards (quirk q3), iff
 *     emit_pop and n > n_step ] + the n records with mc_return;  then, iff her, the same block again
 *   for the hindsight copy (desired_goal := achieved_goal[goal_row]; reward, task_done,
 *   episode_step relabelled as in fdql_episode_her_relabel; mc_return over the relabelled rewards).
 * The mc_return column of the input rows is ignored.  *appended = ring rows written.  No host sync
 * (the pinned staging buffer of the previous episode is waited for before it is reused).       */
typedef struct {
  int32_t reward_key;     /* key index of "reward" (width 1)                                         */
  int32_t return_key;     /* key index of "mc_return" (width 1), or -1: no n-step return              */
  int32_t emit_pop;       /* 1: also emit NStepReturn._pop's record (nstep_return.py:33-34, 50-57)    */
  int32_t n_step;         /* conf.nStep_return_steps                                                  */
  float gamma;
  int32_t her;            /* 0: real records only; 1: append the hindsight copy (her.py:55-95)        */
  int32_t achieved_key, desired_key, task_done_key, step_key;
  int32_t goal_row;       /* record whose achieved_goal is the virtual goal ("final": n-1)            */
  fdql_reward_fn_t reward_fn;
} fdql_episode_spec_t;
int fdql_ring_append_episode(fdql_ring_t *ring, const float *host_rows, int64_t n, const fdql_episode_spec_t *spec,
                             int64_t *appended, void *stream);

/* her_vmap.py:30-45,66-90 for one finished episode (oldest-first device arrays): K virtual goals
 * achieved_goal[goal_idx[k]]; per step i and virtual goal k
 *   r'[i,k] = (reward[i] - R(ag_i, dg_i)) + R(ag_i, g_k),
 *   d'[i,k] = (task_done[i] && !done(ag_i, dg_i)) || done(ag_i, g_k),
 * plus the real column last.  Outputs (f32): virtual_goals[n, (K+1)*g] (the K goals then the step's own
 * desired goal), virtual_rewards[n, K+1], virtual_dones[n, K+1].  Shim-pinned: the reference file needs jax,
 * which the build image lacks; it ran on a numpy-backed jax stand-in and the kernel is checked against the
 * vectors it produced (tests/golden/her_vmap.npz).                                                      */
int fdql_episode_her_vmap(const float *reward, const float *task_done, const float *achieved_goal,
                          const float *desired_goal, const int32_t *goal_idx, int32_t n, int32_t goal_dim,
                          int32_t K, const fdql_reward_fn_t *fn, float *virtual_goals, float *virtual_rewards,
                          float *virtual_dones, void *stream);
/* nstep_return_vmap.py:61-74 per column (oldest-first rows [n, cols]):
 *   ret[i] = r[i] + ret[i+1] * gamma * done[i]      (quirk q10: multiplies by done, not 1-done) */
int fdql_episode_mc_return_vmap(const float *rewards, const float *dones, float *ret, int32_t n, int32_t cols,
                                float gamma, void *stream);

/* A finished episode of the "vmap" hindsight stack (franQ/Replay/wrappers/her_vmap.py:66-90 over nstep_return_vmap.py:26-74) in
 * ONE call: the packed host rows [n, row_floats] (oldest first; the virtual columns' contents are ignored) are staged through
 * pinned memory with the K goal indices (her_vmap.py:75 draws them on the host: numpy's global generator), copied to the device
 * once, relabelled IN PLACE there (fdql_episode_her_vmap's arithmetic on the rows' own columns), given their per-column returns
 * (vreturn_key >= 0; with the one-shot _pop record of quirk q3 in front when n > n_step) and scattered into the ring - what
 * HindsightVmapWrite -> NStepReturnVmap -> ReplayMemory.add would have written record by record.  *appended = n (+ 1). */
typedef struct {
  int32_t reward_key, task_done_key, achieved_key, desired_key;   /* keys of the episode's own columns            */
  int32_t vgoals_key, vrewards_key, vdones_key;                   /* [(K+1)*g], [K+1], [K+1]: written here        */
  int32_t vreturn_key;                                            /* [K+1] or -1 (no n-step wrapper underneath)   */
  int32_t K, n_step;
  float gamma;
  fdql_reward_fn_t reward_fn;
} fdql_episode_vmap_spec_t;
int fdql_ring_append_episode_vmap(fdql_ring_t *ring, const float *host_rows, int64_t n, const int32_t *goal_idx_host,
                                  const fdql_episode_vmap_spec_t *spec, int64_t *appended, void *stream);

/* ------------------------------------------------------------------------------------ */
/* Agent update: one SAC/TQC gradient step                                               */
/* replaces franQ/Agent/deepQlearning.py:105-127 (train_step), :198-258 (get_losses),    */
/*          components/distributional_soft_actor_critic.py:40-103,                       */
/*          components/soft_actor_critic.py:63-154, models/{mlp,gaussian_mlp}.py,        */
/*          utils/common.py:10-19 (polyak) and torch.optim.Adam (deepQlearning.py:100)   */
/* ------------------------------------------------------------------------------------ */
typedef struct fdql_agent fdql_agent_t;

#define FDQL_MAX_HIDDEN 4
#define FDQL_MAX_CONV 3

typedef struct {
  /* shapes — franQ/Agent/conf.py:8-98, encoder.py:26-32 */
  int32_t obs_dim, goal_dim, act_dim;
  int32_t discrete;                 /* 1: Gumbel-softmax actor over act_dim actions          */
  int32_t n_critics, n_quantiles;   /* conf.num_critics, conf.num_q_predictions              */
  int32_t latent, enc_features;     /* conf.latent_state_dim, EncoderConf.hidden_features    */
  int32_t n_enc_hidden, enc_hidden[FDQL_MAX_HIDDEN];
  int32_t n_joint_hidden, joint_hidden[FDQL_MAX_HIDDEN];
  int32_t n_pi_hidden, pi_hidden[FDQL_MAX_HIDDEN];
  int32_t n_critic_hidden, critic_hidden[FDQL_MAX_HIDDEN];
  /* Pixel observations (BASELINE config 5).  THE REFERENCE HAS NO SUCH ENCODER (encoder.py:16-23 is dead code): this is
   * a design of this build - batch.obs_2d [T,B,img_c,img_h,img_w] (values 0..255) is scaled by 1/255 and run through
   * n_conv strided convolutions (conv_out channels, conv_k x conv_k kernels, stride conv_s, no padding) with
   * LeakyReLU(0.01); the last feature map, flattened in (y, x, channel) order, is fed to the obs MLP next to obs_1d /
   * the goals.  img_c == 0: no pixel input.  obs_dim may then be 0.                                                 */
  int32_t img_c, img_h, img_w, n_conv;
  int32_t conv_out[FDQL_MAX_CONV], conv_k[FDQL_MAX_CONV], conv_s[FDQL_MAX_CONV];
  /* encoder joiner: 0 = SkipHeadMLP (EncoderConf.JoinerModeEnum.feedforward), 1 = one nn.GRU layer scanned
   * over the T axis (JoinerModeEnum.gru; encoder.py:40-42, 78-94).  gru_state_mode = where the scan starts
   * (EncoderConf.RnnLatentStateTrainMode): 0 zero, 1 store (batch.agent_state[0]), 2 learned
   * (parameter encoder.hidden_state repeated over the batch).  With the GRU, is_contiguous becomes its
   * cumulative product over t (a window counts up to its first break). */
  int32_t joiner_gru, gru_state_mode;
  /* algorithm switches */
  int32_t distributional;           /* conf.use_distributional_sac                           */
  int32_t use_lowerbound;           /* conf.use_nStep_lowerbounds                            */
  int32_t use_max_entropy;          /* conf.use_max_entropy_q                                */
  int32_t hard_updates;             /* conf.use_hard_updates                                 */
  int32_t keep_frozen_copy;         /* materialise critic_frozen (state_dict parity)         */
  int32_t bootstrap_nstep;          /* conf.use_bootstrap_minibatch_nstep (soft_actor_critic.py:102-132,
                                       deepQlearning.py:226-228): window-long n-step lower bound on
                                       q(t=0); needs !distributional && use_lowerbound, as in the reference */
  int32_t obs_2d_u8;                /* img_c > 0: the batch carries the frames as uint8 (fdql_batch_t.obs_2d_u8), as the ring stores
                                       them (replay_memory.py:26-35 keeps the source dtype), not widened to float32               */
  int32_t burn_in_steps;            /* EncoderConf.use_burn_in: int(T * burn_in_portion) leading rows of
                                       is_contiguous are zeroed (deepQlearning.py:219-220); 0 = off        */
  /* batch geometry: this rank's [T, B, *] minibatch; loss is normalised by B*world_size    */
  int32_t T, B, world_size;
  /* hyper-parameters */
  /* doubles: the reference keeps them as Python floats (e.g. 1 - beta2 is formed in double) */
  double gamma, tau, lr, beta1, beta2, adam_eps, init_log_alpha, drop_frac;
} fdql_agent_config_t;

int fdql_agent_create(fdql_agent_t **out, const fdql_agent_config_t *cfg);
int fdql_agent_destroy(fdql_agent_t *agent);

/* Arena sizes in floats: which = 0 trainable (encoder|actor|critics|log_alpha; also the
 * size of the grad / Adam m / Adam v arenas), 1 targets (actor_target|critic_target),
 * 2 frozen (critic_frozen).  Tensor starts are padded to 4 floats.                       */
int64_t fdql_agent_arena_floats(const fdql_agent_t *agent, int32_t which);
/* Enumerate tensors with the reference's state_dict names.  Returns the
 * number of tensors when index < 0.  shape has 2 entries (bias: [n,0]; scalar: [0,0]). */
int32_t fdql_agent_tensor_info(const fdql_agent_t *agent, int32_t index, char *name, int32_t name_cap,
                               int32_t *arena, int64_t *offset_floats, int32_t *shape2);
int64_t fdql_agent_workspace_bytes(const fdql_agent_t *agent);
/* 1 when the pixel encoder's first layer runs on the implicit-GEMM kernel (csrc/conv.hip) and may therefore read the ring's
 * uint8 block in place (fdql_batch_t.obs_2d_slots); 0: the batch must carry the frames themselves. */
int32_t fdql_agent_conv_reads_ring(const fdql_agent_t *agent);

/* Bind caller-owned device memory.  `frozen` may be NULL unless keep_frozen_copy.
 * grads/adam_m/adam_v/workspace are zeroed here (synchronises the device once).         */
int fdql_agent_bind(fdql_agent_t *agent, float *params, float *grads, float *adam_m, float *adam_v,
                    float *targets, float *frozen, void *workspace, int64_t workspace_bytes);

/* One [T,B,*] f32 minibatch as TorchDataLoader.temporal_sample() would return it
 * (torch_dataloader.py:40-50).  achieved/desired_goal NULL when goal_dim == 0;
 * mc_return NULL when !use_lowerbound.                                                  */
typedef struct {
  const float *obs_1d, *achieved_goal, *desired_goal, *action;
  const float *reward, *mc_return, *task_done, *episode_step;
  const float *obs_2d;        /* [T,B,img_c,img_h,img_w] pixel frames as float32 (0..255), iff img_c > 0 && !cfg.obs_2d_u8      */
  const uint8_t *obs_2d_u8;   /* cfg.obs_2d_u8: the frames as uint8, the ring's storage type: a [T,B,img_c,img_h,img_w] batch
                                 (obs_2d_slots == NULL) or the ring's own block of the key (fdql_ring_key_ptr_u8) read in place  */
  const int32_t *obs_2d_slots;/* ... through one slot index per row, int32 [T,B] (fdql_ring_window_slots); needs
                                 fdql_agent_conv_reads_ring() != 0                                                               */
  const float *agent_state;   /* [T,B,latent]: hidden state the actor had at each step (runner.py:157); read iff
                                 joiner_gru && gru_state_mode == 1, row block t = 0 only (encoder.py:83-84)     */
} fdql_batch_t;

#define FDQL_PHASE_ALL 0   /* loss + backward + Adam + polyak                               */
#define FDQL_PHASE_GRAD 1  /* ... up to the gradient arena (then all-reduce it)             */
#define FDQL_PHASE_APPLY 2 /* Adam + polyak from the gradient arena                          */
/* Data-parallel agents (world_size > 1) can take FDQL_PHASE_GRAD in two parts, so that the all-reduce of the first bucket
 * runs beside the second part (the reference averages nothing: one process, deepQlearning.py:105-127):
 *   FDQL_PHASE_GRAD_CRITICS: ... up to the critics' (and log_alpha's) gradients - arena floats [bucket, n) are final;
 *   FDQL_PHASE_GRAD_REST (batch = NULL): the actor / encoder backward - arena floats [0, bucket) are final.
 * FDQL_PHASE_GRAD = both.  bucket: fdql_agent_grad_bucket (= n when the agent was created with world_size 1: nothing early). */
#define FDQL_PHASE_GRAD_CRITICS 3
#define FDQL_PHASE_GRAD_REST 4

/* train_step() for one shard.  noise_target / noise_actor: the draws the reference takes
 * from torch's global RNG in that order (gaussian_mlp.py:31 / ExpRelaxedCategorical):
 * N(0,1) [T-1,B,act] (continuous) or U(0,1) [T-1,B,act] (discrete).  NULL = generate on
 * device with Philox(seed, step).  No host synchronisation.                              */
int fdql_agent_update(fdql_agent_t *agent, const fdql_batch_t *batch, const float *noise_target,
                      const float *noise_actor, uint64_t seed, int32_t phase, void *stream);

int fdql_agent_grad_bucket(fdql_agent_t *agent, int64_t *first_early_float);

/* How FDQL_PHASE_ALL is issued (franQ/Agent/deepQlearning.py:105-127 is one Python-level step; here it is a fixed launch list):
 * graph = 0: one hipLaunchKernel per stage (default; FDQL_GRAPH=1 in the environment at create time makes 1 the default),
 * graph = 1: the launch list of each plan replayed as ONE hipGraphLaunch (captured on the plan's second run; bit-identical
 * results).  Which is faster depends on the step (17-23 dependent launches of 8-40 us: host issue time matters) and on the
 * host: callers decide by timing both (bench.py / NativeAgent.calibrate_launch_mode).  Split-phase calls are always eager. */
int fdql_agent_set_launch_mode(fdql_agent_t *agent, int32_t graph);

/* Scalars of the last update (device -> host copy; synchronises `stream`):
 * [0] loss (deepQlearning.py:249)  [1] mean q_loss  [2] mean pi_loss  [3] mean alpha_loss
 * [4] q_pred mean  [5] mc-constraint violation rate  [6] alpha used  [7] optimiser step   */
int fdql_agent_scalars(fdql_agent_t *agent, float *host_out8, void *stream);
/* The rest of what the reference's trainer logs every log_interval steps (deepQlearning.py:114-122, 231-247;
 * distributional_soft_actor_critic.py:65-67), computed on demand from the last update's buffers - never inside a step
 * (device -> host copy; synchronises `stream`): [0] q_pred.var(-1).mean()  [1] Valid_Portion mean  [2] max  [3] min;
 * with_grad_norms: [4 + i] = L2 norm of the gradient of trainable tensor i (fdql_agent_tensor_info order, arena 0), as the
 * gradient arena holds it (after FDQL_PHASE_GRAD / FDQL_PHASE_ALL).  Returns the number of floats written (< 0: error). */
int fdql_agent_summaries(fdql_agent_t *agent, float *host_out, int32_t cap, int32_t with_grad_norms, void *stream);
/* curr_alpha carried between steps (soft_actor_critic.py:41,152). */
int fdql_agent_set_alpha(fdql_agent_t *agent, float alpha, void *stream);
int fdql_agent_set_step(fdql_agent_t *agent, int32_t step, void *stream);

/* Named intermediate of the last update, for parity tests ("state", "next_action",
 * "next_log_pi", "next_z", "q_pred", "pi", "log_pi", "q_frozen", "q_loss", "pi_loss",
 * "alpha_loss", "is_contiguous", "td_target").  Pointer into the workspace.             */
int fdql_agent_debug_ptr(fdql_agent_t *agent, const char *name, const float **dev_ptr, int64_t *count);

/* franQ.Agent.DeepQLearning.act (Agent/deepQlearning.py:155-187; called by Runner._agent_handler,
 * Runner/runner.py:133): encoder.forward_eval (components/encoder.py:52-76, feed-forward joiner)
 * -> actor (models/gaussian_mlp.py:15-39 | gumbel_mlp.py:7-54) on `rows` observations (one per env
 * instance), reading the ONLINE weights straight from the bound parameter arena - the actors see
 * the trainer's current weights with no state_dict hop (the reference ships a full state_dict
 * through a queue every 50 steps, deepQlearning.py:136-148).
 *   obs_1d [rows, obs_dim] iff obs_dim > 0; achieved_goal / desired_goal [rows, goal_dim] iff goal_dim > 0;
 *   obs_2d [rows, img_c, img_h, img_w] iff img_c > 0;
 *   exploit_mask [rows] bytes (1 = return the greedy action; Runner/runner.py:120-123) or NULL;
 *   noise [rows, act_dim]: N(0,1) draws (continuous) / U(0,1) draws (discrete) or NULL -> Philox4x32
 *   keyed by (seed, counter);
 *   action [rows, act_dim] (continuous) or [rows] action index as float (discrete) =
 *   exploit*mask + explore*!mask; log_prob [rows], explore_action, exploit_action: optional (the
 *   reference's `info` dict).  GRU joiner: agent_state [rows, latent] is the hidden state the runner carries
 *   (NULL = zeros) and hidden_state [rows, latent] receives the next one (encoder.py:72-76); both are NULL
 *   for the feed-forward joiner, whose hidden state is None in the reference's return triple.  workspace: caller-owned device scratch of at least
 *   fdql_agent_act_workspace_bytes(agent, rows) bytes, 16-byte aligned, private to this call
 *   stream (it may run beside fdql_agent_update on another stream; it then reads whatever mix of
 *   pre/post-step weights is in the arena, like any lock-free actor).  No host sync.          */
int64_t fdql_agent_act_workspace_bytes(const fdql_agent_t *agent, int32_t rows);
int fdql_agent_act(fdql_agent_t *agent, const float *obs_1d, const float *achieved_goal, const float *desired_goal,
                   const float *obs_2d, const float *agent_state, const uint8_t *exploit_mask, const float *noise,
                   uint64_t seed, uint64_t counter, int32_t rows, float *action, float *log_prob, float *explore_action,
                   float *exploit_action, float *hidden_state, void *workspace, int64_t workspace_bytes, void *stream);

/* Algorithmic work of one update, for roofline accounting: dense GEMM flops
 * (2*MAC) and the number/flops of launches by kind. */
typedef struct {
  double gemm_flops;       /* useful dense flops issued through the MFMA GEMM kernel         */
  double skinny_flops;     /* head layers (N<=32)                                            */
  int32_t n_launches;
  int32_t n_gemm_launches;
  int64_t params;
  int64_t plans_built;     /* launch plans built so far (one per new set of batch pointers; a handful are cached) */
  int64_t graph_launches;  /* updates replayed as one hipGraph launch (opt-in: FDQL_GRAPH=1 at create) */
} fdql_agent_stats_t;
int fdql_agent_stats(const fdql_agent_t *agent, fdql_agent_stats_t *out);

/* Time every kernel launch of the next update with hipEvents on `stream` (synchronises):
 * writes up to cap entries (name, milliseconds, flops, bytes). */
typedef struct {
  char name[48];
  float ms;
  double flops;
  double bytes;
} fdql_kernel_time_t;
int32_t fdql_agent_profile_update(fdql_agent_t *agent, const fdql_batch_t *batch, const float *noise_target,
                                  const float *noise_actor, uint64_t seed, fdql_kernel_time_t *out,
                                  int32_t cap, void *stream);

/* ------------------------------------------------------------------------------------ */
/* Test hook: the grouped MFMA GEMM on its own (C = A * op(B) + bias, fp32)               */
int fdql_test_gemm(const float *A, int32_t lda, int32_t a_kc, const float *B, int32_t ldb, int32_t b_kc,
                   const float *bias, float *C, int32_t ldc, int32_t M, int32_t N, int32_t K,
                   int32_t epilogue, const float *ref, int32_t ldref, int32_t ksplit, void *stream);

/* Test hook: one SkipHeadMLP forward (franQ/Agent/models/mlp.py:88-94) through the row-block chain kernel
 * (csrc/chain.hip).  weights: the MLP's tensors packed as the agent's arena packs them (per hidden layer W [h, in]
 * then b [h], then head W [dout, din + sum(h)] and b [dout]; each padded to a multiple of 4 floats).
 * h_out: NULL or nh device pointers [rows, h_i].  Synchronises `stream`. */
int fdql_test_chain_mlp(const float *x, int32_t rows, int32_t din, const int32_t *hid, int32_t nh, int32_t dout,
                        const float *weights, float *const *h_out, float *out, void *stream);
/* Diagnostic: fdql_debug_chain_stamps(NULL, 1) switches the recording on (NULL, 0: off): the middle workgroup of every chain
 * launch then records the shader clock at its entry, after its program fetch and after each operation; with a buffer: returns
 * the count copied. */
int fdql_debug_chain_stamps(uint64_t *out, int32_t cap);

/* Test hook: one launch of the weight-stationary row-block kernel (csrc/wstat.hip) over `ninst` instances of a Linear layer
 * with 256 inputs (+ up to two narrow input blocks of k1, k2 <= 8 columns) and 256 outputs; instance i owns rows
 * [i*M, (i+1)*M) of every array.  ks: weights stored [k][n] (dgrad) else [n][k]; grad: LeakyReLU' gate from `ref` and
 * column sums instead of bias + LeakyReLU; dual: C = f(all but the last narrow block), C2 = f(all); hf_*: head fusion
 * (GemmProblem::hf_* in csrc/common.h); fz_h != null: the head dgrad of the layer above fused into the loader - A0's rows
 * are then OUTPUT, LeakyReLU'(fz_h) * (A1 . fz_w[i]) with A1 = dY (k1 = 2), their per-64-row column sums in fz_colsum.
 * planes: head-sum planes per instance in hf_out; NEGATIVE: the kernel adds a tile's planes itself (the update's default) and
 * plane 0 holds the total.  Asynchronous on `stream`.  FDQL_EINVAL: the kernel does not take the form. */
int fdql_test_rowgemm(const float *A0, const float *A1, int32_t k1, const float *A2, int32_t k2, const float *W0, int32_t ldw0,
                      const float *W1, const float *W2, const float *bias, float *C, float *C2, const float *ref, float *colsum,
                      const float *hf_w, int32_t hf_ldw, int32_t hf_q, float *hf_out, float *hf_out2, int32_t M, int32_t ninst,
                      int32_t ks, int32_t grad, int32_t dual, int32_t planes, const float *fz_h, const float *fz_w, int32_t fz_ldw,
                      float *fz_colsum, void *stream);

/* Test hook: one launch of the output-stationary weight-gradient kernel (csrc/wgrad.hip): nprob blocks
 * dW[i] [256, ldw] = G[i]^T X[i] over M rows (G, X: [nprob * M, 256] row-major), block i's slab 0 at dW + i * 256 * ldw, the
 * nslab K-split slabs slab_stride floats apart: every slab is written (a workgroup's partial result, or zeros) and their sum
 * is the gradient (franQ: autograd's addmm backward of mlp.py:88-94).  Asynchronous on `stream`. */
int fdql_test_wgrad_stat(const float *G, const float *X, float *dW, int32_t M, int32_t nprob, int32_t ldw, int32_t nslab,
                         int64_t slab_stride, void *stream);
/* The same with riders (csrc/wgrad.h): block i (every x2_every-th when X2 is given) also produces
 *   dW2[i] [256, ldw2] (columns < nx2 <= 8) = G[i]^T X2[i]   (X2: [nprob * M, ldx2], the block's few extra input columns) and
 *   dW3[i] [ng2 <= 4, ldw3] (256 columns)    = G2[i]^T X[i]   (G2: [nprob * M, ldg2], a few extra output rows over the same input)
 * in the K-split slabs like dW.  X2 / G2 may be null. */
int fdql_test_wgrad_stat_riders(const float *G, const float *X, float *dW, int32_t M, int32_t nprob, int32_t ldw, int32_t nslab,
                                int64_t slab_stride, const float *X2, int32_t nx2, int32_t ldx2, float *dW2, int32_t ldw2, int32_t x2_every,
                                const float *G2, int32_t ng2, int32_t ldg2, float *dW3, int32_t ldw3, void *stream);

/* Test hook for the implicit-GEMM convolutions of the pixel encoder (csrc/conv.hip; BASELINE config 5 - the reference's conv
 * branch is dead code, franQ/Agent/components/encoder.py:16-23: the parity target is torch conv2d).  One layer on nimg images:
 *   mode 0  forward        out[nimg, OH*OW, cout] = LeakyReLU(conv(in, W) + bias)                                  (NHWC)
 *   mode 1  data gradient  out[nimg, H*W, C] = LeakyReLU'(act_prev) * conv^T(dpre, W); `in` unused
 *   mode 2  weight grad.   out[cout*K + cout] = (dW, db) = sum over images and positions; scratch = partial slabs
 * in: u8 != 0: uint8 NCHW frames - [nimg, C*H*W] back to back (slots == NULL) or a ring block [*, C*H*W] read through one
 *     int32 slot index per image (fdql_ring_window_slots); W is [cout, K] with K = (c, ky, kx);
 *     u8 == 0: float32 NHWC maps [nimg, H, W, C]; K = (ky, kx, c).
 * FDQL_EINVAL when the layer geometry has no kernel instantiation.  Asynchronous on `stream`. */
int fdql_test_conv(int32_t mode, const void *in, int32_t u8, const int32_t *slots,
                   const float *W, const float *bias, const float *dpre, const float *act_prev, float *out, float *scratch,
                   int64_t scratch_floats, int64_t nimg, int32_t C, int32_t H, int32_t Wd, int32_t k, int32_t s, int32_t cout,
                   void *stream);

/* Diagnostic (tools/dp_overlap.py): a stand-in for the channel kernels of a collective - `workgroups` workgroups of 256
 * threads (each holding lds_bytes of LDS) copy n floats from src to dst `passes` times (the ring steps of an all-reduce
 * re-read their chunk) and then stay resident until hold_us microseconds have passed since they started (a collective's
 * workgroups mostly wait for their peers and links while they occupy their compute units).  Launched on a
 * side stream beside the update's launches it shows, on ONE GPU, whether a small kernel gets compute units while the
 * persistent one-workgroup-per-CU kernels run and what it costs them (the data-parallel step of SURVEY 8e all-reduces the
 * gradient arena beside FDQL_PHASE_GRAD_REST).  Asynchronous on `stream`. */
int fdql_debug_side_copy(const float *src, float *dst, int64_t n, int32_t workgroups, int32_t passes, int32_t hold_us,
                         int32_t lds_bytes, void *stream);

/* Tuning / test hook: tile shape of the dense problems: 5 = 64x64 (default), 3 = 64x128, 0 = 128x128, 7 = the small-batch
 * kernel (csrc/smallgemm.hip) on every problem that has its form, whatever the size.  Applies to plans built afterwards
 * and to fdql_test_gemm. */
int fdql_debug_set_gemm_dense_shape(int32_t shape);

#ifdef __cplusplus
}
#endif
#endif /* FDQL_H */
