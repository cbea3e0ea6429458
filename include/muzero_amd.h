/*
 * muzero_amd - C ABI of the MI355X (gfx950) MuZero self-play search path.
 *
 * Every entry point takes raw device pointers, sizes and a hipStream_t, is
 * asynchronous on that stream, and returns 0 on success or a negative MZ_E_*
 * code (mz_last_error() then holds a message for the calling thread).  Nothing
 * throws across this boundary.  All device memory is owned by the caller; the
 * library allocates none.
 *
 * The reference (JannisFengler/MuZero) is pure Python and has no FFI; each
 * entry point below names the reference function it replaces (file:line in
 * the reference checkout).
 */
#ifndef MUZERO_AMD_H
#define MUZERO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mz_stream_t; /* a hipStream_t; NULL = the default stream */

#define MZ_OK 0
#define MZ_E_ARG -1     /* bad argument (null pointer, size out of range) */
#define MZ_E_LAUNCH -2  /* the HIP runtime rejected a launch */
#define MZ_E_STATE -3   /* object used in the wrong state */

#define MZ_MAX_ACTIONS 64  /* one wavefront lane per action */
#define MZ_MAX_NODES 1024  /* expanded nodes per tree: the backup stages two doubles per node per wavefront in LDS */

int mz_version(void);
const char* mz_last_error(void);

/* ------------------------------------------------------------------ *
 * Search trees: struct-of-arrays for B games in ONE caller-owned
 * workspace.  Replaces the per-node Python objects of model.py:373-416
 * (Node) and model.py:351-370 (MinMaxStats).
 *
 * A game's tree holds at most max_nodes expanded nodes (node 0 = root;
 * one node is added per mz_tree_expand_backup call).  An edge record
 * (parent node, action) carries what the reference keeps in an unexpanded
 * child Node: prior, visit_count, value_sum; `child` is the expanded-node
 * id behind the edge or -1.
 * ------------------------------------------------------------------ */
typedef struct mz_tree_layout {
  int32_t B, A, max_nodes, reserved;
  int64_t total_bytes;
  /* byte offsets into the workspace */
  int64_t child;        /* int32  [B][max_nodes][A] */
  int64_t visit;        /* int32  [B][max_nodes][A] */
  int64_t value_sum;    /* double [B][max_nodes][A] */
  int64_t prior;        /* float  [B][max_nodes][A] */
  int64_t root_prior64; /* double [B][A]   noised root priors */
  int64_t reward;       /* double [B][max_nodes]  reward stored in the node */
  int64_t num_nodes;    /* int32  [B] */
  int64_t root_visit;   /* int32  [B] */
  int64_t root_value_sum; /* double [B] */
  int64_t minmax;       /* double [B][2]  {minimum, maximum} */
  int64_t root_is_f64;  /* int32  [B]  1 = root row scored in double */
  int64_t path_node;    /* int32  [B][max_nodes] nodes a descent selected at */
  int64_t path_action;  /* int32  [B][max_nodes] action taken at path_node */
  int64_t path_len;     /* int32  [B] */
  int64_t leaf_parent;  /* int32  [B] node whose edge leaf_action is unexpanded */
  int64_t leaf_action;  /* int32  [B] */
  int64_t status;       /* int32  [B] 0 ok, 1 no selectable child, 2 tree full, 3 score tables shorter than a visit count */
} mz_tree_layout;

/* Host-only: fills *out for (B, A, max_nodes). 1 <= A <= MZ_MAX_ACTIONS, 2 <= max_nodes <= MZ_MAX_NODES. */
int mz_tree_layout_query(int32_t B, int32_t A, int32_t max_nodes, mz_tree_layout* out);

/* Root construction for B games: Node(0).expand(...) with reward 0
 * (train.py:240-249, model.py:391-403) and, if noise != NULL,
 * add_exploration_noise (model.py:405-416): prior' =
 * double(float(prior * keep_f32)) + noise * frac, scored in double from then
 * on.  priors: float [B][A] (softmax of the initial policy logits);
 * noise: double [B][A] Dirichlet rows or NULL; keep_f32 = (float)(1 - frac).
 * Also resets visit counts, value sums and the min/max statistics
 * (a fresh MinMaxStats per root, train.py:261). */
int mz_tree_root_init(void* ws, const mz_tree_layout* L, const float* priors,
                      const double* noise, float keep_f32, double frac,
                      mz_stream_t stream);

/* One descent per game from the root to the first unexpanded edge with the
 * statistics as they are: the `while node.is_expanded()` loop of
 * model.py:439-442 with select_child (model.py:482-497) and ucb_score
 * (model.py:499-513), including its mixed float32 / float64 arithmetic and the
 * lowest-action tie break.  pbc_table[n] = log((n + c_base + 1) / c_base) +
 * c_init and sqrt_table[n] = sqrt(n), both double [table_len], computed on
 * the host with libm so that they equal the reference's math.log / math.sqrt;
 * table_len must exceed the largest visit count a node can reach.
 * Results go to path_*, leaf_parent, leaf_action in the workspace. */
int mz_tree_select(void* ws, const mz_tree_layout* L, const double* pbc_table,
                   const double* sqrt_table, int32_t table_len, mz_stream_t stream);

/* Expansion of the selected leaf from network output and `wave` backups along
 * the selected path: Node.expand (model.py:391-403) + backpropagate
 * (model.py:515-527) as run() applies them per expansion (model.py:459-469).
 * priors float [B][A]; value, reward float [B] (already support-decoded);
 * wave >= 1 is the number of identical backups (run() with num_simulations =
 * wave performs `wave` identical descents on frozen statistics). */
int mz_tree_expand_backup(void* ws, const mz_tree_layout* L, const float* priors,
                          const float* value, const float* reward, int32_t wave,
                          double gamma, mz_stream_t stream);

/* mz_tree_expand_backup for simulation s followed by mz_tree_select for simulation s + 1 in
 * one launch (both work wavefront-per-game on the same game, so no grid-wide ordering is
 * needed).  Same results as the two calls in sequence; saves a launch per simulation.  Trees of
 * at most 64 nodes and 1024 edge records (max_nodes * A) are copied to LDS first and walked there. */
int mz_tree_expand_backup_select(void* ws, const mz_tree_layout* L, const float* priors, const float* value,
                                 const float* reward, int32_t wave, double gamma, const double* pbc_table,
                                 const double* sqrt_table, int32_t table_len, mz_stream_t stream);

/* MCTS.ucb_score (model.py:499-513) for n independent edges, with the arithmetic the descent uses (the same
 * device function): scores[i] from the parent's visit count, the child's visit count and value sum, and the
 * MinMaxStats bounds (minimum > maximum = nothing recorded yet).  Exactly one of prior32 (float priors: float32
 * product and sum, returned widened to double) and prior64 (noised-root priors: double throughout) is given.
 * The caller guarantees 0 <= parent_visits[i] < table_len (device data). */
int mz_ucb_score(const int32_t* parent_visits, const int32_t* child_visits, const double* value_sum,
                 const double* minimum, const double* maximum, const float* prior32, const double* prior64,
                 const double* pbc_table, const double* sqrt_table, int32_t table_len, int32_t n, double* scores,
                 mz_stream_t stream);

/* Search result per game (model.py:472-480 and Node.value, model.py:386-389):
 * visits int32 [B][A], root_value double [B], probs double [B][A]
 * (uniform when no child was visited). Any output may be NULL. */
int mz_tree_readout(const void* ws, const mz_tree_layout* L, int32_t* visits,
                    double* root_value, double* probs, mz_stream_t stream);

/* select_action (model.py:530-546) for B roots. temperature == 0: first
 * maximum.  Otherwise p = counts^(1/T) / sum, cdf = cumsum(p) / cdf[-1],
 * action = #(cdf <= u) with u[b] the uniform [0,1) draw np.random.choice
 * would consume.  Bit-exact for 1/T in {1,2,3,4} (integer powers). */
int mz_select_action(const int32_t* visits, int32_t B, int32_t A, double temperature,
                     const double* u, int32_t* actions, mz_stream_t stream);

/* ------------------------------------------------------------------ *
 * Networks.  Activations are NHWC; BatchNorm is applied in inference
 * form (running statistics) folded into the convolution weights and
 * bias by the host (model.py:143-175 blocks in eval mode).
 * ------------------------------------------------------------------ */

/* fp32 3x3 convolution, padding 1, stride 1 or 2, any shape: the validation
 * precision of ResidualBlock / DownsampleBlock / conv_state (model.py:146-148,
 * :165, :182, :213).  out = act(conv(x) + bias [+ residual]).
 * w: float [9][Cin][Cout] (tap = ky*3+kx); bias float [Cout].
 * in_slot (optional, int32 [B]): image b is read from image index
 * in_slot[b] * slot_stride + b of `in` (gather of parent hidden states from
 * the per-game node pool, replacing torch.cat at model.py:452).
 * action_vec (optional, float [n_actions][Cin]) with actions int32 [B]:
 * x = in + action_vec[actions[b]] inside the image before the convolution,
 * which equals hidden + conv_action(one_hot planes) (model.py:231-236).
 * Preconditions the caller guarantees (device data the host wrapper cannot see):
 * 0 <= actions[b] < n_actions and in_slot[b] * slot_stride + b inside `in`. */
int mz_conv3x3_f32(const float* in, const int32_t* in_slot, int32_t slot_stride, const float* w,
                   const float* bias, const float* residual, float* out, const float* action_vec,
                   const int32_t* actions, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                   int32_t stride, int32_t relu, mz_stream_t stream);

/* bf16 MFMA 3x3 convolution, the throughput precision: same contract as
 * mz_conv3x3_f32 with bf16 NHWC activations (in, residual, out), float32
 * accumulation and bias, and weights pre-packed in MFMA fragment order:
 *   wpk[((cc*9 + tap)*KC + kc)*NT + nt][lane][j] (bf16) =
 *     w[cout = nt*16 + lane%16][cin = cc*chunk + kc*32 + 8*(lane/16) + j][tap]
 * with KC = chunk/32, NT = Cout/16 and `chunk` from mz_conv3x3_bf16_info; the
 * buffer must be followed by 2*NT*1024 readable bytes (prefetch overrun).
 * mean_out (optional, float [B][Cout]): spatial mean of the output, only for
 * shapes whose kernel holds a whole image (whole_image = 1).
 * Kernels exist for the reference architecture's layer shapes (96x96x96->256 s2,
 * 48x48x256->128 s2, 24x24x128->128, 24x24x128->256 s2, 12x12x256->256); any
 * other shape returns MZ_E_ARG. */
int mz_conv3x3_bf16_info(int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t stride, int32_t* cin_chunk,
                         int32_t* whole_image);
int mz_conv3x3_bf16(const void* in, const int32_t* in_slot, int32_t slot_stride, const void* wpk,
                    const float* bias, const void* residual, void* out, const float* action_vec,
                    const int32_t* actions, float* mean_out, int32_t B, int32_t H, int32_t W, int32_t Cin,
                    int32_t Cout, int32_t stride, int32_t relu, mz_stream_t stream);

/* The reference architecture's first convolution (96x96, 3*D = 96 input channels -> 256,
 * stride 2; model.py:182) reading the rotating, tiled observation of mz_obs_slot_write
 * directly.  wpk: fragments packed as for mz_conv3x3_bf16 from weights whose input channel
 * 3s + r is the reference's channel 3*((s - newest - 1) mod D) + r.  out: bf16 NHWC
 * [B][48][48][256] = ReLU(conv + bias) when relu != 0. */
int mz_conv3x3_bf16_rot(const void* obs_rot, const void* wpk, const float* bias, void* out, int32_t B,
                        int32_t relu, mz_stream_t stream);

/* mz_conv3x3_bf16_rot with the weight copy chosen on the device: wpk_all holds one packed copy per possible
 * newest slot, copy_elems bf16 elements apart (a multiple of 8), and *newest (device int32, 0 <= *newest < D,
 * guaranteed by the caller) picks the copy when the kernel runs - so a recorded HIP graph of the search stays
 * valid while the ring advances. */
int mz_conv3x3_bf16_rot_sel(const void* obs_rot, const void* wpk_all, int64_t copy_elems, const int32_t* newest,
                            const float* bias, void* out, int32_t B, int32_t relu, mz_stream_t stream);

/* A chain of 12x12 256->256 convolutions fused into one kernel with the image
 * resident in LDS: the whole dynamics network (conv_state, bn, resblocks;
 * model.py:236-241) or the 12x12 tail of the representation network
 * (model.py:201-202 and the last downsample stage's blocks).  Layer l reads
 * image buffer l%2 and writes buffer (l+1)%2; add_residual adds what the
 * destination buffer still holds (the input of the residual block's first
 * convolution) before the ReLU.  wpk / bias as for mz_conv3x3_bf16.  in_slot,
 * action_vec + actions and mean_out as for mz_conv3x3_bf16 (applied to the
 * first layer's input and the last layer's output).  n_layers <= 20. */
typedef struct mz_chain_layer {
  const void* wpk;
  const float* bias;
  int32_t relu;
  int32_t add_residual;
} mz_chain_layer;
int mz_conv_chain_bf16(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer* layers,
                       int32_t n_layers, void* out, const float* action_vec, const int32_t* actions,
                       float* mean_out, int32_t B, mz_stream_t stream);

/* The residual blocks of the first downsample stage (24x24 images, 128 -> 128 channels:
 * DownsampleBlock.blocks, model.py:167, :173-174; ResidualBlock, model.py:143-159) as one kernel with the
 * image resident in LDS, every layer written over its own input.  in / out: bf16 NHWC [B][24][24][128];
 * layers as for mz_conv_chain_bf16 (add_residual on layer l adds the input of layer l - 1, so it cannot be
 * set on layer 0); n_layers <= 8. */
int mz_conv_chain_mid_bf16(const void* in, const mz_chain_layer* layers, int32_t n_layers, void* out, int32_t B,
                           mz_stream_t stream);

/* ------------------------------------------------------------------ *
 * The same kernels with operands in IEEE half ("f16": 11 significant bits against bf16's 8, the same
 * v_mfma_f32_16x16x32 rate).  Same contracts as the bf16 entry points they are named after
 * (representation / dynamics / prediction forward, model.py:195-204, :228-245, :273-281), with half where
 * those take bf16: activations (in, residual, out), packed weights (mz_weights_fold_pack mode 3, the same
 * fragment order), the rotating observation (mz_obs_slot_write format 3).  float32 accumulation, bias and
 * spatial mean as before.  Results that exceed the largest finite half (65 504) are stored as that value
 * instead of infinity; BatchNorm-folded weights and post-ReLU activations of the reference architecture are
 * orders of magnitude below it.
 * ------------------------------------------------------------------ */
int mz_conv3x3_f16(const void* in, const int32_t* in_slot, int32_t slot_stride, const void* wpk,
                   const float* bias, const void* residual, void* out, const float* action_vec,
                   const int32_t* actions, float* mean_out, int32_t B, int32_t H, int32_t W, int32_t Cin,
                   int32_t Cout, int32_t stride, int32_t relu, mz_stream_t stream);
int mz_conv3x3_f16_rot(const void* obs_rot, const void* wpk, const float* bias, void* out, int32_t B,
                       int32_t relu, mz_stream_t stream);
int mz_conv3x3_f16_rot_sel(const void* obs_rot, const void* wpk_all, int64_t copy_elems, const int32_t* newest,
                           const float* bias, void* out, int32_t B, int32_t relu, mz_stream_t stream);
int mz_conv_chain_f16(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer* layers,
                      int32_t n_layers, void* out, const float* action_vec, const int32_t* actions,
                      float* mean_out, int32_t B, mz_stream_t stream);
int mz_conv_chain_mid_f16(const void* in, const mz_chain_layer* layers, int32_t n_layers, void* out, int32_t B,
                          mz_stream_t stream);

/* ------------------------------------------------------------------ *
 * float32-accurate networks on the bf16 matrix cores ("bf16x3"): a float32 tensor travels as
 * bf16 pairs hi = bf16(x), lo = bf16(x - hi), laid out [image][2 = hi, lo][pixels][channels], and
 * every product is w_hi*x_hi + w_lo*x_hi + w_hi*x_lo accumulated in float32 (error ~2^-17
 * relative per product).  This is the precision in which initial_inference / recurrent_inference
 * (model.py:332-348) meet the reference's float32 outputs to 1e-3 at matrix-core speed.
 * Weights: two fragment arrays in mz_conv3x3_bf16's packing, from bf16(w) and bf16(w - bf16(w)).
 * ------------------------------------------------------------------ */

/* float32 [n_images][per_image] -> split pairs [n_images][2][per_image] (bf16) and back (hi + lo, exact). */
int mz_split_f32(const float* x, void* out, int64_t n_images, int64_t per_image, mz_stream_t stream);
int mz_join_f32(const void* x, float* out, int64_t n_images, int64_t per_image, mz_stream_t stream);

/* mz_conv3x3_bf16 on split tensors for the four layer shapes outside the 12x12 trunk (model.py:165-167, :182):
 * out = act(conv(in) + bias [+ residual]), bias / residual / ReLU applied in float32 before the split. */
int mz_conv3x3_bf16x3(const void* in, const void* wpk_hi, const void* wpk_lo, const float* bias, const void* residual,
                      void* out, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t stride,
                      int32_t relu, mz_stream_t stream);

/* mz_conv_chain_bf16 on split tensors: the dynamics network or the 12x12 tail of the representation network
 * (model.py:236-241, :201-202) as one kernel with a game's (hi, lo) pair resident in LDS.  in_slot, action_vec +
 * actions (added in float32) and mean_out (of the float32 outputs) as for mz_conv_chain_bf16. */
typedef struct mz_chain_layer_x3 {
  const void* wpk_hi;
  const void* wpk_lo;
  const float* bias;
  int32_t relu;
  int32_t add_residual;
} mz_chain_layer_x3;
int mz_conv_chain_bf16x3(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer_x3* layers,
                         int32_t n_layers, void* out, const float* action_vec, const int32_t* actions,
                         float* mean_out, int32_t B, mz_stream_t stream);

/* "f16x2": the float32-accurate path at two MFMAs per product - the same (hi, lo) pairs in IEEE half (22 significant
 * bits together) with the weights rounded to half once: w * x = w_hi * x_hi + w_hi * x_lo.  Same contracts as the
 * bf16x3 entry points above (model.py:332-348 to 1e-3), one weight array per layer (mz_weights_fold_pack mode 3;
 * mz_chain_layer instead of mz_chain_layer_x3), tensors split by mz_split_f32_half. */
int mz_split_f32_half(const float* x, void* out, int64_t n_images, int64_t per_image, mz_stream_t stream);
int mz_join_f32_half(const void* x, float* out, int64_t n_images, int64_t per_image, mz_stream_t stream);
int mz_conv3x3_f16x2(const void* in, const void* wpk, const float* bias, const void* residual, void* out, int32_t B,
                     int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t stride, int32_t relu, mz_stream_t stream);
int mz_conv_chain_f16x2(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer* layers,
                        int32_t n_layers, void* out, const float* action_vec, const int32_t* actions, float* mean_out,
                        int32_t B, mz_stream_t stream);

/* x.mean(dim=(2,3)) (model.py:243, :275): float NHWC [B][HW][C] -> [B][C]. */
int mz_spatial_mean_f32(const float* in, float* out, int32_t B, int32_t HW, int32_t C, mz_stream_t stream);

/* One fully connected head with a single hidden layer (model.py:217-224,
 * :253-269), weights packed in float32 MFMA fragment order:
 *   w1p[(t*(C/4) + ks)*64 + lane]  = W1[t*16 + lane%16][4*ks + lane/16]   (Hd/16 tiles)
 *   w2p[(t*(Hd/4) + ks)*64 + lane] = W2[t*16 + lane%16][4*ks + lane/16]   (ceil(O/16) tiles,
 *                                                       rows >= O are zero)
 * with W1 [Hd][C], W2 [O][Hd] the torch Linear weights; b1 float [Hd];
 * b2 float [ceil(O/16)*16] zero padded.  With bf16 != 0 the two weight buffers
 * hold 16-bit fragments for the 16x16x32 MFMA instead - bf16 (1, the throughput precision) or IEEE half (2):
 *   w1p[(t*(C/32) + ks)*64 + lane][j] = W1[t*16 + lane%16][32*ks + 8*(lane/16) + j]
 * (same for w2p with Hd); C and Hd must then be multiples of 32, and the layer inputs (spatial means, hidden
 * layer) are rounded to the same format. */
typedef struct mz_head_weights {
  const void* w1p;
  const float* b1;
  const void* w2p;
  const float* b2;
  int32_t bf16;
  int32_t reserved;
} mz_head_weights;

/* Policy, value and (optional) reward heads on the spatial mean [B][C], fused
 * with what the search consumes: priors = softmax(policy logits)
 * (model.py:397), value / reward = support_to_scalar(logits, S)
 * (model.py:98-109, :399, :467).  Logit outputs ([B][A], [B][2S+1]) are optional
 * (NULL = not written). reward_w == NULL (initial inference) skips the reward. */
int mz_heads(const float* mean, int32_t B, int32_t C, int32_t Hd, int32_t A, int32_t S,
             const mz_head_weights* policy, const mz_head_weights* value_w, const mz_head_weights* reward_w,
             float* priors, float* value, float* reward, float* policy_logits, float* value_logits,
             float* reward_logits, mz_stream_t stream);

/* support_to_scalar (model.py:98-109): float logits [n][2S+1] -> float [n]. */
int mz_support_to_scalar(const float* logits, int32_t n, int32_t S, float* out, mz_stream_t stream);

/* softmax(policy_logits, dim=1) as Node.expand applies it (model.py:397):
 * float logits [n][A] -> float priors [n][A]; the same code the heads use. */
int mz_policy_softmax(const float* logits, float* priors, int32_t n, int32_t A, mz_stream_t stream);

/* Layout conversion between the reference's NCHW float tensors and NHWC. */
int mz_nchw_to_nhwc_f32(const float* in, float* out, int32_t B, int32_t C, int32_t HW, mz_stream_t stream);
int mz_nhwc_to_nchw_f32(const float* in, float* out, int32_t B, int32_t C, int32_t HW, mz_stream_t stream);

/* ------------------------------------------------------------------ *
 * Frame-stack preprocessing (CustomFrameStack train.py:22-63,
 * prepare_observation train.py:529-540).
 * ------------------------------------------------------------------ */

/* ring: uint8 [B][D][frame_bytes] device ring of the last D frames per game;
 * frames: uint8 [B][frame_bytes] the new frame of every game (device memory,
 * typically the target of a pinned async copy).  Writes slot `slot`, or every
 * slot when fill_all != 0 (reset replicates the first frame, train.py:49-50).
 * fill_mask (optional, int32 [B]): games with a non-zero entry are reset that way while the
 * others push - concurrent games end their episodes at different moves. */
int mz_ring_push(uint8_t* ring, const uint8_t* frames, int32_t B, int32_t D, int64_t frame_bytes, int32_t slot,
                 int32_t fill_all, const int32_t* fill_mask, mz_stream_t stream);

/* Stacked, normalised observation from the ring: out[b][pixel][age*3 + rgb] =
 * frame(age)[pixel][rgb] / 255 with age 0 = oldest = slot (newest_slot+1) % D
 * (train.py:62-63 then :534-540). out is NHWC: float32 (out_is_bf16 = 0), bf16 (1), IEEE half (3), or the
 * float32-accurate precisions' (hi, lo) pairs of the float32 value, [B][2][HW][3*D], in bf16 (2) or half (4); the
 * pair formats need 3*D % 8 == 0 and HW % 64 == 0. */
int mz_ring_to_obs(const uint8_t* ring, void* out, int32_t out_is_bf16, int32_t B, int32_t D, int32_t HW,
                   int32_t newest_slot, mz_stream_t stream);

/* Rotating, tiled form of the stacked observation: obs (float: out_is_bf16 = 0, bf16: 1, IEEE half: 3)
 * [B][H][W/16][D][56] holds, for row y
 * and 16-pixel tile of x, frame slot s as one run of 17 pixels x 3 values + 5 values of padding: the last pixel
 * of the tile to the left (zero for a row's first tile) followed by the tile's 16 pixels.  This call writes
 * frames uint8 [B][H*W][3], normalised by /255, into slot `slot` (all D slots when fill_all != 0, or for the
 * games flagged in the optional fill_mask int32 [B]: the reset rule, train.py:49-50), whole runs at a time
 * (halo and zero padding included).  W must be a multiple of 16; frames and obs 16-byte aligned.
 * With `newest` the last slot written, slot s holds what CustomFrameStack's output (train.py:62-63) has at
 * channels 3*((s - newest - 1) mod D) + {0,1,2}; a consumer (mz_conv3x3_bf16_rot) reads it through
 * input-channel weights permuted the same way. */
int mz_obs_slot_write(const uint8_t* frames, void* obs, int32_t out_is_bf16, int32_t B, int32_t D, int32_t H,
                      int32_t W, int32_t slot, int32_t fill_all, const int32_t* fill_mask, mz_stream_t stream);

/* prepare_observation (train.py:529-540) for a batch: uint8 NHWC [B][HW][C]
 * -> float32 NCHW [B][C][HW], x / 255 (bit-identical to the float64 divide
 * followed by the float32 cast for all 256 byte values). */
int mz_prepare_observation(const uint8_t* obs_nhwc, float* out_nchw, int32_t B, int32_t C, int32_t HW,
                           mz_stream_t stream);

/* The observations of a training batch from device-resident episode frames (Game.get_observation + the
 * prepare_observation of compute_losses, train.py:128-129, :343, :529-540; model.py:603): frames uint8
 * [positions][HW][3] holds the single new frame of every stored position, episodes back to back; game g's first
 * frame is frames[frame_off[g]] (int64 [n_games + 1]).  Sample b = position q_pos[b] of game q_game[b]: the D
 * frames ending there, the episode's first frame repeated in front (train.py:49-50), oldest first.  out: float32
 * NCHW [n][3*D][HW], x / 255.  The caller guarantees 0 <= q_pos[b] < length of game q_game[b]. */
int mz_gather_stacked_obs(const uint8_t* frames, const int64_t* frame_off, const int32_t* q_game, const int32_t* q_pos,
                          int32_t n, int32_t D, int32_t HW, float* out_nchw, mz_stream_t stream);

/* ------------------------------------------------------------------ *
 * Weight hand-over on the device: from the float32 parameters of a state_dict (e.g. views into
 * the flat buffer a weight broadcast filled) to what the kernels read.  The reference shares one
 * network object between training and self-play (train.py:155); here the search engine's copy is
 * refreshed by these calls after every update, without the parameters visiting the host.
 * ------------------------------------------------------------------ */

/* One 3x3 convolution (+ the BatchNorm behind it, when its four vectors are given: inference form,
 * model.py:143-175 in eval mode; fold in double as scale = gamma / sqrt(var + eps)):
 *   mode 0: bf16 fragments in mz_conv3x3_bf16's packing, followed by its two k-steps of zeros;
 *   mode 1: the same for the residue w - bf16(w) (the lo part of the bf16x3 precision);
 *   mode 2: float32 [tap][Cin][Cout] (mz_conv3x3_f32);
 *   mode 3: IEEE-half fragments in mode 0's order (mz_conv3x3_f16 and the other f16 entry points).
 * transpose != 0 packs W'[o][i][tap] = w[i][o][8 - tap], the weights with which a convolution over the
 * output gradient gives the input gradient (square layers, no BatchNorm).  rot_depth = D > 0 writes D copies
 * mz_packed_conv_elems(...) elements apart, copy r with input channel j taken from channel
 * 3*((j/3 - r - 1) mod D) + j%3 (mz_conv3x3_bf16_rot).  bias_out (optional): float [Cout] folded bias.
 * Results equal the host's torch arithmetic bit for bit. */
int64_t mz_packed_conv_elems(int32_t Cout, int32_t Cin, int32_t mode);
int mz_weights_fold_pack(const float* w, const float* conv_bias, const float* bn_weight, const float* bn_bias,
                         const float* bn_mean, const float* bn_var, double eps, int32_t Cout, int32_t Cin,
                         int32_t cin_chunk, int32_t mode, int32_t transpose, int32_t rot_depth, void* out,
                         float* bias_out, mz_stream_t stream);

/* mz_weights_fold_pack(mode 0, no BatchNorm, no rotation) for n weights of one shape in a single launch: w[i]
 * float [Cout][Cin][3][3] -> out[i] (mz_packed_conv_elems(Cout, Cin, 0) bf16 elements), in the transposed,
 * mirrored data-gradient form where transpose[i] != 0 (transpose may be null).  w, out, transpose: HOST arrays
 * of n entries (device pointers inside).  The unrolled training step (train.py:303-313) changes every weight
 * once per update; its native convolutions need both forms of all of them again. */
int mz_weights_pack_many(const float* const* w, void* const* out, const int32_t* transpose, int32_t n, int32_t Cout,
                         int32_t Cin, int32_t cin_chunk, mz_stream_t stream);

/* A torch Linear layer (weight [O][K], bias [O]) in the fragment order of mz_head_weights (bf16 = 0: float32,
 * 1: bf16, 2: IEEE half) with the bias zero-padded to a multiple of 16. */
int mz_pack_linear(const float* w, const float* bias, int32_t O, int32_t K, int32_t bf16, void* w_packed,
                   float* bias_padded, mz_stream_t stream);

/* dynamics.conv_action (1x1, weight [C][A], bias [C]) as the per-action vectors the dynamics kernels add:
 * out[a][c] = w[c][a] + bias[c] (model.py:231-236). */
int mz_action_vectors(const float* w, const float* bias, int32_t C, int32_t A, float* out, mz_stream_t stream);

/* The weight broadcast itself, RCCL over xGMI behind the C ABI (the one collective of the path; the
 * reference shares one network object in one process, train.py:155).  RCCL is bound at run time, preferring the
 * copy already loaded in the process.  One rank calls mz_comm_unique_id (128 bytes) and hands the bytes to the
 * others by any side channel; every rank then calls mz_comm_init with its rank (its GPU current) and later
 * mz_comm_bcast_weights on the flat parameter buffer: in place, `bytes` bytes from rank `root` to all, on the
 * given stream.  MZ_E_STATE when RCCL cannot be loaded. */
int mz_comm_unique_id(void* id128);
int mz_comm_init(const void* id128, int32_t rank, int32_t world, void** comm_out);
int mz_comm_bcast_weights(void* comm, void* buffer, int64_t bytes, int32_t root, mz_stream_t stream);
int mz_comm_destroy(void* comm);

/* ------------------------------------------------------------------ *
 * Training side (the callers after the search path).
 * ------------------------------------------------------------------ */

/* scalar_to_support_loss (model.py:132-140) with scalar_to_support
 * (model.py:112-129) and transform_value (model.py:84-86) fused: float logits
 * [n][2S+1], float targets [n] -> loss float [n] = -(two_hot * log_softmax).sum(1)
 * and, when dlogits != NULL, d loss / d logits float [n][2S+1]. */
int mz_support_ce(const float* logits, const float* targets, int32_t n, int32_t S, float* loss, float* dlogits,
                  mz_stream_t stream);

/* Game.make_target / compute_target_value (train.py:89-118) for n_queries positions of a
 * device-resident episode store.  Episodes lie back to back: game g owns
 * rewards[reward_off[g] .. reward_off[g+1]) and values / policy rows
 * [value_off[g] .. value_off[g+1]) (double; policies double [*][A]; offsets int64 [n_games+1]).
 * (q_game, q_pos) int32 [n_queries].  Outputs double [n_queries][K+1]: n-step value targets,
 * reward targets (0 past the end) and, when policies / target_policies are given,
 * [n_queries][K+1][A] policy targets (zeros past the end). */
int mz_nstep_targets(const double* rewards, const double* values, const double* policies, const int64_t* reward_off,
                     const int64_t* value_off, int32_t n_games, int32_t A, const int32_t* q_game,
                     const int32_t* q_pos, int32_t n_queries, int32_t K, int32_t td_steps, double gamma,
                     double* target_values, double* target_rewards, double* target_policies, mz_stream_t stream);

/* What the unrolled training step (train.py:343-372) does with a hidden state besides convolving it, in one pass:
 * mean_out float [B][C] = its spatial mean (the input of the prediction heads, model.py:273-281) and x_out =
 * bf16(h + table[actions[b]]) on every pixel of image b - the dynamics network's "hidden + conv_action(one-hot
 * action planes)" (model.py:231-236).  h, x_out: bf16 NHWC [B][HW][C]; table: float [A][C] = conv_action's
 * weight column of each action + its bias (mz_action_vectors' layout); actions int32 [B], 0 <= actions[b] < A
 * guaranteed by the caller.  Either output may be NULL (x_out == NULL needs neither table nor actions).
 * Gradient: dh bf16 = dy + dmean / HW on every pixel (dy: gradient of x_out, dmean float [B][C]: gradient of
 * mean_out; either may be NULL), and, when dtable is given, dtable float [A][C] = sum over the images with that
 * action and over their pixels of dy (image_sums: caller-owned float [B][C] scratch).  C: 8 times a power of two,
 * 32 to 2048.  No atomics: the same bits on every run. */
int mz_step_input_bf16(const void* h, const float* table, const int32_t* actions, void* x_out, float* mean_out,
                       int32_t B, int32_t HW, int32_t C, mz_stream_t stream);
int mz_step_input_grad_bf16(const void* dy, const float* dmean, const int32_t* actions, int32_t B, int32_t HW, int32_t C,
                            int32_t A, void* dh, float* image_sums, float* dtable, mz_stream_t stream);

/* Weight gradient of a 3x3 convolution (padding 1) for the unrolled training step (compute_losses' backward
 * pass, train.py:303-308): dw float [Cout][Cin][3][3] = sum over images and output pixels of
 * grad_out[n][y][x][co] * x[n][s*y+ky-1][s*x+kx-1][ci]; grad_out (bf16 NHWC [B][H/s][W/s][Cout]) and x (bf16 NHWC
 * [B][H][W][Cin]), float32 accumulation; dbias (optional, float [Cout]) = sum over images and pixels of grad_out,
 * from the same pass.  H, W: the layer's INPUT size.  Kernels exist for the reference architecture's layer shapes
 * (model.py:165, :182, :146-148), the stride being the layer's: 12x12 256->256 (s = 1, 23 of the 31
 * convolutions), 96x96 96->256 (s = 2), 48x48 256->128 (s = 2), 24x24 128->128 (s = 1), 24x24 128->256 (s = 2);
 * other shapes return MZ_E_ARG.  workspace: caller-owned device memory of at least
 * mz_conv_wgrad_workspace_bytes_for(B, H, W, Cin, Cout) bytes (partial sums of the shares of the batch; 0 = no
 * kernel for the shape; mz_conv_wgrad_workspace_bytes(B) is the 12x12 shape's); dw and workspace 16-byte aligned;
 * a tensor at most 4 GiB.  The matching data gradient of a stride-1 layer is mz_conv3x3_bf16 applied to grad_out
 * with weights packed by mz_weights_fold_pack(transpose = 1) and a zero bias; of a stride-2 layer,
 * mz_conv_dgrad_s2_bf16. */
int64_t mz_conv_wgrad_workspace_bytes(int32_t B);
int64_t mz_conv_wgrad_workspace_bytes_for(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int mz_conv_wgrad_bf16(const void* grad_out, const void* x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                       int32_t Cout, float* dw, float* dbias, void* workspace, int64_t workspace_bytes,
                       mz_stream_t stream);
/* Data gradient of a stride-2 3x3 convolution (padding 1): dx bf16 NHWC [B][H][W][Cin] = sum over the taps (ky, kx)
 * that reach the pixel and over co of grad_out[n][(iy+1-ky)/2][(ix+1-kx)/2][co] * w[co][ci][ky][kx], grad_out bf16
 * NHWC [B][H/2][W/2][Cout], float32 accumulation (the backward pass of DownsampleBlock.conv, model.py:165, under
 * train.py:303-308).  H, W: the layer's INPUT size.  Kernels exist for the reference architecture's two stride-2
 * layers whose input needs a gradient: 48x48 256->128 and 24x24 128->256; other shapes return MZ_E_ARG.  wpk: the
 * layer's weights (float [Cout][Cin][3][3]) packed by mz_weights_pack_dgrad_s2 into a caller-owned buffer of
 * mz_packed_dgrad_s2_elems(Cout, Cin) bf16 elements (per parity class of (iy, ix), the taps that reach it, in
 * MFMA fragment order). */
int64_t mz_packed_dgrad_s2_elems(int32_t Cout, int32_t Cin);
int mz_weights_pack_dgrad_s2(const float* w, int32_t Cout, int32_t Cin, void* out, mz_stream_t stream);
int mz_conv_dgrad_s2_bf16(const void* grad_out, const void* wpk, void* dx, int32_t B, int32_t H, int32_t W, int32_t Cin,
                          int32_t Cout, mz_stream_t stream);

/* mz_conv_wgrad_bf16 for one layer applied to several batches - the dynamics network's convolutions run once per unroll
 * step with the same weights (train.py:356-372), and the weight gradient of the update is the sum over the
 * steps: grad_out[i], x[i] (host arrays of n_segments <= 8 device pointers, B images each) are treated as one
 * batch of n_segments * B images, so the sum costs one pass and one reduction instead of one per step.
 * workspace: mz_conv_wgrad_workspace_bytes(n_segments * B). */
int mz_conv_wgrad_bf16_multi(const void* const* grad_out, const void* const* x, int32_t n_segments, int32_t B,
                             int32_t H, int32_t W, int32_t Cin, int32_t Cout, float* dw, float* dbias, void* workspace,
                             int64_t workspace_bytes, mz_stream_t stream);

/* The same for SEVERAL 12x12 256->256 layers in one launch (and one reduction): the backward pass of an update
 * (train.py:303-308) produces the operands of the network's 23 such layers one after the other; computed together
 * after the pass, a workgroup gets tens of images of ONE layer instead of a handful, and the partial sums that
 * pass through memory shrink accordingly.  Per layer: host arrays of n_segments (<= 8) device pointers to grad_out
 * and x (B images each, summed as in mz_conv_wgrad_bf16_multi), dw float [256][256][3][3], dbias optional.  At
 * most 32 layers and 64 (layer, segment) pairs.  workspace: mz_conv_wgrad_group_workspace_bytes(images per layer =
 * n_segments * B, n_layers) bytes, 16-byte aligned. */
typedef struct mz_wgrad_layer {
  const void* const* grad_out;
  const void* const* x;
  int32_t n_segments;
  int32_t B;
  float* dw;
  float* dbias;
} mz_wgrad_layer;
int64_t mz_conv_wgrad_group_workspace_bytes(const int32_t* images_per_layer, int32_t n_layers);
int mz_conv_wgrad_bf16_group(const mz_wgrad_layer* layers, int32_t n_layers, void* workspace, int64_t workspace_bytes,
                             mz_stream_t stream);

/* Training-mode BatchNorm fused with its surroundings in the reference's blocks (ResidualBlock / DownsampleBlock /
 * DynamicsNetwork, model.py:143-175, :237, as compute_losses runs them, train.py:343, :411):
 *   y = act( batch_norm(x; batch statistics) [+ residual] ),  act = ReLU when relu != 0
 * x, residual, y: bf16 [rows][C] (NHWC / channels_last, rows = N*H*W).  Batch mean and biased variance over the
 * rows (accumulated in double); running_mean / running_var (optional) move by `momentum` with the unbiased
 * variance as torch.nn.BatchNorm2d does; save_mean / save_invstd float [C] are kept for the backward pass.
 * Backward: from dy (gradient w.r.t. y), y (for the ReLU mask) and x: dx, dgamma, dbeta and, when dresidual is
 * given, the gradient of the residual branch (dy masked by the ReLU).  workspace: mz_bn_workspace_bytes(C). */
int64_t mz_bn_workspace_bytes(int32_t C);
int mz_bn_relu_train_fwd(const void* x, const void* residual, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, double eps, double momentum, int32_t relu,
                         int64_t rows, int32_t C, void* y, float* save_mean, float* save_invstd, void* workspace,
                         int64_t workspace_bytes, mz_stream_t stream);
/* The same with the column sums supplied by the producer of x instead of a pass over x: stats double
 * [stat_blocks][2][C] = partial sums of x and of x * x (any split of the rows), e.g. per image from
 * mz_conv3x3_bf16_stats. */
int mz_bn_relu_train_fwd_stats(const void* x, const void* residual, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, double eps, double momentum, int32_t relu,
                               int64_t rows, int32_t C, void* y, float* save_mean, float* save_invstd,
                               const double* stats, int32_t stat_blocks, void* workspace, int64_t workspace_bytes,
                               mz_stream_t stream);
/* mz_conv3x3_bf16 for the 12x12, 256 -> 256, stride-1 layer without residual or ReLU (a convolution in front of
 * a training-mode BatchNorm, model.py:152-157 under train.py:343 / :411) that also writes, per image, the sums
 * the BatchNorm needs: stats double [B][2][256] = sum and sum of squares over the image's 144 pixels of the
 * bf16-rounded outputs. */
int mz_conv3x3_bf16_stats(const void* in, const void* wpk, const float* bias, void* out, double* stats, int32_t B,
                          mz_stream_t stream);
int mz_bn_relu_train_bwd(const void* x, const void* dy, const void* y, const float* gamma, const float* save_mean,
                         const float* save_invstd, int32_t relu, int64_t rows, int32_t C, void* dx, void* dresidual,
                         float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, mz_stream_t stream);
/* The same with accumulate != 0: dgamma and dbeta are ADDED to what the buffers hold (float32) - a layer applied at
 * every unroll step (the dynamics network's, train.py:356-372) sums its steps' gradients without a launch per
 * addition; accumulate == 0 is mz_bn_relu_train_bwd. */
int mz_bn_relu_train_bwd_acc(const void* x, const void* dy, const void* y, const float* gamma, const float* save_mean,
                             const float* save_invstd, int32_t relu, int64_t rows, int32_t C, void* dx,
                             void* dresidual, float* dgamma, float* dbeta, int32_t accumulate, void* workspace,
                             int64_t workspace_bytes, mz_stream_t stream);

/* ReplayBuffer.sample_batch's draw (model.py:580-588) on device-resident priorities: double prio [n], one per
 * stored position, oldest game first.  u: double [batch], the [0,1) draws np.random.choice would consume.
 * Outputs: flat int32 [batch] = #(cdf <= u) with cdf the normalised cumulative sum of prio / sum(prio)
 * (np.random.choice's inversion; uniform over the positions when every priority is zero, model.py:581-583);
 * weights double [batch] = (n * prob[flat]) ** -beta divided by their maximum (model.py:587-588); and, when
 * game_start (int64 [n_games + 1], first flat index of every game, game_start[n_games] = n) is given, the game
 * and in-game position each index falls into (the scan of model.py:592-596).  The cumulative sums are formed in
 * a fixed two-level order rather than left to right, so they equal the reference's to rounding and an index can
 * differ only for a draw within a few ulp of a boundary of the distribution.
 * workspace: caller-owned device memory of at least mz_replay_workspace_bytes(n) bytes, 8-byte aligned (chunk
 * sums and their scan); its contents need not be preserved between calls. */
int64_t mz_replay_workspace_bytes(int64_t n);
int mz_replay_sample(const double* prio, int64_t n, const double* u, int32_t batch, double beta,
                     const int64_t* game_start, int32_t n_games, int32_t* flat, int32_t* game, int32_t* pos,
                     double* weights, void* workspace, int64_t workspace_bytes, mz_stream_t stream);

/* mz_replay_sample without a workspace argument and without any allocation: the chunk sums and their scan go
 * to a file-scope device array inside the library (128 KiB, held by every GPU the library has run on until the
 * process ends).  No counterpart in the reference; it is for callers that cannot allocate device memory or
 * carry a workspace around - code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  n must not exceed mz_replay_static_max_positions() (16 Mi positions; MZ_E_ARG otherwise).  All
 * calls on one GPU share the array: the results of a call are complete once its stream work has finished, but
 * two calls running at the same time on different streams of one GPU race on it.
 * The array's device address is looked up once per GPU (hipGetSymbolAddress, a runtime query that does not belong
 * inside a stream capture): mz_replay_static_max_positions() does it for the current device, so call that once
 * per GPU at set-up; without it the first mz_replay_sample_static on a GPU does it, and that first call must
 * then be made outside capture. */
int64_t mz_replay_static_max_positions(void);
int mz_replay_sample_static(const double* prio, int64_t n, const double* u, int32_t batch, double beta,
                            const int64_t* game_start, int32_t n_games, int32_t* flat, int32_t* game,
                            int32_t* pos, double* weights, mz_stream_t stream);

/* ReplayBuffer.update_priorities (model.py:612-614): prio[flat[i]] = (values[i] + 1e-6) ** alpha for i in list
 * order (of several entries naming one slot the last wins).  flat holds FLAT position indices as
 * mz_replay_sample returns them (the reference passes in-game offsets here, model.py:595 / :607, and so writes
 * other slots than it sampled; that is not reproduced).  Entries outside [0, n) are ignored.  Linear in m:
 * the entries bid for their slot with their list index (an integer atomic on the slot), the highest writes. */
int mz_replay_update(double* prio, int64_t n, const int32_t* flat, const double* values, int32_t m, double alpha,
                     mz_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MUZERO_AMD_H */
