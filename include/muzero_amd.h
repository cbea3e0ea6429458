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
  int64_t status;       /* int32  [B] 0 ok, 1 no selectable child, 2 tree full */
} mz_tree_layout;

/* Host-only: fills *out for (B, A, max_nodes). 1 <= A <= MZ_MAX_ACTIONS. */
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

#ifdef __cplusplus
}
#endif
#endif /* MUZERO_AMD_H */
