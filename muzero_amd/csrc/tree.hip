// Device-resident search trees: one wavefront (64 lanes) per game, one lane
// per action.  Replaces the per-node Python tree of the reference
// (model.py:351-527).  This file must be compiled with -ffp-contract=off: the
// selection score and the backup reproduce the reference's float32 / float64
// arithmetic step by step, and a fused multiply-add would change the rounding.
#pragma clang fp contract(off)

#include <cstdlib>

#include "common.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWavesPerBlock * 64;

struct TreeView {
  int B, A, N;  // N = max_nodes
  int* child;
  int* visit;
  double* value_sum;
  float* prior;
  double* root_prior64;
  double* reward;
  int* num_nodes;
  int* root_visit;
  double* root_value_sum;
  double* minmax;
  int* root_is_f64;
  int* path_node;
  int* path_action;
  int* path_len;
  int* leaf_parent;
  int* leaf_action;
  int* status;
};

TreeView make_view(void* ws, const mz_tree_layout* L) {
  char* p = static_cast<char*>(ws);
  TreeView t;
  t.B = L->B; t.A = L->A; t.N = L->max_nodes;
  t.child = reinterpret_cast<int*>(p + L->child);
  t.visit = reinterpret_cast<int*>(p + L->visit);
  t.value_sum = reinterpret_cast<double*>(p + L->value_sum);
  t.prior = reinterpret_cast<float*>(p + L->prior);
  t.root_prior64 = reinterpret_cast<double*>(p + L->root_prior64);
  t.reward = reinterpret_cast<double*>(p + L->reward);
  t.num_nodes = reinterpret_cast<int*>(p + L->num_nodes);
  t.root_visit = reinterpret_cast<int*>(p + L->root_visit);
  t.root_value_sum = reinterpret_cast<double*>(p + L->root_value_sum);
  t.minmax = reinterpret_cast<double*>(p + L->minmax);
  t.root_is_f64 = reinterpret_cast<int*>(p + L->root_is_f64);
  t.path_node = reinterpret_cast<int*>(p + L->path_node);
  t.path_action = reinterpret_cast<int*>(p + L->path_action);
  t.path_len = reinterpret_cast<int*>(p + L->path_len);
  t.leaf_parent = reinterpret_cast<int*>(p + L->leaf_parent);
  t.leaf_action = reinterpret_cast<int*>(p + L->leaf_action);
  t.status = reinterpret_cast<int*>(p + L->status);
  return t;
}

__device__ __forceinline__ int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

// max over the 64 lanes of a wavefront, returned in every lane.  Four data-parallel-primitive
// steps reduce each row of 16 lanes (xor 1, xor 2, half-row mirror, row mirror: every lane of a
// row ends with the row's maximum), four scalar lane reads combine the rows; about a third of
// the latency of six cross-lane permutes through the LDS crossbar.  No NaN inputs.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                          __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_max_f64(double v) {
  double o;
  o = dpp_f64<0xB1>(v);  v = o > v ? o : v;   // quad_perm [1,0,3,2]
  o = dpp_f64<0x4E>(v);  v = o > v ? o : v;   // quad_perm [2,3,0,1]
  o = dpp_f64<0x141>(v); v = o > v ? o : v;   // row_half_mirror
  o = dpp_f64<0x140>(v); v = o > v ? o : v;   // row_mirror
  const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
  const double ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}

// ---------------------------------------------------------------- root init
__global__ __launch_bounds__(kBlock) void root_init_kernel(TreeView t, const float* __restrict__ priors,
                                                          const double* __restrict__ noise, float keep,
                                                          double frac) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (b >= t.B) return;
  const size_t row = (size_t)b * t.N * t.A;  // node 0
  if (lane < t.A) {
    const float p = priors[(size_t)b * t.A + lane];
    t.prior[row + lane] = p;
    t.visit[row + lane] = 0;
    t.value_sum[row + lane] = 0.0;
    t.child[row + lane] = -1;
    if (noise != nullptr) {
      const float kept = p * keep;  // float32 product, as prior * (1 - frac) under NumPy promotion
      t.root_prior64[(size_t)b * t.A + lane] = (double)kept + noise[(size_t)b * t.A + lane] * frac;
    }
  }
  if (lane == 0) {
    t.num_nodes[b] = 1;
    t.root_visit[b] = 0;
    t.root_value_sum[b] = 0.0;
    t.minmax[2 * b] = __builtin_inf();
    t.minmax[2 * b + 1] = -__builtin_inf();
    t.root_is_f64[b] = noise != nullptr;
    t.reward[(size_t)b * t.N] = 0.0;
    t.path_len[b] = 0;
    t.status[b] = 0;
  }
}

// ------------------------------------------------------------------ score
// ucb_score (model.py:499-513) with MinMaxStats.normalize (model.py:364-370) and Node.value
// (model.py:386-389) for one edge, in the reference's own mix of float32 and float64: the
// exploration factor and the normalised value are doubles; with a float32 prior (every node but a
// noised root) the product and the sum are float32 (NumPy >= 2 scalar promotion), with a float64
// prior (noised root) they stay double.  pbc = log((N + c_base + 1) / c_base) + c_init and
// sqrt_n = sqrt(N) of the parent's visit count N come from host tables.
__device__ __forceinline__ double edge_score(double pbc, double sqrt_n, int n, double w, double lo, double hi,
                                             bool wide, float prior32, double prior64) {
  const double c = pbc * (sqrt_n / (double)(n + 1));
  double vs = 0.0;
  if (n > 0) vs = hi > lo ? (w / (double)n - lo) / (hi - lo) : 0.5;
  if (wide) return c * prior64 + vs;
  const float ps = (float)c * prior32;
  return (double)(ps + (float)vs);  // float32 sum, widened only to compare
}

__global__ void ucb_score_kernel(const int* __restrict__ parent_visits, const int* __restrict__ child_visits,
                                 const double* __restrict__ value_sum, const double* __restrict__ lo,
                                 const double* __restrict__ hi, const float* __restrict__ prior32,
                                 const double* __restrict__ prior64, const double* __restrict__ pbc,
                                 const double* __restrict__ sq, int n, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int pn = parent_visits[i];
  out[i] = edge_score(pbc[pn], sq[pn], child_visits[i], value_sum[i], lo[i], hi[i], prior64 != nullptr,
                      prior32 ? prior32[i] : 0.f, prior64 ? prior64[i] : 0.0);
}

// ------------------------------------------------------------------ select
__device__ __forceinline__ void select_game(const TreeView& t, const double* __restrict__ pbc,
                                            const double* __restrict__ sq, int table_len, int b, int lane) {
  const int A = t.A;
  const double lo = t.minmax[2 * b], hi = t.minmax[2 * b + 1];
  const bool root64 = t.root_is_f64[b] != 0;
  const double ninf = -__builtin_inf();

  int node = 0;
  int parent_n = t.root_visit[b];
  int depth = 0;
  int bad = 0;
  while (true) {
    const size_t row = ((size_t)b * t.N + node) * A;
    int n = 0, kid = -1;
    double score = ninf;
    if (lane < A) {
      n = t.visit[row + lane];
      kid = t.child[row + lane];
      const double w = t.value_sum[row + lane];
      const int pn = parent_n < table_len ? parent_n : table_len - 1;  // too-short tables are reported below
      const bool wide = root64 && node == 0;
      score = edge_score(pbc[pn], sq[pn], n, w, lo, hi, wide, t.prior[row + lane],
                         wide ? t.root_prior64[(size_t)b * A + lane] : 0.0);
      if (!(score > ninf)) score = ninf;  // NaN and -inf never beat the initial -inf
    }
    // wavefront argmax, strict '>' in action order == lowest action among equals: the maximum,
    // then the first lane that holds it
    const double best = wave_max_f64(score);
    const unsigned long long holders = __ballot(score == best);
    int best_a = holders ? __builtin_ctzll(holders) : 0;
    if (parent_n >= table_len) bad = 3;
    if (!(best > ninf)) {  // the reference would return action -1 / child None here
      bad = 1;
      best_a = 0;
    }
    if (lane == 0) {
      t.path_node[(size_t)b * t.N + depth] = node;
      t.path_action[(size_t)b * t.N + depth] = best_a;
    }
    ++depth;
    const int next = __builtin_amdgcn_readlane(kid, best_a);  // best_a is wave-uniform: a scalar lane read, no LDS trip
    if (next < 0 || depth >= t.N) {
      if (lane == 0) {
        t.leaf_parent[b] = node;
        t.leaf_action[b] = best_a;
        t.path_len[b] = depth;
        if (bad) t.status[b] = bad;
      }
      break;
    }
    parent_n = __builtin_amdgcn_readlane(n, best_a);
    node = next;
  }
}

__global__ __launch_bounds__(kBlock) void select_kernel(TreeView t, const double* __restrict__ pbc,
                                                       const double* __restrict__ sq, int table_len) {
  const int b = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (b < t.B) select_game(t, pbc, sq, table_len, b, threadIdx.x & 63);
}

// ----------------------------------------------------------- expand + backup
// rwd / val: per-wave LDS arrays of max_nodes doubles (reward of each node on the path, value
// added at each node).  Returns false when the game's tree is full (status 2).
__device__ __forceinline__ bool expand_backup_game(const TreeView& t, const float* __restrict__ priors,
                                                   const float* __restrict__ value,
                                                   const float* __restrict__ reward, int wave, double gamma,
                                                   double* rwd, double* val, int b, int lane) {
  const int A = t.A;
  const int e = wave_uniform(t.num_nodes[b]);
  const int len = wave_uniform(t.path_len[b]);
  if (e >= t.N || len <= 0) {
    if (lane == 0) t.status[b] = 2;
    return false;
  }
  const int parent = wave_uniform(t.leaf_parent[b]);
  const int act = wave_uniform(t.leaf_action[b]);
  const double leaf_reward = (double)reward[b];

  // expansion: a fresh row of A edge records behind (parent, act)
  const size_t erow = ((size_t)b * t.N + e) * A;
  if (lane < A) {
    t.prior[erow + lane] = priors[(size_t)b * A + lane];
    t.visit[erow + lane] = 0;
    t.value_sum[erow + lane] = 0.0;
    t.child[erow + lane] = -1;
  }
  if (lane == 0) {
    t.child[((size_t)b * t.N + parent) * A + act] = e;
    t.reward[(size_t)b * t.N + e] = leaf_reward;
    t.num_nodes[b] = e + 1;
  }

  // rewards of the nodes on the path below the root: node behind edge d is
  // path_node[d + 1], or the new node for the last edge
  const int* pn = t.path_node + (size_t)b * t.N;
  const int* pa = t.path_action + (size_t)b * t.N;
  for (int d = lane; d < len; d += 64)
    rwd[d] = d == len - 1 ? leaf_reward : t.reward[(size_t)b * t.N + pn[d + 1]];
  __builtin_amdgcn_wave_barrier();

  // value = reward + gamma * value, leaf first (model.py:522-526); every lane
  // runs the same recurrence on LDS broadcasts
  double v = (double)value[b];
  for (int d = len - 1; d >= 0; --d) {
    if (lane == 0) val[d] = v;
    v = rwd[d] + gamma * v;
  }
  const double root_add = v;
  __builtin_amdgcn_wave_barrier();

  double lo = __builtin_inf(), hi = -__builtin_inf();
  for (int d = lane; d < len; d += 64) {
    const size_t idx = ((size_t)b * t.N + pn[d]) * A + pa[d];
    double ws = t.value_sum[idx];
    int n = t.visit[idx];
    const double add = val[d];
    for (int k = 0; k < wave; ++k) {
      ws += add;
      n += 1;
      const double q = ws / (double)n;
      lo = q < lo ? q : lo;
      hi = q > hi ? q : hi;
    }
    t.value_sum[idx] = ws;
    t.visit[idx] = n;
  }
  if (lane == 0) {
    double ws = t.root_value_sum[b];
    int n = t.root_visit[b];
    for (int k = 0; k < wave; ++k) {
      ws += root_add;
      n += 1;
      const double q = ws / (double)n;
      lo = q < lo ? q : lo;
      hi = q > hi ? q : hi;
    }
    t.root_value_sum[b] = ws;
    t.root_visit[b] = n;
  }
  lo = -wave_max_f64(-lo);
  hi = wave_max_f64(hi);
  if (lane == 0) {
    const double plo = t.minmax[2 * b], phi = t.minmax[2 * b + 1];
    t.minmax[2 * b] = lo < plo ? lo : plo;
    t.minmax[2 * b + 1] = hi > phi ? hi : phi;
  }
  return true;
}

// Dynamic LDS: per wave two double arrays of max_nodes entries.
__global__ __launch_bounds__(kBlock) void expand_backup_kernel(TreeView t, const float* __restrict__ priors,
                                                              const float* __restrict__ value,
                                                              const float* __restrict__ reward, int wave,
                                                              double gamma) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = threadIdx.x >> 6;
  const int b = blockIdx.x * kWavesPerBlock + w;
  if (b >= t.B) return;
  double* rwd = reinterpret_cast<double*>(smem) + (size_t)w * 2 * t.N;
  expand_backup_game(t, priors, value, reward, wave, gamma, rwd, rwd + t.N, b, threadIdx.x & 63);
}

// expand + backup of simulation s followed by the descent of simulation s + 1 for the same
// game: both are wavefront-per-game, so the only ordering needed is this wavefront's own
// stores before its own loads.  A workgroup-scope fence is enough (one compute unit, one
// write-through L1); a device-scope fence would write back the whole L2 slice per wavefront.
__global__ __launch_bounds__(kBlock) void expand_backup_select_kernel(TreeView t, const float* __restrict__ priors,
                                                                     const float* __restrict__ value,
                                                                     const float* __restrict__ reward, int wave,
                                                                     double gamma, const double* __restrict__ pbc,
                                                                     const double* __restrict__ sq, int table_len) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * kWavesPerBlock + w;
  if (b >= t.B) return;
  double* rwd = reinterpret_cast<double*>(smem) + (size_t)w * 2 * t.N;
  if (!expand_backup_game(t, priors, value, reward, wave, gamma, rwd, rwd + t.N, b, lane)) return;
  __threadfence_block();
  select_game(t, pbc, sq, table_len, b, lane);
}

// The same step with the game's tree staged in LDS.  The kernel above is a chain of dependent global loads
// (bookkeeping scalars -> path -> edge records -> one row of edge records per level of the descent), each a
// round trip to L2 or beyond, because the convolutions between two steps flush the tree out of the nearer
// caches.  Here one round trip fetches everything whose address is known up front (scalars, the path, every
// node's reward, the new priors, the first entries of the score tables), a second one fetches the edge records
// of all num_nodes rows into LDS, and backup and descent then run on the LDS copy; the records they change are
// written to LDS and to global memory.  Same arithmetic in the same order as the kernel above.
constexpr int kStagedIters = 16;      // edge records per lane: trees of up to 1024 edges
constexpr int kStagedTable = 256;     // score-table entries kept in LDS; larger parent counts read global memory

__host__ __device__ inline size_t staged_wave_bytes(int N, int A) {
  return ((size_t)N * A * 20 + (size_t)N * 32 + 15) / 16 * 16;
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64) void expand_backup_select_staged_kernel(
    TreeView t, const float* __restrict__ priors, const float* __restrict__ value, const float* __restrict__ reward,
    int wave, double gamma, const double* __restrict__ pbc, const double* __restrict__ sq, int table_len) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * WPB + w;
  const int A = t.A, N = t.N;
  const int staged = table_len < kStagedTable ? table_len : kStagedTable;
  double* tab_pbc = reinterpret_cast<double*>(smem);
  double* tab_sq = tab_pbc + kStagedTable;
  // loads below are unconditional (indices clamped): a load inside a branch makes the compiler wait for every
  // outstanding load where the branch rejoins, which would turn one round trip into several
  static_assert(WPB * 64 >= kStagedTable, "one table entry per thread");
  const int ti = (int)threadIdx.x < staged ? (int)threadIdx.x : staged - 1;
  const double my_pbc = pbc[ti], my_sq = sq[ti];
  char* mine = smem + 2 * kStagedTable * sizeof(double) + (size_t)w * staged_wave_bytes(N, A);
  double* l_ws = reinterpret_cast<double*>(mine);
  double* rwd = l_ws + (size_t)N * A;
  double* val = rwd + N;
  double* nrew = val + N;
  int* l_visit = reinterpret_cast<int*>(nrew + N);
  int* l_child = l_visit + (size_t)N * A;
  float* l_prior = reinterpret_cast<float*>(l_child + (size_t)N * A);
  int* l_pn = reinterpret_cast<int*>(l_prior + (size_t)N * A);
  int* l_pa = l_pn + N;

  bool live = b < t.B;
  const int bb = live ? b : 0;
  // ---- round trip 1: everything with a known address
  const int e = wave_uniform(t.num_nodes[bb]);
  const int len = wave_uniform(t.path_len[bb]);
  const int parent = wave_uniform(t.leaf_parent[bb]);
  const int act = wave_uniform(t.leaf_action[bb]);
  const double leaf_reward = (double)reward[bb];
  const double leaf_value = (double)value[bb];
  const int root_n = t.root_visit[bb];
  const double root_ws = t.root_value_sum[bb];
  const double plo = t.minmax[2 * bb], phi = t.minmax[2 * bb + 1];
  const bool root64 = t.root_is_f64[bb] != 0;
  const size_t nbase = (size_t)bb * N;
  const int ln = lane < N ? lane : N - 1, la = lane < A ? lane : A - 1;
  const int my_pn = t.path_node[nbase + ln];
  const int my_pa = t.path_action[nbase + ln];
  const double my_rew = t.reward[nbase + ln];
  const float new_prior = priors[(size_t)bb * A + la];
  const double root_p64 = t.root_prior64[(size_t)bb * A + la];
  if (live && (e >= N || len <= 0)) {
    if (lane == 0) t.status[b] = 2;
    live = false;
  }
  // ---- round trip 2: the edge records of rows 0 .. e-1
  const size_t ebase = nbase * A;
  const int cnt = live ? e * A : 0;
  int r_visit[kStagedIters], r_child[kStagedIters];
  float r_prior[kStagedIters];
  double r_ws[kStagedIters];
  const int jmax = cnt > 0 ? cnt - 1 : 0;
#pragma unroll
  for (int it = 0; it < kStagedIters; ++it) {  // no branches: records past the last one re-read it
    const int i = it * 64 + lane;
    const int j = i < jmax ? i : jmax;
    r_visit[it] = t.visit[ebase + j];
    r_child[it] = t.child[ebase + j];
    r_prior[it] = t.prior[ebase + j];
    r_ws[it] = t.value_sum[ebase + j];
  }
  if ((int)threadIdx.x < staged) {
    tab_pbc[threadIdx.x] = my_pbc;
    tab_sq[threadIdx.x] = my_sq;
  }
  if (lane < N) {
    l_pn[lane] = my_pn;
    l_pa[lane] = my_pa;
    nrew[lane] = my_rew;
  }
#pragma unroll
  for (int it = 0; it < kStagedIters; ++it) {
    const int i = it * 64 + lane;
    if (i < cnt) {
      l_visit[i] = r_visit[it];
      l_child[i] = r_child[it];
      l_prior[i] = r_prior[it];
      l_ws[i] = r_ws[it];
    }
  }
  __syncthreads();  // the score tables (and, within each wavefront, its own tree) are in LDS
  if (!live) return;

  // ---- expansion: a fresh row of A edge records behind (parent, act)
  const int erow = e * A;
  if (lane < A) {
    l_prior[erow + lane] = new_prior;
    l_visit[erow + lane] = 0;
    l_ws[erow + lane] = 0.0;
    l_child[erow + lane] = -1;
    t.prior[ebase + erow + lane] = new_prior;
    t.visit[ebase + erow + lane] = 0;
    t.value_sum[ebase + erow + lane] = 0.0;
    t.child[ebase + erow + lane] = -1;
  }
  if (lane == 0) {
    l_child[parent * A + act] = e;
    t.child[ebase + parent * A + act] = e;
    t.reward[nbase + e] = leaf_reward;
    t.num_nodes[b] = e + 1;
  }
  for (int d = lane; d < len; d += 64) rwd[d] = d == len - 1 ? leaf_reward : nrew[l_pn[d + 1]];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- value = reward + gamma * value, leaf first (model.py:522-526)
  double v = leaf_value;
  for (int d = len - 1; d >= 0; --d) {
    if (lane == 0) val[d] = v;
    v = rwd[d] + gamma * v;
  }
  const double root_add = v;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  double lo = __builtin_inf(), hi = -__builtin_inf();
  for (int d = lane; d < len; d += 64) {
    const int idx = l_pn[d] * A + l_pa[d];
    double ws = l_ws[idx];
    int n = l_visit[idx];
    const double add = val[d];
    for (int k = 0; k < wave; ++k) {
      ws += add;
      n += 1;
      const double q = ws / (double)n;
      lo = q < lo ? q : lo;
      hi = q > hi ? q : hi;
    }
    l_ws[idx] = ws;
    l_visit[idx] = n;
    t.value_sum[ebase + idx] = ws;
    t.visit[ebase + idx] = n;
  }
  int parent_n = root_n;
  if (lane == 0) {
    double ws = root_ws;
    int n = root_n;
    for (int k = 0; k < wave; ++k) {
      ws += root_add;
      n += 1;
      const double q = ws / (double)n;
      lo = q < lo ? q : lo;
      hi = q > hi ? q : hi;
    }
    t.root_value_sum[b] = ws;
    t.root_visit[b] = n;
    parent_n = n;
  }
  parent_n = wave_uniform(parent_n);
  lo = -wave_max_f64(-lo);
  hi = wave_max_f64(hi);
  lo = lo < plo ? lo : plo;
  hi = hi > phi ? hi : phi;
  if (lane == 0) {
    t.minmax[2 * b] = lo;
    t.minmax[2 * b + 1] = hi;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- descent of the next simulation (select_game on the LDS copy)
  const double ninf = -__builtin_inf();
  int node = 0, depth = 0, bad = 0;
  while (true) {
    const int row = node * A;
    int n = 0, kid = -1;
    double score = ninf;
    if (lane < A) {
      n = l_visit[row + lane];
      kid = l_child[row + lane];
      const double wsum = l_ws[row + lane];
      const int pn = parent_n < table_len ? parent_n : table_len - 1;
      const bool wide = root64 && node == 0;
      const double c_pbc = pn < staged ? tab_pbc[pn] : pbc[pn];
      const double c_sq = pn < staged ? tab_sq[pn] : sq[pn];
      score = edge_score(c_pbc, c_sq, n, wsum, lo, hi, wide, l_prior[row + lane], wide ? root_p64 : 0.0);
      if (!(score > ninf)) score = ninf;
    }
    const double best = wave_max_f64(score);
    const unsigned long long holders = __ballot(score == best);
    int best_a = holders ? __builtin_ctzll(holders) : 0;
    if (parent_n >= table_len) bad = 3;
    if (!(best > ninf)) {
      bad = 1;
      best_a = 0;
    }
    if (lane == 0) {
      t.path_node[nbase + depth] = node;
      t.path_action[nbase + depth] = best_a;
    }
    ++depth;
    const int next = __builtin_amdgcn_readlane(kid, best_a);  // best_a is wave-uniform: a scalar lane read, no LDS trip
    if (next < 0 || depth >= N) {
      if (lane == 0) {
        t.leaf_parent[b] = node;
        t.leaf_action[b] = best_a;
        t.path_len[b] = depth;
        if (bad) t.status[b] = bad;
      }
      break;
    }
    parent_n = __builtin_amdgcn_readlane(n, best_a);
    node = next;
  }
}

// ----------------------------------------------------------------- readout
__global__ __launch_bounds__(kBlock) void readout_kernel(TreeView t, int* __restrict__ visits,
                                                        double* __restrict__ root_value,
                                                        double* __restrict__ probs) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (b >= t.B) return;
  const int A = t.A;
  const int n = lane < A ? t.visit[(size_t)b * t.N * A + lane] : 0;
  int tot = n;
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) tot += __shfl_xor(tot, m);
  if (lane < A) {
    if (visits) visits[(size_t)b * A + lane] = n;
    if (probs) probs[(size_t)b * A + lane] = tot == 0 ? 1.0 / (double)A : (double)n / (double)tot;
  }
  if (lane == 0 && root_value) {
    const int rn = t.root_visit[b];
    root_value[b] = rn == 0 ? 0.0 : t.root_value_sum[b] / (double)rn;
  }
}

// ----------------------------------------------------------- select_action
// One thread per game: the arithmetic is a short serial chain (cumulative sum).
__device__ double numpy_pairwise_sum(const double* a, int n) {
  // Order of additions of NumPy's float64 pairwise summation for n <= 128.
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i];
  return res;
}

__global__ void select_action_kernel(const int* __restrict__ visits, int B, int A, double temperature,
                                     const double* __restrict__ u, int* __restrict__ actions) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int* c = visits + (size_t)b * A;
  if (temperature == 0.0) {
    int best = 0;
    for (int a = 1; a < A; ++a)
      if (c[a] > c[best]) best = a;
    actions[b] = best;
    return;
  }
  double p[MZ_MAX_ACTIONS];
  const double ex = 1.0 / temperature;
  const int iex = (int)ex;
  const bool integral = (double)iex == ex && iex >= 1 && iex <= 4;
  for (int a = 0; a < A; ++a) {
    const double x = (double)c[a];
    double y;
    if (integral) {
      y = x;
      for (int k = 1; k < iex; ++k) y *= x;
    } else {
      y = pow(x, ex);
    }
    p[a] = y;
  }
  // ndarray.sum(): the first element seeds the reduction, the rest is summed pairwise
  const double tot = A > 1 ? p[0] + numpy_pairwise_sum(p + 1, A - 1) : p[0];
  double run = 0.0;
  for (int a = 0; a < A; ++a) {
    run += p[a] / tot;
    p[a] = run;
  }
  const double last = p[A - 1];
  const double x = u[b];
  int k = 0;
  for (int a = 0; a < A; ++a) k += (p[a] / last) <= x;
  actions[b] = k < A ? k : A - 1;
}

int blocks_for(int B) { return (B + kWavesPerBlock - 1) / kWavesPerBlock; }

int check_tree_args(const void* ws, const mz_tree_layout* L) {
  MZ_REQUIRE(ws != nullptr && L != nullptr, "tree: null workspace or layout");
  MZ_REQUIRE(L->B > 0 && L->A >= 1 && L->A <= MZ_MAX_ACTIONS && L->max_nodes >= 2 && L->max_nodes <= MZ_MAX_NODES,
             "tree: bad layout (B=%d A=%d max_nodes=%d)", L->B, L->A, L->max_nodes);
  return MZ_OK;
}

}  // namespace

extern "C" {

int mz_tree_layout_query(int32_t B, int32_t A, int32_t max_nodes, mz_tree_layout* out) {
  MZ_REQUIRE(out != nullptr, "mz_tree_layout_query: out is null");
  MZ_REQUIRE(B > 0 && A >= 1 && A <= MZ_MAX_ACTIONS && max_nodes >= 2 && max_nodes <= MZ_MAX_NODES,
             "mz_tree_layout_query: need B>0, 1<=A<=%d, 2<=max_nodes<=%d (got %d, %d, %d)", MZ_MAX_ACTIONS,
             MZ_MAX_NODES, B, A, max_nodes);
  mz_tree_layout L{};
  L.B = B; L.A = A; L.max_nodes = max_nodes;
  int64_t off = 0;
  auto take = [&off](int64_t bytes) {
    int64_t at = off;
    off += (bytes + 255) / 256 * 256;
    return at;
  };
  const int64_t edges = (int64_t)B * max_nodes * A, nodes = (int64_t)B * max_nodes;
  L.value_sum = take(edges * 8);
  L.child = take(edges * 4);
  L.visit = take(edges * 4);
  L.prior = take(edges * 4);
  L.root_prior64 = take((int64_t)B * A * 8);
  L.reward = take(nodes * 8);
  L.root_value_sum = take((int64_t)B * 8);
  L.minmax = take((int64_t)B * 16);
  L.num_nodes = take((int64_t)B * 4);
  L.root_visit = take((int64_t)B * 4);
  L.root_is_f64 = take((int64_t)B * 4);
  L.path_node = take(nodes * 4);
  L.path_action = take(nodes * 4);
  L.path_len = take((int64_t)B * 4);
  L.leaf_parent = take((int64_t)B * 4);
  L.leaf_action = take((int64_t)B * 4);
  L.status = take((int64_t)B * 4);
  L.total_bytes = off;
  *out = L;
  return MZ_OK;
}

int mz_tree_root_init(void* ws, const mz_tree_layout* L, const float* priors, const double* noise,
                      float keep_f32, double frac, mz_stream_t stream) {
  if (int rc = check_tree_args(ws, L)) return rc;
  MZ_REQUIRE(priors != nullptr, "mz_tree_root_init: priors is null");
  TreeView t = make_view(ws, L);
  hipLaunchKernelGGL(root_init_kernel, dim3(blocks_for(L->B)), dim3(kBlock), 0, mz::as_stream(stream), t,
                     priors, noise, keep_f32, frac);
  return mz::check_launch("mz_tree_root_init");
}

int mz_tree_select(void* ws, const mz_tree_layout* L, const double* pbc_table, const double* sqrt_table,
                   int32_t table_len, mz_stream_t stream) {
  if (int rc = check_tree_args(ws, L)) return rc;
  MZ_REQUIRE(pbc_table && sqrt_table && table_len >= 2, "mz_tree_select: bad score tables");
  TreeView t = make_view(ws, L);
  hipLaunchKernelGGL(select_kernel, dim3(blocks_for(L->B)), dim3(kBlock), 0, mz::as_stream(stream), t,
                     pbc_table, sqrt_table, table_len);
  return mz::check_launch("mz_tree_select");
}

int mz_tree_expand_backup(void* ws, const mz_tree_layout* L, const float* priors, const float* value,
                          const float* reward, int32_t wave, double gamma, mz_stream_t stream) {
  if (int rc = check_tree_args(ws, L)) return rc;
  MZ_REQUIRE(priors && value && reward, "mz_tree_expand_backup: null network output");
  MZ_REQUIRE(wave >= 1, "mz_tree_expand_backup: wave must be >= 1 (got %d)", wave);
  TreeView t = make_view(ws, L);
  const size_t lds = (size_t)kWavesPerBlock * 2 * L->max_nodes * sizeof(double);
  hipLaunchKernelGGL(expand_backup_kernel, dim3(blocks_for(L->B)), dim3(kBlock), lds, mz::as_stream(stream),
                     t, priors, value, reward, wave, gamma);
  return mz::check_launch("mz_tree_expand_backup");
}

int mz_tree_expand_backup_select(void* ws, const mz_tree_layout* L, const float* priors, const float* value,
                                 const float* reward, int32_t wave, double gamma, const double* pbc_table,
                                 const double* sqrt_table, int32_t table_len, mz_stream_t stream) {
  if (int rc = check_tree_args(ws, L)) return rc;
  MZ_REQUIRE(priors && value && reward, "mz_tree_expand_backup_select: null network output");
  MZ_REQUIRE(wave >= 1, "mz_tree_expand_backup_select: wave must be >= 1 (got %d)", wave);
  MZ_REQUIRE(pbc_table && sqrt_table && table_len >= 2, "mz_tree_expand_backup_select: bad score tables");
  TreeView t = make_view(ws, L);
  static const bool no_staging = getenv("MZ_TREE_STAGED") && atoi(getenv("MZ_TREE_STAGED")) == 0;  // kernel comparisons
  constexpr int WPB = 4;  // one, two and four wavefronts per workgroup measured the same
  const size_t staged_lds = 2 * kStagedTable * sizeof(double) + WPB * staged_wave_bytes(L->max_nodes, L->A);
  if (!no_staging && L->max_nodes <= 64 && L->max_nodes * L->A <= 64 * kStagedIters && staged_lds <= 160 * 1024) {
    static bool configured[mz::kMaxDevices] = {false};
    auto kern = expand_backup_select_staged_kernel<WPB>;
    if (int rc = mz::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)staged_lds, configured,
                                        "expand_backup_select_staged"))
      return rc;
    hipLaunchKernelGGL(kern, dim3((L->B + WPB - 1) / WPB), dim3(WPB * 64), staged_lds, mz::as_stream(stream), t,
                       priors, value, reward, wave, gamma, pbc_table, sqrt_table, table_len);
    return mz::check_launch("mz_tree_expand_backup_select");
  }
  const size_t lds = (size_t)kWavesPerBlock * 2 * L->max_nodes * sizeof(double);
  hipLaunchKernelGGL(expand_backup_select_kernel, dim3(blocks_for(L->B)), dim3(kBlock), lds, mz::as_stream(stream),
                     t, priors, value, reward, wave, gamma, pbc_table, sqrt_table, table_len);
  return mz::check_launch("mz_tree_expand_backup_select");
}

int mz_tree_readout(const void* ws, const mz_tree_layout* L, int32_t* visits, double* root_value,
                    double* probs, mz_stream_t stream) {
  if (int rc = check_tree_args(ws, L)) return rc;
  TreeView t = make_view(const_cast<void*>(ws), L);
  hipLaunchKernelGGL(readout_kernel, dim3(blocks_for(L->B)), dim3(kBlock), 0, mz::as_stream(stream), t,
                     visits, root_value, probs);
  return mz::check_launch("mz_tree_readout");
}

int mz_select_action(const int32_t* visits, int32_t B, int32_t A, double temperature, const double* u,
                     int32_t* actions, mz_stream_t stream) {
  MZ_REQUIRE(visits && actions && B > 0 && A >= 1 && A <= MZ_MAX_ACTIONS, "mz_select_action: bad arguments");
  MZ_REQUIRE(temperature >= 0.0, "mz_select_action: negative temperature");
  MZ_REQUIRE(temperature == 0.0 || u != nullptr, "mz_select_action: u is required when temperature > 0");
  hipLaunchKernelGGL(select_action_kernel, dim3((B + 63) / 64), dim3(64), 0, mz::as_stream(stream), visits, B,
                     A, temperature, u, actions);
  return mz::check_launch("mz_select_action");
}

int mz_ucb_score(const int32_t* parent_visits, const int32_t* child_visits, const double* value_sum,
                 const double* minimum, const double* maximum, const float* prior32, const double* prior64,
                 const double* pbc_table, const double* sqrt_table, int32_t table_len, int32_t n, double* scores,
                 mz_stream_t stream) {
  MZ_REQUIRE(parent_visits && child_visits && value_sum && minimum && maximum && pbc_table && sqrt_table && scores,
             "mz_ucb_score: null argument");
  MZ_REQUIRE((prior32 != nullptr) != (prior64 != nullptr), "mz_ucb_score: exactly one of prior32 / prior64");
  MZ_REQUIRE(n > 0 && table_len > 0, "mz_ucb_score: bad sizes");
  hipLaunchKernelGGL(ucb_score_kernel, dim3((n + 255) / 256), dim3(256), 0, mz::as_stream(stream), parent_visits,
                     child_visits, value_sum, minimum, maximum, prior32, prior64, pbc_table, sqrt_table, n, scores);
  return mz::check_launch("mz_ucb_score");
}

}  // extern "C"
