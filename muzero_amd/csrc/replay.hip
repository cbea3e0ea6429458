// Prioritised sampling over the positions of the replay window on the device
// (ReplayBuffer.sample_batch / update_priorities, model.py:575-614): the priorities stay in
// HBM, a training batch is drawn by inverting the cumulative distribution for `batch` uniform
// draws, and the new priorities of the sampled positions are written back in place.
//
// The reference sums the priorities left to right in Python and inverts np.cumsum; this file
// sums in a fixed two-level order (chunks of 2048 positions, then the chunk sums), so the
// cumulative values agree with the reference's to rounding (a few ulp of the total) and the
// drawn index can differ only for a uniform that lies that close to a boundary of the
// cumulative distribution.  The order is fixed: the same inputs always give the same draws.
//
// HBM-bound: one pass over the n priorities (8 B each) for the chunk sums plus 16 KB per draw.
#pragma clang fp contract(off)

#include <atomic>

#include "common.h"

namespace {

constexpr int kChunk = 2048;           // positions per chunk: 64 lanes x 32, or 256 threads x 8
constexpr int kPerLane = kChunk / 64;

// scratch layout (doubles): [0] total, [1 .. 1+nchunks] exclusive chunk offsets (one past the end = total),
// then nchunks chunk sums
__host__ __device__ inline long scratch_doubles(long n) {
  const long nchunks = (n + kChunk - 1) / kChunk;
  return 1 + (nchunks + 1) + nchunks;
}

__global__ __launch_bounds__(256) void chunk_sums_kernel(const double* __restrict__ prio, long n,
                                                        double* __restrict__ ws) {
  __shared__ double part[4];
  const long nchunks = (n + kChunk - 1) / kChunk;
  double* sums = ws + 1 + (nchunks + 1);
  const long base = (long)blockIdx.x * kChunk;
  const int tid = threadIdx.x, lane = tid & 63;
  // thread t owns positions base + 8t .. +8: its sum, then lanes, then the four wavefronts, always
  // in this order
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const long i = base + tid * 8 + k;
    s += i < n ? prio[i] : 0.0;
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
  if (lane == 0) part[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) sums[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// one workgroup: exclusive scan of the chunk sums, left to right in tiles of 1024
__global__ __launch_bounds__(1024) void chunk_scan_kernel(long n, double* __restrict__ ws) {
  __shared__ double tile[1024];
  __shared__ double carry_s;
  const long nchunks = (n + kChunk - 1) / kChunk;
  double* off = ws + 1;
  const double* sums = ws + 1 + (nchunks + 1);
  const int tid = threadIdx.x;
  if (tid == 0) carry_s = 0.0;
  __syncthreads();
  for (long t0 = 0; t0 < nchunks; t0 += 1024) {
    const long c = t0 + tid;
    tile[tid] = c < nchunks ? sums[c] : 0.0;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
      const double v = tid >= d ? tile[tid - d] : 0.0;
      __syncthreads();
      tile[tid] += v;
      __syncthreads();
    }
    const double carry = carry_s;
    if (c < nchunks) off[c] = carry + (tid ? tile[tid - 1] : 0.0);
    __syncthreads();
    if (tid == 1023) carry_s = carry + tile[1023];
    __syncthreads();
  }
  if (tid == 0) {
    off[nchunks] = carry_s;
    ws[0] = carry_s;
  }
}

// one wavefront per draw
__global__ __launch_bounds__(256) void draw_kernel(const double* __restrict__ prio, long n,
                                                  const double* __restrict__ u, int batch, double beta,
                                                  const long* __restrict__ game_start, int n_games,
                                                  const double* __restrict__ ws, int* __restrict__ flat,
                                                  int* __restrict__ game, int* __restrict__ pos,
                                                  double* __restrict__ weights) {
  const int d = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (d >= batch) return;
  const long nchunks = (n + kChunk - 1) / kChunk;
  const double total = ws[0];
  const double* off = ws + 1;
  const double x = u[d];
  long idx;
  double p_idx;
  if (!(total > 0.0)) {  // all priorities zero: uniform over the positions (model.py:581-583)
    long i = (long)(x * (double)n);
    idx = i < 0 ? 0 : (i >= n ? n - 1 : i);
    p_idx = 0.0;
  } else {
    // chunk: the first one whose end lies above the draw (the last chunk if rounding leaves none)
    long lo = 0, hi = nchunks - 1;
    while (lo < hi) {
      const long mid = (lo + hi) >> 1;
      if (off[mid + 1] / total > x) hi = mid; else lo = mid + 1;
    }
    const long base = lo * kChunk;
    const double start = off[lo];
    double v[kPerLane];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < kPerLane; ++k) {
      const long i = base + lane * kPerLane + k;
      v[k] = i < n ? prio[i] : 0.0;
      s += v[k];
    }
    double incl = s;  // inclusive scan of the lane sums over the wavefront
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const double t = __shfl_up(incl, m);
      if (lane >= m) incl += t;
    }
    const double excl = incl - s;
    const unsigned long long above = __ballot((start + incl) / total > x);
    const long last_valid = (n - base < kChunk ? n - base : kChunk) - 1;
    // A draw must never land on a position of priority zero: its importance weight would be 1 / 0.  The running
    // sum below restarts from start + (incl - s), which can come out an ulp away from the neighbouring lane's
    // inclusive sum, so "the first k whose cumulative sum exceeds the draw" is taken among the lane's POSITIVE
    // entries, and every fallback (rounding left no crossing in the lane, or none in the chunk) takes the last
    // positive entry at or before where the crossing should have been.
    int last_pos = -1;  // this lane's last positive entry
#pragma unroll
    for (int k = 0; k < kPerLane; ++k)
      if (v[k] > 0.0) last_pos = k;
    const unsigned long long has_pos = __ballot(last_pos >= 0);
    int src_lane = -1;
    int k_hit = -1;
    if (above) {
      src_lane = __builtin_ctzll(above);
      // the lane of the crossing holds a positive entry unless rounding put the crossing one lane late: then
      // the nearest lane below it that has one
      if (!((has_pos >> src_lane) & 1ull)) {
        const unsigned long long below = has_pos & ((1ull << src_lane) - 1ull);
        src_lane = below ? 63 - __builtin_clzll(below) : (has_pos ? __builtin_ctzll(has_pos) : -1);
      }
      if (lane == src_lane) {
        double r = start + excl;
        k_hit = last_pos;
        bool found = false;
#pragma unroll
        for (int k = 0; k < kPerLane; ++k) {
          r += v[k];
          if (!found && v[k] > 0.0 && r / total > x) { k_hit = k; found = true; }
        }
      }
    } else if (has_pos) {
      src_lane = 63 - __builtin_clzll(has_pos);
      if (lane == src_lane) k_hit = last_pos;
    }
    if (src_lane >= 0) {
      k_hit = __shfl(k_hit, src_lane);
      idx = base + (long)src_lane * kPerLane + k_hit;
    } else {
      idx = base + last_valid;  // a chunk without any positive entry: only a draw outside [0, 1) gets here
    }
    p_idx = prio[idx];
  }
  if (lane == 0) {
    flat[d] = (int)idx;
    const double prob = total > 0.0 ? p_idx / total : 1.0 / (double)n;
    const double t = (double)n * prob;
    // normalised by the batch maximum afterwards; a position of priority zero while others are positive can
    // only come from a draw outside [0, 1): weight zero (the sample is ignored) instead of infinity
    weights[d] = t > 0.0 ? (beta == 1.0 ? 1.0 / t : pow(t, -beta)) : 0.0;
    if (game_start) {
      int a = 0, b = n_games - 1;  // last game whose first position is <= idx
      while (a < b) {
        const int mid = (a + b + 1) >> 1;
        if (game_start[mid] <= idx) a = mid; else b = mid - 1;
      }
      if (game) game[d] = a;
      if (pos) pos[d] = (int)(idx - game_start[a]);
    }
  }
}

// weights / max(weights) (model.py:588); one workgroup
__global__ __launch_bounds__(1024) void normalise_kernel(double* __restrict__ weights, int batch) {
  __shared__ double red[1024];
  const int tid = threadIdx.x;
  double m = 0.0;
  for (int i = tid; i < batch; i += 1024) m = fmax(m, weights[i]);
  red[tid] = m;
  __syncthreads();
  for (int d = 512; d > 0; d >>= 1) {
    if (tid < d) red[tid] = fmax(red[tid], red[tid + d]);
    __syncthreads();
  }
  m = red[0];
  if (!(m > 0.0)) return;
  for (int i = tid; i < batch; i += 1024) weights[i] = weights[i] / m;
}

// prio[flat[i]] = (values[i] + 1e-6) ** alpha in list order: of several entries naming one slot the
// last one wins, as in the reference's loop (model.py:612-614).  Two passes, O(m): first every entry bids
// for its slot with the NEGATIVE number -(i + 1) by an integer minimum on the slot's bit pattern (priorities
// are non-negative doubles, whose patterns are non-negative integers, so any bid beats the old priority and
// the highest list index beats every other bid); then the entry that finds its own bid in the slot writes
// its value there.
__global__ void update_bid_kernel(double* __restrict__ prio, long n, const int* __restrict__ flat, int m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int slot = flat[i];
  if (slot < 0 || slot >= n) return;
  atomicMin(reinterpret_cast<long long*>(prio) + slot, -(long long)(i + 1));
}

__global__ void update_write_kernel(double* __restrict__ prio, long n, const int* __restrict__ flat,
                                    const double* __restrict__ values, int m, double alpha) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int slot = flat[i];
  if (slot < 0 || slot >= n) return;
  if (reinterpret_cast<const long long*>(prio)[slot] != -(long long)(i + 1)) return;
  const double v = values[i] + 1e-6;
  prio[slot] = alpha == 1.0 ? v : pow(v, alpha);
}

int sample_impl(const double* prio, long n, const double* u, int batch, double beta, const long* game_start,
                int n_games, int* flat, int* game, int* pos, double* weights, double* ws, hipStream_t s) {
  const long nchunks = (n + kChunk - 1) / kChunk;
  hipLaunchKernelGGL(chunk_sums_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, prio, n, ws);
  hipLaunchKernelGGL(chunk_scan_kernel, dim3(1), dim3(1024), 0, s, n, ws);
  hipLaunchKernelGGL(draw_kernel, dim3((batch + 3) / 4), dim3(256), 0, s, prio, n, u, batch, beta, game_start,
                     n_games, ws, flat, game, pos, weights);
  hipLaunchKernelGGL(normalise_kernel, dim3(1), dim3(1024), 0, s, weights, batch);
  return mz::check_launch("mz_replay_sample");
}

int check_sample_args(const double* prio, int64_t n, const double* u, int32_t batch, const int64_t* game_start,
                      int32_t n_games, const int32_t* flat, const int32_t* game, const int32_t* pos,
                      const double* weights) {
  MZ_REQUIRE(prio && u && flat && weights, "mz_replay_sample: null argument");
  MZ_REQUIRE(n >= 1 && n <= 0x7fffffffLL, "mz_replay_sample: %lld positions (1 .. 2^31-1 supported)", (long long)n);
  MZ_REQUIRE(batch >= 1, "mz_replay_sample: empty batch");
  MZ_REQUIRE((game_start != nullptr) == (n_games > 0), "mz_replay_sample: game_start and n_games go together");
  MZ_REQUIRE(game_start || (!game && !pos), "mz_replay_sample: game / pos outputs need game_start");
  return MZ_OK;
}

// --- the allocation-free variant's scratch: one array per GPU, held until the process ends ---
constexpr long kStaticMaxPositions = 16L << 20;  // 16 Mi positions -> 8192 chunks -> 128 KiB + 16 B
constexpr long kStaticDoubles = 1 + (kStaticMaxPositions / kChunk + 1) + kStaticMaxPositions / kChunk;
static __device__ __align__(256) double scratch[kStaticDoubles];

double* static_scratch_ptr(int* rc) {
  static std::atomic<double*> ptr[mz::kMaxDevices];
  const int dev = mz::current_device();
  *rc = MZ_OK;
  double* p = ptr[dev].load(std::memory_order_acquire);
  if (!p) {  // once per device, with that device current: each GPU has its own copy of the array.  Two threads
             // racing here both ask the runtime and store the same address.
    const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
      *rc = mz::fail(MZ_E_LAUNCH, "mz_replay_sample_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
      return nullptr;
    }
    ptr[dev].store(p, std::memory_order_release);
  }
  return p;
}

}  // namespace

extern "C" {

int64_t mz_replay_workspace_bytes(int64_t n) { return n < 1 ? 0 : scratch_doubles(n) * 8; }

int mz_replay_sample(const double* prio, int64_t n, const double* u, int32_t batch, double beta,
                     const int64_t* game_start, int32_t n_games, int32_t* flat, int32_t* game, int32_t* pos,
                     double* weights, void* workspace, int64_t workspace_bytes, mz_stream_t stream) {
  if (int rc = check_sample_args(prio, n, u, batch, game_start, n_games, flat, game, pos, weights)) return rc;
  MZ_REQUIRE(workspace && workspace_bytes >= mz_replay_workspace_bytes(n),
             "mz_replay_sample: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
             (long long)mz_replay_workspace_bytes(n));
  MZ_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "mz_replay_sample: workspace must be 8-byte aligned");
  return sample_impl(prio, n, u, batch, beta, reinterpret_cast<const long*>(game_start), n_games, flat, game, pos,
                     weights, static_cast<double*>(workspace), mz::as_stream(stream));
}

int64_t mz_replay_static_max_positions(void) {
  // doubles as the set-up call: looks up the scratch array's address on the current device now, so that the
  // first mz_replay_sample_static there - possibly under stream capture - makes no runtime query
  int rc = MZ_OK;
  static_scratch_ptr(&rc);
  return kStaticMaxPositions;
}

int mz_replay_sample_static(const double* prio, int64_t n, const double* u, int32_t batch, double beta,
                            const int64_t* game_start, int32_t n_games, int32_t* flat, int32_t* game,
                            int32_t* pos, double* weights, mz_stream_t stream) {
  if (int rc = check_sample_args(prio, n, u, batch, game_start, n_games, flat, game, pos, weights)) return rc;
  MZ_REQUIRE(n <= kStaticMaxPositions,
             "mz_replay_sample_static: %lld positions, at most %lld fit the built-in scratch (use mz_replay_sample "
             "with a workspace)", (long long)n, (long long)kStaticMaxPositions);
  int rc = MZ_OK;
  double* ws = static_scratch_ptr(&rc);
  if (!ws) return rc;
  return sample_impl(prio, n, u, batch, beta, reinterpret_cast<const long*>(game_start), n_games, flat, game, pos,
                     weights, ws, mz::as_stream(stream));
}

int mz_replay_update(double* prio, int64_t n, const int32_t* flat, const double* values, int32_t m, double alpha,
                     mz_stream_t stream) {
  MZ_REQUIRE(prio && flat && values && n >= 1 && m >= 1, "mz_replay_update: bad arguments");
  hipLaunchKernelGGL(update_bid_kernel, dim3((m + 255) / 256), dim3(256), 0, mz::as_stream(stream), prio, (long)n,
                     flat, m);
  hipLaunchKernelGGL(update_write_kernel, dim3((m + 255) / 256), dim3(256), 0, mz::as_stream(stream), prio, (long)n,
                     flat, values, m, alpha);
  return mz::check_launch("mz_replay_update");
}

}  // extern "C"
