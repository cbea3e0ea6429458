// Training-side kernels: the support cross-entropy of the unrolled loss
// (scalar_to_support_loss, model.py:112-140, with transform_value, model.py:84-86) and the
// n-step value targets (Game.compute_target_value / make_target, train.py:89-118).
// Compiled with -ffp-contract=off: both restate the reference's rounding sequence.
#pragma clang fp contract(off)

#include "common.h"

namespace {

// One wavefront per row, the row held in registers (one pass over HBM in, one out).
// loss = -sum_o t[o] * log_softmax(x)[o] with t the two-hot encoding of the transformed, clamped
// target; dlogits = softmax(x) * sum(t) - t  (d loss / d x).
// The two-hot row is built as the reference's two scatter_ calls do: the upper entry is
// written after the lower one, so when both indices coincide (target clamped at +S) only
// prob_upper survives.
constexpr int kRowRegs = 16;  // 64 lanes x 16 = rows of up to 1024 logits

__global__ __launch_bounds__(256) void support_ce_kernel(const float* __restrict__ logits,
                                                        const float* __restrict__ targets, int n, int S,
                                                        float* __restrict__ loss, float* __restrict__ dlogits) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const int O = 2 * S + 1;
  const float* x = logits + (long)row * O;
  const float v = targets[row];
  const float sg = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
  float t = sg * (sqrtf(fabsf(v) + 1.0f) - 1.0f + 0.001f * v);   // transform_value
  t = fminf(fmaxf(t, (float)-S), (float)S);
  const float fl = floorf(t);
  const float p_up = t - fl, p_lo = 1.0f - p_up;
  int lo = (int)(fl + (float)S);
  int up = lo + 1;
  up = up < 0 ? 0 : (up > 2 * S ? 2 * S : up);
  lo = lo < 0 ? 0 : (lo > 2 * S ? 2 * S : lo);
  const float t_lo = lo == up ? p_up : p_lo;  // second scatter overwrites the first
  const float t_sum = lo == up ? p_up : p_lo + p_up;

  float xr[kRowRegs];
  float m = -__builtin_inff();
#pragma unroll
  for (int j = 0; j < kRowRegs; ++j) {
    const int o = lane + 64 * j;
    xr[j] = o < O ? x[o] : -__builtin_inff();
    m = fmaxf(m, xr[j]);
  }
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) m = fmaxf(m, __shfl_xor(m, k));
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < kRowRegs; ++j) {
    xr[j] = expf(xr[j] - m);  // exp(-inf) = 0 past the end of the row
    den += xr[j];
  }
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) den += __shfl_xor(den, k);
  const float lse = logf(den);
  if (lane == 0) {
    float l = -(t_lo * (x[lo] - m - lse));
    if (up != lo) l = l - p_up * (x[up] - m - lse);
    loss[row] = l;
  }
  if (dlogits) {
    float* d = dlogits + (long)row * O;
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j) {
      const int o = lane + 64 * j;
      if (o >= O) break;
      float g = xr[j] / den * t_sum;
      if (o == lo) g -= t_lo;
      if (o == up && up != lo) g -= p_up;
      d[o] = g;
    }
  }
}

// One thread per (query, unroll step).  Episodes are stored back to back: game g owns
// rewards[r_off[g] .. r_off[g+1]) and values / policy rows [v_off[g] .. v_off[g+1]).
__global__ void nstep_targets_kernel(const double* __restrict__ rewards, const double* __restrict__ values,
                                     const double* __restrict__ policies, const long* __restrict__ r_off,
                                     const long* __restrict__ v_off, int A, const int* __restrict__ q_game,
                                     const int* __restrict__ q_pos, int nq, int K, int td, double gamma,
                                     double* __restrict__ tv, double* __restrict__ tr, double* __restrict__ tp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * (K + 1)) return;
  const int q = i / (K + 1), k = i - q * (K + 1);
  const int g = q_game[q], cur = q_pos[q] + k;
  const double* r = rewards + r_off[g];
  const double* v = values + v_off[g];
  const int nr = (int)(r_off[g + 1] - r_off[g]), nv = (int)(v_off[g + 1] - v_off[g]);
  double value = 0.0, discount = 1.0;
  for (int j = 0; j < td; ++j) {  // train.py:92-98
    const int idx = cur + j;
    if (idx >= nr) break;
    value += r[idx] * discount;
    discount *= gamma;
  }
  if (cur + td < nv) value += v[cur + td] * discount;  // train.py:99-100
  tv[i] = value;
  tr[i] = cur < nr ? r[cur] : 0.0;                     // train.py:111-116
  if (tp) {
    const bool has = cur < nr && cur < nv;             // a position past the end has an all-zero policy
    const double* p = policies + (v_off[g] + cur) * (long)A;
    for (int a = 0; a < A; ++a) tp[(long)i * A + a] = has ? p[a] : 0.0;
  }
}

// ---- what the unrolled training step (train.py:343-372) does with a hidden state h_t besides convolving it, in one
// pass over it: the spatial mean the prediction heads read (model.py:273-281) and the dynamics network's input
// "hidden + conv1x1(one-hot action planes)" (model.py:231-236) - on every pixel of image b the planes contribute
// table[actions[b]][c] (table = conv_action's weight column + bias).  bf16 NHWC activations, float32 table and
// means; the sum is rounded to bf16 once.  One workgroup per image: thread = 8-channel group x row lane.
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                               // nearest even
  return (unsigned short)(u >> 16);
}

constexpr int kStepThreads = 1024;   // 16 wavefronts per image: 32 row lanes at 256 channels, 4-5 rows each

// column sums of the workgroup's per-thread partials s[8] -> out[c] * scale, through LDS
__device__ __forceinline__ void combine_row_lanes(float* red, const float (&s)[8], int groups, int lanes, int C,
                                                  float scale, float* __restrict__ out) {
  const int cg = threadIdx.x % groups, rl = threadIdx.x / groups;
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl * C + cg * 8 + j] = s[j];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += kStepThreads) {
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[l * C + c];
    out[c] = t * scale;
  }
}

__global__ __launch_bounds__(kStepThreads) void step_input_kernel(const unsigned short* __restrict__ h,
                                                        const float* __restrict__ table, const int* __restrict__ actions,
                                                        int HW, int C, unsigned short* __restrict__ x_out,
                                                        float* __restrict__ mean_out) {
  __shared__ float red[kStepThreads * 8];
  const int groups = C / 8, lanes = kStepThreads / groups;
  const int cg = threadIdx.x % groups, rl = threadIdx.x / groups;
  const long base = (long)blockIdx.x * HW * C;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (x_out) {
    const float* t = table + (long)actions[blockIdx.x] * C + cg * 8;
    const float4 a = *reinterpret_cast<const float4*>(t), b = *reinterpret_cast<const float4*>(t + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // three rows per trip, their loads in flight together
  for (int r0 = rl; r0 < HW; r0 += 3 * lanes) {
    u16x8 in[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = r0 + k * lanes < HW ? r0 + k * lanes : r0;
      in[k] = *reinterpret_cast<const u16x8*>(h + base + (long)r * C + cg * 8);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = r0 + k * lanes;
      if (r >= HW) break;
      u16x8 y;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = bf16_bits_to_f32(in[k][j]);
        s[j] += f;
        y[j] = f32_to_bf16_bits(f + v[j]);
      }
      if (x_out) *reinterpret_cast<u16x8*>(x_out + base + (long)r * C + cg * 8) = y;
    }
  }
  if (mean_out) combine_row_lanes(red, s, groups, lanes, C, 1.f / (float)HW, mean_out + (long)blockIdx.x * C);
}

// The gradient of both: dh = bf16(dy + dmean / HW) on every pixel (either term may be absent), and per image the
// pixel sums of dy, from which dtable[a][c] = sum over the images with actions[b] == a, in index order - no
// atomics, the same bits every time.
__global__ __launch_bounds__(kStepThreads) void step_input_grad_kernel(const unsigned short* __restrict__ dy,
                                                             const float* __restrict__ dmean, int HW, int C,
                                                             unsigned short* __restrict__ dh,
                                                             float* __restrict__ image_sums) {
  __shared__ float red[kStepThreads * 8];
  const int groups = C / 8, lanes = kStepThreads / groups;
  const int cg = threadIdx.x % groups, rl = threadIdx.x / groups;
  const long base = (long)blockIdx.x * HW * C;
  float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (dmean) {
    const float* t = dmean + (long)blockIdx.x * C + cg * 8;
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = t[j] * inv;
  }
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r0 = rl; r0 < HW; r0 += 3 * lanes) {  // three rows per trip, as in step_input_kernel
    u16x8 in[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = r0 + k * lanes < HW ? r0 + k * lanes : r0;
      in[k] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (dy) in[k] = *reinterpret_cast<const u16x8*>(dy + base + (long)r * C + cg * 8);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = r0 + k * lanes;
      if (r >= HW) break;
      u16x8 y;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = bf16_bits_to_f32(in[k][j]);
        s[j] += f;
        y[j] = f32_to_bf16_bits(f + m[j]);
      }
      *reinterpret_cast<u16x8*>(dh + base + (long)r * C + cg * 8) = y;
    }
  }
  if (image_sums) combine_row_lanes(red, s, groups, lanes, C, 1.f, image_sums + (long)blockIdx.x * C);
}

// dtable[a][c]: one workgroup per (action, 32 channels); thread = channel x one of 8 image lanes that adds every
// eighth image (all loads unconditional and in flight together, the action test picks what counts), the lanes
// combined in a fixed order.
__global__ __launch_bounds__(256) void action_group_sum_kernel(const float* __restrict__ sums,
                                                              const int* __restrict__ actions, int B, int C, int A,
                                                              float* __restrict__ dtable) {
  __shared__ float red[8][32];
  const int per_a = C / 32;
  const int a = blockIdx.x / per_a, c = (blockIdx.x % per_a) * 32 + (threadIdx.x & 31), bl = threadIdx.x >> 5;
  float t = 0.f;
  for (int b0 = bl; b0 < B; b0 += 64) {
    float v[8];
    int act[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b = b0 + 8 * k < B ? b0 + 8 * k : b0;
      v[k] = sums[(long)b * C + c];
      act[k] = actions[b];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (b0 + 8 * k < B && act[k] == a) t += v[k];
  }
  red[bl][threadIdx.x & 31] = t;
  __syncthreads();
  if (bl == 0) {
    float r = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) r += red[l][threadIdx.x];
    dtable[(long)a * C + c] = r;
  }
}

int check_step_shape(const char* who, int32_t B, int32_t HW, int32_t C) {
  MZ_REQUIRE(B > 0 && HW > 0, "%s: empty batch", who);
  MZ_REQUIRE(C >= 32 && C % 32 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0,
             "%s: C = %d must be 8 times a power of two, from 32 to 2048", who, C);
  return MZ_OK;
}

}  // namespace

extern "C" {

int mz_step_input_bf16(const void* h, const float* table, const int32_t* actions, void* x_out, float* mean_out,
                       int32_t B, int32_t HW, int32_t C, mz_stream_t stream) {
  MZ_REQUIRE(h && (x_out || mean_out), "mz_step_input_bf16: null tensor");
  MZ_REQUIRE(!x_out || (table && actions), "mz_step_input_bf16: x_out needs table and actions");
  if (int rc = check_step_shape("mz_step_input_bf16", B, HW, C)) return rc;
  hipLaunchKernelGGL(step_input_kernel, dim3(B), dim3(kStepThreads), 0, mz::as_stream(stream),
                     static_cast<const unsigned short*>(h), table, actions, HW, C, static_cast<unsigned short*>(x_out),
                     mean_out);
  return mz::check_launch("mz_step_input_bf16");
}

int mz_step_input_grad_bf16(const void* dy, const float* dmean, const int32_t* actions, int32_t B, int32_t HW, int32_t C,
                            int32_t A, void* dh, float* image_sums, float* dtable, mz_stream_t stream) {
  MZ_REQUIRE(dh && (dy || dmean), "mz_step_input_grad_bf16: null tensor");
  MZ_REQUIRE(!dtable || (dy && actions && image_sums && A > 0),
             "mz_step_input_grad_bf16: dtable needs dy, actions, image_sums and A > 0");
  if (int rc = check_step_shape("mz_step_input_grad_bf16", B, HW, C)) return rc;
  hipStream_t s = mz::as_stream(stream);
  hipLaunchKernelGGL(step_input_grad_kernel, dim3(B), dim3(kStepThreads), 0, s, static_cast<const unsigned short*>(dy), dmean, HW,
                     C, static_cast<unsigned short*>(dh), dtable ? image_sums : nullptr);
  if (dtable)
    hipLaunchKernelGGL(action_group_sum_kernel, dim3(A * (C / 32)), dim3(256), 0, s, image_sums, actions, B, C, A, dtable);
  return mz::check_launch("mz_step_input_grad_bf16");
}

int mz_support_ce(const float* logits, const float* targets, int32_t n, int32_t S, float* loss, float* dlogits,
                  mz_stream_t stream) {
  MZ_REQUIRE(logits && targets && loss && n > 0 && S >= 1, "mz_support_ce: bad arguments");
  MZ_REQUIRE(2 * S + 1 <= 64 * kRowRegs, "mz_support_ce: support_size %d above the supported 511", S);
  hipLaunchKernelGGL(support_ce_kernel, dim3((n + 3) / 4), dim3(256), 0, mz::as_stream(stream), logits, targets, n,
                     S, loss, dlogits);
  return mz::check_launch("mz_support_ce");
}

int mz_nstep_targets(const double* rewards, const double* values, const double* policies, const int64_t* reward_off,
                     const int64_t* value_off, int32_t n_games, int32_t A, const int32_t* q_game,
                     const int32_t* q_pos, int32_t n_queries, int32_t K, int32_t td_steps, double gamma,
                     double* target_values, double* target_rewards, double* target_policies, mz_stream_t stream) {
  MZ_REQUIRE(rewards && values && reward_off && value_off && q_game && q_pos && target_values && target_rewards,
             "mz_nstep_targets: null argument");
  MZ_REQUIRE((policies != nullptr) == (target_policies != nullptr),
             "mz_nstep_targets: policies and target_policies go together");
  MZ_REQUIRE(n_games > 0 && n_queries > 0 && K >= 0 && td_steps >= 0 && A >= 1, "mz_nstep_targets: bad sizes");
  const int total = n_queries * (K + 1);
  hipLaunchKernelGGL(nstep_targets_kernel, dim3((total + 255) / 256), dim3(256), 0, mz::as_stream(stream), rewards,
                     values, policies, reinterpret_cast<const long*>(reward_off),
                     reinterpret_cast<const long*>(value_off), A, q_game, q_pos, n_queries, K, td_steps, gamma,
                     target_values, target_rewards, target_policies);
  return mz::check_launch("mz_nstep_targets");
}

}  // extern "C"
