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

// ---- "hidden + conv1x1(one-hot action planes)" of the dynamics network (model.py:231-236) in the training step:
// on every pixel of image b the planes contribute table[actions[b]][c] (table = conv_action's weight column +
// bias).  bf16 NHWC activations, float32 table; the sum is rounded to bf16 once.
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                               // nearest even
  return (unsigned short)(u >> 16);
}

__global__ __launch_bounds__(256) void add_action_vector_kernel(const unsigned short* __restrict__ h,
                                                               const float* __restrict__ table,
                                                               const int* __restrict__ actions, long units, int per_image,
                                                               int C, unsigned short* __restrict__ out) {
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long)gridDim.x * 256) {
    const int b = (int)(u / per_image), c8 = (int)(u % (C / 8));
    const float* v = table + (long)actions[b] * C + c8 * 8;
    const float4 v0 = *reinterpret_cast<const float4*>(v), v1 = *reinterpret_cast<const float4*>(v + 4);
    const u16x8 x = *reinterpret_cast<const u16x8*>(h + u * 8);
    u16x8 y;
    y[0] = f32_to_bf16_bits(bf16_bits_to_f32(x[0]) + v0.x); y[1] = f32_to_bf16_bits(bf16_bits_to_f32(x[1]) + v0.y);
    y[2] = f32_to_bf16_bits(bf16_bits_to_f32(x[2]) + v0.z); y[3] = f32_to_bf16_bits(bf16_bits_to_f32(x[3]) + v0.w);
    y[4] = f32_to_bf16_bits(bf16_bits_to_f32(x[4]) + v1.x); y[5] = f32_to_bf16_bits(bf16_bits_to_f32(x[5]) + v1.y);
    y[6] = f32_to_bf16_bits(bf16_bits_to_f32(x[6]) + v1.z); y[7] = f32_to_bf16_bits(bf16_bits_to_f32(x[7]) + v1.w);
    *reinterpret_cast<u16x8*>(out + u * 8) = y;
  }
}

// Its gradient with respect to the table: dtable[a][c] = sum over the images b with actions[b] == a and over
// their pixels of dy[b][pixel][c].  Two launches: per-image pixel sums (one workgroup per image: thread = channel
// group x row lane, lanes combined through LDS), then per (a, c) the images in index order - no atomics, the
// same bits every time.
__global__ __launch_bounds__(256) void pixel_sum_kernel(const unsigned short* __restrict__ dy, int HW, int C,
                                                       float* __restrict__ sums) {
  __shared__ float red[256 * 8];
  const int groups = C / 8, lanes = 256 / groups;
  const int cg = threadIdx.x % groups, rl = threadIdx.x / groups;
  const unsigned short* img = dy + (long)blockIdx.x * HW * C;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < lanes)
    for (int r = rl; r < HW; r += lanes) {
      const u16x8 v = *reinterpret_cast<const u16x8*>(img + (long)r * C + cg * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += bf16_bits_to_f32(v[j]);
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[(rl * groups + cg) * 8 + j] = rl < lanes ? s[j] : 0.f;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[l * C + c];
    sums[(long)blockIdx.x * C + c] = t;
  }
}

__global__ __launch_bounds__(256) void action_group_sum_kernel(const float* __restrict__ sums,
                                                              const int* __restrict__ actions, int B, int C, int A,
                                                              float* __restrict__ dtable) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A * C) return;
  const int a = i / C, c = i - a * C;
  float t = 0.f;
  for (int b = 0; b < B; ++b)
    if (actions[b] == a) t += sums[(long)b * C + c];
  dtable[i] = t;
}

}  // namespace

extern "C" {

int mz_add_action_vector_bf16(const void* h, const float* table, const int32_t* actions, void* out, int32_t B,
                              int32_t HW, int32_t C, mz_stream_t stream) {
  MZ_REQUIRE(h && table && actions && out && B > 0 && HW > 0, "mz_add_action_vector_bf16: null tensor or empty batch");
  MZ_REQUIRE(C >= 8 && C % 8 == 0, "mz_add_action_vector_bf16: C = %d must be a multiple of 8", C);
  const long units = (long)B * HW * (C / 8);
  const long want = (units + 255) / 256, cap = 8L * mz::compute_units();
  hipLaunchKernelGGL(add_action_vector_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0,
                     mz::as_stream(stream), static_cast<const unsigned short*>(h), table, actions, units, HW * (C / 8), C,
                     static_cast<unsigned short*>(out));
  return mz::check_launch("mz_add_action_vector_bf16");
}

int mz_action_vector_grad_bf16(const void* dy, const int32_t* actions, int32_t B, int32_t HW, int32_t C, int32_t A,
                               float* image_sums, float* dtable, mz_stream_t stream) {
  MZ_REQUIRE(dy && actions && image_sums && dtable && B > 0 && HW > 0 && A > 0,
             "mz_action_vector_grad_bf16: null tensor or empty batch");
  MZ_REQUIRE(C >= 8 && C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0,
             "mz_action_vector_grad_bf16: C = %d must be 8 times a power of two, at most 2048", C);
  hipStream_t s = mz::as_stream(stream);
  hipLaunchKernelGGL(pixel_sum_kernel, dim3(B), dim3(256), 0, s, static_cast<const unsigned short*>(dy), HW, C, image_sums);
  hipLaunchKernelGGL(action_group_sum_kernel, dim3((A * C + 255) / 256), dim3(256), 0, s, image_sums, actions, B, C, A,
                     dtable);
  return mz::check_launch("mz_action_vector_grad_bf16");
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
