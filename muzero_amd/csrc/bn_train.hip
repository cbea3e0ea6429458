// Training-mode BatchNorm fused with what surrounds it in the reference's blocks (model.py:143-175, :237):
//     y = relu( batch_norm(x) [+ residual] )
// forward and backward, for bf16 NHWC activations [rows = N*H*W][C] (torch's channels_last), in the unrolled
// training step (train.py:330-431).  Batch statistics and the two gradient sums are accumulated in double;
// running statistics move with momentum 0.1 and the unbiased variance, as torch.nn.BatchNorm2d does.
// HBM-bound: forward reads x twice (+ residual) and writes y; backward reads dy, y, x twice and writes dx
// (+ the residual branch's gradient).  Three short launches each way: column sums per workgroup, a
// one-workgroup finalisation, an elementwise pass.
#include "common.h"

namespace {

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                               // nearest even
  return (unsigned short)(u >> 16);
}

constexpr int kStatsThreads = 256;
constexpr int kMaxStatBlocks = 256;

// Column sums over a strided share of the rows.  MODE 0: sum x, sum x^2.  MODE 1: with g = dy * (y > 0) (or dy when
// relu == 0) and xhat = (x - mean) * invstd: sum g, sum g * xhat.  partial: double [blocks][2][C].
template <int MODE>
__global__ __launch_bounds__(kStatsThreads) void bn_colsum_kernel(const unsigned short* __restrict__ x,
                                                                 const unsigned short* __restrict__ dy,
                                                                 const unsigned short* __restrict__ y,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, int relu,
                                                                 long rows, int C, double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int groups = C / 8;                       // 16-byte channel groups per row
  const int lanes = kStatsThreads / groups;       // rows in flight per workgroup
  const int cg = threadIdx.x % groups, rl = threadIdx.x / groups;
  double s0[8], s1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.0;
  float mu[8], is[8];
  if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = mean[cg * 8 + j]; is[j] = invstd[cg * 8 + j]; }
  }
  auto accumulate = [&](const u16x8& xv, const u16x8& gv, const u16x8& yv) __attribute__((always_inline)) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double v = (double)bf16_to_f32(xv[j]);
        s0[j] += v;
        s1[j] += v * v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float g = bf16_to_f32(gv[j]);
        if (relu && !(bf16_to_f32(yv[j]) > 0.f)) g = 0.f;
        const float xh = (bf16_to_f32(xv[j]) - mu[j]) * is[j];
        s0[j] += (double)g;
        s1[j] += (double)g * (double)xh;
      }
    }
  };
  if (rl < lanes) {
    // three rows per trip: their loads are in flight together (a thread walks ~9 rows at the training batch, and a
    // round trip to HBM per row was most of the kernel's time)
    const long step = (long)gridDim.x * lanes;
    long r = (long)blockIdx.x * lanes + rl;
    for (; r + 2 * step < rows; r += 3 * step) {
      u16x8 xv[3], gv[3], yv[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const long o = (r + k * step) * C + cg * 8;
        xv[k] = *reinterpret_cast<const u16x8*>(x + o);
        if (MODE == 1) {
          gv[k] = *reinterpret_cast<const u16x8*>(dy + o);
          yv[k] = gv[k];
        }
      }
      if (MODE == 1 && relu) {
#pragma unroll
        for (int k = 0; k < 3; ++k) yv[k] = *reinterpret_cast<const u16x8*>(y + (r + k * step) * C + cg * 8);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) accumulate(xv[k], gv[k], yv[k]);
    }
    for (; r < rows; r += step) {
      const u16x8 xv = *reinterpret_cast<const u16x8*>(x + r * C + cg * 8);
      u16x8 gv = xv, yv = xv;
      if (MODE == 1) {
        gv = *reinterpret_cast<const u16x8*>(dy + r * C + cg * 8);
        yv = gv;
        if (relu) yv = *reinterpret_cast<const u16x8*>(y + r * C + cg * 8);
      }
      accumulate(xv, gv, yv);
    }
  }
  // combine the row lanes of this workgroup: [lane][2][C] doubles in LDS
  double* red = reinterpret_cast<double*>(smem);
  if (rl < lanes) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(rl * 2 + 0) * C + cg * 8 + j] = s0[j];
      red[(rl * 2 + 1) * C + cg * 8 + j] = s1[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += kStatsThreads) {
    double s = 0.0;
    for (int l = 0; l < lanes; ++l) s += red[l * 2 * C + i];
    partial[(long)blockIdx.x * 2 * C + i] = s;
  }
}

// Sum of the workgroups' partial column sums for channel c (both statistics), by the 32 threads that share
// the channel: thread (lane, channel) adds blocks lane, lane + 32, ...; the lanes are combined through LDS.
// One workgroup of 256 threads = 8 channels.  Returns the totals in every thread of the channel.
__device__ __forceinline__ void sum_partials(const double* __restrict__ partial, int blocks, int C, int c, int lane,
                                             double& s, double& q) {
  __shared__ double red[2][32][8];
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int k = lane; k < blocks; k += 32) {
      a += partial[(long)k * 2 * C + c];
      b += partial[(long)k * 2 * C + C + c];
    }
  red[0][lane][threadIdx.x & 7] = a;
  red[1][lane][threadIdx.x & 7] = b;
  __syncthreads();
  s = q = 0.0;
#pragma unroll
  for (int l = 0; l < 32; ++l) {
    s += red[0][l][threadIdx.x & 7];
    q += red[1][l][threadIdx.x & 7];
  }
}

// forward finalisation: batch mean / inverse standard deviation, running statistics, and the affine pair the
// elementwise pass applies (y = x * scale + shift)
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const double* __restrict__ partial, int blocks, long rows, int C, double eps,
                                       double momentum, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* __restrict__ running_mean,
                                       float* __restrict__ running_var, float* __restrict__ save_mean,
                                       float* __restrict__ save_invstd, float* __restrict__ scale,
                                       float* __restrict__ shift) {
  const int c = blockIdx.x * 8 + (threadIdx.x & 7), lane = threadIdx.x >> 3;
  double s, q;
  sum_partials(partial, blocks, C, c, lane, s, q);
  if (c >= C || lane != 0) return;
  const double m = s / (double)rows;
  double var = q / (double)rows - m * m;
  var = var > 0.0 ? var : 0.0;
  const double inv = 1.0 / sqrt(var + eps);
  save_mean[c] = (float)m;
  save_invstd[c] = (float)inv;
  const double sc = (double)gamma[c] * inv;
  scale[c] = (float)sc;
  shift[c] = (float)((double)beta[c] - m * sc);
  if (running_mean) {
    const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
  }
}

// Elementwise passes: a thread owns one 16-byte channel group, keeps that group's per-channel parameters in
// registers and walks down the rows (the grid's thread count is a multiple of the groups per row).
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const unsigned short* __restrict__ x,
                                                      const unsigned short* __restrict__ residual,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      int relu, long units, int C, unsigned short* __restrict__ y) {
  const long first = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
  const int c0 = (int)(first % (C / 8)) * 8;
  float sc[8], sh[8];
  load8(scale + c0, sc);
  load8(shift + c0, sh);
  for (long u = first; u < units; u += stride) {
    const u16x8 xv = *reinterpret_cast<const u16x8*>(x + u * 8);
    u16x8 rv = xv;
    if (residual) rv = *reinterpret_cast<const u16x8*>(residual + u * 8);
    u16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = bf16_to_f32(xv[j]) * sc[j] + sh[j];
      if (residual) v += bf16_to_f32(rv[j]);
      if (relu) v = fmaxf(v, 0.f);
      out[j] = f32_to_bf16(v);
    }
    *reinterpret_cast<u16x8*>(y + u * 8) = out;
  }
}

// backward finalisation: dgamma, dbeta and the three per-channel coefficients of dx = a * (g - b - xhat * c)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int blocks, long rows, int C,
                                       const float* __restrict__ gamma, const float* __restrict__ invstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ coef, int accumulate) {
  const int c = blockIdx.x * 8 + (threadIdx.x & 7), lane = threadIdx.x >> 3;
  double sg, sgx;
  sum_partials(partial, blocks, C, c, lane, sg, sgx);
  if (c >= C || lane != 0) return;
  // accumulate: a layer applied at several unroll steps adds each step's gradient to the running sum (the same
  // float32 additions, in the same order, that autograd's accumulation would make - without a launch each)
  dbeta[c] = accumulate ? dbeta[c] + (float)sg : (float)sg;
  dgamma[c] = accumulate ? dgamma[c] + (float)sgx : (float)sgx;
  coef[c] = gamma[c] * invstd[c];
  coef[C + c] = (float)(sg / (double)rows);
  coef[2 * C + c] = (float)(sgx / (double)rows);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const unsigned short* __restrict__ x,
                                                          const unsigned short* __restrict__ dy,
                                                          const unsigned short* __restrict__ y,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ coef, int relu, long units, int C,
                                                          unsigned short* __restrict__ dx,
                                                          unsigned short* __restrict__ dres) {
  const long first = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
  const int c0 = (int)(first % (C / 8)) * 8;
  float mu[8], is[8], ka[8], kb[8], kc[8];
  load8(mean + c0, mu);
  load8(invstd + c0, is);
  load8(coef + c0, ka);
  load8(coef + C + c0, kb);
  load8(coef + 2 * C + c0, kc);
  for (long u = first; u < units; u += stride) {
    const u16x8 xv = *reinterpret_cast<const u16x8*>(x + u * 8);
    const u16x8 gv = *reinterpret_cast<const u16x8*>(dy + u * 8);
    u16x8 yv = gv;
    if (relu) yv = *reinterpret_cast<const u16x8*>(y + u * 8);
    u16x8 ox, og;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float g = bf16_to_f32(gv[j]);
      if (relu && !(bf16_to_f32(yv[j]) > 0.f)) g = 0.f;
      const float xh = (bf16_to_f32(xv[j]) - mu[j]) * is[j];
      ox[j] = f32_to_bf16(ka[j] * (g - kb[j] - xh * kc[j]));
      og[j] = f32_to_bf16(g);
    }
    *reinterpret_cast<u16x8*>(dx + u * 8) = ox;
    if (dres) *reinterpret_cast<u16x8*>(dres + u * 8) = og;
  }
}

// workgroups of an elementwise pass: enough to fill the chip, thread count a multiple of the groups per row
int apply_blocks(long units) {
  const long want = (units + 255) / 256;
  const long cap = 4L * mz::compute_units();
  return (int)(want < cap ? want : cap);
}

int stat_blocks(long rows, int C) {
  const int lanes = kStatsThreads / (C / 8);
  long b = (rows + lanes - 1) / lanes;
  const int cap = mz::compute_units() < kMaxStatBlocks ? mz::compute_units() : kMaxStatBlocks;
  return (int)(b < cap ? b : cap);
}

int check_shape(const char* who, int64_t rows, int32_t C) {
  MZ_REQUIRE(rows > 0 && C >= 8 && C % 8 == 0 && C <= 1024 && kStatsThreads % (C / 8) == 0,
             "%s: rows > 0 and C in {8, 16, 32, 64, 128, 256, 512, 1024} expected (rows = %lld, C = %d)", who,
             (long long)rows, C);
  return MZ_OK;
}

}  // namespace

extern "C" {

int64_t mz_bn_workspace_bytes(int32_t C) { return C > 0 ? (int64_t)kMaxStatBlocks * 2 * C * 8 + 3 * (int64_t)C * 4 : 0; }

static int bn_fwd(const char* who, const void* x, const void* residual, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, double eps, double momentum, int32_t relu, int64_t rows,
                  int32_t C, void* y, float* save_mean, float* save_invstd, const double* stats, int32_t stat_blocks_in,
                  void* workspace, int64_t workspace_bytes, mz_stream_t stream) {
  MZ_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && workspace, "%s: null argument", who);
  MZ_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "%s: running statistics go together", who);
  if (int rc = check_shape(who, rows, C)) return rc;
  MZ_REQUIRE(workspace_bytes >= mz_bn_workspace_bytes(C), "%s: workspace too small", who);
  hipStream_t s = mz::as_stream(stream);
  double* partial = static_cast<double*>(workspace);
  float* scale = reinterpret_cast<float*>(partial + (long)kMaxStatBlocks * 2 * C);
  float* shift = scale + C;
  const auto* xs = static_cast<const unsigned short*>(x);
  int blocks = stat_blocks_in;
  const double* sums = stats;
  if (!stats) {  // column sums of x per workgroup
    blocks = stat_blocks(rows, C);
    const int lanes = kStatsThreads / (C / 8);
    hipLaunchKernelGGL(bn_colsum_kernel<0>, dim3(blocks), dim3(kStatsThreads), (size_t)lanes * 2 * C * 8, s, xs, nullptr,
                       nullptr, nullptr, nullptr, 0, (long)rows, C, partial);
    sums = partial;
  }
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(C / 8), dim3(256), 0, s, sums, blocks, (long)rows, C, eps, momentum,
                     gamma, beta, running_mean, running_var, save_mean, save_invstd, scale, shift);
  const long units = rows * (C / 8);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(apply_blocks(units)), dim3(256), 0, s, xs,
                     static_cast<const unsigned short*>(residual), scale, shift, relu, units, C,
                     static_cast<unsigned short*>(y));
  return mz::check_launch(who);
}

int mz_bn_relu_train_fwd(const void* x, const void* residual, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, double eps, double momentum, int32_t relu,
                         int64_t rows, int32_t C, void* y, float* save_mean, float* save_invstd, void* workspace,
                         int64_t workspace_bytes, mz_stream_t stream) {
  return bn_fwd("mz_bn_relu_train_fwd", x, residual, gamma, beta, running_mean, running_var, eps, momentum, relu, rows, C,
                y, save_mean, save_invstd, nullptr, 0, workspace, workspace_bytes, stream);
}

int mz_bn_relu_train_fwd_stats(const void* x, const void* residual, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, double eps, double momentum, int32_t relu,
                               int64_t rows, int32_t C, void* y, float* save_mean, float* save_invstd,
                               const double* stats, int32_t stat_blocks, void* workspace, int64_t workspace_bytes,
                               mz_stream_t stream) {
  MZ_REQUIRE(stats && stat_blocks > 0, "mz_bn_relu_train_fwd_stats: stats [stat_blocks][2][C] expected");
  return bn_fwd("mz_bn_relu_train_fwd_stats", x, residual, gamma, beta, running_mean, running_var, eps, momentum, relu,
                rows, C, y, save_mean, save_invstd, stats, stat_blocks, workspace, workspace_bytes, stream);
}

int mz_bn_relu_train_bwd_acc(const void* x, const void* dy, const void* y, const float* gamma, const float* save_mean,
                             const float* save_invstd, int32_t relu, int64_t rows, int32_t C, void* dx,
                             void* dresidual, float* dgamma, float* dbeta, int32_t accumulate, void* workspace,
                             int64_t workspace_bytes, mz_stream_t stream);

int mz_bn_relu_train_bwd(const void* x, const void* dy, const void* y, const float* gamma, const float* save_mean,
                         const float* save_invstd, int32_t relu, int64_t rows, int32_t C, void* dx, void* dresidual,
                         float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, mz_stream_t stream) {
  return mz_bn_relu_train_bwd_acc(x, dy, y, gamma, save_mean, save_invstd, relu, rows, C, dx, dresidual, dgamma, dbeta, 0,
                                  workspace, workspace_bytes, stream);
}

int mz_bn_relu_train_bwd_acc(const void* x, const void* dy, const void* y, const float* gamma, const float* save_mean,
                             const float* save_invstd, int32_t relu, int64_t rows, int32_t C, void* dx,
                             void* dresidual, float* dgamma, float* dbeta, int32_t accumulate, void* workspace,
                             int64_t workspace_bytes, mz_stream_t stream) {
  MZ_REQUIRE(x && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace,
             "mz_bn_relu_train_bwd: null argument");
  MZ_REQUIRE(!relu || y, "mz_bn_relu_train_bwd: the ReLU mask needs the forward output y");
  if (int rc = check_shape("mz_bn_relu_train_bwd", rows, C)) return rc;
  MZ_REQUIRE(workspace_bytes >= mz_bn_workspace_bytes(C), "mz_bn_relu_train_bwd: workspace too small");
  hipStream_t s = mz::as_stream(stream);
  double* partial = static_cast<double*>(workspace);
  float* coef = reinterpret_cast<float*>(partial + (long)kMaxStatBlocks * 2 * C);
  const int blocks = stat_blocks(rows, C);
  const int lanes = kStatsThreads / (C / 8);
  const auto* xs = static_cast<const unsigned short*>(x);
  const auto* gs = static_cast<const unsigned short*>(dy);
  const auto* ys = static_cast<const unsigned short*>(y);
  hipLaunchKernelGGL(bn_colsum_kernel<1>, dim3(blocks), dim3(kStatsThreads), (size_t)lanes * 2 * C * 8, s, xs, gs, ys,
                     save_mean, save_invstd, relu, (long)rows, C, partial);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C / 8), dim3(256), 0, s, partial, blocks, (long)rows, C, gamma,
                     save_invstd, dgamma, dbeta, coef, accumulate ? 1 : 0);
  const long units = rows * (C / 8);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(apply_blocks(units)), dim3(256), 0, s, xs, gs, ys, save_mean,
                     save_invstd, coef, relu, units, C, static_cast<unsigned short*>(dx),
                     static_cast<unsigned short*>(dresidual));
  return mz::check_launch("mz_bn_relu_train_bwd");
}

}  // extern "C"
