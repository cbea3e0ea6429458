// Weight gradient of the 12x12 256->256 3x3 convolution (the shape of 23 of the network's 31
// convolutions) for the unrolled training step (compute_losses / update_weights, train.py:295-431):
//     dW[co][ci][ky][kx] = sum over images n and pixels (y, x) of
//                          gout[n][y][x][co] * x[n][y + ky - 1][x + kx - 1][ci]      (zero outside the image)
// bf16 operands, float32 accumulation on the matrix cores.  The data gradient needs no kernel of its
// own: it is the forward convolution of gout with the transposed, mirrored weights
// (mz_weights_fold_pack with transpose = 1, then mz_conv3x3_bf16).
//
// As a GEMM the reduction runs over pixels, which are the SLOW index of both NHWC operands; gfx950's
// transposing LDS read (ds_read_b64_tr_b16) delivers exactly that: per 16 lanes it reads 4 rows
// (pixels) x 16 columns (channels) and hands every lane one column - 4 pixels of one channel.  Both
// operands take the same pixels for the same k index, so the k order is free: lane group kq of a
// 32-pixel k-step holds pixels k0 + 4kq + {0..3} and k0 + 16 + 4kq + {0..3}, which makes the 8 rows a
// 32-lane half reads consecutive pixels (conflict-free with the row strides below).
//
// One workgroup owns a 64 (co) x 128 (ci) x 9 (taps) tile of dW - 8 tiles cover the layer - and a
// share of the images; its 8 wavefronts split the 128 input channels, each keeping 4 x 9 accumulator
// tiles (144 registers).  Per image the tile's slices of gout (144 x 64) and of x with a zero halo
// (14 x 14 x 128) are staged in LDS; the next image travels in registers meanwhile.  Partial sums
// of the image shares go to a float32 scratch buffer [share][co][ci][ky][kx] (the reference's layout of
// dW); a second kernel adds the shares up.
#include "common.h"
#include "conv_common.h"

namespace {

constexpr int kCoT = 64, kCiT = 128;          // tile of dW per workgroup (x 9 taps)
constexpr int kGRow = kCoT * 2 + 32;          // LDS bytes per gout pixel row: 160 (row r starts at bank 40 r)
constexpr int kXRow = kCiT * 2 + 32;          // LDS bytes per x pixel row: 288 (bank 72 r = 8 r mod 64)
constexpr int kGRows = 160;                   // 144 pixels + 16 zero rows (the fifth k-step's upper half)
constexpr int kPadSide = kSide + 2;           // 14
constexpr int kXRows = kPadSide * kPadSide;   // 196 rows incl. the zero halo
constexpr int kGBytes = kGRows * kGRow;       // 25 600
constexpr int kXBytes = kXRows * kXRow;       // 56 448
constexpr int kWgradLds = kGBytes + kXBytes;  // 82 048
constexpr int kKSteps = kGRows / 32;          // 5
constexpr int kGUnits = kPix * (kCoT / 8);    // 16-byte units of a gout slice: 1152
constexpr int kXUnits = kPix * (kCiT / 8);    // of an x slice: 2304

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s16x4 read_tr(int lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_addr));
}

// The images may come in several equally sized segments (the same layer applied at several unroll steps:
// one pass then produces the sum of their weight gradients): image n is image n % seg_images of segment
// n / seg_images.
constexpr int kMaxSegments = 8;
struct WgradSrc {
  const __bf16* gout[kMaxSegments];
  const __bf16* x[kMaxSegments];
  int seg_images;
};

__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(WgradSrc src, int n_images, float* __restrict__ partial,
                                                           float* __restrict__ db_partial) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* gs = lds;
  char* xs = lds + kGBytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x & 7, share = blockIdx.x >> 3, shares = gridDim.x >> 3;
  const int co0 = (tile >> 1) * kCoT, ci0 = (tile & 1) * kCiT;
  const int kq = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;

  // zero the whole staging area once: the halo of x and gout's rows 144..159 are never written again
  for (int i = tid; i < kWgradLds / 16; i += 512) *reinterpret_cast<uint4*>(lds + i * 16) = make_uint4(0, 0, 0, 0);

  // this thread's 16-byte units of an image: 3 of the gout slice, 5 of the x slice (the last round of each is
  // partial: threads past the end reload the last unit and never store it).  Eight named registers
  // rather than an array: the array form ended up in scratch memory.
  static_assert((kGUnits + 511) / 512 == 3 && (kXUnits + 511) / 512 == 5, "staging rounds");
  auto gsrc = [&](int j) { int u = tid + 512 * j; u = u < kGUnits ? u : kGUnits - 1; return (u >> 3) * kCh + (u & 7) * 8; };
  auto xsrc = [&](int j) { int v = tid + 512 * j; v = v < kXUnits ? v : kXUnits - 1; return (v >> 4) * kCh + (v & 15) * 8; };
  const int gs0 = gsrc(0), gs1 = gsrc(1), gs2 = gsrc(2);
  const int xs0 = xsrc(0), xs1 = xsrc(1), xs2 = xsrc(2), xs3 = xsrc(3), xs4 = xsrc(4);
  auto gdst = [&](int j) { const int u = tid + 512 * j; return (u >> 3) * kGRow + (u & 7) * 16; };
  auto xdst = [&](int j) {
    const int v = tid + 512 * j, pix = v >> 4;
    const int py = pix / kSide, px = pix - py * kSide;
    return ((py + 1) * kPadSide + px + 1) * kXRow + (v & 15) * 16;
  };
  const int gd0 = gdst(0), gd1 = gdst(1), gd2 = gdst(2);
  const int xd0 = xdst(0), xd1 = xdst(1), xd2 = xdst(2), xd3 = xdst(3), xd4 = xdst(4);
  const bool g2_ok = tid + 1024 < kGUnits, x4_ok = tid + 2048 < kXUnits;
  uint4 g0, g1, g2, x0, x1, x2, x3, x4;
#define MZ_LOAD_IMAGE(n_)                                                      \
  {                                                                            \
    const int seg_ = (n_) / src.seg_images, in_seg_ = (n_) - seg_ * src.seg_images;  \
    const __bf16* gi = src.gout[seg_] + (long)in_seg_ * (kPix * kCh) + co0;    \
    const __bf16* xi = src.x[seg_] + (long)in_seg_ * (kPix * kCh) + ci0;       \
    g0 = *reinterpret_cast<const uint4*>(gi + gs0);                            \
    g1 = *reinterpret_cast<const uint4*>(gi + gs1);                            \
    g2 = *reinterpret_cast<const uint4*>(gi + gs2);                            \
    x0 = *reinterpret_cast<const uint4*>(xi + xs0);                            \
    x1 = *reinterpret_cast<const uint4*>(xi + xs1);                            \
    x2 = *reinterpret_cast<const uint4*>(xi + xs2);                            \
    x3 = *reinterpret_cast<const uint4*>(xi + xs3);                            \
    x4 = *reinterpret_cast<const uint4*>(xi + xs4);                            \
  }

  f32x4 acc[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool want_db = db_partial != nullptr && (tile & 1) == 0;  // one of the two input-channel tiles does it
  float dbacc = 0.f;

  const int lds_base = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  // lane 4q + p4 of a 16-lane group supplies row q, columns 4 p4 .. 4 p4 + 3 of the block it reads
  const int g_lane = lds_base + (4 * kq + q) * kGRow + p4 * 8;                         // + pixel0 * kGRow + co tile * 32
  const int x_col = lds_base + kGBytes + wave * 32 + p4 * 8;                           // + padded row * kXRow

  int n = share;
  if (n < n_images) MZ_LOAD_IMAGE(n);
  __syncthreads();  // zeroing done
  for (; n < n_images; n += shares) {
    *reinterpret_cast<uint4*>(gs + gd0) = g0;
    *reinterpret_cast<uint4*>(gs + gd1) = g1;
    if (g2_ok) *reinterpret_cast<uint4*>(gs + gd2) = g2;
    *reinterpret_cast<uint4*>(xs + xd0) = x0;
    *reinterpret_cast<uint4*>(xs + xd1) = x1;
    *reinterpret_cast<uint4*>(xs + xd2) = x2;
    *reinterpret_cast<uint4*>(xs + xd3) = x3;
    if (x4_ok) *reinterpret_cast<uint4*>(xs + xd4) = x4;
    __syncthreads();
    if (n + shares < n_images) MZ_LOAD_IMAGE(n + shares);  // travels while this image is multiplied
    if (want_db) {  // bias gradient = column sums of gout: 8 row groups x 64 channels of the staged slice
#pragma unroll
      for (int r = tid >> 6; r < kPix; r += 8)
        dbacc += __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(gs + r * kGRow + (tid & 63) * 2) << 16);
    }
#pragma unroll 1
    for (int ks = 0; ks < kKSteps; ++ks) {
      // the two pixel rows this lane addresses in this k-step, and where their tap-(0,0) neighbours sit in xs
      const int pa = ks * 32 + 4 * kq + q, pb = pa + 16;
      const int pya = pa / kSide, pxa = pa - pya * kSide;
      const int pyb = pb / kSide, pxb = pb - pyb * kSide;
      const int xa = x_col + (pa < kPix ? pya * kPadSide + pxa : 0) * kXRow;   // rows past the image: gout is zero there
      const int xb = x_col + (pb < kPix ? pyb * kPadSide + pxb : 0) * kXRow;
      s16x8 af[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const s16x4 lo = read_tr(g_lane + ks * 32 * kGRow + c * 32);
        const s16x4 hi = read_tr(g_lane + (ks * 32 + 16) * kGRow + c * 32);
        af[c] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int toff = ((t / 3) * kPadSide + t % 3) * kXRow;
        const s16x4 lo = read_tr(xa + toff);
        const s16x4 hi = read_tr(xb + toff);
        const s16x8 bfr = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[c]),
                                                              __builtin_bit_cast(bf16x8, bfr), acc[c][t], 0, 0, 0);
      }
    }
    __syncthreads();  // everyone has read this image
  }

  if (want_db) {  // (the loop's last barrier has passed: the staging area is free)
    float* red = reinterpret_cast<float*>(lds);
    red[tid] = dbacc;
    __syncthreads();
    if (tid < 64) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) s += red[g * 64 + tid];
      db_partial[share * kCh + co0 + tid] = s;
    }
  }
  // partial[share][co][ci][tap] (the layout of dW itself).  A lane holds rows (co) 4 kq .. 4 kq + 3 of column
  // (ci) lane % 16: for one accumulator row r the wavefront owns, in each of 4 rows of dW, 16 ci x 9 taps = 576
  // consecutive bytes.  They pass through a wavefront-private piece of LDS (the staging area is free by now; the
  // first 2 KB may still hold the bias sums) so that they leave as whole 16-byte units of consecutive addresses.
  float* dst = partial + (long)share * 9 * kCh * kCh;
  float* tr = reinterpret_cast<float*>(lds + 2048) + wave * 576;
  const int l16 = lane & 15;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int t = 0; t < 9; ++t) tr[kq * 144 + l16 * 9 + t] = acc[c][t][r];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int f = lane + 64 * j;  // float4 index within the 4 x 144 floats
        if (f < 144) {
          const int row = f / 36, col4 = f - row * 36;
          const float4 v = *reinterpret_cast<const float4*>(tr + row * 144 + col4 * 4);
          const long o = ((long)(co0 + c * 16 + 4 * row + r) * kCh + ci0 + wave * 16) * 9 + col4 * 4;
          *reinterpret_cast<float4*>(dst + o) = v;
        }
      }
    }
}

// dW = sum over shares of partial[share], in share order; four consecutive floats per thread
constexpr int kDw4 = 9 * kCh * kCh / 4;
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float4* __restrict__ partial, int shares,
                                                          float4* __restrict__ dw, const float* __restrict__ db_partial,
                                                          float* __restrict__ db) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // the grid covers kDw4 exactly
  if (db && blockIdx.x == 0) {
    float s = 0.f;
    for (int g = 0; g < shares; ++g) s += db_partial[g * kCh + threadIdx.x];
    db[threadIdx.x] = s;
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* p = partial + i;
  int g = 0;
  for (; g + 8 <= shares; g += 8) {  // eight loads in flight per thread
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[(long)(g + k) * kDw4];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
  }
  for (; g < shares; ++g) {
    const float4 v = p[(long)g * kDw4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  dw[i] = s;
}

}  // namespace

extern "C" {

int32_t mz_conv_wgrad_shares(int32_t B) {
  if (B <= 0) return 0;
  const int by_cu = mz::compute_units() / 8;
  const int s = B < by_cu ? B : by_cu;
  return s < 1 ? 1 : s;
}

int64_t mz_conv_wgrad_workspace_bytes(int32_t B) {
  return (int64_t)mz_conv_wgrad_shares(B) * (9 * kCh * kCh + kCh) * 4;  // dW partials, then bias partials
}

static int wgrad_launch(const WgradSrc& src, int n_images, float* dw, float* dbias, void* workspace,
                        int64_t workspace_bytes, hipStream_t s, const char* who) {
  MZ_REQUIRE(((uintptr_t)dw | (uintptr_t)workspace) % 16 == 0, "%s: dw and workspace must be 16-byte aligned", who);
  MZ_REQUIRE(workspace_bytes >= mz_conv_wgrad_workspace_bytes(n_images), "%s: workspace of %lld bytes, %lld needed", who,
             (long long)workspace_bytes, (long long)mz_conv_wgrad_workspace_bytes(n_images));
  static bool configured[mz::kMaxDevices] = {false};
  if (int rc = mz::ensure_dynamic_lds(reinterpret_cast<const void*>(conv_wgrad_kernel), kWgradLds, configured,
                                      "conv_wgrad"))
    return rc;
  const int shares = mz_conv_wgrad_shares(n_images);
  float* dbp = dbias ? static_cast<float*>(workspace) + (long)shares * 9 * kCh * kCh : nullptr;
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3(8 * shares), dim3(512), kWgradLds, s, src, n_images,
                     static_cast<float*>(workspace), dbp);
  static_assert(kDw4 % 256 == 0, "reduce grid");
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(kDw4 / 256), dim3(256), 0, s, static_cast<const float4*>(workspace), shares,
                     reinterpret_cast<float4*>(dw), dbp, dbias);
  return mz::check_launch(who);
}

int mz_conv_wgrad_bf16(const void* grad_out, const void* x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                       int32_t Cout, float* dw, float* dbias, void* workspace, int64_t workspace_bytes,
                       mz_stream_t stream) {
  MZ_REQUIRE(grad_out && x && dw && workspace && B > 0, "mz_conv_wgrad_bf16: null tensor or empty batch");
  MZ_REQUIRE(H == kSide && W == kSide && Cin == kCh && Cout == kCh,
             "mz_conv_wgrad_bf16: only the 12x12, 256->256 layer shape has a kernel (got %dx%d, %d->%d)", H, W, Cin, Cout);
  WgradSrc src{};
  src.gout[0] = static_cast<const __bf16*>(grad_out);
  src.x[0] = static_cast<const __bf16*>(x);
  src.seg_images = B;
  return wgrad_launch(src, B, dw, dbias, workspace, workspace_bytes, mz::as_stream(stream), "mz_conv_wgrad_bf16");
}

int mz_conv_wgrad_bf16_multi(const void* const* grad_out, const void* const* x, int32_t n_segments, int32_t B,
                             int32_t H, int32_t W, int32_t Cin, int32_t Cout, float* dw, float* dbias, void* workspace,
                             int64_t workspace_bytes, mz_stream_t stream) {
  MZ_REQUIRE(grad_out && x && dw && workspace && B > 0, "mz_conv_wgrad_bf16_multi: null tensor or empty batch");
  MZ_REQUIRE(n_segments >= 1 && n_segments <= kMaxSegments, "mz_conv_wgrad_bf16_multi: n_segments = %d outside 1..%d",
             n_segments, kMaxSegments);
  MZ_REQUIRE(H == kSide && W == kSide && Cin == kCh && Cout == kCh,
             "mz_conv_wgrad_bf16_multi: only the 12x12, 256->256 layer shape has a kernel (got %dx%d, %d->%d)", H, W, Cin,
             Cout);
  WgradSrc src{};
  for (int i = 0; i < n_segments; ++i) {
    MZ_REQUIRE(grad_out[i] && x[i], "mz_conv_wgrad_bf16_multi: segment %d has a null tensor", i);
    src.gout[i] = static_cast<const __bf16*>(grad_out[i]);
    src.x[i] = static_cast<const __bf16*>(x[i]);
  }
  src.seg_images = B;
  return wgrad_launch(src, n_segments * B, dw, dbias, workspace, workspace_bytes, mz::as_stream(stream),
                      "mz_conv_wgrad_bf16_multi");
}

}  // extern "C"
