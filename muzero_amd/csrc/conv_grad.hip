// Weight gradient of the 3x3 convolutions for the unrolled training step (compute_losses / update_weights,
// train.py:295-431); first the 12x12 256->256 layer (the shape of 23 of the network's 31 convolutions), then
// - same scheme, a band of output rows at a time - the four other shapes of the representation network:
//     dW[co][ci][ky][kx] = sum over images n and output pixels (y, x) of
//                          gout[n][y][x][co] * x[n][s y + ky - 1][s x + kx - 1][ci]  (zero outside the image)
// bf16 operands, float32 accumulation on the matrix cores.  The data gradient of a stride-1 layer needs no
// kernel of its own: it is the forward convolution of gout with the transposed, mirrored weights
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

// acc + (operand tile `which` of af)^T x ones: the column sums of that tile's 16 channels over the k-step's 32
// pixels, in every column of the result.  `which` is uniform in the wavefront.
__device__ __forceinline__ f32x4 column_sums(s16x8 a0, s16x8 a1, s16x8 a2, s16x8 a3, int which, f32x4 acc) {
  const s16x8 ones = s16x8{0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};   // bf16 1.0
  const s16x8 a = which == 0 ? a0 : which == 1 ? a1 : which == 2 ? a2 : a3;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, ones), acc, 0, 0, 0);
}

// The images may come in several equally sized segments (the same layer applied at several unroll steps:
// one pass then produces the sum of their weight gradients): image n is image n % seg_images of segment
// n / seg_images.
//
// One launch may serve several LAYERS (a group): each layer has its own segments, its own number of shares and
// its own run of workgroups (8 tiles x shares, layer after layer) and of partial-sum slots.  The backward pass of an
// update produces the operands of 23 such layers one after the other; computed layer by layer, every launch
// spreads one batch over the whole chip and writes 75 MB of partial sums for 4 images' worth of matrix work per
// workgroup - computed together after the pass, a workgroup gets 64 images and the partial sums shrink with it.
constexpr int kMaxSegments = 8;
constexpr int kMaxGroupLayers = 32, kMaxGroupUnits = 64;   // units: (layer, segment) pairs
struct WgradGroup {
  const __bf16* gout[kMaxGroupUnits];
  const __bf16* x[kMaxGroupUnits];
  int first_unit[kMaxGroupLayers];       // the layer's first (layer, segment) pair
  int n_images[kMaxGroupLayers];         // all its segments together
  int seg_images[kMaxGroupLayers];
  int want_db[kMaxGroupLayers];
  int first_block[kMaxGroupLayers + 1];  // 8 x (shares of the layers before it)
  int n_layers;
};

__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(WgradGroup grp, float* __restrict__ partial_base,
                                                           float* __restrict__ db_partial_base) {
  // which layer this workgroup works for (uniform: scalar registers)
  int layer = 0;
  while (layer + 1 < grp.n_layers && (int)blockIdx.x >= grp.first_block[layer + 1]) ++layer;
  const int local_block = (int)blockIdx.x - grp.first_block[layer];
  const int shares = (grp.first_block[layer + 1] - grp.first_block[layer]) >> 3;
  const int n_images = grp.n_images[layer], seg_images = grp.seg_images[layer], unit0 = grp.first_unit[layer];
  const int slot = (grp.first_block[layer] >> 3) + (local_block >> 3);   // this workgroup's partial-sum slot
  float* __restrict__ partial = partial_base;
  float* __restrict__ db_partial = grp.want_db[layer] ? db_partial_base : nullptr;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* gs = lds;
  char* xs = lds + kGBytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = local_block & 7, share = local_block >> 3;
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
    const int seg_ = (n_) / seg_images, in_seg_ = (n_) - seg_ * seg_images;    \
    const __bf16* gi = grp.gout[unit0 + seg_] + (long)in_seg_ * (kPix * kCh) + co0;  \
    const __bf16* xi = grp.x[unit0 + seg_] + (long)in_seg_ * (kPix * kCh) + ci0;     \
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
  // bias gradient = column sums of gout = gout^T x (a vector of ones): one more MFMA per k-step in four of the
  // wavefronts of one of the two input-channel tiles (wavefront w < 4 takes the 16 channels of operand tile w)
  const bool want_db = db_partial != nullptr && (tile & 1) == 0 && wave < 4;
  f32x4 dbv = f32x4{0.f, 0.f, 0.f, 0.f};

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
      if (want_db) dbv = column_sums(af[0], af[1], af[2], af[3], wave, dbv);
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

  if (want_db && (lane & 15) == 0) {  // every column of the product holds the sums; rows 4 kq .. 4 kq + 3
#pragma unroll
    for (int r = 0; r < 4; ++r) db_partial[slot * kCh + co0 + wave * 16 + 4 * kq + r] = dbv[r];
  }
  // partial[share][co][ci][tap] (the layout of dW itself).  A lane holds rows (co) 4 kq .. 4 kq + 3 of column
  // (ci) lane % 16: for one accumulator row r the wavefront owns, in each of 4 rows of dW, 16 ci x 9 taps = 576
  // consecutive bytes.  They pass through a wavefront-private piece of LDS (the staging area is free by now)
  // so that they leave as whole 16-byte units of consecutive addresses.
  float* dst = partial + (long)slot * 9 * kCh * kCh;
  float* tr = reinterpret_cast<float*>(lds) + wave * 576;
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

// dW = sum over shares of partial[share]: a workgroup owns 64 float4 columns of dW; its 16 wavefronts add up every
// 16th share each (up to eight loads in flight per thread: with 128 shares the whole read is one round trip),
// then wavefront 0 adds the 16 sums in a fixed order.  dw4: float4s of dW (the grid covers them in 64s), cout:
// output channels (at most 256).
// blockIdx.y = layer of a group: layer l owns the partial-sum slots first_slot[l] .. first_slot[l + 1] - 1.
constexpr int kDw4 = 9 * kCh * kCh / 4;
constexpr int kRedGroups = 16;
struct ReduceGroup {
  float4* dw[kMaxGroupLayers];
  float* db[kMaxGroupLayers];
  int first_slot[kMaxGroupLayers + 1];
};
__global__ __launch_bounds__(64 * kRedGroups) void wgrad_reduce_kernel(ReduceGroup rg,
                                                                      const float4* __restrict__ partial_base,
                                                                      const float* __restrict__ db_partial_base,
                                                                      int dw4, int cout) {
  __shared__ float4 part[kRedGroups - 1][64];
  const int layer = blockIdx.y;
  const int slot0 = rg.first_slot[layer], shares = rg.first_slot[layer + 1] - slot0;
  const float4* __restrict__ partial = partial_base + (long)slot0 * dw4;
  const float* __restrict__ db_partial = db_partial_base + (long)slot0 * cout;
  float4* __restrict__ dw = rg.dw[layer];
  float* __restrict__ db = rg.db[layer];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  if (db && blockIdx.x == 0) {  // the bias gradient: four threads per channel, every fourth share each
    __shared__ float dbs[4][256];
    const int c = threadIdx.x & 255, q = threadIdx.x >> 8;
    float t = 0.f;
    if (c < cout) {
      int g = q;
      for (; g + 28 < shares; g += 32) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = db_partial[(g + 4 * k) * cout + c];
#pragma unroll
        for (int k = 0; k < 8; ++k) t += v[k];
      }
      for (; g < shares; g += 4) t += db_partial[g * cout + c];
    }
    dbs[q][c] = t;
    __syncthreads();
    if (q == 0 && c < cout) db[c] = (dbs[0][c] + dbs[1][c]) + (dbs[2][c] + dbs[3][c]);
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < dw4) {
    const float4* p = partial + col;
    int g = grp;
    for (; g + 7 * kRedGroups < shares; g += 8 * kRedGroups) {
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p[(long)(g + kRedGroups * k) * dw4];
#pragma unroll
      for (int k = 0; k < 8; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
    }
    for (; g < shares; g += kRedGroups) {
      const float4 v = p[(long)g * dw4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  if (grp > 0) part[grp - 1][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp == 0 && col < dw4) {
#pragma unroll
    for (int k = 0; k < kRedGroups - 1; ++k) {
      const float4 v = part[k][threadIdx.x];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    dw[col] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The other layer shapes (model.py:165, :182): the same scheme with the image cut into bands of R output rows.
// A workgroup owns a (64 WCO) x (16 WCI) x 9 tile of dW - its WCO x WCI = 8 wavefronts one 64 (co) x 16 (ci) x 9
// piece each - and every `shares`-th unit (image, band).  Per unit it stages the band of gout (R WO pixels x CoT
// channels) and the rows of x the band's taps touch (S (R - 1) + 3 rows of S (WO - 1) + 3 pixels incl. the zero
// halo, x CiT channels); the next unit travels in registers meanwhile.  Row pitches: consecutive output pixels
// are S window pixels apart, so S times the x pitch (and the gout pitch) is an odd multiple of 32 bytes mod 256 -
// the 8 rows a 32-lane half reads through ds_read_b64_tr_b16 fall on 8 different bank groups.
template <int HO_, int WO_, int S_, int CIN_, int COUT_, int WCO_, int WCI_, int R_>
struct WgShape {
  static constexpr int HO = HO_, WO = WO_, S = S_, CIN = CIN_, COUT = COUT_, WCO = WCO_, WCI = WCI_, R = R_;
  static constexpr int H = HO * S, W = WO * S;  // input size (even sizes: (H - 1) / S + 1 == H / S)
  static constexpr int CoT = 64 * WCO, CiT = 16 * WCI;
  static constexpr int TILES_CO = COUT / CoT, TILES_CI = CIN / CiT, TILES = TILES_CO * TILES_CI;
  static constexpr int BANDS = HO / R, PIX = R * WO, KSTEPS = (PIX + 31) / 32, GROWS = KSTEPS * 32;
  static constexpr int XR = S * (R - 1) + 3, XW = S * (WO - 1) + 3;
  static constexpr int GROW = CoT * 2 + 32, XROW = CiT * 2 + (S == 2 ? 16 : 32);
  static constexpr int GBYTES = GROWS * GROW, XBYTES = XR * XW * XROW, LDS = GBYTES + XBYTES;
  static constexpr int GUPP = CoT / 8, XUPP = CiT / 8;             // 16-byte units per pixel of a slice
  static constexpr int GUNITS = PIX * GUPP, XUNITS = XR * W * XUPP;
  static constexpr int GROUNDS = (GUNITS + 511) / 512, XROUNDS = (XUNITS + 511) / 512;
  static constexpr int DW4 = COUT * CIN * 9 / 4;
  static_assert(WCO * WCI == 8 && COUT % CoT == 0 && CIN % CiT == 0 && HO % R == 0, "tiling");
  static_assert(LDS <= 160 * 1024 && LDS >= 8 * 576 * 4, "staging area");
};
//                       HO  WO  S  CIN COUT WCO WCI R
using WgStem  = WgShape<48, 48, 2,  96, 256, 4, 2, 2>;   // representation.conv1: 96x96x96 -> 48x48x256
using WgDown0 = WgShape<24, 24, 2, 256, 128, 2, 4, 4>;   // downsample.0.conv:    48x48x256 -> 24x24x128
using WgMid   = WgShape<24, 24, 1, 128, 128, 1, 8, 6>;   // downsample.0.blocks:  24x24x128 -> 24x24x128
using WgDown1 = WgShape<12, 12, 2, 128, 256, 2, 4, 4>;   // downsample.1.conv:    24x24x128 -> 12x12x256

template <class C>
__global__ __launch_bounds__(512, 2) void conv_wgrad_band_kernel(const __bf16* __restrict__ gout,
                                                                const __bf16* __restrict__ x, int n_images, int shares,
                                                                float* __restrict__ partial,
                                                                float* __restrict__ db_partial) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* gs = lds;
  char* xs = lds + C::GBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave / C::WCI, wci = wave % C::WCI;
  // consecutive workgroups land on consecutive XCDs: the tiles of one share sit 8 ids apart, i.e. on one XCD,
  // and read the same bands at about the same time (one fetch into that XCD's L2 serves all of them)
  const int tile = (blockIdx.x >> 3) % C::TILES;
  const int share = (blockIdx.x / (8 * C::TILES)) * 8 + (blockIdx.x & 7);
  if (share >= shares) return;
  const int co0 = (tile / C::TILES_CI) * C::CoT, ci0 = (tile % C::TILES_CI) * C::CiT;
  const int kq = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
  const int units = n_images * C::BANDS;

  for (int i = tid; i < C::LDS / 16; i += 512) *reinterpret_cast<uint4*>(lds + i * 16) = make_uint4(0, 0, 0, 0);

  // Staging loads are buffer loads: one descriptor per tensor, the unit's base (its first window row inside the
  // image) in a scalar register, the thread's own offsets fixed for the whole launch.  The window row above the
  // first band gets a negative offset - as an unsigned number past the end of the buffer, for which the hardware
  // returns zeros without touching memory; the row below the last band (stride 1 only) is pointed there by hand.
  constexpr unsigned kOutside = 0xffffff00u;
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(gout), 0, (unsigned)n_images * (unsigned)(C::HO * C::WO * C::COUT * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(x), 0, (unsigned)n_images * (unsigned)(C::H * C::W * C::CIN * 2), 0x00020000);
  unsigned goff[C::GROUNDS], xoff[C::XROUNDS];
#pragma unroll
  for (int j = 0; j < C::GROUNDS; ++j) {
    int v = tid + 512 * j;
    v = v < C::GUNITS ? v : C::GUNITS - 1;
    goff[j] = (unsigned)(((v / C::GUPP) * C::COUT + co0 + (v % C::GUPP) * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < C::XROUNDS; ++j) {
    int v = tid + 512 * j;
    v = v < C::XUNITS ? v : C::XUNITS - 1;
    const int pl = v / C::XUPP, r = pl / C::W, ix = pl - r * C::W;
    xoff[j] = (unsigned)(((r * C::W + ix) * C::CIN + ci0 + (v % C::XUPP) * 8) * 2);
  }
  u32x4_t gv[C::GROUNDS], xv[C::XROUNDS];
  auto load_unit = [&](int u) __attribute__((always_inline)) {
    const int n = u / C::BANDS, band = u - n * C::BANDS;
    const int gbase = __builtin_amdgcn_readfirstlane((int)((unsigned)((n * C::HO + band * C::R) * C::WO) * (unsigned)(C::COUT * 2)));
    const int top = band == 0 ? 0 : C::S * band * C::R - 1;   // first window row that exists
    const int xbase = __builtin_amdgcn_readfirstlane((int)((unsigned)(n * C::H + top) * (unsigned)(C::W * C::CIN * 2)));
    const unsigned shift = band == 0 ? (unsigned)(C::W * C::CIN * 2) : 0u;   // first band: window row r is image row r - 1
#pragma unroll
    for (int j = 0; j < C::GROUNDS; ++j) gv[j] = __builtin_amdgcn_raw_buffer_load_b128(grs, goff[j], gbase, 0);
#pragma unroll
    for (int j = 0; j < C::XROUNDS; ++j) {
      unsigned o = xoff[j] - shift;
      if constexpr (C::S == 1) {   // the window's last row of the last band lies below the image
        const int v = tid + 512 * j;
        const int r = (v < C::XUNITS ? v : C::XUNITS - 1) / (C::XUPP * C::W);
        o = (band == C::BANDS - 1 && r == C::XR - 1) ? kOutside : o;
      }
      xv[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, o, xbase, 0);
    }
  };
  auto store_unit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < C::GROUNDS; ++j) {
      const int v = tid + 512 * j;
      if (v < C::GUNITS) *reinterpret_cast<u32x4_t*>(gs + (v / C::GUPP) * C::GROW + (v % C::GUPP) * 16) = gv[j];
    }
#pragma unroll
    for (int j = 0; j < C::XROUNDS; ++j) {
      const int v = tid + 512 * j;
      if (v < C::XUNITS) {
        const int pl = v / C::XUPP, r = pl / C::W, ix = pl - r * C::W;
        *reinterpret_cast<u32x4_t*>(xs + (r * C::XW + ix + 1) * C::XROW + (v % C::XUPP) * 16) = xv[j];
      }
    }
  };

  f32x4 acc[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient as in the 12x12 kernel: among the WCI wavefronts that share gout's columns, wavefront wci takes
  // NDB of the four operand tiles (one input-channel tile per output-channel tile does it)
  constexpr int NDB = C::WCI >= 4 ? 1 : 4 / C::WCI;
  const bool want_db = db_partial != nullptr && tile % C::TILES_CI == 0 && wci * NDB < 4;
  f32x4 dbv[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i) dbv[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lds_base = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int g_lane = lds_base + (4 * kq + q) * C::GROW + wco * 128 + p4 * 8;  // + pixel0 * GROW + co tile * 32
  const int x_col = lds_base + C::GBYTES + wci * 32 + p4 * 8;                 // + window pixel * XROW

  int u = share;
  if (u < units) load_unit(u);
  __syncthreads();  // zeroing done
  for (; u < units; u += shares) {
    store_unit();
    __syncthreads();
    if (u + shares < units) load_unit(u + shares);  // travels while this unit is multiplied
#pragma unroll 1
    for (int ks = 0; ks < C::KSTEPS; ++ks) {
      const int pa = ks * 32 + 4 * kq + q, pb = pa + 16;
      const int pya = pa / C::WO, pxa = pa - pya * C::WO;
      const int pyb = pb / C::WO, pxb = pb - pyb * C::WO;
      // pixels past the band: gout is zero there, any window pixel will do
      const int xa = x_col + (pa < C::PIX ? (C::S * pya * C::XW + C::S * pxa) : 0) * C::XROW;
      const int xb = x_col + (pb < C::PIX ? (C::S * pyb * C::XW + C::S * pxb) : 0) * C::XROW;
      s16x8 af[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const s16x4 lo = read_tr(g_lane + ks * 32 * C::GROW + c * 32);
        const s16x4 hi = read_tr(g_lane + (ks * 32 + 16) * C::GROW + c * 32);
        af[c] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      if (want_db) {
#pragma unroll
        for (int i = 0; i < NDB; ++i) dbv[i] = column_sums(af[0], af[1], af[2], af[3], wci * NDB + i, dbv[i]);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int toff = ((t / 3) * C::XW + t % 3) * C::XROW;
        const s16x4 lo = read_tr(xa + toff);
        const s16x4 hi = read_tr(xb + toff);
        const s16x8 bfr = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[c]),
                                                              __builtin_bit_cast(bf16x8, bfr), acc[c][t], 0, 0, 0);
      }
    }
    __syncthreads();  // everyone has read this unit
  }

  if (want_db && (lane & 15) == 0) {
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        db_partial[share * C::COUT + co0 + wco * 64 + (wci * NDB + i) * 16 + 4 * kq + r] = dbv[i][r];
  }
  // partial[share][co][ci][tap], through a wavefront-private piece of LDS as in the 12x12 kernel
  float* dst = partial + (long)share * 9 * C::COUT * C::CIN;
  float* tr = reinterpret_cast<float*>(lds) + wave * 576;
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
          const long o = ((long)(co0 + wco * 64 + c * 16 + 4 * row + r) * C::CIN + ci0 + wci * 16) * 9 + col4 * 4;
          *reinterpret_cast<float4*>(dst + o) = v;
        }
      }
    }
}

template <class C>
bool wg_matches(int H, int W, int Cin, int Cout) { return H == C::H && W == C::W && Cin == C::CIN && Cout == C::COUT; }

// Workgroups of one launch: shares x TILES, one per compute unit at most (a second workgroup does not fit beside
// the first and would run after it).  Shares go to the XCDs in turn (see the kernel), so a whole number of
// rounds of 8 keeps every XCD within its own compute units.
template <class C>
int wg_shares(int B) {
  int by_cu = mz::compute_units() / C::TILES;
  if (by_cu >= 8) by_cu -= by_cu % 8;
  const int units = B * C::BANDS;
  const int s = units < by_cu ? units : by_cu;
  return s < 1 ? 1 : s;
}

template <class C>
int64_t wg_workspace(int B) { return (int64_t)wg_shares<C>(B) * (9 * C::COUT * C::CIN + C::COUT) * 4; }

template <class C>
int wg_launch(const void* gout, const void* x, int B, float* dw, float* dbias, void* workspace, int64_t workspace_bytes,
              hipStream_t s) {
  MZ_REQUIRE(workspace_bytes >= wg_workspace<C>(B), "mz_conv_wgrad_bf16: workspace of %lld bytes, %lld needed",
             (long long)workspace_bytes, (long long)wg_workspace<C>(B));
  MZ_REQUIRE((int64_t)B * C::H * C::W * C::CIN * 2 < (1LL << 32) && (int64_t)B * C::HO * C::WO * C::COUT * 2 < (1LL << 32),
             "mz_conv_wgrad_bf16: a tensor of %d images exceeds 4 GiB", B);
  static bool configured[mz::kMaxDevices] = {false};
  auto kern = conv_wgrad_band_kernel<C>;
  if (int rc = mz::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), C::LDS, configured, "conv_wgrad_band"))
    return rc;
  const int shares = wg_shares<C>(B);
  float* dbp = dbias ? static_cast<float*>(workspace) + (long)shares * 9 * C::COUT * C::CIN : nullptr;
  hipLaunchKernelGGL(kern, dim3((shares + 7) / 8 * 8 * C::TILES), dim3(512), C::LDS, s, static_cast<const __bf16*>(gout),
                     static_cast<const __bf16*>(x), B, shares, static_cast<float*>(workspace), dbp);
  ReduceGroup rg{};
  rg.dw[0] = reinterpret_cast<float4*>(dw);
  rg.db[0] = dbias;
  rg.first_slot[1] = shares;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((C::DW4 + 63) / 64), dim3(64 * kRedGroups), 0, s, rg,
                     static_cast<const float4*>(workspace), dbp, C::DW4, C::COUT);
  return mz::check_launch("mz_conv_wgrad_bf16");
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

// one launch of the 12x12 kernel + one of the reduction for a group of layers; shares[l] workgroups x 8 tiles each
struct HostWgradLayer {
  const void* const* gout;
  const void* const* x;
  int n_segments, B;
  float* dw;
  float* db;
};

static int wgrad_group_launch(const HostWgradLayer* layers, int n_layers, const int* shares, void* workspace,
                              int64_t workspace_bytes, hipStream_t s, const char* who) {
  WgradGroup grp{};
  ReduceGroup rg{};
  int units = 0, slots = 0;
  bool any_db = false;
  for (int l = 0; l < n_layers; ++l) {
    const HostWgradLayer& L = layers[l];
    MZ_REQUIRE(L.gout && L.x && L.dw && L.B > 0, "%s: layer %d has a null tensor or an empty batch", who, l);
    MZ_REQUIRE(L.n_segments >= 1 && L.n_segments <= kMaxSegments, "%s: n_segments = %d outside 1..%d", who, L.n_segments,
               kMaxSegments);
    MZ_REQUIRE(units + L.n_segments <= kMaxGroupUnits, "%s: more than %d (layer, segment) pairs", who, kMaxGroupUnits);
    MZ_REQUIRE((uintptr_t)L.dw % 16 == 0, "%s: dw must be 16-byte aligned", who);
    grp.first_unit[l] = units;
    for (int i = 0; i < L.n_segments; ++i) {
      MZ_REQUIRE(L.gout[i] && L.x[i], "%s: segment %d has a null tensor", who, i);
      grp.gout[units] = static_cast<const __bf16*>(L.gout[i]);
      grp.x[units] = static_cast<const __bf16*>(L.x[i]);
      ++units;
    }
    grp.n_images[l] = L.n_segments * L.B;
    grp.seg_images[l] = L.B;
    grp.want_db[l] = L.db != nullptr;
    any_db |= L.db != nullptr;
    grp.first_block[l] = 8 * slots;
    rg.dw[l] = reinterpret_cast<float4*>(L.dw);
    rg.db[l] = L.db;
    rg.first_slot[l] = slots;
    slots += shares[l];
  }
  grp.first_block[n_layers] = 8 * slots;
  grp.n_layers = n_layers;
  rg.first_slot[n_layers] = slots;
  const int64_t need = (int64_t)slots * (9 * kCh * kCh + kCh) * 4;
  MZ_REQUIRE((uintptr_t)workspace % 16 == 0, "%s: workspace must be 16-byte aligned", who);
  MZ_REQUIRE(workspace_bytes >= need, "%s: workspace of %lld bytes, %lld needed", who, (long long)workspace_bytes,
             (long long)need);
  static bool configured[mz::kMaxDevices] = {false};
  if (int rc = mz::ensure_dynamic_lds(reinterpret_cast<const void*>(conv_wgrad_kernel), kWgradLds, configured,
                                      "conv_wgrad"))
    return rc;
  float* partial = static_cast<float*>(workspace);
  float* dbp = partial + (long)slots * 9 * kCh * kCh;   // (written only by layers that want a bias gradient)
  (void)any_db;
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3(8 * slots), dim3(512), kWgradLds, s, grp, partial, dbp);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((kDw4 + 63) / 64, n_layers), dim3(64 * kRedGroups), 0, s, rg,
                     reinterpret_cast<const float4*>(partial), dbp, kDw4, kCh);
  return mz::check_launch(who);
}

// shares per layer of a group: 64 images per workgroup - unless that leaves compute units without work (then as
// many workgroups as there are units) or needs more than kMaxGroupSlots partial-sum slots (then more images each)
constexpr int kMaxGroupSlots = 128;
static void group_shares(const int* images, int n_layers, int* shares) {
  long total = 0;
  for (int l = 0; l < n_layers; ++l) total += images[l];
  const int cus = mz::compute_units();
  long quota = 64;
  if (total * 8 / quota < cus) quota = (total * 8 + cus - 1) / cus;
  if (quota < 1) quota = 1;
  for (;;) {
    long slots = 0;
    for (int l = 0; l < n_layers; ++l) slots += (images[l] + quota - 1) / quota;
    if (slots <= kMaxGroupSlots) break;
    quota += quota / 4 + 1;
  }
  for (int l = 0; l < n_layers; ++l) {
    const long sh = (images[l] + quota - 1) / quota;
    shares[l] = (int)(sh < 1 ? 1 : sh);
  }
}

int64_t mz_conv_wgrad_workspace_bytes_for(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
  if (B <= 0) return 0;
  if (H == kSide && W == kSide && Cin == kCh && Cout == kCh) return mz_conv_wgrad_workspace_bytes(B);
  if (wg_matches<WgStem>(H, W, Cin, Cout)) return wg_workspace<WgStem>(B);
  if (wg_matches<WgDown0>(H, W, Cin, Cout)) return wg_workspace<WgDown0>(B);
  if (wg_matches<WgMid>(H, W, Cin, Cout)) return wg_workspace<WgMid>(B);
  if (wg_matches<WgDown1>(H, W, Cin, Cout)) return wg_workspace<WgDown1>(B);
  return 0;
}

int mz_conv_wgrad_bf16(const void* grad_out, const void* x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                       int32_t Cout, float* dw, float* dbias, void* workspace, int64_t workspace_bytes,
                       mz_stream_t stream) {
  MZ_REQUIRE(grad_out && x && dw && workspace && B > 0, "mz_conv_wgrad_bf16: null tensor or empty batch");
  if (!(H == kSide && W == kSide && Cin == kCh && Cout == kCh)) {
    MZ_REQUIRE(((uintptr_t)dw | (uintptr_t)workspace) % 16 == 0, "mz_conv_wgrad_bf16: dw and workspace must be 16-byte aligned");
    hipStream_t s = mz::as_stream(stream);
    if (wg_matches<WgStem>(H, W, Cin, Cout)) return wg_launch<WgStem>(grad_out, x, B, dw, dbias, workspace, workspace_bytes, s);
    if (wg_matches<WgDown0>(H, W, Cin, Cout)) return wg_launch<WgDown0>(grad_out, x, B, dw, dbias, workspace, workspace_bytes, s);
    if (wg_matches<WgMid>(H, W, Cin, Cout)) return wg_launch<WgMid>(grad_out, x, B, dw, dbias, workspace, workspace_bytes, s);
    if (wg_matches<WgDown1>(H, W, Cin, Cout)) return wg_launch<WgDown1>(grad_out, x, B, dw, dbias, workspace, workspace_bytes, s);
    return mz::fail(MZ_E_ARG, "mz_conv_wgrad_bf16: no kernel for a %dx%d input, %d->%d channels (the reference "
                    "architecture's five layer shapes have one)", H, W, Cin, Cout);
  }
  const HostWgradLayer layer{&grad_out, &x, 1, B, dw, dbias};
  const int shares = mz_conv_wgrad_shares(B);
  return wgrad_group_launch(&layer, 1, &shares, workspace, workspace_bytes, mz::as_stream(stream), "mz_conv_wgrad_bf16");
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
  const HostWgradLayer layer{grad_out, x, n_segments, B, dw, dbias};
  const int shares = mz_conv_wgrad_shares(n_segments * B);
  return wgrad_group_launch(&layer, 1, &shares, workspace, workspace_bytes, mz::as_stream(stream),
                            "mz_conv_wgrad_bf16_multi");
}

int64_t mz_conv_wgrad_group_workspace_bytes(const int32_t* images_per_layer, int32_t n_layers) {
  if (!images_per_layer || n_layers <= 0 || n_layers > kMaxGroupLayers) return 0;
  int shares[kMaxGroupLayers];
  group_shares(images_per_layer, n_layers, shares);
  int64_t slots = 0;
  for (int l = 0; l < n_layers; ++l) slots += shares[l];
  return slots * (9 * kCh * kCh + kCh) * 4;
}

int mz_conv_wgrad_bf16_group(const mz_wgrad_layer* layers, int32_t n_layers, void* workspace, int64_t workspace_bytes,
                             mz_stream_t stream) {
  MZ_REQUIRE(layers && workspace, "mz_conv_wgrad_bf16_group: null argument");
  MZ_REQUIRE(n_layers >= 1 && n_layers <= kMaxGroupLayers, "mz_conv_wgrad_bf16_group: n_layers = %d outside 1..%d", n_layers,
             kMaxGroupLayers);
  HostWgradLayer host[kMaxGroupLayers];
  int images[kMaxGroupLayers], shares[kMaxGroupLayers];
  for (int l = 0; l < n_layers; ++l) {
    host[l] = HostWgradLayer{layers[l].grad_out, layers[l].x, layers[l].n_segments, layers[l].B, layers[l].dw,
                             layers[l].dbias};
    MZ_REQUIRE(layers[l].n_segments >= 1 && layers[l].B > 0, "mz_conv_wgrad_bf16_group: layer %d is empty", l);
    images[l] = layers[l].n_segments * layers[l].B;
  }
  group_shares(images, n_layers, shares);
  return wgrad_group_launch(host, n_layers, shares, workspace, workspace_bytes, mz::as_stream(stream),
                            "mz_conv_wgrad_bf16_group");
}

}  // extern "C"
