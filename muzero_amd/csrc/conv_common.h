// Shared by the convolution translation units: vector types, the tiling description of the
// patch kernels and the reference architecture's layer shapes, the geometry of the 12x12 trunk.
#pragma once
#include <hip/hip_runtime.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- the 16-bit operand format of a matrix-core kernel: __bf16 (8 significant bits, float32's exponent range) or
// _Float16 (IEEE half: 11 significant bits, largest finite value 65 504).  v_mfma_f32_16x16x32 runs both at the
// same rate; the kernels are templates over the format and differ only in the helpers below.
template <typename T> using vec8 = T __attribute__((ext_vector_type(8)));
template <typename T> using vec4 = T __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_16x16x32(vec8<__bf16> a, vec8<__bf16> b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_16x16x32(vec8<_Float16> a, vec8<_Float16> b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ vec8<T> zero8() {
  return __builtin_bit_cast(vec8<T>, u32x4_t{0u, 0u, 0u, 0u});
}

// float32 -> operand format, round to nearest even.  IEEE half saturates at its largest finite value instead of
// overflowing to infinity (one v_med3_f32 in front of the conversion).
template <typename T> __device__ __forceinline__ T to_op(float v);
template <> __device__ __forceinline__ __bf16 to_op<__bf16>(float v) { return (__bf16)v; }
template <> __device__ __forceinline__ _Float16 to_op<_Float16>(float v) {
  return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
}
// max(v, floor) for an epilogue whose result is converted next: for IEEE half the saturation rides in the same
// instruction (v_med3_f32 between the floor and the largest finite value), and to_op_unclamped follows
template <typename T> __device__ __forceinline__ float floor_for_op(float v, float floor_v);
template <> __device__ __forceinline__ float floor_for_op<__bf16>(float v, float floor_v) { return fmaxf(v, floor_v); }
template <> __device__ __forceinline__ float floor_for_op<_Float16>(float v, float floor_v) {
  return __builtin_amdgcn_fmed3f(v, floor_v, 65504.f);      // floor_v is 0 or -65504 (op_floor)
}
template <typename T> __device__ __forceinline__ float op_floor(bool relu);
template <> __device__ __forceinline__ float op_floor<__bf16>(bool relu) { return relu ? 0.f : -__builtin_inff(); }
template <> __device__ __forceinline__ float op_floor<_Float16>(bool relu) { return relu ? 0.f : -65504.f; }
template <typename T> __device__ __forceinline__ vec4<T> to_op4_unclamped(float a, float b, float c, float d) {
  return vec4<T>{(T)a, (T)b, (T)c, (T)d};
}
template <typename T> __device__ __forceinline__ vec4<T> to_op4(float a, float b, float c, float d) {
  return vec4<T>{to_op<T>(a), to_op<T>(b), to_op<T>(c), to_op<T>(d)};
}

// The two 16-bit values of a dword as float32 - or two zeros when the kernel does not want them (a layer whose
// successor adds no residual).  bf16: one instruction each (AND for the upper half, byte permute for the lower;
// selector 0x0c = zero); half: an AND on the dword, then the two conversions.
template <typename T> struct PairToFloat;
template <> struct PairToFloat<__bf16> {
  unsigned hi, lo;
  __device__ __forceinline__ explicit PairToFloat(bool keep)
      : hi(keep ? 0xffff0000u : 0u), lo(keep ? 0x01000c0cu : 0x0c0c0c0cu) {}
  __device__ __forceinline__ float even(unsigned r) const { return __uint_as_float(__builtin_amdgcn_perm(r, r, lo)); }
  __device__ __forceinline__ float odd(unsigned r) const { return __uint_as_float(r & hi); }
};
template <> struct PairToFloat<_Float16> {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  unsigned mask;
  __device__ __forceinline__ explicit PairToFloat(bool keep) : mask(keep ? 0xffffffffu : 0u) {}
  __device__ __forceinline__ float even(unsigned r) const { return (float)__builtin_bit_cast(h2, r & mask)[0]; }
  __device__ __forceinline__ float odd(unsigned r) const { return (float)__builtin_bit_cast(h2, r & mask)[1]; }
};

// Sum over the 16 lanes of a row (lanes with the same lane >> 4), returned in each of them: quad permutes and row
// mirrors pair the same groups of lanes as a butterfly of exchanges at distances 1, 2, 4, 8 would - the same tree
// of additions, the same bits - without four trips through the LDS crossbar.
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}

#ifndef MZ_ROWPAD
#define MZ_ROWPAD 1
#endif
template <int CIN_, int CINC_, int COUT_, int S_, int TH_, int TW_, int HIN_, int WIN_, int MINW_, int PF_ = 2, int NW_ = 4>
struct Cfg {
  static constexpr int NW = NW_;          // wavefronts per workgroup; they split the output channels
  static constexpr int THREADS = 64 * NW_;
  static constexpr int PF = PF_;  // weight prefetch distance in k-steps; 9 * KC must be a multiple of PF + 1
  static constexpr int CIN = CIN_, CINC = CINC_, COUT = COUT_, S = S_, TH = TH_, TW = TW_, HIN = HIN_, WIN = WIN_;
  static constexpr int MINW = MINW_;
  static constexpr int HO = (HIN - 1) / S + 1, WO = (WIN - 1) / S + 1;
  static constexpr int PY = HO / TH, PX = WO / TW;
  static constexpr int M = TH * TW, MT = (M + 15) / 16;
  static constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;  // input window incl. the halo
  static constexpr int ROWB = CINC * 2 + 16;
  static constexpr int KC = CINC / 32, NCC = CIN / CINC;
  static constexpr int NTILES = COUT / 16, WN = NTILES / NW;
  // bytes between window rows.  A stride-2 operand read puts the 16 lanes of a pixel tile on every second pixel
  // of two window rows two apart: with an odd pixel pitch (in 16-byte units) the eight pixels of a row fall on
  // eight different even units of the 16 the banks hold, and the two rows coincide - no conflict - only if twice
  // the row pitch is a multiple of 16 units, so the row pitch is padded to a multiple of 128 bytes
  static constexpr int ROWP = (S == 2 && MZ_ROWPAD) ? (IW * ROWB + 127) / 128 * 128 : IW * ROWB;
  static constexpr int LDS_BYTES = IH * ROWP;
  static constexpr int KSTEPS = NCC * 9 * KC;
  static_assert(HO % TH == 0 && WO % TW == 0, "patches must tile the output");
  static_assert(CIN % CINC == 0 && CINC % 32 == 0 && COUT % (16 * NW) == 0, "channel tiling");
};

// The reference architecture's convolution shapes (model.py:178-215 with the default config)
// other than the 12x12 trunk.  Eight wavefronts per workgroup measured 10-20 % faster than four
// on the three strided layers (more loads in flight per CU); the 24x24 layer did not care.
//                     CIN CINC COUT S TH TW  HIN WIN waves/SIMD PF NW
#ifndef MZ_STEM_TH
#define MZ_STEM_TH 8
#define MZ_STEM_TW 8
#define MZ_STEM_NW 8
#define MZ_STEM_MINW 2
#endif
using ConvStem = Cfg<96, 96, 256, 2, MZ_STEM_TH, MZ_STEM_TW, 96, 96, MZ_STEM_MINW, 2, MZ_STEM_NW>;  // representation.conv1
using ConvStemRot = Cfg<96, 96, 256, 2, 8, 8, 96, 96, 2, 2, 8>;   // the same layer on the rotating, tiled observation
using ConvDown0 = Cfg<256, 64, 128, 2, 8, 8, 48, 48, 2, 2, 8>; // downsample.0.conv (64-channel chunks: -7 % against 128)
using ConvMid = Cfg<128, 128, 128, 1, 12, 12, 24, 24, 3, 2, 4>;   // downsample.0.blocks
using ConvDown1 = Cfg<128, 128, 256, 2, 4, 12, 24, 24, 2, 2, 8>;  // downsample.1.conv


// ---- the 12x12x256 trunk (conv_trunk.hip, conv_x3.hip): planar LDS image [16-byte channel chunk][pixel]
constexpr int kPix = 144, kSide = 12, kCh = 256;
constexpr int kPixPad = 160;                 // pixels per plane incl. the zero pixels
constexpr int kPlaneB = kPixPad * 16;        // bytes per channel-chunk plane
constexpr int kPlanes = kCh / 8;             // 32 planes of 8 channels
constexpr int kLdsBytes = kPlanes * kPlaneB; // 81 920
constexpr int kMT = kPix / 16;               // 9 pixel tiles
constexpr int kKC = kCh / 32;                // 8 k-steps per tap
constexpr int kMaxChain = 20;

// lanes (bit l: lane l, pixel mt*16 + l%16 of a row-major 12-wide image) whose pixel sits in column x_edge
__host__ __device__ constexpr unsigned long long edge_lanes(int mt, int x_edge) {
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if ((mt * 16 + (l & 15)) % kSide == x_edge) m |= 1ull << l;
  return m;
}

}  // namespace
