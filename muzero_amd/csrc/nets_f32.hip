// fp32 network kernels: the validation-precision path (exact float32
// arithmetic, any layer shape) plus the pieces both precisions share
// (spatial mean, prediction / reward heads with fused softmax and support
// decode, layout conversion).  Activations are NHWC.
#include "common.h"

namespace {

// ------------------------------------------------------------ conv3x3, fp32
// One block = TP consecutive output pixels of one output row x all Cout
// channels; thread c owns output channel c.  The 3 x (TP*stride+2) input
// window is staged in LDS (zero outside the image), so every weight that a
// thread loads is reused TP times and every activation read is an LDS
// broadcast.  Weights are [tap][cin][cout] so the per-thread loads coalesce.
constexpr int TP = 4;

__global__ void conv3x3_f32_kernel(const float* __restrict__ in, const int* __restrict__ in_slot, int slot_stride,
                                   const float* __restrict__ w, const float* __restrict__ bias,
                                   const float* __restrict__ residual, float* __restrict__ out,
                                   const float* __restrict__ avec, const int* __restrict__ actions, int H, int W,
                                   int Cin, int Cout, int stride, int HO, int WO, int relu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);  // [3][TW][Cin]
  const int TW = TP * stride + 2;
  const int b = blockIdx.z, oy = blockIdx.y, ox0 = blockIdx.x * TP;
  const int c = threadIdx.x;
  const long img = in_slot ? (long)in_slot[b] * slot_stride + b : b;
  const float* src = in + img * (long)H * W * Cin;
  const float* av = avec ? avec + (long)actions[b] * Cin : nullptr;
  const int iy0 = oy * stride - 1, ix0 = ox0 * stride - 1;
  for (int i = threadIdx.x; i < 3 * TW * Cin; i += blockDim.x) {
    const int ci = i % Cin, px = (i / Cin) % TW, py = i / (Cin * TW);
    const int iy = iy0 + py, ix = ix0 + px;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
      v = src[((long)iy * W + ix) * Cin + ci];
      if (av) v += av[ci];  // hidden + conv1x1(one-hot action) inside the image only (model.py:236)
    }
    tile[i] = v;
  }
  __syncthreads();
  if (c >= Cout) return;
  float acc[TP];
#pragma unroll
  for (int p = 0; p < TP; ++p) acc[p] = 0.f;
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap % 3;
    const float* wt = w + (long)tap * Cin * Cout + c;
    const float* trow = tile + (dy * TW + dx) * Cin;
    for (int ci = 0; ci < Cin; ++ci) {
      const float wv = wt[(long)ci * Cout];
#pragma unroll
      for (int p = 0; p < TP; ++p) acc[p] = fmaf(trow[p * stride * Cin + ci], wv, acc[p]);
    }
  }
  const float bv = bias[c];
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const long o = (((long)b * HO + oy) * WO + ox0 + p) * Cout + c;
    float v = acc[p] + bv;
    if (residual) v += residual[o];
    if (relu) v = fmaxf(v, 0.f);
    out[o] = v;
  }
}

// -------------------------------------------------------------- spatial mean
__global__ void spatial_mean_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int HW, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float* p = in + (long)b * HW * C + c;
    float s = 0.f;
    for (int i = 0; i < HW; ++i) s += p[(long)i * C];
    out[(long)b * C + c] = s / (float)HW;
  }
}

// --------------------------------------------------------------------- heads
// Policy / value / reward MLPs (one hidden layer each) on the spatial mean, with
// the softmax over actions and the support decode fused (model.py:243, :275-280,
// :397, :89-109).  Exact float32 on the matrix cores: v_mfma_f32_16x16x4_f32 is
// bit-for-bit a k-ordered fmaf chain.  One workgroup = 16 samples x one head; the
// weights are the row operand, pre-packed on the host in fragment order
// ([out tile][k/4][lane]: lane l holds W[tile*16 + l%16][4*kstep + l/16]) so that a
// fragment is one coalesced 256-byte load; the 16 samples are the column operand,
// read from LDS.  A lane ends up with 4 consecutive outputs of one sample.
constexpr int kHeadsBlock = 1024;
constexpr int kHeadSamples = 16;
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct HeadW {
  const float* w1p;  // [Hd/16][C/4][64]
  const float* b1;   // [Hd]
  const float* w2p;  // [ceil(O/16)][Hd/4][64]
  const float* b2;   // [ceil(O/16)*16]  (zero padded)
  int out;           // O: number of outputs of this head
  int bf16;          // w1p / w2p hold bf16 fragments for the 16x16x32 MFMA instead
};

// Sum and maximum over the 64 lanes of a wavefront, returned in every lane.  The additions form the same
// balanced tree as a butterfly of lane exchanges at distances 1, 2, 4, ... 32 (quad permutes, half-row and row
// mirrors pair the same groups; float addition commutes), so the sum has the butterfly's bits - at a quarter of
// its latency: four data-parallel-primitive steps and four scalar lane reads instead of six trips through the
// LDS crossbar.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float readlane_f32(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
  v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);  // row_half_mirror
  v += dpp_f32<0x140>(v);  // row_mirror
  return (readlane_f32(v, 0) + readlane_f32(v, 16)) + (readlane_f32(v, 32) + readlane_f32(v, 48));
}
__device__ __forceinline__ float wave_max_f32(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}

// softmax over the (up to 64) logits held one per lane of a wavefront
__device__ __forceinline__ float wave_softmax(float lg, bool active) {
  const float m = wave_max_f32(active ? lg : -__builtin_inff());
  const float e = active ? expf(lg - m) : 0.f;
  return e / wave_sum_f32(e);
}

// float32 operation sequence of inverse_transform_value (model.py:93-95);
// contraction is off so that no multiply-add is fused.
__device__ float inverse_value_transform(float x) {
#pragma clang fp contract(off)
  const float eps = 0.001f;
  float t = fabsf(x) + 1.0f;
  t = t + eps;
  t = 0.004f * t;
  t = 1.0f + t;
  t = sqrtf(t) - 1.0f;
  t = t / 0.002f;
  t = t * t - 1.0f;
  const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
  return sg * t;
}

// softmax expectation over bins -S..S of lg[0..O) by ONE wavefront, then the inverse transform
__device__ float wave_decode_support(const float* lg, int O) {
  const int lane = threadIdx.x & 63, S = O / 2;
  float m = -__builtin_inff();
  for (int o = lane; o < O; o += 64) m = fmaxf(m, lg[o]);
  m = wave_max_f32(m);
  float den = 0.f, num = 0.f;
  for (int o = lane; o < O; o += 64) {
    const float e = expf(lg[o] - m);
    den += e;
    num += e * (float)(o - S);
  }
  den = wave_sum_f32(den);
  num = wave_sum_f32(num);
  return inverse_value_transform(num / den);
}

// one dense layer for 16 samples: dst[s][o] = act(bias[o] + sum_k src[s][k] * W[o][k]);
// out tiles are dealt round-robin to the wavefronts, up to 4 tiles in flight per wavefront;
// U k-steps of weight fragments are loaded ahead of their MFMAs to cover the L2 latency
template <int U>
__device__ __forceinline__ void dense16(const float* __restrict__ wp, const float* __restrict__ bias,
                                        const float* src, int lds, float* dst, int ldd, int K, int tiles,
                                        bool relu) {
  constexpr int NW = kHeadsBlock / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ks_n = K / 4;
  const float* sp = src + (lane & 15) * lds + (lane >> 4);
  for (int t0 = wave; t0 < tiles; t0 += 4 * NW) {
    f32x4 acc[4];
    const float* w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int t = t0 + NW * i < tiles ? t0 + NW * i : t0;  // idle slots recompute tile t0
      w[i] = wp + (long)t * ks_n * 64 + lane;
    }
    const bool multi = t0 + NW < tiles;  // wave-uniform: more than one live tile
    for (int ks = 0; ks < ks_n; ks += U) {
      float wf[4][U], bf[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        bf[u] = sp[(ks + u) * 4];
        wf[0][u] = w[0][(long)(ks + u) * 64];
      }
      if (multi) {
#pragma unroll
        for (int i = 1; i < 4; ++i)
#pragma unroll
          for (int u = 0; u < U; ++u) wf[i][u] = w[i][(long)(ks + u) * 64];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0][u], bf[u], acc[0], 0, 0, 0);
        if (multi) {
#pragma unroll
          for (int i = 1; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][u], bf[u], acc[i], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = t0 + NW * i;
      if (t < tiles && (i == 0 || multi)) {
        const int o = t * 16 + (lane >> 4) * 4;
        const float4 bv = *reinterpret_cast<const float4*>(bias + o);
        float v0 = acc[i][0] + bv.x, v1 = acc[i][1] + bv.y, v2 = acc[i][2] + bv.z, v3 = acc[i][3] + bv.w;
        if (relu) {
          v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
        }
        float* d = dst + (lane & 15) * ldd + o;
        d[0] = v0; d[1] = v1; d[2] = v2; d[3] = v3;
      }
    }
  }
}

// the same layer on the matrix cores with 16-bit operands (T = __bf16: the throughput precision; T = _Float16:
// IEEE half, three more significant bits at the same rate): weights are fragments
// [out tile][K/32][lane][8] (lane l: W[tile*16 + l%16][32*kstep + 8*(l/16) + j]); the 16
// samples are converted from the float32 LDS copy on the fly; float32 accumulation
template <typename T> using hvec8 = T __attribute__((ext_vector_type(8)));
template <typename T> using hvec4 = T __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 head_mfma(hvec8<__bf16> a, hvec8<__bf16> b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 head_mfma(hvec8<_Float16> a, hvec8<_Float16> b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// float32 -> operand format (nearest even; IEEE half saturates at its largest finite value)
template <typename T> __device__ __forceinline__ T head_op(float v);
template <> __device__ __forceinline__ __bf16 head_op<__bf16>(float v) { return (__bf16)v; }
template <> __device__ __forceinline__ _Float16 head_op<_Float16>(float v) {
  return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
}

template <typename T>
__device__ __forceinline__ void dense16_bf16(const hvec8<T>* __restrict__ wp, const float* __restrict__ bias,
                                             const float* src, int lds, float* dst, int ldd, int K, int tiles,
                                             bool relu) {
  typedef hvec8<T> hbf16x8;
  constexpr int NW = kHeadsBlock / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ks_n = K / 32;
  const float* sp = src + (lane & 15) * lds + (lane >> 4) * 8;
  for (int t = wave; t < tiles; t += NW) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const hbf16x8* w = wp + (long)t * ks_n * 64 + lane;
    for (int ks = 0; ks < ks_n; ++ks) {
      hbf16x8 bfrag;
#pragma unroll
      for (int j = 0; j < 8; ++j) bfrag[j] = head_op<T>(sp[ks * 32 + j]);
      acc = head_mfma(w[(long)ks * 64], bfrag, acc);
    }
    const int o = t * 16 + (lane >> 4) * 4;
    const float4 bv = *reinterpret_cast<const float4*>(bias + o);
    float v0 = acc[0] + bv.x, v1 = acc[1] + bv.y, v2 = acc[2] + bv.z, v3 = acc[3] + bv.w;
    if (relu) {
      v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
    }
    float* d = dst + (lane & 15) * ldd + o;
    d[0] = v0; d[1] = v1; d[2] = v2; d[3] = v3;
  }
}

// The weight fragments of a whole output tile (K = 32*KS), requested in one go: the kernel is bound by the
// latency of its dependent phases, so a tile's fragments are loaded before the data they multiply exists.
template <typename T, int KS>
struct TileFrags {
  hvec8<T> f[KS];
};
template <typename T, int KS>
__device__ __forceinline__ void load_tile_frags(TileFrags<T, KS>& fr, const hvec8<T>* __restrict__ wp, int t) {
  const hvec8<T>* w = wp + (long)t * KS * 64 + (threadIdx.x & 63);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) fr.f[ks] = w[ks * 64];
}

// One 16-output tile with its fragments and bias already in registers, on a bf16 image of the layer's input in
// LDS ([16 samples][kActPitch bytes]: a lane's operand of a k-step is one 16-byte read; rounding the inputs to
// bf16 once when they are written gives the same values as rounding them at every read, and re-reading the
// float32 image - 64 four-byte reads per lane and tile - was what the layers spent their time on).
constexpr int kActPitch = 256 * 2 + 16;  // bytes per sample: 33 units of 16 bytes, so 16 rows fall on 16 bank groups
template <typename T, int KS, bool TO_BF16>
__device__ __forceinline__ void dense16_tile(const TileFrags<T, KS>& fr, const float4 bv, const char* src, char* dst_act,
                                             float* dst, int ldd, int t, bool relu) {
  typedef hvec8<T> hbf16x8;
  const int lane = threadIdx.x & 63;
  const char* sp = src + (lane & 15) * kActPitch + (lane >> 4) * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    acc = head_mfma(fr.f[ks], *reinterpret_cast<const hbf16x8*>(sp + ks * 64), acc);
  float v0 = acc[0] + bv.x, v1 = acc[1] + bv.y, v2 = acc[2] + bv.z, v3 = acc[3] + bv.w;
  if (relu) {
    v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
  }
  const int o = t * 16 + (lane >> 4) * 4;
  if constexpr (TO_BF16) {
    *reinterpret_cast<hvec4<T>*>(dst_act + (lane & 15) * kActPitch + o * 2) =
        hvec4<T>{head_op<T>(v0), head_op<T>(v1), head_op<T>(v2), head_op<T>(v3)};
  } else {
    float* d = dst + (lane & 15) * ldd + o;
    d[0] = v0; d[1] = v1; d[2] = v2; d[3] = v3;
  }
}

__device__ __forceinline__ void dense16_any(const float* wp, const float* bias, const float* src, int lds,
                                            float* dst, int ldd, int K, int tiles, bool relu) {
  if (K % 32 == 0) dense16<8>(wp, bias, src, lds, dst, ldd, K, tiles, relu);
  else dense16<1>(wp, bias, src, lds, dst, ldd, K, tiles, relu);
}

// FAST: every head has 16-bit weights of format T, C == Hd == 256 and at most 48 output tiles (checked by the
// host), so the kernel holds only that path's registers.
template <bool FAST, typename T = __bf16>
__global__ __launch_bounds__(kHeadsBlock) void heads_kernel(
    const float* __restrict__ mean, int B, int C, int Hd, int A, HeadW pol, HeadW val, HeadW rew,
    float* __restrict__ priors, float* __restrict__ value, float* __restrict__ reward,
    float* __restrict__ policy_logits, float* __restrict__ value_logits, float* __restrict__ reward_logits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int head = blockIdx.y;  // 0 policy, 1 value, 2 reward
  const HeadW hw = head == 0 ? pol : (head == 1 ? val : rew);
  const int O = hw.out, otiles = (O + 15) / 16;
  const int ldm = C + 1, ldh = Hd + 1, ldo = otiles * 16 + 1;
  float* hid_s = reinterpret_cast<float*>(smem);  // [16][Hd+1]
  float* mean_s = hid_s + kHeadSamples * ldh;     // [16][C+1]
  float* logit_s = mean_s;                        // [16][otiles*16+1], reuses the mean once it is consumed
  const int b0 = blockIdx.x * kHeadSamples;
  const int ns = B - b0 < kHeadSamples ? B - b0 : kHeadSamples;
  constexpr bool fast = FAST;
  if constexpr (fast) {
    // the reference architecture's sizes (256 -> 256 -> O <= 768) in bf16.  The kernel is one workgroup's chain
    // of dependent phases, and every weight fragment comes from beyond L2 (the convolutions between two calls
    // flush it): a wavefront requests what it will multiply as early as registers allow - its layer-1 tile and
    // bias before the spatial means are staged, its first layer-2 tile and all biases before the barrier that
    // publishes the means, its second layer-2 tile once the layer-1 registers are spent and its third once the
    // first is spent.  Means and hidden layer live in LDS as bf16 images; the logits follow them as float32.
    constexpr int NW = kHeadsBlock / 64;
    static_assert(kHeadSamples * 256 / 4 == kHeadsBlock, "one float4 of the means per thread");
    char* x_s = smem;
    char* h_s = smem + kHeadSamples * kActPitch;
    logit_s = reinterpret_cast<float*>(smem + 2 * kHeadSamples * kActPitch);
    const int wv = threadIdx.x >> 6, lq = (threadIdx.x & 63) >> 4;
    typedef hvec8<T> hbf16x8;   // eight weights / activations of format T
    TileFrags<T, 8> fa, fb;
    const hbf16x8* w2 = reinterpret_cast<const hbf16x8*>(hw.w2p);
    // Every load below is unconditional (sample and tile indices are clamped instead): a load inside a branch
    // makes the compiler wait for ALL outstanding loads where the branch rejoins, and the barrier would then
    // wait for the weight fragments too.  The means go first: loads return in order.
    const int last = otiles - 1;
    const int ta = wv < last ? wv : last, tb = wv + NW < last ? wv + NW : last, tc = wv + 2 * NW < last ? wv + 2 * NW : last;
    {
      const int s = threadIdx.x >> 6, k = (threadIdx.x & 63) * 4;
      float4 m = *reinterpret_cast<const float4*>(mean + (long)(b0 + (s < ns ? s : ns - 1)) * 256 + k);
      load_tile_frags<T, 8>(fa, reinterpret_cast<const hbf16x8*>(hw.w1p), wv);
      if (s >= ns) m = float4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<hvec4<T>*>(x_s + s * kActPitch + k * 2) =
          hvec4<T>{head_op<T>(m.x), head_op<T>(m.y), head_op<T>(m.z), head_op<T>(m.w)};
    }
    __builtin_amdgcn_sched_barrier(0);
    const float4 bv1 = *reinterpret_cast<const float4*>(hw.b1 + wv * 16 + lq * 4);
    load_tile_frags<T, 8>(fb, w2, ta);
    const float4 bv2a = *reinterpret_cast<const float4*>(hw.b2 + ta * 16 + lq * 4);
    const float4 bv2b = *reinterpret_cast<const float4*>(hw.b2 + tb * 16 + lq * 4);
    const float4 bv2c = *reinterpret_cast<const float4*>(hw.b2 + tc * 16 + lq * 4);
    __syncthreads();
    dense16_tile<T, 8, true>(fa, bv1, x_s, h_s, nullptr, 0, wv, true);
    load_tile_frags<T, 8>(fa, w2, tb);
    __syncthreads();
    if (wv < otiles) dense16_tile<T, 8, false>(fb, bv2a, h_s, nullptr, logit_s, ldo, wv, false);
    load_tile_frags<T, 8>(fb, w2, tc);
    if (wv + NW < otiles) dense16_tile<T, 8, false>(fa, bv2b, h_s, nullptr, logit_s, ldo, wv + NW, false);
    if (wv + 2 * NW < otiles) dense16_tile<T, 8, false>(fb, bv2c, h_s, nullptr, logit_s, ldo, wv + 2 * NW, false);
    __syncthreads();
  } else {
  for (int i = threadIdx.x; i < kHeadSamples * C; i += kHeadsBlock) {
    const int s = i / C, k = i % C;
    mean_s[s * ldm + k] = s < ns ? mean[(long)(b0 + s) * C + k] : 0.f;
  }
  __syncthreads();
  if (hw.bf16 == 1) dense16_bf16<__bf16>(reinterpret_cast<const hvec8<__bf16>*>(hw.w1p), hw.b1, mean_s, ldm, hid_s, ldh, C, Hd / 16, true);
  else if (hw.bf16 == 2) dense16_bf16<_Float16>(reinterpret_cast<const hvec8<_Float16>*>(hw.w1p), hw.b1, mean_s, ldm, hid_s, ldh, C, Hd / 16, true);
  else dense16_any(hw.w1p, hw.b1, mean_s, ldm, hid_s, ldh, C, Hd / 16, true);
  __syncthreads();
  if (hw.bf16 == 1) dense16_bf16<__bf16>(reinterpret_cast<const hvec8<__bf16>*>(hw.w2p), hw.b2, hid_s, ldh, logit_s, ldo, Hd, otiles, false);
  else if (hw.bf16 == 2) dense16_bf16<_Float16>(reinterpret_cast<const hvec8<_Float16>*>(hw.w2p), hw.b2, hid_s, ldh, logit_s, ldo, Hd, otiles, false);
  else dense16_any(hw.w2p, hw.b2, hid_s, ldh, logit_s, ldo, Hd, otiles, false);
  __syncthreads();
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* lg_out = head == 0 ? policy_logits : (head == 1 ? value_logits : reward_logits);
  for (int s = wave; s < ns; s += kHeadsBlock / 64) {  // one wavefront per sample
    const float* lg = logit_s + s * ldo;
    if (head == 0) {
      const float pr = wave_softmax(lane < A ? lg[lane] : 0.f, lane < A);
      if (lane < A) priors[(long)(b0 + s) * A + lane] = pr;
    } else {
      const float y = wave_decode_support(lg, O);
      if (lane == 0) (head == 1 ? value : reward)[b0 + s] = y;
    }
    if (lg_out)
      for (int o = lane; o < O; o += 64) lg_out[(long)(b0 + s) * O + o] = lg[o];
  }
}

// decode only: logits [n][O] -> scalar [n]  (support_to_scalar, model.py:98-109), one wavefront per row
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ logits, int n, int O,
                                                     float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float y = wave_decode_support(logits + (long)row * O, O);
  if ((threadIdx.x & 63) == 0) out[row] = y;
}

// softmax(policy_logits, dim=1) (model.py:397): one wavefront per row
__global__ void policy_softmax_kernel(const float* __restrict__ logits, float* __restrict__ priors, int n, int A) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), a = threadIdx.x & 63;
  if (row >= n) return;
  const float lg = a < A ? logits[(long)row * A + a] : 0.f;
  const float pr = wave_softmax(lg, a < A);
  if (a < A) priors[(long)row * A + a] = pr;
}

// ------------------------------------------------------- layout conversions
// NCHW f32 -> NHWC f32 (tile transpose through LDS so both sides coalesce)
__global__ void nchw_to_nhwc_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW) {
  __shared__ float t[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    t[r][tx] = (c < C && p < HW) ? in[((long)b * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    if (c < C && p < HW) out[((long)b * HW + p) * C + c] = t[tx][r];
  }
}

__global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW) {
  __shared__ float t[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    t[r][tx] = (c < C && p < HW) ? in[((long)b * HW + p) * C + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    if (c < C && p < HW) out[((long)b * C + c) * HW + p] = t[tx][r];
  }
}

}  // namespace

extern "C" {

int mz_conv3x3_f32(const float* in, const int32_t* in_slot, int32_t slot_stride, const float* w,
                   const float* bias, const float* residual, float* out, const float* action_vec,
                   const int32_t* actions, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                   int32_t stride, int32_t relu, mz_stream_t stream) {
  MZ_REQUIRE(in && w && bias && out, "mz_conv3x3_f32: null tensor");
  MZ_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cout <= 1024, "mz_conv3x3_f32: bad shape");
  MZ_REQUIRE(stride == 1 || stride == 2, "mz_conv3x3_f32: stride must be 1 or 2");
  MZ_REQUIRE((action_vec == nullptr) == (actions == nullptr), "mz_conv3x3_f32: action_vec and actions go together");
  const int HO = (H + 2 - 3) / stride + 1, WO = (W + 2 - 3) / stride + 1;
  MZ_REQUIRE(WO % TP == 0, "mz_conv3x3_f32: output width %d must be a multiple of %d", WO, TP);
  MZ_REQUIRE(B <= 65535, "mz_conv3x3_f32: batch %d exceeds the grid limit", B);
  const size_t lds = (size_t)3 * (TP * stride + 2) * Cin * sizeof(float);
  MZ_REQUIRE(lds <= 64 * 1024, "mz_conv3x3_f32: Cin %d too large for the LDS window", Cin);
  const int threads = (Cout + 63) / 64 * 64;
  hipLaunchKernelGGL(conv3x3_f32_kernel, dim3(WO / TP, HO, B), dim3(threads), lds, mz::as_stream(stream), in,
                     in_slot, slot_stride, w, bias, residual, out, action_vec, actions, H, W, Cin, Cout, stride,
                     HO, WO, relu);
  return mz::check_launch("mz_conv3x3_f32");
}

int mz_spatial_mean_f32(const float* in, float* out, int32_t B, int32_t HW, int32_t C, mz_stream_t stream) {
  MZ_REQUIRE(in && out && B > 0 && HW > 0 && C > 0, "mz_spatial_mean_f32: bad arguments");
  hipLaunchKernelGGL(spatial_mean_f32_kernel, dim3(B), dim3(256), 0, mz::as_stream(stream), in, out, HW, C);
  return mz::check_launch("mz_spatial_mean_f32");
}

int mz_heads(const float* mean, int32_t B, int32_t C, int32_t Hd, int32_t A, int32_t S,
             const mz_head_weights* policy, const mz_head_weights* value_w, const mz_head_weights* reward_w,
             float* priors, float* value, float* reward, float* policy_logits, float* value_logits,
             float* reward_logits, mz_stream_t stream) {
  MZ_REQUIRE(mean && policy && value_w && priors && value, "mz_heads: null argument");
  MZ_REQUIRE((reward_w == nullptr) == (reward == nullptr), "mz_heads: reward weights and output go together");
  MZ_REQUIRE(B > 0 && A >= 1 && A <= MZ_MAX_ACTIONS && S >= 1, "mz_heads: bad sizes");
  MZ_REQUIRE(C > 0 && C % 4 == 0 && Hd > 0 && Hd % 16 == 0, "mz_heads: C must be a multiple of 4 and Hd of 16");
  for (const mz_head_weights* h : {policy, value_w, reward_w})
    MZ_REQUIRE(!h || !h->bf16 || (C % 32 == 0 && Hd % 32 == 0), "mz_heads: 16-bit weights need C and Hd multiples of 32");
  const int O = 2 * S + 1;
  const int omax = ((O > A ? O : A) + 15) / 16 * 16;
  const size_t lds = (size_t)kHeadSamples * ((Hd + 1) + (C > omax ? C + 1 : omax + 1)) * sizeof(float);
  MZ_REQUIRE(lds <= 64 * 1024, "mz_heads: layer sizes exceed the LDS budget");
  auto cvt = [](const mz_head_weights* h, int out) {
    HeadW r{nullptr, nullptr, nullptr, nullptr, out, 0};
    if (h) r = HeadW{static_cast<const float*>(h->w1p), h->b1, static_cast<const float*>(h->w2p), h->b2, out, h->bf16};
    return r;
  };
  bool fast = C == 256 && Hd == 256 && (O + 15) / 16 <= 3 * (kHeadsBlock / 64) &&
              reinterpret_cast<uintptr_t>(mean) % 16 == 0;
  const int fmt = policy->bf16;   // 0 float32, 1 bf16, 2 IEEE half; the fast kernel wants one 16-bit format throughout
  for (const mz_head_weights* h : {policy, value_w, reward_w}) {
    MZ_REQUIRE(!h || (h->bf16 >= 0 && h->bf16 <= 2), "mz_heads: weight format must be 0 (float32), 1 (bf16) or 2 (half)");
    fast = fast && (!h || (h->bf16 != 0 && h->bf16 == fmt));
  }
  const dim3 grid((B + kHeadSamples - 1) / kHeadSamples, reward_w ? 3 : 2);
  const size_t fast_lds = (size_t)kHeadSamples * (2 * kActPitch + (omax + 1) * sizeof(float));
  if (fast && fmt == 2)
    hipLaunchKernelGGL((heads_kernel<true, _Float16>), grid, dim3(kHeadsBlock), fast_lds, mz::as_stream(stream), mean, B, C,
                       Hd, A, cvt(policy, A), cvt(value_w, O), cvt(reward_w, O), priors, value, reward, policy_logits,
                       value_logits, reward_logits);
  else if (fast)
    hipLaunchKernelGGL((heads_kernel<true, __bf16>), grid, dim3(kHeadsBlock), fast_lds, mz::as_stream(stream), mean, B, C,
                       Hd, A, cvt(policy, A), cvt(value_w, O), cvt(reward_w, O), priors, value, reward, policy_logits,
                       value_logits, reward_logits);
  else
    hipLaunchKernelGGL(heads_kernel<false>, grid, dim3(kHeadsBlock), lds, mz::as_stream(stream), mean, B, C, Hd, A,
                       cvt(policy, A), cvt(value_w, O), cvt(reward_w, O), priors, value, reward, policy_logits,
                       value_logits, reward_logits);
  return mz::check_launch("mz_heads");
}

int mz_support_to_scalar(const float* logits, int32_t n, int32_t S, float* out, mz_stream_t stream) {
  MZ_REQUIRE(logits && out && n > 0 && S >= 1, "mz_support_to_scalar: bad arguments");
  hipLaunchKernelGGL(decode_kernel, dim3((n + 3) / 4), dim3(256), 0, mz::as_stream(stream), logits, n,
                     2 * S + 1, out);
  return mz::check_launch("mz_support_to_scalar");
}

int mz_policy_softmax(const float* logits, float* priors, int32_t n, int32_t A, mz_stream_t stream) {
  MZ_REQUIRE(logits && priors && n > 0 && A >= 1 && A <= MZ_MAX_ACTIONS, "mz_policy_softmax: bad arguments");
  hipLaunchKernelGGL(policy_softmax_kernel, dim3((n + 3) / 4), dim3(256), 0, mz::as_stream(stream), logits, priors,
                     n, A);
  return mz::check_launch("mz_policy_softmax");
}

int mz_nchw_to_nhwc_f32(const float* in, float* out, int32_t B, int32_t C, int32_t HW, mz_stream_t stream) {
  MZ_REQUIRE(in && out && B > 0 && C > 0 && HW > 0 && B <= 65535, "mz_nchw_to_nhwc_f32: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0,
                     mz::as_stream(stream), in, out, C, HW);
  return mz::check_launch("mz_nchw_to_nhwc_f32");
}

int mz_nhwc_to_nchw_f32(const float* in, float* out, int32_t B, int32_t C, int32_t HW, mz_stream_t stream) {
  MZ_REQUIRE(in && out && B > 0 && C > 0 && HW > 0 && B <= 65535, "mz_nhwc_to_nchw_f32: bad arguments");
  hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0,
                     mz::as_stream(stream), in, out, C, HW);
  return mz::check_launch("mz_nhwc_to_nchw_f32");
}

}  // extern "C"
