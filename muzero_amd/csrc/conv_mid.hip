// The residual blocks of the first downsample stage (24x24 images, 128 channels:
// DownsampleBlock.blocks, model.py:167, :173-174, built from ResidualBlock, model.py:143-159) fused
// into ONE kernel with the image resident in LDS from the first convolution to the last - the
// same idea as the 12x12 chain (conv_trunk.hip), on the other geometry that fits a CU:
//
//   * a 24x24x128 bf16 image is 147 456 bytes: one image per workgroup, held once, every layer
//     written over its own input (main loop - barrier - epilogue - barrier).  The input of a
//     residual block is picked up by the epilogue of the block's first convolution just before it
//     is overwritten, and the second convolution starts its accumulation from there.
//   * planar layout [16-byte channel chunk][pixel], 32 zero pixels behind every plane (and in front
//     of the first): taps that leave the image through its top or bottom row land in them by plain
//     address arithmetic; only a pixel in column 0 / 23 reading its left / right neighbour needs a
//     redirect (fixed lane patterns per pixel tile, period three tiles = two image rows).
//   * eight wavefronts = 2 image halves (18 pixel tiles each) x 4 pairs of 16-channel output tiles:
//     36 accumulator tiles per wavefront, every weight fragment feeds 18 MFMAs and every LDS
//     operand 2 - the instruction mix of the two-image 12x12 kernel.
//
// Per image and layer 169 869 312 FLOP (2 * 576 * 128 * 128 * 9), the same as a 12x12x256 layer;
// HBM traffic per image 144 KB in + 144 KB out per chain instead of per layer.
#include "common.h"
#include "conv_common.h"

namespace {

constexpr int mSide = 24, mPix = 576, mCh = 128;
constexpr int mZero = 32;                          // zero pixels behind every plane (>= 25: one row + one pixel)
constexpr int mPlaneB = (mPix + mZero) * 16;       // 9 728 bytes per channel-chunk plane
constexpr int mPlanes = mCh / 8;                   // 16
constexpr int mFront = mZero * 16;                 // zero pixels in front of plane 0
constexpr int mLdsBytes = mFront + mPlanes * mPlaneB;  // 156 160 of the CU's 163 840
constexpr int mKC = mCh / 32;                      // 4 k-steps per tap
constexpr int mGT = 9;                             // pixel tiles per accumulator group (two groups per wavefront)
constexpr int mWStep = (mCh / 16) * 1024;          // bytes of weight fragments per k-step: 8 output tiles x 1 KB
constexpr int mMaxChain = 8;

template <typename T>
struct MidArgs {
  const T* in;
  T* out;
  int B;
  int n_layers;
  const T* wpk[mMaxChain];
  const float* bias[mMaxChain];
  unsigned relu_mask;  // bit l: ReLU after layer l
  unsigned res_mask;   // bit l: layer l adds the input of layer l - 1 before the ReLU
};

// lanes (replicated over the four 16-lane groups) of pixel tile T whose pixel sits in image column 0 / 23;
// T * 16 mod 24 has period 3, and a wavefront's tiles start at multiples of 9, so T % 3 == mt % 3
__host__ __device__ constexpr unsigned long long mid_edge_lanes(int mt, bool right) {
  const int r = mt % 3;
  const int lane = right ? (r == 1 ? 7 : r == 2 ? 15 : -1) : (r == 0 ? 0 : r == 1 ? 8 : -1);
  return lane < 0 ? 0ull : 0x0001000100010001ull << lane;
}

template <typename T>
__global__ __launch_bounds__(512, 2) void conv_mid_chain_kernel(MidArgs<T> a) {
  typedef vec8<T> bf16x8;   // eight / four operands of format T (the names predate the second format)
  typedef vec4<T> bf16x4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NW = 8, WN = 2;
  constexpr int KOFF = mPlaneB;       // LDS bytes between consecutive 32-channel k-steps
  constexpr int GOFF = mGT * 256;     // ... between a wavefront's two accumulator groups (9 pixel tiles)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, cpair = wave & 3;   // image half; output tiles 2 cpair, 2 cpair + 1
  const int r16 = lane & 15, kq = lane >> 4;
  char* img = lds + mFront;

  // operand addresses (LDS byte offsets including the array's base, used through LDS-typed pointers):
  // channel chunk q = 4 kc + kq lives in plane 4 kq + kc, so the k-step is an immediate offset
  typedef const __attribute__((address_space(3))) bf16x8* lds_frag_ptr;
  const int lds_base = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + mFront;
  const int vb0 = lds_base + kq * (mKC * mPlaneB) + (half * (2 * mGT) * 16 + r16) * 16;
  const int vz0 = lds_base + kq * (mKC * mPlaneB) + mPix * 16;
  // addr[mt]: the lane's operand of pixel tile mt of the wavefront's FIRST accumulator group; the same tile of
  // the second group sits GOFF further - except for a redirected lane, whose zero pixel serves both groups:
  // goff[mt % 3] is GOFF or, for the redirected lanes of that kind of tile, 0
  auto tap_addresses = [&](int tap, int (&addr)[mGT], int (&goff)[3]) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const int shift = (dy - 1) * mSide + (dx - 1);
    const int t = vb0 + shift * 16;
    // a redirected lane reads the zero pixel that shares a bank with the pixel it would have read
    const int z = vz0 + ((r16 + shift) & 15) * 16;
#pragma unroll
    for (int mt = 0; mt < mGT; ++mt) {
      const unsigned long long bad = dx == 0 ? mid_edge_lanes(mt, false) : dx == 2 ? mid_edge_lanes(mt, true) : 0ull;
      addr[mt] = __builtin_amdgcn_inverse_ballot_w64(bad) ? z : t + mt * 256;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const unsigned long long bad = dx == 0 ? mid_edge_lanes(r, false) : dx == 2 ? mid_edge_lanes(r, true) : 0ull;
      goff[r] = __builtin_amdgcn_inverse_ballot_w64(bad) ? 0 : GOFF;
    }
  };
  auto plane_of = [](int q) { return (q & 3) * mKC + (q >> 2); };  // channel chunk q -> plane

  {  // zero pixels: the 32 in front and the 32 behind every plane (never written afterwards)
    const bf16x8 z = zero8<T>();
    for (int s = tid; s < (mPlanes + 1) * mZero; s += NW * 64) {
      const int p = s / mZero, r = s - p * mZero;  // p == 0: the front block
      char* dst = p == 0 ? lds + r * 16 : img + (p - 1) * mPlaneB + (mPix + r) * 16;
      *reinterpret_cast<bf16x8*>(dst) = z;
    }
  }

  const int wlane = cpair * (WN * 1024) + lane * 16;
  auto wfrag = [&](__amdgpu_buffer_rsrc_t r, int step_off, int nt) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, wlane + nt * 1024, step_off, 0));
  };
  auto wdesc = [&](int l) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a.wpk[l]), 0, (9 * mKC + 2) * mWStep, 0x00020000);
  };
  const int pl = lane & 15, qq = lane >> 4;

  int touched = 0;  // sink of the next-image prefetch loads (keeps them alive; never stored)
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const int nb = b + (int)gridDim.x;
    const char* nsrc = nb < a.B ? reinterpret_cast<const char*>(a.in + (long)nb * (mPix * mCh)) : nullptr;

    __amdgpu_buffer_rsrc_t wr = wdesc(0);
    bf16x8 wq[WN];  // the current k-step's fragments; the next step's travel in w2
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) wq[nt] = wfrag(wr, 0, nt);
    int wp = mWStep;

    // (the lane's block coordinates pass through an empty asm so that the addresses derived from
    // them are computed here and now instead of living in registers through every layer)
    int plv = pl, qqv = qq;
    asm volatile("" : "+v"(plv), "+v"(qqv));
    {  // image -> LDS (planar); 144 blocks of (16 pixels x 4 channel chunks)
      const T* src = a.in + (long)b * (mPix * mCh);
#pragma unroll
      for (int j = 0; j < 144 / NW; ++j) {
        const int it = wave + NW * j;
        const int pix = (it >> 2) * 16 + plv, q = (it & 3) * 4 + qqv;
        *reinterpret_cast<bf16x8*>(img + plane_of(q) * mPlaneB + pix * 16) =
            *reinterpret_cast<const bf16x8*>(src + pix * mCh + q * 8);
      }
    }
    __syncthreads();

    f32x4 acc[2][mGT][WN];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int mt = 0; mt < mGT; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[g][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {
      const bool last = l == a.n_layers - 1;
      const __amdgpu_buffer_rsrc_t wr_next = wdesc(last ? l : l + 1);

      int aaddr[mGT], goff[3];
      tap_addresses(0, aaddr, goff);
      bf16x8 af[mGT];
#pragma unroll
      for (int mt = 0; mt < mGT; ++mt) af[mt] = *(lds_frag_ptr)(aaddr[mt]);

      int pf = 0;
      if (last && nsrc) {  // 1152 lines of 128 bytes, 512 threads: pull the next image towards L2
        pf ^= *reinterpret_cast<const int*>(nsrc + tid * 128);
        pf ^= *reinterpret_cast<const int*>(nsrc + (512 + tid) * 128);
        // unconditional (threads past the last line repeat it): a guarded load would be waited for on its own
        constexpr int kLines = mPix * mCh * 2 / 128;
        pf ^= *reinterpret_cast<const int*>(nsrc + (1024 + tid < kLines ? 1024 + tid : kLines - 1) * 128);
      }

#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int kc = 0; kc < mKC; ++kc) {
          bf16x8 w2[WN];
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) w2[nt] = wfrag(wr, wp, nt);
          wp += mWStep;
          // group 0; each operand register is refilled with group 1's operand of this k-step
#pragma unroll
          for (int mt = 0; mt < mGT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
              acc[0][mt][nt] = mfma_16x16x32(wq[nt], af[mt], acc[0][mt][nt]);
            af[mt] = *(lds_frag_ptr)(aaddr[mt] + goff[mt % 3] + kc * KOFF);
          }
          __builtin_amdgcn_sched_group_barrier(0x020, WN, 0);
#pragma unroll
          for (int i = 0; i < mGT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, WN, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (kc == mKC - 1) tap_addresses(tap < 8 ? tap + 1 : 8, aaddr, goff);  // the next step is the next tap's first
          // group 1; refill with group 0's operand of the next k-step
#pragma unroll
          for (int mt = 0; mt < mGT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
              acc[1][mt][nt] = mfma_16x16x32(wq[nt], af[mt], acc[1][mt][nt]);
            af[mt] = *(lds_frag_ptr)(aaddr[mt] + (kc < mKC - 1 ? (kc + 1) * KOFF : 0));
          }
#pragma unroll
          for (int i = 0; i < mGT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, WN, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) wq[nt] = w2[nt];
        }
      }
      touched ^= pf;

      float4 bv[WN];  // requested before the barrier, needed after it
#pragma unroll
      for (int nt = 0; nt < WN; ++nt)
        bv[nt] = *reinterpret_cast<const float4*>(a.bias[l] + (cpair * WN + nt) * 16 + kq * 4);
      __syncthreads();  // every wavefront has read the layer's input: it may be overwritten
      const bool relu = (a.relu_mask >> l) & 1u;
      const bool keep_input = !last && ((a.res_mask >> (l + 1)) & 1u);  // the next layer adds this layer's input
      {  // next layer's first weight fragments travel while the epilogue runs; not guarded by `!last` (the last
         // layer requests its own first fragments again, unused): loads in a branch cannot be counted, and the
         // epilogue's first use of the bias would wait for them too.  The empty asm looks at the biases here.
        wr = wr_next;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) wq[nt] = wfrag(wr, 0, nt);
        wp = mWStep;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) asm volatile("" ::"v"(bv[nt].x));
      }
      // epilogue: a lane holds channels cbase..cbase+3 (half a 16-byte chunk) of pixel (tile * 16 + r16);
      // results replace the layer's input in place
      const float floor_v = op_floor<T>(relu);
      // the input's 16-bit pairs -> two floats, or two zeros (PairToFloat: one or two instructions per value)
      const PairToFloat<T> kept(keep_input);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
          const int cbase = (cpair * WN + nt) * 16 + kq * 4;
          char* plane = img + plane_of(cbase >> 3) * mPlaneB + (cbase & 4) * 2 + ((half * 2 + g) * mGT * 16 + r16) * 16;
#pragma unroll
          for (int mt = 0; mt < mGT; ++mt) {
            char* slot = plane + mt * 256;
            const float v0 = floor_for_op<T>(acc[g][mt][nt][0] + bv[nt].x, floor_v), v1 = floor_for_op<T>(acc[g][mt][nt][1] + bv[nt].y, floor_v);
            const float v2 = floor_for_op<T>(acc[g][mt][nt][2] + bv[nt].z, floor_v), v3 = floor_for_op<T>(acc[g][mt][nt][3] + bv[nt].w, floor_v);
            // the next layer's starting value: this layer's input where the next layer adds it, else zero
            const uint2 r = *reinterpret_cast<const uint2*>(slot);
            acc[g][mt][nt] = f32x4{kept.even(r.x), kept.odd(r.x), kept.even(r.y), kept.odd(r.y)};
            const bf16x4 ov = to_op4_unclamped<T>(v0, v1, v2, v3);
            *reinterpret_cast<bf16x4*>(slot) = ov;
          }
        }
      __syncthreads();
    }

    plv = pl, qqv = qq;
    asm volatile("" : "+v"(plv), "+v"(qqv));
    {  // result -> global, same (16 pixels x 4 chunks) blocks as the load
      T* dst = a.out + (long)b * (mPix * mCh);
#pragma unroll
      for (int j = 0; j < 144 / NW; ++j) {
        const int it = wave + NW * j;
        const int pix = (it >> 2) * 16 + plv, q = (it & 3) * 4 + qqv;
        *reinterpret_cast<bf16x8*>(dst + pix * mCh + q * 8) =
            *reinterpret_cast<const bf16x8*>(img + plane_of(q) * mPlaneB + pix * 16);
      }
    }
    __syncthreads();  // the next image overwrites this one
  }
  if (a.B < 0) a.out[tid] = (T)__int_as_float(touched);  // never true: B > 0 is checked by the host
}

}  // namespace

namespace {
template <typename T>
int conv_mid_chain_any(const char* what, const void* in, const mz_chain_layer* layers, int32_t n_layers, void* out,
                       int32_t B, mz_stream_t stream) {
  MZ_REQUIRE(in && layers && out && B > 0, "%s: null tensor or empty batch", what);
  MZ_REQUIRE(n_layers >= 1 && n_layers <= mMaxChain, "%s: n_layers = %d outside 1..%d", what, n_layers, mMaxChain);
  MidArgs<T> a{};
  a.in = static_cast<const T*>(in);
  a.out = static_cast<T*>(out);
  a.B = B;
  a.n_layers = n_layers;
  for (int l = 0; l < n_layers; ++l) {
    MZ_REQUIRE(layers[l].wpk && layers[l].bias, "%s: layer %d has a null tensor", what, l);
    MZ_REQUIRE(!(layers[l].add_residual && l == 0), "%s: the first layer has no block input to add", what);
    a.wpk[l] = static_cast<const T*>(layers[l].wpk);
    a.bias[l] = layers[l].bias;
    if (layers[l].relu) a.relu_mask |= 1u << l;
    if (layers[l].add_residual) a.res_mask |= 1u << l;
  }
  static bool configured[mz::kMaxDevices] = {false};
  if (int rc = mz::ensure_dynamic_lds(reinterpret_cast<const void*>(conv_mid_chain_kernel<T>), mLdsBytes, configured,
                                      "conv_mid_chain"))
    return rc;
  const int n_cu = mz::compute_units();
  hipLaunchKernelGGL(conv_mid_chain_kernel<T>, dim3(B < n_cu ? B : n_cu), dim3(512), mLdsBytes, mz::as_stream(stream), a);
  return mz::check_launch(what);
}
}  // namespace

extern "C" int mz_conv_chain_mid_bf16(const void* in, const mz_chain_layer* layers, int32_t n_layers, void* out,
                                      int32_t B, mz_stream_t stream) {
  return conv_mid_chain_any<__bf16>("mz_conv_chain_mid_bf16", in, layers, n_layers, out, B, stream);
}

extern "C" int mz_conv_chain_mid_f16(const void* in, const mz_chain_layer* layers, int32_t n_layers, void* out,
                                     int32_t B, mz_stream_t stream) {
  return conv_mid_chain_any<_Float16>("mz_conv_chain_mid_f16", in, layers, n_layers, out, B, stream);
}
