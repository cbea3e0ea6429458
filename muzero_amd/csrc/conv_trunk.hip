// The hot kernel of the search: 3x3 convolution 256 -> 256 channels on 12x12
// images (every layer of the dynamics network and 18 of the 25 representation
// layers; model.py:146-148, :213), bf16 in / fp32 accumulate on the matrix cores.
//
// One workgroup = one image; its 8 wavefronts split the 256 output channels
// (32 each), so a wavefront's weight fragments are private and stream from L2
// straight into registers in host-packed fragment order.  The image lives in LDS
// for all 9 taps in a PLANAR layout [16-byte channel chunk][pixel]: the 16 pixels
// of an MFMA operand fragment are then 16 consecutive 16-byte slots for every
// tap shift, which makes each ds_read_b128 lane group conflict-free (a
// pixel-major layout, padded or not, always puts two lanes of a group on one
// bank group because half of the group reads the next channel chunk).  Pixels
// 144..159 of every plane are zero: a lane whose tap falls outside the image reads
// the zero pixel that shares a bank group with the pixel it would have read.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_common.h"

namespace {

template <typename T>
struct TrunkArgs {
  const T* in;
  const int* in_slot;
  int slot_stride;
  const T* wpk;
  const float* bias;
  const T* residual;
  T* out;
  const float* avec;
  const int* actions;
  float* mean_out;
  int relu;
  double* stats;  // optional [image][2][256]: per-channel sum and sum of squares of the image's (rounded) outputs
};

template <typename T, int NW>
__device__ __forceinline__ void stage_image(char* lds, const T* src, const float* av, int wave, int lane) {
  typedef vec8<T> bf16x8;   // eight operands of format T (the name predates the second format)
  // 72 blocks of (16 pixels x 4 channel chunks); lane = chunk*16 + pixel: global reads are
  // 64-byte runs per pixel, LDS writes are 16 consecutive slots per 16 lanes
  // all loads first, in straight-line code (a load inside `if (av)` would be waited for where the branch
  // rejoins, one round trip per round); without an action vector the additions are skipped, not the loop
  const int pl = lane & 15, qq = lane >> 4;
  bf16x8 v[72 / NW];
#pragma unroll
  for (int j = 0; j < 72 / NW; ++j) {
    const int it = wave + NW * j;
    v[j] = *reinterpret_cast<const bf16x8*>(src + ((it >> 3) * 16 + pl) * kCh + ((it & 7) * 4 + qq) * 8);
  }
  if (av) {
#pragma unroll
    for (int j = 0; j < 72 / NW; ++j) {
      const int q = ((wave + NW * j) & 7) * 4 + qq;
      const float4 a0 = *reinterpret_cast<const float4*>(av + q * 8);
      const float4 a1 = *reinterpret_cast<const float4*>(av + q * 8 + 4);
      v[j][0] = to_op<T>((float)v[j][0] + a0.x); v[j][1] = to_op<T>((float)v[j][1] + a0.y);
      v[j][2] = to_op<T>((float)v[j][2] + a0.z); v[j][3] = to_op<T>((float)v[j][3] + a0.w);
      v[j][4] = to_op<T>((float)v[j][4] + a1.x); v[j][5] = to_op<T>((float)v[j][5] + a1.y);
      v[j][6] = to_op<T>((float)v[j][6] + a1.z); v[j][7] = to_op<T>((float)v[j][7] + a1.w);
    }
  }
#pragma unroll
  for (int j = 0; j < 72 / NW; ++j) {
    const int it = wave + NW * j;
    const int pix = (it >> 3) * 16 + pl, q = (it & 7) * 4 + qq;
    *reinterpret_cast<bf16x8*>(lds + q * kPlaneB + pix * 16) = v[j];
  }
  // zero pixels 144..159 of every plane: 32 planes x 16 slots = 512 slots
  const bf16x8 z = zero8<T>();
  for (int s = wave * 64 + lane; s < kPlanes * 16; s += NW * 64)
    *reinterpret_cast<bf16x8*>(lds + (s >> 4) * kPlaneB + (kPix + (s & 15)) * 16) = z;
}

// LDS region B: the residual image on the way in, the output image on the way out,
// pixel-major [pixel][256 ch] bf16 with 16 bytes of row padding
constexpr int kRowB = kCh * 2 + 16;
constexpr int kOutBytes = kPix * kRowB;           // 76 032
constexpr int kLdsTotal = kLdsBytes + kOutBytes;  // 157 952 of the CU's 163 840

// SPLIT workgroups share one image, each computing 256 / SPLIT of the output channels (its own staging of the
// image, the same accumulation order per output: results do not depend on SPLIT).  SPLIT = 2 serves batches
// of at most half as many images as there are CUs - a training batch of 128.
template <typename T, int NW, int PF, int SPLIT = 1>
__global__ __launch_bounds__(64 * NW, NW / 4) void conv_trunk_kernel(TrunkArgs<T> a) {
  typedef vec8<T> bf16x8;   // eight / four operands of format T (the names predate the second format)
  typedef vec4<T> bf16x4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* ldsB = lds + kLdsBytes;
  constexpr int WN = 16 / (NW * SPLIT);  // 16-channel output tiles per wavefront
  static_assert(WN >= 1, "at least one channel tile per wavefront");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / SPLIT;
  const int tile0 = ((blockIdx.x % SPLIT) * NW + wave) * WN;  // this wavefront's first 16-channel tile
  const long img = a.in_slot ? (long)a.in_slot[b] * a.slot_stride + b : b;
  const T* src = a.in + img * (long)(kPix * kCh);
  const float* av = a.avec ? a.avec + (long)a.actions[b] * kCh : nullptr;
  const int r16 = lane & 15, kq = lane >> 4;

  // weight fragments: [k-step][n-tile][lane][8]; prefetched PF k-steps ahead
  const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wpk) + (long)tile0 * 64 + lane;
  constexpr int WSTEP = 16 * 64;
  bf16x8 w0[WN], w1[WN];
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    w0[nt] = wp[nt * 64];
    if constexpr (PF == 2) w1[nt] = wp[WSTEP + nt * 64];
  }
  wp += PF * WSTEP;

  stage_image<T, NW>(lds, src, av, wave, lane);
  if (a.residual) {  // whole rows, 16 bytes per lane: coalesced in global memory and in LDS
    const T* rs = a.residual + (long)b * (kPix * kCh);
#pragma unroll
    for (int j = 0; j < kPix * 32 / (64 * NW); ++j) {
      const int u = tid + 64 * NW * j, pix = u >> 5, c16 = u & 31;
      *reinterpret_cast<bf16x8*>(ldsB + pix * kRowB + c16 * 16) = *reinterpret_cast<const bf16x8*>(rs + u * 8);
    }
  }

  f32x4 acc[kMT][WN];
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LDS byte address of this lane's operand pixel in each 16-pixel tile for a tap; lanes whose
  // tap falls outside the image read a zero pixel on the same bank group as the real one
  auto tap_addresses = [&](int tap, int (&addr)[kMT]) {
    const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
      const int m = mt * 16 + r16;
      const int py = m / kSide, px = m - py * kSide;
      const bool ok = (unsigned)(py + dy - 1) < (unsigned)kSide && (unsigned)(px + dx - 1) < (unsigned)kSide;
      const int q = m + (dy - 1) * kSide + (dx - 1);
      addr[mt] = kq * kPlaneB + (ok ? q : kPix + (q & 15)) * 16;
    }
  };
  constexpr int KOFF = 4 * kPlaneB;  // LDS bytes between consecutive 32-channel k-steps

  int aaddr[kMT];
  tap_addresses(0, aaddr);
  __syncthreads();
  if (a.residual) {  // the accumulation starts from the residual (as in the chain kernels)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
      const int cbase = (tile0 + nt) * 16 + kq * 4;
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) {
        const bf16x4 r = *reinterpret_cast<const bf16x4*>(ldsB + (mt * 16 + r16) * kRowB + cbase * 2);
        acc[mt][nt] = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
      }
    }
  }
  bf16x8 afA[kMT], afB[kMT];
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt) afA[mt] = *reinterpret_cast<const bf16x8*>(lds + aaddr[mt]);

  // Software pipeline, one k-step deep for the LDS operand and PF deep for the weights: each
  // step first issues the loads of LATER steps, then runs its 18 MFMAs on registers that were
  // loaded a whole step ago.  sched_barrier keeps hipcc from sinking the loads back down to
  // their uses (it otherwise waits for every ds_read right before the MFMA that needs it).
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    int anext[kMT];
    tap_addresses(tap < 8 ? tap + 1 : 8, anext);
#pragma unroll
    for (int kc = 0; kc < kKC; ++kc) {
      bf16x8(&cur)[kMT] = (kc & 1) ? afB : afA;
      bf16x8(&nxt)[kMT] = (kc & 1) ? afA : afB;
      bf16x8 w2[WN];
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) w2[nt] = wp[nt * 64];  // PF k-steps ahead (the buffer is padded)
      wp += WSTEP;
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt)
        nxt[mt] = *reinterpret_cast<const bf16x8*>(lds + (kc < kKC - 1 ? aaddr[mt] + (kc + 1) * KOFF : anext[mt]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
          acc[mt][nt] = mfma_16x16x32(w0[nt], cur[mt], acc[mt][nt]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) {
        if constexpr (PF == 2) {
          w0[nt] = w1[nt];
          w1[nt] = w2[nt];
        } else {
          w0[nt] = w2[nt];
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) aaddr[mt] = anext[mt];
  }

  // epilogue, part 1: a lane holds channels cbase..cbase+3 of pixel (mt*16 + r16); add bias,
  // apply ReLU, round to bf16 and park the result in LDS region B
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int cbase = (tile0 + nt) * 16 + kq * 4;
    const float4 bv = *reinterpret_cast<const float4*>(a.bias + cbase);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    float t[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};  // sums of the ROUNDED outputs and of their squares
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
      char* slot = ldsB + (mt * 16 + r16) * kRowB + cbase * 2;
      float v0 = acc[mt][nt][0] + bv.x, v1 = acc[mt][nt][1] + bv.y;
      float v2 = acc[mt][nt][2] + bv.z, v3 = acc[mt][nt][3] + bv.w;
      if (a.relu) {
        v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
      }
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
      const bf16x4 ov = to_op4<T>(v0, v1, v2, v3);
      if (a.stats) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float r = (float)ov[j];
          t[j] += r;
          q[j] += r * r;
        }
      }
      *reinterpret_cast<bf16x4*>(slot) = ov;
    }
    if (a.stats) {  // what a following training-mode BatchNorm needs of this image: per lane 9 pixels, then the 16 lanes
      double td[4], qd[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        td[j] = (double)t[j];
        qd[j] = (double)q[j];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          td[j] += __shfl_xor(td[j], m);
          qd[j] += __shfl_xor(qd[j], m);
        }
      }
      if (r16 == 0) {
        double* st = a.stats + (long)b * 2 * kCh + cbase;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          st[j] = td[j];
          st[kCh + j] = qd[j];
        }
      }
    }
    if (a.mean_out) {
      s0 = row16_sum(s0); s1 = row16_sum(s1); s2 = row16_sum(s2); s3 = row16_sum(s3);
      if (r16 == 0) {
        const float inv = 1.0f / (float)kPix;
        *reinterpret_cast<float4*>(a.mean_out + (long)b * kCh + cbase) = make_float4(s0 * inv, s1 * inv, s2 * inv, s3 * inv);
      }
    }
  }
  __syncthreads();
  // epilogue, part 2: the finished image leaves LDS as whole 512-byte rows, 16 bytes per lane
  T* dst = a.out + (long)b * (kPix * kCh);
  constexpr int UPP = 32 / SPLIT;  // 16-byte units per pixel that this workgroup produced
#pragma unroll
  for (int j = 0; j < (kPix * UPP + 64 * NW - 1) / (64 * NW); ++j) {
    const int u = tid + 64 * NW * j, pix = u / UPP, c16 = (blockIdx.x % SPLIT) * UPP + u % UPP;
    if (u < kPix * UPP)
      *reinterpret_cast<bf16x8*>(dst + (pix * 32 + c16) * 8) = *reinterpret_cast<const bf16x8*>(ldsB + pix * kRowB + c16 * 16);
  }
}

// ---------------------------------------------------------------------------
// Fused chain of trunk convolutions: the whole dynamics network (conv_state + its
// residual blocks, model.py:236-241) or the representation network's 12x12 tail
// run as ONE kernel with the image resident in LDS from the first layer to the
// last.  Two planar image buffers fill the CU's 160 KiB of LDS: layer l reads
// buffer l%2 and writes buffer (l+1)%2; the second convolution of a residual block
// finds the block's input still sitting in its destination buffer and adds it in
// place.  Per image the kernel reads 72 KB and writes 72 KB of HBM instead of
// (2..3 x 72 KB) per layer, has no per-layer launch, and needs one workgroup
// barrier per layer.  Workgroups are persistent: each walks images b, b+grid, ...
// ---------------------------------------------------------------------------
#ifndef MZ_CHAIN_PF
#define MZ_CHAIN_PF 2
#endif
#ifndef MZ_CHAIN_NW
#define MZ_CHAIN_NW 8
#endif
#ifndef MZ_PAIR_PF
#define MZ_PAIR_PF 1  // weight queue depth of the two-image kernel: its k-steps are twice as long, one step ahead is the same lead
#endif
// -DMZ_CHAIN_STAMPS builds a diagnostic variant (tools/chain_phase_stamps.py): two workgroups
// record the shader clock at the phase boundaries of every layer.
#ifdef MZ_CHAIN_STAMPS
__device__ long long g_stamps[8192];
#define MZ_STAMP(i) if (stamp) g_stamps[sidx + (i)] = clock64()
// one record per launch of the two-image kernel (workgroup 0): layers, shader-clock cycles and constant-rate
// (100 MHz) ticks from its first instruction to its last - their ratio is the clock the launch ran at
__device__ long long g_launches[3 * 4096];
__device__ unsigned g_launch_count;
// per workgroup of the most recent 5-layer and 18-layer launch: first and last instruction in 100 MHz ticks
__device__ long long g_wg_span[2][512][2];
#define MZ_LAUNCH_BEGIN() const long long lc0 = clock64(), lw0 = wall_clock64()
#define MZ_LAUNCH_END(layers)                                                     \
  if (threadIdx.x == 0 && blockIdx.x < 512 && ((layers) == 5 || (layers) == 18)) {  \
    g_wg_span[(layers) == 18][blockIdx.x][0] = lw0;                                 \
    g_wg_span[(layers) == 18][blockIdx.x][1] = wall_clock64();                      \
  }                                                                                \
  if (blockIdx.x == 0 && threadIdx.x == 0) {                                       \
    const unsigned slot = atomicAdd(&g_launch_count, 1u) % 4096u;                   \
    g_launches[3 * slot] = (layers);                                                \
    g_launches[3 * slot + 1] = clock64() - lc0;                                     \
    g_launches[3 * slot + 2] = wall_clock64() - lw0;                                \
  }
#else
#define MZ_STAMP(i)
#define MZ_LAUNCH_BEGIN()
#define MZ_LAUNCH_END(layers)
#endif

template <typename T>
struct ChainArgs {
  const T* in;
  const int* in_slot;
  int slot_stride;
  T* out;
  const float* avec;
  const int* actions;
  float* mean_out;
  int B;
  int n_layers;
  const T* wpk[kMaxChain];
  const float* bias[kMaxChain];
  unsigned relu_mask;  // bit l: ReLU after layer l
  unsigned res_mask;   // bit l: add the destination buffer's content (residual) before the ReLU
};

template <typename T, int NW, int PF>
__global__ __launch_bounds__(64 * NW, NW / 4) void conv_chain_kernel(ChainArgs<T> a) {
  typedef vec8<T> bf16x8;   // eight / four operands of format T (the names predate the second format)
  typedef vec4<T> bf16x4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int WN = 16 / NW;
  constexpr int WSTEP = 16 * 1024;  // bytes of weight fragments per k-step (16 channel tiles x 1 KB)
  constexpr int KOFF = kPlaneB;  // LDS bytes between consecutive 32-channel k-steps
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;

  // LDS byte address of the 16-byte operand a lane reads for pixel tile mt under a tap: the pixel
  // tile's own slot shifted by the tap, which for a neighbour above the first or below the last
  // row lands in the 16 zero pixels between planes.  Only two kinds of lane need redirecting to a
  // zero pixel by hand: those whose left / right neighbour belongs to another row (a fixed lane
  // pattern per tile: edge_lanes) and, for the upward taps, the first row of the very first plane,
  // which has no zero pixels in front of it.  Planes are ordered so that a k-step's four planes are
  // 8 planes apart and the k-step index is a small immediate offset.
  // (addresses are LDS byte offsets including the array's own base, used through LDS-typed
  // pointers, so that no read pays an add of the base)
  typedef const __attribute__((address_space(3))) bf16x8* lds_frag_ptr;
  const int lds_base = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int vb0 = lds_base + kq * (8 * kPlaneB) + r16 * 16;
  const int vz0 = lds_base + kq * (8 * kPlaneB) + kPix * 16;
  auto tap_addresses = [&](int tap, int buf_off, int (&addr)[kMT]) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const int shift = (dy - 1) * kSide + (dx - 1);
    const int t = vb0 + (buf_off + shift * 16);
    // the zero pixel a redirected lane reads sits in the bank its own pixel would have used, so
    // that redirected and ordinary lanes of a read never collide
    const int z = vz0 + buf_off + ((r16 + shift) & 15) * 16;
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
      unsigned long long bad = dx == 0 ? edge_lanes(mt, 0) : dx == 2 ? edge_lanes(mt, kSide - 1) : 0ull;
      if (mt == 0 && dy == 0) bad |= 0x0FFF0FFF0FFF0FFFull;
      addr[mt] = __builtin_amdgcn_inverse_ballot_w64(bad) ? z : t + mt * 256;
    }
  };
  auto plane_of = [](int q) { return (q & 3) * 8 + (q >> 2); };  // channel chunk q -> plane

  // zero pixels of both buffers (never written afterwards)
  {
    const bf16x8 z = zero8<T>();
    for (int s = tid; s < 2 * kPlanes * 16; s += NW * 64) {
      const int buf = s >> 9, r = s & 511;
      *reinterpret_cast<bf16x8*>(lds + buf * kLdsBytes + (r >> 4) * kPlaneB + (kPix + (r & 15)) * 16) = z;
    }
  }

  int touched = 0;  // sink of the next-image prefetch loads (keeps them alive; never stored)
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const long img = a.in_slot ? (long)a.in_slot[b] * a.slot_stride + b : b;
    const T* src = a.in + img * (long)(kPix * kCh);
    const float* av = a.avec ? a.avec + (long)a.actions[b] * kCh : nullptr;
    // this workgroup's next image: one 4-byte load per 128-byte line pulls all 72 KB of it towards
    // L2 while the last layer of the current image computes (every workgroup otherwise fetches its
    // image from HBM in the same instant, and waits for that burst)
    const int nb = b + gridDim.x;
    const char* nsrc = nullptr;
    if (nb < a.B) {
      const long nimg = a.in_slot ? (long)a.in_slot[nb] * a.slot_stride + nb : nb;
      nsrc = reinterpret_cast<const char*>(a.in + nimg * (long)(kPix * kCh));
    }

    // first layer's weights are requested before the image so that both are in flight together
    // (buffer loads: the layer's base sits in a scalar resource descriptor, the k-step offset in a
    // scalar register that the scalar unit advances, the lane's own offset never changes - no
    // vector instruction is spent on weight addresses; reads past the packed array return zero)
    const int wlane = wave * (WN * 1024) + lane * 16;
    auto wfrag = [&](__amdgpu_buffer_rsrc_t r, int step_off, int nt) {
      return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, wlane + nt * 1024, step_off, 0));
    };
    auto wdesc = [&](int l) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a.wpk[l]), 0, (9 * kKC + 2) * WSTEP, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t wr = wdesc(0);
    int wp = 0;
    bf16x8 wq[PF][WN];  // wq[0] feeds the current k-step, wq[PF-1] was requested last
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) wq[p][nt] = wfrag(wr, p * WSTEP, nt);
    wp = PF * WSTEP;

    {  // image -> buffer 0 (planar); 72 blocks of (16 pixels x 4 channel chunks)
      // all loads first, in straight-line code: a load inside `if (av)` is waited for where the branch rejoins
      const int pl = lane & 15, qq = lane >> 4;
      bf16x8 v[72 / NW];
#pragma unroll
      for (int j = 0; j < 72 / NW; ++j) {
        const int it = wave + NW * j;
        v[j] = *reinterpret_cast<const bf16x8*>(src + ((it >> 3) * 16 + pl) * kCh + ((it & 7) * 4 + qq) * 8);
      }
      if (av) {
#pragma unroll
        for (int j = 0; j < 72 / NW; ++j) {
          const int q = ((wave + NW * j) & 7) * 4 + qq;
          const float4 a0 = *reinterpret_cast<const float4*>(av + q * 8);
          const float4 a1 = *reinterpret_cast<const float4*>(av + q * 8 + 4);
          v[j][0] = to_op<T>((float)v[j][0] + a0.x); v[j][1] = to_op<T>((float)v[j][1] + a0.y);
          v[j][2] = to_op<T>((float)v[j][2] + a0.z); v[j][3] = to_op<T>((float)v[j][3] + a0.w);
          v[j][4] = to_op<T>((float)v[j][4] + a1.x); v[j][5] = to_op<T>((float)v[j][5] + a1.y);
          v[j][6] = to_op<T>((float)v[j][6] + a1.z); v[j][7] = to_op<T>((float)v[j][7] + a1.w);
        }
      }
#pragma unroll
      for (int j = 0; j < 72 / NW; ++j) {
        const int it = wave + NW * j;
        const int pix = (it >> 3) * 16 + pl, q = (it & 7) * 4 + qq;
        *reinterpret_cast<bf16x8*>(lds + plane_of(q) * kPlaneB + pix * 16) = v[j];
      }
    }
    __syncthreads();

#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {
#ifdef MZ_CHAIN_STAMPS
      const bool stamp = (blockIdx.x == 7 || blockIdx.x == 130) && lane == 0 && (wave == 0 || wave == 5) && l < 8;
      const int sidx = ((((blockIdx.x == 130) * 2 + (wave == 5)) * 4 + ((b / (int)gridDim.x) & 3)) * 8 + l) * 4;
#endif
      MZ_STAMP(0);  // layer start
      const int sbuf_off = (l & 1) * kLdsBytes;
      char* dbuf = lds + ((l + 1) & 1) * kLdsBytes;
      const bool last = l == a.n_layers - 1;

      f32x4 acc[kMT][WN];
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if ((a.res_mask >> l) & 1u) {  // the accumulation starts from the residual: what the destination buffer holds
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
          const int cbase = (wave * WN + nt) * 16 + kq * 4;
          const char* plane = dbuf + plane_of(cbase >> 3) * kPlaneB + (cbase & 4) * 2;
#pragma unroll
          for (int mt = 0; mt < kMT; ++mt) {
            const bf16x4 r = *reinterpret_cast<const bf16x4*>(plane + (mt * 16 + r16) * 16);
            acc[mt][nt] = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
          }
        }
      }

      // this layer's bias and the next layer's weight descriptor are fetched now, not when the
      // epilogue needs them
      float4 bv[WN];
#pragma unroll
      for (int nt = 0; nt < WN; ++nt)
        bv[nt] = *reinterpret_cast<const float4*>(a.bias[l] + (wave * WN + nt) * 16 + kq * 4);
      const __amdgpu_buffer_rsrc_t wr_next = wdesc(last ? l : l + 1);

      int aaddr[kMT];
      tap_addresses(0, sbuf_off, aaddr);
      bf16x8 af[kMT];
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) af[mt] = *(lds_frag_ptr)(aaddr[mt]);

      int pf0 = 0, pf1 = 0;
      if (last && nsrc) {  // 576 lines of 128 bytes, 512 threads
        pf0 = *reinterpret_cast<const int*>(nsrc + tid * 128);
        if (tid < kPix * kCh * 2 / 128 - 64 * NW) pf1 = *reinterpret_cast<const int*>(nsrc + (64 * NW + tid) * 128);
      }

#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int kc = 0; kc < kKC; ++kc) {
          bf16x8 w2[WN];
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) w2[nt] = wfrag(wr, wp, nt);
          wp += WSTEP;
          if (kc == kKC - 1) tap_addresses(tap < 8 ? tap + 1 : 8, sbuf_off, aaddr);  // the next step is the next tap's first
          // a pixel tile's operand register is refilled for the NEXT step as soon as this step's
          // MFMAs have read it: one live copy per tile instead of two
#pragma unroll
          for (int mt = 0; mt < kMT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
              acc[mt][nt] = mfma_16x16x32(wq[0][nt], af[mt], acc[mt][nt]);
            af[mt] = *(lds_frag_ptr)(aaddr[mt] + (kc < kKC - 1 ? (kc + 1) * KOFF : 0));
          }
          // issue order inside the step: the two weight loads, then one LDS read of the NEXT
          // step behind every pair of MFMAs of this one (measured +2.5 % over loads-then-MFMAs)
          __builtin_amdgcn_sched_group_barrier(0x020, WN, 0);
#pragma unroll
          for (int i = 0; i < kMT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, WN, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) {
#pragma unroll
            for (int p = 0; p + 1 < PF; ++p) wq[p][nt] = wq[p + 1][nt];
            wq[PF - 1][nt] = w2[nt];
          }
        }
      }

      touched ^= pf0 ^ pf1;
      MZ_STAMP(1);  // main loop done
      // next layer's first weight fragments travel while this layer's epilogue runs
      const bool relu = (a.relu_mask >> l) & 1u;
      {  // not guarded by `!last` (the last layer requests its own first fragments again, unused): loads in a
         // branch cannot be counted, so a later wait for anything older would wait for them too
        wr = wr_next;
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) wq[p][nt] = wfrag(wr, p * WSTEP, nt);
        wp = PF * WSTEP;
      }

      // epilogue: lane holds channels cbase..cbase+3 (8 bytes = half a 16-byte chunk) of pixel
      // mt*16 + r16; the result goes to the destination buffer in the planar layout.  Only the
      // last layer sums its outputs over the pixels (the heads' input).
      auto epilogue = [&](auto with_mean) {
        constexpr bool kMean = decltype(with_mean)::value;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
          const int cbase = (wave * WN + nt) * 16 + kq * 4;
          char* plane = dbuf + plane_of(cbase >> 3) * kPlaneB + (cbase & 4) * 2;
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
          for (int mt = 0; mt < kMT; ++mt) {
            char* slot = plane + (mt * 16 + r16) * 16;
            float v0 = acc[mt][nt][0] + bv[nt].x, v1 = acc[mt][nt][1] + bv[nt].y;
            float v2 = acc[mt][nt][2] + bv[nt].z, v3 = acc[mt][nt][3] + bv[nt].w;
            if (relu) {
              v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
            }
            if (kMean) { s0 += v0; s1 += v1; s2 += v2; s3 += v3; }
            const bf16x4 ov = to_op4<T>(v0, v1, v2, v3);
            *reinterpret_cast<bf16x4*>(slot) = ov;
          }
          if (kMean) {
            s0 = row16_sum(s0); s1 = row16_sum(s1); s2 = row16_sum(s2); s3 = row16_sum(s3);
            if (r16 == 0) {
              const float inv = 1.0f / (float)kPix;
              *reinterpret_cast<float4*>(a.mean_out + (long)b * kCh + cbase) =
                  make_float4(s0 * inv, s1 * inv, s2 * inv, s3 * inv);
            }
          }
        }
      };
      if (last && a.mean_out) epilogue(std::true_type{}); else epilogue(std::false_type{});
      MZ_STAMP(2);  // epilogue done
      __syncthreads();
      MZ_STAMP(3);  // barrier passed
    }

    {  // final buffer -> global, same (16 pixels x 4 chunks) blocks as the load
      const char* fbuf = lds + (a.n_layers & 1) * kLdsBytes;
      T* dst = a.out + (long)b * (kPix * kCh);
      const int pl = lane & 15, qq = lane >> 4;
#pragma unroll
      for (int j = 0; j < 72 / NW; ++j) {
        const int it = wave + NW * j;
        const int pix = (it >> 3) * 16 + pl, q = (it & 7) * 4 + qq;
        *reinterpret_cast<bf16x8*>(dst + pix * kCh + q * 8) = *reinterpret_cast<const bf16x8*>(fbuf + plane_of(q) * kPlaneB + pix * 16);
      }
    }
    __syncthreads();  // the next image overwrites buffer 0
  }
  if (a.B < 0) a.mean_out[tid] = __int_as_float(touched);  // never true: B > 0 is checked by the host
}

// ---------------------------------------------------------------------------
// The same chain for TWO images per workgroup pass.  A weight fragment then feeds
// 18 MFMAs instead of 9: half the weight-fragment traffic (L2 -> registers) per
// MFMA, which is what the chip's clock management rewards - this loop sustains
// ~11 % more than the one-image loop on random data at the same MFMA count.
//
// Both images live in LDS once (no second buffer): 64 planes, image i's plane p at
// (2p + i) * kPlaneB, each layer written over its own input.  A layer is: main loop
// (reads everything) - barrier - epilogue (each wavefront overwrites the 32 channels
// it owns, both images) - barrier.  What the single-image kernel finds in its second
// buffer - the input of a residual block, to be added to the block's second
// convolution - is picked up here just before it is overwritten: the epilogue of
// the block's first convolution loads it into the accumulators, which the second
// convolution then starts from (instead of from zero).
// Per k-step a wavefront runs two half-steps of 18 MFMAs with the same two weight
// fragments, image 0 then image 1; a pixel tile's operand register is refilled
// right after its MFMAs with the other image's operand (same k-step after image 0,
// next k-step after image 1), so 9 operand registers serve both images with the
// same 18-MFMA lead as in the single-image loop.
// ---------------------------------------------------------------------------
template <typename T, int PF>
__global__ __launch_bounds__(512, 2) void conv_chain_pair_kernel(ChainArgs<T> a) {
  typedef vec8<T> bf16x8;   // eight / four operands of format T (the names predate the second format)
  typedef vec4<T> bf16x4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  MZ_LAUNCH_BEGIN();
  constexpr int NW = 8, WN = 2;
  constexpr int WSTEP = 16 * 1024;       // bytes of weight fragments per k-step
  constexpr int PSTRIDE = 2 * kPlaneB;   // LDS bytes between consecutive planes of one image
  constexpr int KOFF = PSTRIDE;          // ... between consecutive 32-channel k-steps
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;

  // operand addresses as in conv_chain_kernel (image 0; image 1 is kPlaneB further)
  typedef const __attribute__((address_space(3))) bf16x8* lds_frag_ptr;
  const int lds_base = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int vb0 = lds_base + kq * (8 * PSTRIDE) + r16 * 16;
  const int vz0 = lds_base + kq * (8 * PSTRIDE) + kPix * 16;
  auto tap_addresses = [&](int tap, int (&addr)[kMT]) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const int shift = (dy - 1) * kSide + (dx - 1);
    const int t = vb0 + shift * 16;
    const int z = vz0 + ((r16 + shift) & 15) * 16;
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
      unsigned long long bad = dx == 0 ? edge_lanes(mt, 0) : dx == 2 ? edge_lanes(mt, kSide - 1) : 0ull;
      if (mt == 0 && dy == 0) bad |= 0x0FFF0FFF0FFF0FFFull;
      addr[mt] = __builtin_amdgcn_inverse_ballot_w64(bad) ? z : t + mt * 256;
    }
  };
  auto plane_of = [](int q) { return (q & 3) * 8 + (q >> 2); };  // channel chunk q -> plane of its image

  {  // zero pixels of all 64 planes (never written afterwards)
    const bf16x8 z = zero8<T>();
    for (int s = tid; s < 2 * kPlanes * 16; s += NW * 64)
      *reinterpret_cast<bf16x8*>(lds + (s >> 4) * kPlaneB + (kPix + (s & 15)) * 16) = z;
  }

  const int wlane = wave * (WN * 1024) + lane * 16;
  auto wfrag = [&](__amdgpu_buffer_rsrc_t r, int step_off, int nt) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, wlane + nt * 1024, step_off, 0));
  };
  auto wdesc = [&](int l) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a.wpk[l]), 0, (9 * kKC + 2) * WSTEP, 0x00020000);
  };
  // Loads in this kernel are unconditional, with a valid stand-in address where the real one does not apply: a
  // load inside a branch makes the compiler wait for every outstanding load where the branch rejoins, which
  // turned the 18 loads of a pass's images (and the bookkeeping before them) into as many round trips.
  const int* const slot_p = a.in_slot ? a.in_slot : reinterpret_cast<const int*>(a.bias[0]);
  const int* const act_p = a.avec ? a.actions : reinterpret_cast<const int*>(a.bias[0]);
  const int pl = lane & 15, qq = lane >> 4;

  for (int b0 = blockIdx.x; b0 < a.B; b0 += 2 * gridDim.x) {
    // the pass's images (a missing second image repeats the first and is never stored) and the next pass's,
    // which are pulled towards L2 during the last layer (a missing one repeats this pass's first)
    const bool two = b0 + (int)gridDim.x < a.B;
    int bsel[4] = {b0, two ? b0 + (int)gridDim.x : b0, b0, b0};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int nb = b0 + (2 + i) * (int)gridDim.x;
      if (nb < a.B) bsel[2 + i] = nb;
    }
    const int bimg[2] = {bsel[0], bsel[1]};
    int slot[4], act[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) slot[i] = slot_p[a.in_slot ? bsel[i] : 0];
#pragma unroll
    for (int i = 0; i < 2; ++i) act[i] = act_p[a.avec ? bimg[i] : 0];
    const T* isrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long img = a.in_slot ? (long)__builtin_amdgcn_readfirstlane(slot[i]) * a.slot_stride + bsel[i] : bsel[i];
      isrc[i] = a.in + img * (long)(kPix * kCh);
    }
    const char* nsrc[2] = {reinterpret_cast<const char*>(isrc[2]), reinterpret_cast<const char*>(isrc[3])};

    __amdgpu_buffer_rsrc_t wr = wdesc(0);
    int wp = 0;
    bf16x8 wq[PF][WN];  // wq[0] feeds the current k-step, wq[PF-1] was requested last
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) wq[p][nt] = wfrag(wr, p * WSTEP, nt);
    wp = PF * WSTEP;

    // (the lane's block coordinates pass through an empty asm so that the addresses derived from
    // them are computed here and now: hoisted to the top of the kernel they would sit in
    // registers, or rather in scratch memory, through every layer of every pass)
    int plv = pl, qqv = qq;
    asm volatile("" : "+v"(plv), "+v"(qqv));
    {  // images -> LDS (planar); 72 blocks of (16 pixels x 4 channel chunks) each.  A lane's channel chunk is the
       // same in all nine rounds (eight wavefronts: block it = wave + 8 j), so one pair of action-vector loads
       // per image serves them all; all 18 image loads and the 4 vector loads travel together.
      static_assert(NW == 8, "a lane's channel chunk must not depend on the round");
      const int q = wave * 4 + qqv;
      bf16x8 v[2][72 / NW];
      float4 a0[2], a1[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 72 / NW; ++j)
          v[i][j] = *reinterpret_cast<const bf16x8*>(isrc[i] + (j * 16 + plv) * kCh + q * 8);
        const float* av = a.avec ? a.avec + (long)__builtin_amdgcn_readfirstlane(act[i]) * kCh : a.bias[0];
        a0[i] = *reinterpret_cast<const float4*>(av + q * 8);
        a1[i] = *reinterpret_cast<const float4*>(av + q * 8 + 4);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 72 / NW; ++j) {
          bf16x8 w = v[i][j];
          if (a.avec) {
            w[0] = to_op<T>((float)w[0] + a0[i].x); w[1] = to_op<T>((float)w[1] + a0[i].y);
            w[2] = to_op<T>((float)w[2] + a0[i].z); w[3] = to_op<T>((float)w[3] + a0[i].w);
            w[4] = to_op<T>((float)w[4] + a1[i].x); w[5] = to_op<T>((float)w[5] + a1[i].y);
            w[6] = to_op<T>((float)w[6] + a1[i].z); w[7] = to_op<T>((float)w[7] + a1[i].w);
          }
          *reinterpret_cast<bf16x8*>(lds + (plane_of(q) * 2 + i) * kPlaneB + (j * 16 + plv) * 16) = w;
        }
      }
    }
    __syncthreads();

    f32x4 acc[2][kMT][WN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[i][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {
      const bool last = l == a.n_layers - 1;
#ifdef MZ_CHAIN_STAMPS
      const bool stamp = (blockIdx.x == 7 || blockIdx.x == 130) && lane == 0 && (wave == 0 || wave == 5) && l < 8;
      const int sidx = ((((blockIdx.x == 130) * 2 + (wave == 5)) * 4 + ((b0 / (2 * (int)gridDim.x)) & 3)) * 8 + l) * 4;
#endif
      MZ_STAMP(0);  // layer start
      const __amdgpu_buffer_rsrc_t wr_next = wdesc(last ? l : l + 1);

      int aaddr[kMT];
      tap_addresses(0, aaddr);
      bf16x8 af[kMT];
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) af[mt] = *(lds_frag_ptr)(aaddr[mt]);

      if (last) {  // 2 x 576 lines of 128 bytes: line tid of each image, the 64 tail lines of both images by
                   // wavefronts 0 and 1 (the others repeat theirs).  One register: holding three through the
                   // main loop so that nothing waits here costs more (+2 % per launch) than the wait.
        static_assert(kPix * kCh * 2 / 128 == 64 * NW + 64, "512 lines + 64");
        int pf = *reinterpret_cast<const int*>(nsrc[0] + tid * 128);
        pf ^= *reinterpret_cast<const int*>(nsrc[1] + tid * 128);
        pf ^= *reinterpret_cast<const int*>(nsrc[wave & 1] + (64 * NW + lane) * 128);
        asm volatile("" ::"v"(pf));  // keeps the loads alive
      }

#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int kc = 0; kc < kKC; ++kc) {
          bf16x8 w2[WN];
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) w2[nt] = wfrag(wr, wp, nt);
          wp += WSTEP;
          // image 0; each operand register is refilled with image 1's operand of this k-step
#pragma unroll
          for (int mt = 0; mt < kMT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
              acc[0][mt][nt] = mfma_16x16x32(wq[0][nt], af[mt], acc[0][mt][nt]);
            af[mt] = *(lds_frag_ptr)(aaddr[mt] + (kPlaneB + kc * KOFF));
          }
          __builtin_amdgcn_sched_group_barrier(0x020, WN, 0);
#pragma unroll
          for (int i = 0; i < kMT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, WN, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (kc == kKC - 1) tap_addresses(tap < 8 ? tap + 1 : 8, aaddr);  // the next step is the next tap's first
          // image 1; refill with image 0's operand of the next k-step
#pragma unroll
          for (int mt = 0; mt < kMT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
              acc[1][mt][nt] = mfma_16x16x32(wq[0][nt], af[mt], acc[1][mt][nt]);
            af[mt] = *(lds_frag_ptr)(aaddr[mt] + (kc < kKC - 1 ? (kc + 1) * KOFF : 0));
          }
#pragma unroll
          for (int i = 0; i < kMT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, WN, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) {
#pragma unroll
            for (int p = 0; p + 1 < PF; ++p) wq[p][nt] = wq[p + 1][nt];
            wq[PF - 1][nt] = w2[nt];
          }
        }
      }
      MZ_STAMP(1);  // main loop done

      float4 bv[WN];  // requested before the barrier, needed after it
#pragma unroll
      for (int nt = 0; nt < WN; ++nt)
        bv[nt] = *reinterpret_cast<const float4*>(a.bias[l] + (wave * WN + nt) * 16 + kq * 4);
      __syncthreads();  // every wavefront has read the layer's input: it may be overwritten
      MZ_STAMP(2);  // everyone has finished reading
      const bool relu = (a.relu_mask >> l) & 1u;
      const bool keep_input = !last && ((a.res_mask >> (l + 1)) & 1u);  // the next layer adds this layer's input
      {  // next layer's first weight fragments travel while the epilogue runs.  Not guarded by `!last` (the last
         // layer requests its own first fragments again, unused): with the loads in a branch the compiler cannot
         // count them and the epilogue's first use of the bias would wait for these loads too, a round trip to
         // L2 at the start of every epilogue.  The empty asm looks at the biases here, where the count is known.
        wr = wr_next;
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) wq[p][nt] = wfrag(wr, p * WSTEP, nt);
        wp = PF * WSTEP;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) asm volatile("" ::"v"(bv[nt].x));
      }
      // epilogue: lane holds channels cbase..cbase+3 (8 bytes = half a 16-byte chunk) of pixel
      // mt*16 + r16 of both images; results replace the layer's input in place
      const float floor_v = op_floor<T>(relu);
      // the input's 16-bit pairs -> two floats, or two zeros (PairToFloat: one or two instructions per value)
      const PairToFloat<T> kept(keep_input);
      const bool want_mean = last && a.mean_out;
      // one straight path for every layer kind (selects, not branches, around the accumulators)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
          const int cbase = (wave * WN + nt) * 16 + kq * 4;
          char* plane = lds + (plane_of(cbase >> 3) * 2 + i) * kPlaneB + (cbase & 4) * 2;
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
          for (int mt = 0; mt < kMT; ++mt) {
            char* slot = plane + (mt * 16 + r16) * 16;
            const float v0 = floor_for_op<T>(acc[i][mt][nt][0] + bv[nt].x, floor_v), v1 = floor_for_op<T>(acc[i][mt][nt][1] + bv[nt].y, floor_v);
            const float v2 = floor_for_op<T>(acc[i][mt][nt][2] + bv[nt].z, floor_v), v3 = floor_for_op<T>(acc[i][mt][nt][3] + bv[nt].w, floor_v);
            // the next layer's starting value: this layer's input where the next layer adds it, else zero
            const uint2 r = *reinterpret_cast<const uint2*>(slot);
            acc[i][mt][nt] = f32x4{kept.even(r.x), kept.odd(r.x), kept.even(r.y), kept.odd(r.y)};
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
            const bf16x4 ov = to_op4_unclamped<T>(v0, v1, v2, v3);
            *reinterpret_cast<bf16x4*>(slot) = ov;
          }
          if (want_mean) {  // only the last layer's outputs are averaged over the pixels (the heads' input)
            s0 = row16_sum(s0); s1 = row16_sum(s1); s2 = row16_sum(s2); s3 = row16_sum(s3);
            if (r16 == 0 && (i == 0 || two)) {
              const float inv = 1.0f / (float)kPix;
              *reinterpret_cast<float4*>(a.mean_out + (long)bimg[i] * kCh + cbase) =
                  make_float4(s0 * inv, s1 * inv, s2 * inv, s3 * inv);
            }
          }
        }
      __syncthreads();
      MZ_STAMP(3);  // outputs in place
    }

    plv = pl, qqv = qq;
    asm volatile("" : "+v"(plv), "+v"(qqv));
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // results -> global, same (16 pixels x 4 chunks) blocks as the load
      if (i == 1 && !two) break;
      T* dst = a.out + (long)bimg[i] * (kPix * kCh);
#pragma unroll
      for (int j = 0; j < 72 / NW; ++j) {
        const int it = wave + NW * j;
        const int pix = (it >> 3) * 16 + plv, q = (it & 7) * 4 + qqv;
        *reinterpret_cast<bf16x8*>(dst + pix * kCh + q * 8) =
            *reinterpret_cast<const bf16x8*>(lds + (plane_of(q) * 2 + i) * kPlaneB + pix * 16);
      }
    }
    __syncthreads();  // the next images overwrite these
  }
  MZ_LAUNCH_END(a.n_layers);
}

template <typename T, int NW, int PF, int SPLIT>
int launch_trunk(const TrunkArgs<T>& a, int B, hipStream_t stream) {
  static bool configured[mz::kMaxDevices] = {false};
  auto kern = conv_trunk_kernel<T, NW, PF, SPLIT>;
  if (int rc = mz::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), kLdsTotal, configured, "conv_trunk")) return rc;
  hipLaunchKernelGGL(kern, dim3(B * SPLIT), dim3(64 * NW), kLdsTotal, stream, a);
  return mz::check_launch("mz_conv3x3_bf16 (trunk)");
}

}  // namespace

#ifdef MZ_CHAIN_STAMPS
extern "C" int mz_debug_read_stamps(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(long long) * n);
}
extern "C" int mz_debug_read_wg_spans(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_span), sizeof(long long) * 2 * 512 * 2);
}
// the last up to 4096 launch records {layers, shader cycles, 100 MHz ticks}; returns the launch count so far
extern "C" int mz_debug_read_launches(long long* host, int n) {
  unsigned count = 0;
  if (hipMemcpyFromSymbol(&count, HIP_SYMBOL(g_launch_count), sizeof(count)) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_launches), sizeof(long long) * 3 * n) != hipSuccess) return -1;
  return (int)count;
}
#endif

namespace {

template <typename T>
int conv_trunk_any(const void* in, const int32_t* in_slot, int32_t slot_stride, const void* wpk, const float* bias,
                   const void* residual, void* out, const float* action_vec, const int32_t* actions, float* mean_out,
                   int32_t B, int32_t relu, hipStream_t stream, double* stats) {
  TrunkArgs<T> a{static_cast<const T*>(in), in_slot, slot_stride, static_cast<const T*>(wpk), bias,
                 static_cast<const T*>(residual), static_cast<T*>(out), action_vec, actions, mean_out, relu, stats};
  // two workgroups per image while that still leaves CUs idle otherwise
  if (2 * B <= mz::compute_units()) return launch_trunk<T, 8, 2, 2>(a, B, stream);
  return launch_trunk<T, 8, 2, 1>(a, B, stream);
}

template <typename T>
int conv_chain_any(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer* layers,
                   int32_t n_layers, void* out, const float* action_vec, const int32_t* actions, float* mean_out,
                   int32_t B, hipStream_t stream) {
  ChainArgs<T> a{};
  a.in = static_cast<const T*>(in);
  a.in_slot = in_slot;
  a.slot_stride = slot_stride;
  a.out = static_cast<T*>(out);
  a.avec = action_vec;
  a.actions = actions;
  a.mean_out = mean_out;
  a.B = B;
  a.n_layers = n_layers;
  for (int l = 0; l < n_layers; ++l) {
    a.wpk[l] = static_cast<const T*>(layers[l].wpk);
    a.bias[l] = layers[l].bias;
    if (layers[l].relu) a.relu_mask |= 1u << l;
    if (layers[l].add_residual) a.res_mask |= 1u << l;
  }
  constexpr int lds = 2 * kLdsBytes;
  const int n_cu = mz::compute_units();
  // one persistent workgroup per CU (all of its LDS); with more images than CUs the workgroups
  // take them two at a time (a pass of two costs ~1.85 passes of one)
  static const bool no_pairs = getenv("MZ_CHAIN_NO_PAIRS") != nullptr;  // kernel comparisons
  if (B > n_cu && !no_pairs) {
    static bool configured2[mz::kMaxDevices] = {false};
    auto kern2 = conv_chain_pair_kernel<T, MZ_PAIR_PF>;
    if (int rc = mz::ensure_dynamic_lds(reinterpret_cast<const void*>(kern2), lds, configured2, "conv_chain_pair"))
      return rc;
    const int pairs = (B + 1) / 2;
    hipLaunchKernelGGL(kern2, dim3(pairs < n_cu ? pairs : n_cu), dim3(512), lds, stream, a);
    return mz::check_launch("mz_conv_chain");
  }
  static bool configured[mz::kMaxDevices] = {false};
  auto kern = conv_chain_kernel<T, MZ_CHAIN_NW, MZ_CHAIN_PF>;
  if (int rc = mz::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, configured, "conv_chain")) return rc;
  const int grid = B < n_cu ? B : n_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * MZ_CHAIN_NW), lds, stream, a);
  return mz::check_launch("mz_conv_chain");
}

}  // namespace

namespace mz {
// called by mz_conv3x3_bf16 / mz_conv3x3_f16 for the 12x12x256->256 stride-1 shape (half: operands in IEEE half)
int conv_trunk_bf16(const void* in, const int32_t* in_slot, int32_t slot_stride, const void* wpk, const float* bias,
                    const void* residual, void* out, const float* action_vec, const int32_t* actions,
                    float* mean_out, int32_t B, int32_t relu, hipStream_t stream, double* stats) {
  return conv_trunk_any<__bf16>(in, in_slot, slot_stride, wpk, bias, residual, out, action_vec, actions, mean_out, B, relu,
                                stream, stats);
}
int conv_trunk_f16(const void* in, const int32_t* in_slot, int32_t slot_stride, const void* wpk, const float* bias,
                   const void* residual, void* out, const float* action_vec, const int32_t* actions,
                   float* mean_out, int32_t B, int32_t relu, hipStream_t stream) {
  return conv_trunk_any<_Float16>(in, in_slot, slot_stride, wpk, bias, residual, out, action_vec, actions, mean_out, B,
                                  relu, stream, nullptr);
}
int conv_chain_bf16(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer* layers,
                    int32_t n_layers, void* out, const float* action_vec, const int32_t* actions, float* mean_out,
                    int32_t B, hipStream_t stream) {
  return conv_chain_any<__bf16>(in, in_slot, slot_stride, layers, n_layers, out, action_vec, actions, mean_out, B, stream);
}
int conv_chain_f16(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer* layers,
                   int32_t n_layers, void* out, const float* action_vec, const int32_t* actions, float* mean_out,
                   int32_t B, hipStream_t stream) {
  return conv_chain_any<_Float16>(in, in_slot, slot_stride, layers, n_layers, out, action_vec, actions, mean_out, B, stream);
}
}  // namespace mz
