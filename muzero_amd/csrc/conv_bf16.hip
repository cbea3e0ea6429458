// bf16 MFMA 3x3 convolution (padding 1, stride 1 or 2) for gfx950, NHWC.
//
// Implicit GEMM with the operands arranged for CDNA4:
//   * one workgroup (4 or 8 wavefronts) owns an output patch of TH x TW pixels of one
//     image and ALL output channels; the input window of the patch (with its
//     zero halo) is staged ONCE in LDS and then read 9 times, once per filter
//     tap, as MFMA operand fragments (ds_read_b128, rows padded by 16 B so that
//     the 16 pixels of a fragment fall on different bank groups);
//   * the wavefronts split the output channels, so weight fragments are
//     private to a wavefront: they are pre-packed on the host in fragment order
//     and stream straight from L2 into registers (1 KiB contiguous per
//     wave-instruction), never through LDS;
//   * v_mfma_f32_16x16x32_bf16 with the weights as the row operand: a lane ends
//     up with 4 consecutive output channels of one pixel, so bias + residual +
//     ReLU + bf16 packing and the 8-byte NHWC store need no cross-lane traffic.
// Fused: folded-BatchNorm bias, residual add, ReLU, the dynamics network's
// "hidden + conv1x1(one-hot action)" prologue, the gather of parent hidden
// states from the node pool, and the spatial mean for the heads.
#include "common.h"
#include "conv_common.h"

namespace mz {
int conv_trunk_bf16(const void* in, const int32_t* in_slot, int32_t slot_stride, const void* wpk, const float* bias,
                    const void* residual, void* out, const float* action_vec, const int32_t* actions,
                    float* mean_out, int32_t B, int32_t relu, hipStream_t stream, double* stats = nullptr);
int conv_chain_bf16(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer* layers,
                    int32_t n_layers, void* out, const float* action_vec, const int32_t* actions, float* mean_out,
                    int32_t B, hipStream_t stream);
int conv_trunk_f16(const void* in, const int32_t* in_slot, int32_t slot_stride, const void* wpk, const float* bias,
                   const void* residual, void* out, const float* action_vec, const int32_t* actions,
                   float* mean_out, int32_t B, int32_t relu, hipStream_t stream);
int conv_chain_f16(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer* layers,
                   int32_t n_layers, void* out, const float* action_vec, const int32_t* actions, float* mean_out,
                   int32_t B, hipStream_t stream);
}

namespace {

template <typename T>
struct ConvArgs {
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
  const int* wsel = nullptr;  // optional device index: the weights are wpk + *wsel * wsel_stride (elements)
  long wsel_stride = 0;
};

// Every load of the window is in flight before the first LDS write waits for one.  That only holds when the
// loads sit in straight-line code: a load inside a branch (`if (inside) v = load`) makes the compiler wait for
// ALL outstanding loads where the branch rejoins, one full round trip per unrolled round.  So pixels outside
// the image load a clamped address and are zeroed afterwards, and the action-vector variant is chosen once.
template <typename T, class C, bool AV>
__device__ __forceinline__ void stage_input_impl(char* lds, const T* src, const float* av, int cc, int iy0,
                                                 int ix0, int tid) {
  typedef vec8<T> bf16x8;   // eight operands of format T (the name predates the second format)
  constexpr int UPP = C::CINC / 8;  // 16-byte units per pixel
  constexpr int UNITS = C::IH * C::IW * UPP;
  constexpr int ROUNDS = (UNITS + C::THREADS - 1) / C::THREADS;
  bf16x8 v[ROUNDS];
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    int u = tid + i * C::THREADS;
    u = u < UNITS ? u : UNITS - 1;
    const int pix = u / UPP, cu = u % UPP;
    const int py = pix / C::IW, px = pix % C::IW;
    int iy = iy0 + py, ix = ix0 + px;
    iy = iy < 0 ? 0 : (iy < C::HIN ? iy : C::HIN - 1);
    ix = ix < 0 ? 0 : (ix < C::WIN ? ix : C::WIN - 1);
    v[i] = *reinterpret_cast<const bf16x8*>(src + ((long)iy * C::WIN + ix) * C::CIN + cc * C::CINC + cu * 8);
  }
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    const int u = tid + i * C::THREADS;
    if (u < UNITS) {
      const int pix = u / UPP, cu = u % UPP;
      const int py = pix / C::IW, px = pix % C::IW;
      const int iy = iy0 + py, ix = ix0 + px;
      bf16x8 w = v[i];
      if constexpr (AV) {
        const float4 a0 = *reinterpret_cast<const float4*>(av + cc * C::CINC + cu * 8);
        const float4 a1 = *reinterpret_cast<const float4*>(av + cc * C::CINC + cu * 8 + 4);
        w[0] = to_op<T>((float)w[0] + a0.x); w[1] = to_op<T>((float)w[1] + a0.y);
        w[2] = to_op<T>((float)w[2] + a0.z); w[3] = to_op<T>((float)w[3] + a0.w);
        w[4] = to_op<T>((float)w[4] + a1.x); w[5] = to_op<T>((float)w[5] + a1.y);
        w[6] = to_op<T>((float)w[6] + a1.z); w[7] = to_op<T>((float)w[7] + a1.w);
      }
      if (!(iy >= 0 && iy < C::HIN && ix >= 0 && ix < C::WIN)) w = zero8<T>();
      *reinterpret_cast<bf16x8*>(lds + py * C::ROWP + px * C::ROWB + cu * 16) = w;
    }
  }
}

template <typename T, class C>
__device__ __forceinline__ void stage_input(char* lds, const T* src, const float* av, int cc, int iy0, int ix0,
                                            int tid) {
  if (av) stage_input_impl<T, C, true>(lds, src, av, cc, iy0, ix0, tid);
  else stage_input_impl<T, C, false>(lds, src, av, cc, iy0, ix0, tid);
}

// The same window from the rotating, tiled observation [H][W/16][D][56] (prep.hip): the patch's 17 input
// columns - 16 output-stride columns and the one to their left - are exactly one tile's runs (halo pixel, 16
// pixels, padding: 7 units of 16 bytes per slot).  A run holds one slot's three channels of consecutive pixels;
// the window wants every slot's channels of one pixel side by side.  A lane takes 8 pixels of one slot (48
// bytes), the four lanes of a quad hold four neighbouring slots, and a 4x4 transposition inside the quad (two
// exchange steps over the data-parallel-primitive lane network) leaves each lane with two pixels of all four
// slots: 24 contiguous, 8-byte-aligned bytes of the window per pixel, written as three 8-byte stores.  Writing
// the values two bytes at a time instead costs eight times the LDS instructions, most of them bank conflicts.
// The 17th pixel of each run (three values) is scattered two bytes at a time.
__device__ __forceinline__ unsigned quad_xor1(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
}
__device__ __forceinline__ unsigned quad_xor2(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
}

// (the staging moves 16-bit values without looking inside them: the same code serves both operand formats)
template <typename T, class C>
__device__ __forceinline__ void stage_input_rot(char* lds, const T* src, int iy0, int ix0, int tid) {
  static_assert(C::IW == 17 && C::IH == 17 && C::CIN == 96 && C::CINC == C::CIN && C::WIN % 16 == 0 &&
                C::THREADS == 512 && C::ROWB % 8 == 0, "one 16-pixel tile, 32 slots, eight wavefronts");
  constexpr int D = C::CIN / 3, TPR = C::WIN / 16;  // slots, tiles per image row
  constexpr int RUN = 56;                           // values per slot run
  typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
  typedef unsigned __attribute__((ext_vector_type(2))) u32x2;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xt = (ix0 + 1) / 16;
  const int s = lane & 31, h = lane >> 5;  // slot; which 8 pixels of the run
  const bool b0 = lane & 1, b1 = lane & 2;
  // rows wave and wave + 8 in every wavefront; row 16 in wavefront 0, the 17th pixels in the others
  constexpr int ROUNDS = 3;
  u32x4 v[ROUNDS][3];
  auto row_of = [&](int i) { return i < 2 ? wave + 8 * i : 16; };
  auto clamp_row = [&](int row) {
    const int iy = iy0 + row;
    return (iy >= 0 && iy < C::HIN) ? iy : 0;  // rows outside the image load row 0 and are zeroed afterwards
  };
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    if (i == 2 && wave != 0) break;
    const u32x4* p = reinterpret_cast<const u32x4*>(
        src + (((long)clamp_row(row_of(i)) * TPR + xt) * D + s) * RUN + h * 24);
    v[i][0] = p[0]; v[i][1] = p[1]; v[i][2] = p[2];
  }
  // the 17th pixels: 17 rows x 32 slots, two per lane of wavefronts 1..7
  u32x2 last[2];
  int last_row[2], last_s[2];
  if (wave != 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int n = (wave - 1) * 64 + lane + 448 * i;
      n = n < C::IH * D ? n : C::IH * D - 1;
      last_row[i] = n / D; last_s[i] = n % D;
      last[i] = *reinterpret_cast<const u32x2*>(
          src + (((long)clamp_row(last_row[i]) * TPR + xt) * D + last_s[i]) * RUN + 48);
    }
  }

#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    if (i == 2 && wave != 0) break;
    const int row = row_of(i);
    const int iy = iy0 + row;
    const unsigned keep = (iy >= 0 && iy < C::HIN) ? 0xffffffffu : 0u;
    // blk[d][0..2]: this slot's pixels 2d, 2d+1 (of the lane's eight) - the block lane d of the quad wants
    unsigned blk[4][3];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int c = 0; c < 3; ++c) blk[d][c] = v[i][(d * 3 + c) / 4][(d * 3 + c) % 4] & keep;
    // step 1, partner lane ^ 2: keep the blocks for lanes with this lane's bit 1, receive the partner's
    unsigned ka[3], kb[3], ra[3], rb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ka[c] = b1 ? blk[2][c] : blk[0][c];
      kb[c] = b1 ? blk[3][c] : blk[1][c];
      ra[c] = quad_xor2(b1 ? blk[0][c] : blk[2][c]);
      rb[c] = quad_xor2(b1 ? blk[1][c] : blk[3][c]);
    }
    // step 2, partner lane ^ 1: x[j] is the block of slot (lane ^ j) for this lane's two pixels
    unsigned x[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      x[0][c] = b0 ? kb[c] : ka[c];
      x[1][c] = quad_xor1(b0 ? ka[c] : kb[c]);
      x[2][c] = b0 ? rb[c] : ra[c];
      x[3][c] = quad_xor1(b0 ? ra[c] : rb[c]);
    }
    // order by slot: sl[q] = x[q ^ (lane & 3)]
    unsigned sl[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned p0 = b1 ? x[2][c] : x[0][c], p1 = b1 ? x[3][c] : x[1][c];
      const unsigned p2 = b1 ? x[0][c] : x[2][c], p3 = b1 ? x[1][c] : x[3][c];
      sl[0][c] = b0 ? p1 : p0; sl[1][c] = b0 ? p0 : p1;
      sl[2][c] = b0 ? p3 : p2; sl[3][c] = b0 ? p2 : p3;
    }
    // a block is (p r0, p r1 | p r2, p' r0 | p' r1, p' r2) for the pixel pair p, p'
    const int px = h * 8 + (lane & 3) * 2;
    char* dst = lds + row * C::ROWP + px * C::ROWB + (s & ~3) * 6;
    unsigned o[2][6];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned* e = sl[2 * q];
      const unsigned* f = sl[2 * q + 1];
      o[0][3 * q + 0] = e[0];
      o[0][3 * q + 1] = (e[1] & 0xffffu) | (f[0] << 16);
      o[0][3 * q + 2] = (f[0] >> 16) | (f[1] << 16);
      o[1][3 * q + 0] = (e[1] >> 16) | (e[2] << 16);
      o[1][3 * q + 1] = (e[2] >> 16) | (f[1] & 0xffff0000u);
      o[1][3 * q + 2] = f[2];
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        *reinterpret_cast<u32x2*>(dst + p * C::ROWB + c * 8) = u32x2{o[p][2 * c], o[p][2 * c + 1]};
  }
  if (wave != 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = (wave - 1) * 64 + lane + 448 * i;
      if (n < C::IH * D) {
        const int iy = iy0 + last_row[i];
        const unsigned keep = (iy >= 0 && iy < C::HIN) ? 0xffffffffu : 0u;
        unsigned short* d16 = reinterpret_cast<unsigned short*>(lds + last_row[i] * C::ROWP + 16 * C::ROWB + last_s[i] * 6);
        d16[0] = (unsigned short)(last[i][0] & keep);
        d16[1] = (unsigned short)((last[i][0] & keep) >> 16);
        d16[2] = (unsigned short)(last[i][1] & keep);
      }
    }
  }
}

// -DMZ_PATCH_STAMPS builds a diagnostic variant (tools/patch_phase_stamps.py): wavefront 0 of every 97th
// workgroup records the shader clock at the phase boundaries of its patch (per channel chunk: start, window in
// LDS, barrier passed, main loop done; then epilogue parked, barrier passed, stores issued).
#ifdef MZ_PATCH_STAMPS
__device__ long long g_patch_stamps[64 * 32];
#define MZ_PSTAMP(i) if (pstamp) g_patch_stamps[pslot * 32 + (i)] = clock64()
#else
#define MZ_PSTAMP(i)
#endif

template <typename T, class C, bool ROT = false>
__global__ __launch_bounds__(C::THREADS, C::MINW) void conv3x3_bf16_kernel(ConvArgs<T> a) {
  typedef vec8<T> bf16x8;   // eight / four operands of format T (the names predate the second format)
  typedef vec4<T> bf16x4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int MT = C::MT, WN = C::WN, KC = C::KC;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / (C::PY * C::PX), patch = blockIdx.x % (C::PY * C::PX);
#ifdef MZ_PATCH_STAMPS
  const int pslot = (int)(blockIdx.x / 97) % 64;
  const bool pstamp = blockIdx.x % 97 == 0 && blockIdx.x >= gridDim.x / 2 && tid == 0;   // second half: steady state
#endif
  const int oy0 = (patch / C::PX) * C::TH, ox0 = (patch % C::PX) * C::TW;
  const long img = a.in_slot ? (long)a.in_slot[b] * a.slot_stride + b : b;
  // elements per image: NHWC, or the rotating layout's runs of 56 values per (row, 16-pixel tile, slot)
  constexpr long kImage = ROT ? (long)C::HIN * (C::WIN / 16) * (C::CIN / 3) * 56 : (long)C::HIN * C::WIN * C::CIN;
  const T* src = a.in + img * kImage;
  const float* av = a.avec ? a.avec + (long)a.actions[b] * C::CIN : nullptr;
  const int iy0 = oy0 * C::S - 1, ix0 = ox0 * C::S - 1;
  const int r16 = lane & 15, kq = lane >> 4;

  f32x4 acc[MT][WN];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // weight fragments: [k-step][n-tile][lane][8]; this wave's n-tiles are wave*WN .. +WN.  They
  // come by buffer loads: base in a scalar descriptor, k-step offset in a scalar register, the
  // lane's own offset fixed - no vector address arithmetic (reads past the array return zero)
  constexpr int WSTEP = C::NTILES * 1024;  // bytes per k-step
  const T* wbase = a.wsel ? a.wpk + (long)__builtin_amdgcn_readfirstlane(*a.wsel) * a.wsel_stride : a.wpk;
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(wbase), 0, (C::KSTEPS + 2) * WSTEP, 0x00020000);
  const int wlane = wave * (WN * 1024) + lane * 16;
  auto wfrag = [&](int step_off, int nt) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, wlane + nt * 1024, step_off, 0));
  };
  // a ring of PF + 1 fragment sets: step st multiplies set st % (PF + 1) while the loads of step st + PF land
  // in set (st + PF) % (PF + 1); the k-steps of a channel chunk are a multiple of the ring, so the (unrolled)
  // indices are the same in every chunk
  constexpr int RING = C::PF + 1;
  static_assert((9 * KC) % RING == 0, "k-steps per channel chunk must be a multiple of the prefetch ring");
  bf16x8 wq[RING][WN];
#pragma unroll
  for (int pf = 0; pf < C::PF; ++pf)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) wq[pf][nt] = wfrag(pf * WSTEP, nt);
  int wp = C::PF * WSTEP;

  // LDS byte address of this lane's pixel (tap 0, first k-step) in each 16-pixel tile; every
  // other tap / k-step is a compile-time offset from it, so it rides in the ds_read immediate
  typedef const __attribute__((address_space(3))) bf16x8* lds_frag_ptr;
  const int lds_base = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = mt * 16 + r16;
    if (m >= C::M) m = C::M - 1;  // padding rows recompute the last pixel; never stored
    const int py = m / C::TW, px = m % C::TW;
    abase[mt] = lds_base + py * C::S * C::ROWP + px * C::S * C::ROWB + kq * 16;
  }
  constexpr int NS = 9 * KC;  // k-steps per channel chunk
  auto step_offset = [](int st) {
    const int tap = st / KC, kc = st % KC;
    return (tap / 3) * C::ROWP + (tap % 3) * C::ROWB + kc * 64;
  };

  for (int cc = 0; cc < C::NCC; ++cc) {
    if (cc > 0) __syncthreads();
    MZ_PSTAMP(cc * 4 + 0);
    if constexpr (ROT) stage_input_rot<T, C>(lds, src, iy0, ix0, tid);
    else stage_input<T, C>(lds, src, av, cc, iy0, ix0, tid);
    MZ_PSTAMP(cc * 4 + 1);
    __syncthreads();
    MZ_PSTAMP(cc * 4 + 2);
    // software pipeline: a pixel tile's operand register is refilled with step st+1's operand
    // right after step st's MFMAs have read it
    bf16x8 af[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = *(lds_frag_ptr)(abase[mt]);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) wq[(st + C::PF) % RING][nt] = wfrag(wp, nt);  // PF k-steps ahead
      wp += WSTEP;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
          acc[mt][nt] = mfma_16x16x32(wq[st % RING][nt], af[mt], acc[mt][nt]);
        if (st + 1 < NS) af[mt] = *(lds_frag_ptr)(abase[mt] + step_offset(st + 1));
      }
      __builtin_amdgcn_sched_group_barrier(0x020, WN, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, WN, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    MZ_PSTAMP(cc * 4 + 3);
  }

  // epilogue, part 1: a lane holds channels cbase..cbase+3 of pixel (mt*16 + r16).  Bias,
  // residual, ReLU and the bf16 rounding happen here; the tile is parked in LDS (the input
  // window is dead by now) so that it can leave as whole rows
  constexpr int OROW = C::COUT * 2 + 16;
  static_assert(C::M * OROW <= C::LDS_BYTES, "output tile must fit in the input window's LDS");
  __syncthreads();
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int cbase = (wave * WN + nt) * 16 + kq * 4;
    const float4 bv = *reinterpret_cast<const float4*>(a.bias + cbase);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 16 + r16;
      if (m < C::M) {
        float v0 = acc[mt][nt][0] + bv.x, v1 = acc[mt][nt][1] + bv.y;
        float v2 = acc[mt][nt][2] + bv.z, v3 = acc[mt][nt][3] + bv.w;
        if (a.residual) {
          const int py = m / C::TW, px = m % C::TW;
          const long o = (((long)b * C::HO + oy0 + py) * C::WO + ox0 + px) * C::COUT + cbase;
          const bf16x4 r = *reinterpret_cast<const bf16x4*>(a.residual + o);
          v0 += (float)r[0]; v1 += (float)r[1]; v2 += (float)r[2]; v3 += (float)r[3];
        }
        if (a.relu) {
          v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
        }
        const bf16x4 ov = to_op4<T>(v0, v1, v2, v3);
        *reinterpret_cast<bf16x4*>(lds + m * OROW + cbase * 2) = ov;
      }
    }
  }
  MZ_PSTAMP(16);
  __syncthreads();
  MZ_PSTAMP(17);
  // part 2: 16 bytes per lane, consecutive lanes on consecutive channels of a pixel
  constexpr int UPR = C::COUT / 8;  // 16-byte units per output pixel
  for (int u = tid; u < C::M * UPR; u += C::THREADS) {
    const int m = u / UPR, c16 = u % UPR;
    const int py = m / C::TW, px = m % C::TW;
    const long o = (((long)b * C::HO + oy0 + py) * C::WO + ox0 + px) * C::COUT + c16 * 8;
    *reinterpret_cast<bf16x8*>(a.out + o) = *reinterpret_cast<const bf16x8*>(lds + m * OROW + c16 * 16);
  }
  MZ_PSTAMP(18);
}

template <typename T, class C, bool ROT = false>
int launch(const ConvArgs<T>& a, int B, hipStream_t stream) {
  static bool configured[mz::kMaxDevices] = {false};
  auto kern = conv3x3_bf16_kernel<T, C, ROT>;
  if (int rc = mz::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), C::LDS_BYTES, configured, "conv3x3_bf16"))
    return rc;
  hipLaunchKernelGGL(kern, dim3(B * C::PY * C::PX), dim3(C::THREADS), C::LDS_BYTES, stream, a);
  return mz::check_launch("mz_conv3x3_bf16");
}

template <class C>
bool matches(int H, int W, int Cin, int Cout, int stride) {
  return H == C::HIN && W == C::WIN && Cin == C::CIN && Cout == C::COUT && stride == C::S;
}

template <typename T>
int conv3x3_any(const char* what, const void* in, const int32_t* in_slot, int32_t slot_stride, const void* wpk,
                const float* bias, const void* residual, void* out, const float* action_vec, const int32_t* actions,
                float* mean_out, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t stride, int32_t relu,
                mz_stream_t stream) {
  MZ_REQUIRE(in && wpk && bias && out && B > 0, "%s: null tensor or empty batch", what);
  MZ_REQUIRE((action_vec == nullptr) == (actions == nullptr), "%s: action_vec and actions go together", what);
  ConvArgs<T> a{static_cast<const T*>(in), in_slot, slot_stride, static_cast<const T*>(wpk), bias,
                static_cast<const T*>(residual), static_cast<T*>(out), action_vec, actions, mean_out, relu};
  hipStream_t s = mz::as_stream(stream);
  if (H == 12 && W == 12 && Cin == 256 && Cout == 256 && stride == 1) {  // the trunk shape has its own kernel
    if constexpr (__is_same(T, __bf16))
      return mz::conv_trunk_bf16(in, in_slot, slot_stride, wpk, bias, residual, out, action_vec, actions, mean_out, B,
                                 relu, s);
    else
      return mz::conv_trunk_f16(in, in_slot, slot_stride, wpk, bias, residual, out, action_vec, actions, mean_out, B,
                                relu, s);
  }
  MZ_REQUIRE(mean_out == nullptr, "%s: the fused spatial mean needs a whole-image kernel", what);
  if (matches<ConvStem>(H, W, Cin, Cout, stride)) return launch<T, ConvStem>(a, B, s);
  if (matches<ConvDown0>(H, W, Cin, Cout, stride)) return launch<T, ConvDown0>(a, B, s);
  if (matches<ConvMid>(H, W, Cin, Cout, stride)) return launch<T, ConvMid>(a, B, s);
  if (matches<ConvDown1>(H, W, Cin, Cout, stride)) return launch<T, ConvDown1>(a, B, s);
  return mz::fail(MZ_E_ARG, "%s: no MFMA kernel for %dx%d, %d->%d channels, stride %d", what, H, W, Cin, Cout, stride);
}

template <typename T>
int conv3x3_rot_any(const char* what, const void* obs_rot, const void* wpk, int64_t copy_elems, const int32_t* newest,
                    const float* bias, void* out, int32_t B, int32_t relu, mz_stream_t stream) {
  MZ_REQUIRE(obs_rot && wpk && bias && out && B > 0, "%s: null tensor or empty batch", what);
  MZ_REQUIRE(copy_elems % 8 == 0, "%s: weight copies must be 16-byte aligned", what);
  ConvArgs<T> a{static_cast<const T*>(obs_rot), nullptr, 0, static_cast<const T*>(wpk), bias, nullptr,
                static_cast<T*>(out), nullptr, nullptr, nullptr, relu, newest, (long)copy_elems};
  return launch<T, ConvStemRot, true>(a, B, mz::as_stream(stream));
}

int check_chain_args(const char* what, const void* in, const mz_chain_layer* layers, int32_t n_layers, const void* out,
                     const float* action_vec, const int32_t* actions, int32_t B) {
  MZ_REQUIRE(in && layers && out && B > 0, "%s: null tensor or empty batch", what);
  MZ_REQUIRE(n_layers >= 1 && n_layers <= 20, "%s: n_layers must be 1..20 (got %d)", what, n_layers);
  MZ_REQUIRE((action_vec == nullptr) == (actions == nullptr), "%s: action_vec and actions go together", what);
  for (int l = 0; l < n_layers; ++l) MZ_REQUIRE(layers[l].wpk && layers[l].bias, "%s: layer %d is null", what, l);
  MZ_REQUIRE(!layers[0].add_residual, "%s: the first layer has no residual source", what);
  return MZ_OK;
}

}  // namespace

#ifdef MZ_PATCH_STAMPS
extern "C" int mz_debug_read_patch_stamps(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_patch_stamps), sizeof(long long) * 64 * 32);
}
#endif

extern "C" {

int mz_conv3x3_bf16_info(int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t stride, int32_t* cin_chunk,
                         int32_t* whole_image) {
  int cc = 0, whole = 0;
  if (matches<ConvStem>(H, W, Cin, Cout, stride)) cc = ConvStem::CINC;
  else if (matches<ConvDown0>(H, W, Cin, Cout, stride)) cc = ConvDown0::CINC;
  else if (matches<ConvMid>(H, W, Cin, Cout, stride)) cc = ConvMid::CINC;
  else if (matches<ConvDown1>(H, W, Cin, Cout, stride)) cc = ConvDown1::CINC;
  else if (H == 12 && W == 12 && Cin == 256 && Cout == 256 && stride == 1) { cc = 256; whole = 1; }
  else
    return mz::fail(MZ_E_ARG, "conv3x3_bf16: no MFMA kernel for %dx%d, %d->%d channels, stride %d", H, W, Cin,
                    Cout, stride);
  if (cin_chunk) *cin_chunk = cc;
  if (whole_image) *whole_image = whole;
  return MZ_OK;
}

int mz_conv3x3_bf16(const void* in, const int32_t* in_slot, int32_t slot_stride, const void* wpk,
                    const float* bias, const void* residual, void* out, const float* action_vec,
                    const int32_t* actions, float* mean_out, int32_t B, int32_t H, int32_t W, int32_t Cin,
                    int32_t Cout, int32_t stride, int32_t relu, mz_stream_t stream) {
  return conv3x3_any<__bf16>("mz_conv3x3_bf16", in, in_slot, slot_stride, wpk, bias, residual, out, action_vec, actions,
                             mean_out, B, H, W, Cin, Cout, stride, relu, stream);
}

int mz_conv3x3_f16(const void* in, const int32_t* in_slot, int32_t slot_stride, const void* wpk,
                   const float* bias, const void* residual, void* out, const float* action_vec,
                   const int32_t* actions, float* mean_out, int32_t B, int32_t H, int32_t W, int32_t Cin,
                   int32_t Cout, int32_t stride, int32_t relu, mz_stream_t stream) {
  return conv3x3_any<_Float16>("mz_conv3x3_f16", in, in_slot, slot_stride, wpk, bias, residual, out, action_vec, actions,
                               mean_out, B, H, W, Cin, Cout, stride, relu, stream);
}

int mz_conv3x3_bf16_stats(const void* in, const void* wpk, const float* bias, void* out, double* stats, int32_t B,
                          mz_stream_t stream) {
  MZ_REQUIRE(in && wpk && bias && out && stats && B > 0, "mz_conv3x3_bf16_stats: null tensor or empty batch");
  return mz::conv_trunk_bf16(in, nullptr, 0, wpk, bias, nullptr, out, nullptr, nullptr, nullptr, B, 0, mz::as_stream(stream),
                             stats);
}

int mz_conv3x3_bf16_rot(const void* obs_rot, const void* wpk, const float* bias, void* out, int32_t B,
                        int32_t relu, mz_stream_t stream) {
  return conv3x3_rot_any<__bf16>("mz_conv3x3_bf16_rot", obs_rot, wpk, 0, nullptr, bias, out, B, relu, stream);
}

int mz_conv3x3_bf16_rot_sel(const void* obs_rot, const void* wpk_all, int64_t copy_elems, const int32_t* newest,
                            const float* bias, void* out, int32_t B, int32_t relu, mz_stream_t stream) {
  MZ_REQUIRE(newest && copy_elems > 0, "mz_conv3x3_bf16_rot_sel: null index or empty copies");
  return conv3x3_rot_any<__bf16>("mz_conv3x3_bf16_rot_sel", obs_rot, wpk_all, copy_elems, newest, bias, out, B, relu, stream);
}

int mz_conv3x3_f16_rot(const void* obs_rot, const void* wpk, const float* bias, void* out, int32_t B, int32_t relu,
                       mz_stream_t stream) {
  return conv3x3_rot_any<_Float16>("mz_conv3x3_f16_rot", obs_rot, wpk, 0, nullptr, bias, out, B, relu, stream);
}

int mz_conv3x3_f16_rot_sel(const void* obs_rot, const void* wpk_all, int64_t copy_elems, const int32_t* newest,
                           const float* bias, void* out, int32_t B, int32_t relu, mz_stream_t stream) {
  MZ_REQUIRE(newest && copy_elems > 0, "mz_conv3x3_f16_rot_sel: null index or empty copies");
  return conv3x3_rot_any<_Float16>("mz_conv3x3_f16_rot_sel", obs_rot, wpk_all, copy_elems, newest, bias, out, B, relu, stream);
}

int mz_conv_chain_bf16(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer* layers,
                       int32_t n_layers, void* out, const float* action_vec, const int32_t* actions,
                       float* mean_out, int32_t B, mz_stream_t stream) {
  if (int rc = check_chain_args("mz_conv_chain_bf16", in, layers, n_layers, out, action_vec, actions, B)) return rc;
  return mz::conv_chain_bf16(in, in_slot, slot_stride, layers, n_layers, out, action_vec, actions, mean_out, B,
                             mz::as_stream(stream));
}

int mz_conv_chain_f16(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer* layers,
                      int32_t n_layers, void* out, const float* action_vec, const int32_t* actions,
                      float* mean_out, int32_t B, mz_stream_t stream) {
  if (int rc = check_chain_args("mz_conv_chain_f16", in, layers, n_layers, out, action_vec, actions, B)) return rc;
  return mz::conv_chain_f16(in, in_slot, slot_stride, layers, n_layers, out, action_vec, actions, mean_out, B,
                            mz::as_stream(stream));
}

}  // extern "C"
