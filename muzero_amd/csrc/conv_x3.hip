// float32-accurate 3x3 convolutions on the matrix cores.  "bf16x3": every float32 value
// x travels as a pair of bf16 values (hi, lo) with hi = bf16(x), lo = bf16(x - hi), which
// carries 16-17 significant bits, and a product w * x is formed as
//     w_hi * x_hi + w_lo * x_hi + w_hi * x_lo
// by three v_mfma_f32_16x16x32_bf16 into one float32 accumulator (the dropped lo * lo term and
// the rounding of the lo parts are ~2^-17 relative).  This is the precision in which the
// networks meet the reference's float32 outputs to 1e-3 (model.py:332-348 on the torch CPU
// path); the matrix pipe runs three times the bf16 kernels' work.
//
// "f16x2" (template arguments T = _Float16, WLO = false): the same pairs in IEEE half (hi carries 11 bits,
// hi + lo 22) with the WEIGHTS rounded to half once, w * x = w_hi * x_hi + w_hi * x_lo - two MFMAs per product.
// What is dropped is the weights' own rounding (2^-12 relative per weight, independent over the 2304 terms of a
// sum) instead of a cross term; the activations, whose rounding would otherwise compound through 25 layers, stay
// 22 bits wide.  Meets the same 1e-3 bound at 2/3 of the matrix work.
//
// A split tensor is stored component-major per image: [image][2 = hi, lo][pixels][channels].
// Weights are two fragment arrays in the packing of mz_conv3x3_bf16 (hi and lo parts).
// The kernels are the bf16 ones' designs (conv_bf16.hip, conv_trunk.hip) with the second
// component in the place of the second image / a second LDS window.
#include "common.h"
#include "conv_common.h"

namespace {

template <typename T>
__device__ __forceinline__ void split4(float v0, float v1, float v2, float v3, vec4<T>& hi, vec4<T>& lo) {
  hi = to_op4<T>(v0, v1, v2, v3);
  lo = vec4<T>{(T)(v0 - (float)hi[0]), (T)(v1 - (float)hi[1]), (T)(v2 - (float)hi[2]), (T)(v3 - (float)hi[3])};
}

// ---------------------------------------------------------------------------
// Patch kernel for the four non-trunk layer shapes (cf. conv3x3_bf16_kernel)
// ---------------------------------------------------------------------------
template <typename T>
struct X3Args {
  const T* in;
  const T* wpk_hi;
  const T* wpk_lo;   // unused when the kernel forms two products
  const float* bias;
  const T* residual;
  T* out;
  int relu;
};

template <typename T, class C>
__device__ __forceinline__ void stage_window(char* lds, const T* src, int cc, int iy0, int ix0, int tid) {
  typedef vec8<T> bf16x8;   // eight operands of format T (the name predates the second format)
  constexpr int UPP = C::CINC / 8;
  constexpr int UNITS = C::IH * C::IW * UPP;
  constexpr int ROUNDS = (UNITS + C::THREADS - 1) / C::THREADS;
  // loads in straight-line code (clamped address, zeroed afterwards): a load inside a branch is waited for
  // where the branch rejoins, one round trip per round
  bf16x8 v[ROUNDS];
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    int u = tid + i * C::THREADS;
    u = u < UNITS ? u : UNITS - 1;
    const int pix = u / UPP, cu = u % UPP;
    int iy = iy0 + pix / C::IW, ix = ix0 + pix % C::IW;
    iy = iy < 0 ? 0 : (iy < C::HIN ? iy : C::HIN - 1);
    ix = ix < 0 ? 0 : (ix < C::WIN ? ix : C::WIN - 1);
    v[i] = *reinterpret_cast<const bf16x8*>(src + ((long)iy * C::WIN + ix) * C::CIN + cc * C::CINC + cu * 8);
  }
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    const int u = tid + i * C::THREADS;
    if (u < UNITS) {
      const int pix = u / UPP, cu = u % UPP;
      const int iy = iy0 + pix / C::IW, ix = ix0 + pix % C::IW;
      const bool inside = iy >= 0 && iy < C::HIN && ix >= 0 && ix < C::WIN;
      *reinterpret_cast<bf16x8*>(lds + pix * C::ROWB + cu * 16) = inside ? v[i] : zero8<T>();
    }
  }
}

// WLO: form w_lo * x_hi too (three products); without it the weights are hi only (two products)
template <typename T, bool WLO, class C>
__global__ __launch_bounds__(C::THREADS, 1) void conv3x3_x3_kernel(X3Args<T> a) {
  typedef vec8<T> bf16x8;   // eight / four operands of format T (the names predate the second format)
  typedef vec4<T> bf16x4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int MT = C::MT, WN = C::WN, KC = C::KC;
  constexpr int WIN_B = C::LDS_BYTES;  // one component's window; the lo window follows the hi window
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / (C::PY * C::PX), patch = blockIdx.x % (C::PY * C::PX);
  const int oy0 = (patch / C::PX) * C::TH, ox0 = (patch % C::PX) * C::TW;
  constexpr long IMG_IN = (long)C::HIN * C::WIN * C::CIN, IMG_OUT = (long)C::HO * C::WO * C::COUT;
  const T* src = a.in + (long)b * 2 * IMG_IN;
  const int iy0 = oy0 * C::S - 1, ix0 = ox0 * C::S - 1;
  const int r16 = lane & 15, kq = lane >> 4;

  f32x4 acc[MT][WN];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int WSTEP = C::NTILES * 1024;
  const __amdgpu_buffer_rsrc_t wrh = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a.wpk_hi), 0,
                                                                       (C::KSTEPS + 2) * WSTEP, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrl = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(WLO ? a.wpk_lo : a.wpk_hi), 0,
                                                                       (C::KSTEPS + 2) * WSTEP, 0x00020000);
  const int wlane = wave * (WN * 1024) + lane * 16;
  auto wfrag = [&](__amdgpu_buffer_rsrc_t r, int step_off, int nt) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, wlane + nt * 1024, step_off, 0));
  };
  bf16x8 wh[WN], wl[WN];
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    wh[nt] = wfrag(wrh, 0, nt);
    if constexpr (WLO) wl[nt] = wfrag(wrl, 0, nt);
  }
  int wp = WSTEP;

  typedef const __attribute__((address_space(3))) bf16x8* lds_frag_ptr;
  const int lds_base = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = mt * 16 + r16;
    if (m >= C::M) m = C::M - 1;
    const int py = m / C::TW, px = m % C::TW;
    abase[mt] = lds_base + (py * C::S * C::IW + px * C::S) * C::ROWB + kq * 16;
  }
  constexpr int NS = 9 * KC;
  auto step_offset = [](int st) {
    const int tap = st / KC, kc = st % KC;
    return ((tap / 3) * C::IW + tap % 3) * C::ROWB + kc * 64;
  };

  for (int cc = 0; cc < C::NCC; ++cc) {
    if (cc > 0) __syncthreads();
    stage_window<T, C>(lds, src, cc, iy0, ix0, tid);
    stage_window<T, C>(lds + WIN_B, src + IMG_IN, cc, iy0, ix0, tid);
    __syncthreads();
    bf16x8 af[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = *(lds_frag_ptr)(abase[mt]);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      bf16x8 w2h[WN], w2l[WN];
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) {
        w2h[nt] = wfrag(wrh, wp, nt);
        if constexpr (WLO) w2l[nt] = wfrag(wrl, wp, nt);
      }
      wp += WSTEP;
      // w_hi * x_hi, then (three products) w_lo * x_hi, after which the operand register takes x_lo, then
      // w_hi * x_lo, after which it takes the next step's x_hi
      if constexpr (WLO) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = mfma_16x16x32(wh[nt], af[mt], acc[mt][nt]);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = mfma_16x16x32(WLO ? wl[nt] : wh[nt], af[mt], acc[mt][nt]);
        af[mt] = *(lds_frag_ptr)(abase[mt] + (WIN_B + step_offset(st)));
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = mfma_16x16x32(wh[nt], af[mt], acc[mt][nt]);
        if (st + 1 < NS) af[mt] = *(lds_frag_ptr)(abase[mt] + step_offset(st + 1));
      }
      __builtin_amdgcn_sched_group_barrier(0x020, (WLO ? 2 : 1) * WN, 0);
      if constexpr (WLO) __builtin_amdgcn_sched_group_barrier(0x008, MT * WN, 0);
#pragma unroll
      for (int i = 0; i < 2 * MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, WN, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) {
        wh[nt] = w2h[nt];
        if constexpr (WLO) wl[nt] = w2l[nt];
      }
    }
  }

  // epilogue: bias, residual (hi + lo), ReLU in float32, split, parked in LDS as two tiles, stored as rows
  constexpr int OROW = C::COUT * 2 + 16;
  constexpr int TILE_B = C::M * OROW;
  static_assert(2 * TILE_B <= 2 * C::LDS_BYTES, "output tiles must fit in the input windows' LDS");
  __syncthreads();
  const T* res = a.residual ? a.residual + (long)b * 2 * IMG_OUT : nullptr;
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
        if (res) {
          const int py = m / C::TW, px = m % C::TW;
          const long o = (((long)oy0 + py) * C::WO + ox0 + px) * C::COUT + cbase;
          const bf16x4 rh = *reinterpret_cast<const bf16x4*>(res + o);
          const bf16x4 rl = *reinterpret_cast<const bf16x4*>(res + IMG_OUT + o);
          v0 += (float)rh[0] + (float)rl[0]; v1 += (float)rh[1] + (float)rl[1];
          v2 += (float)rh[2] + (float)rl[2]; v3 += (float)rh[3] + (float)rl[3];
        }
        if (a.relu) {
          v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
        }
        bf16x4 hi, lo;
        split4<T>(v0, v1, v2, v3, hi, lo);
        *reinterpret_cast<bf16x4*>(lds + m * OROW + cbase * 2) = hi;
        *reinterpret_cast<bf16x4*>(lds + TILE_B + m * OROW + cbase * 2) = lo;
      }
    }
  }
  __syncthreads();
  constexpr int UPR = C::COUT / 8;
  T* dst = a.out + (long)b * 2 * IMG_OUT;
  for (int u = tid; u < 2 * C::M * UPR; u += C::THREADS) {
    const int comp = u / (C::M * UPR), r = u % (C::M * UPR);
    const int m = r / UPR, c16 = r % UPR;
    const int py = m / C::TW, px = m % C::TW;
    const long o = (((long)oy0 + py) * C::WO + ox0 + px) * C::COUT + c16 * 8;
    *reinterpret_cast<bf16x8*>(dst + comp * IMG_OUT + o) =
        *reinterpret_cast<const bf16x8*>(lds + comp * TILE_B + m * OROW + c16 * 16);
  }
}

template <typename T, bool WLO, class C>
int launch_x3(const X3Args<T>& a, int B, hipStream_t stream) {
  static bool configured[mz::kMaxDevices] = {false};
  auto kern = conv3x3_x3_kernel<T, WLO, C>;
  constexpr int lds = 2 * C::LDS_BYTES;
  static_assert(lds <= 160 * 1024, "two input windows must fit in a CU's LDS");
  if (int rc = mz::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, configured, "conv3x3_x3")) return rc;
  hipLaunchKernelGGL(kern, dim3(B * C::PY * C::PX), dim3(C::THREADS), lds, stream, a);
  return mz::check_launch(WLO ? "mz_conv3x3_bf16x3" : "mz_conv3x3_f16x2");
}

template <class C>
bool matches(int H, int W, int Cin, int Cout, int stride) {
  return H == C::HIN && W == C::WIN && Cin == C::CIN && Cout == C::COUT && stride == C::S;
}

// ---------------------------------------------------------------------------
// Chain of 12x12 256->256 layers with one game's (hi, lo) pair resident in LDS
// (cf. conv_chain_pair_kernel: the lo component sits where that kernel keeps its second image)
// ---------------------------------------------------------------------------
template <typename T>
struct ChainX3Args {
  const T* in;
  const int* in_slot;
  int slot_stride;
  T* out;
  const float* avec;
  const int* actions;
  float* mean_out;
  int B;
  int n_layers;
  const T* wpk_hi[kMaxChain];
  const T* wpk_lo[kMaxChain];   // unused when the kernel forms two products
  const float* bias[kMaxChain];
  unsigned relu_mask;
  unsigned res_mask;
};

template <typename T, bool WLO>
__global__ __launch_bounds__(512, 2) void conv_chain_x3_kernel(ChainX3Args<T> a) {
  typedef vec8<T> bf16x8;   // eight / four operands of format T (the names predate the second format)
  typedef vec4<T> bf16x4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NW = 8, WN = 2;
  constexpr int WSTEP = 16 * 1024;
  constexpr int PSTRIDE = 2 * kPlaneB;  // LDS bytes between consecutive planes of one component
  constexpr int KOFF = PSTRIDE;
  constexpr long IMG = (long)kPix * kCh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;

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
  auto plane_of = [](int q) { return (q & 3) * 8 + (q >> 2); };

  {  // zero pixels of all 64 planes (never written afterwards)
    const bf16x8 z = zero8<T>();
    for (int s = tid; s < 2 * kPlanes * 16; s += NW * 64)
      *reinterpret_cast<bf16x8*>(lds + (s >> 4) * kPlaneB + (kPix + (s & 15)) * 16) = z;
  }

  const int wlane = wave * (WN * 1024) + lane * 16;
  auto wfrag = [&](__amdgpu_buffer_rsrc_t r, int step_off, int nt) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, wlane + nt * 1024, step_off, 0));
  };
  auto wdesc = [&](const T* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p), 0, (9 * kKC + 2) * WSTEP, 0x00020000);
  };
  const int pl = lane & 15, qq = lane >> 4;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const long img = a.in_slot ? (long)a.in_slot[b] * a.slot_stride + b : b;
    const T* src = a.in + img * 2 * IMG;
    const float* av = a.avec ? a.avec + (long)a.actions[b] * kCh : nullptr;

    __amdgpu_buffer_rsrc_t wrh = wdesc(a.wpk_hi[0]), wrl = wdesc(WLO ? a.wpk_lo[0] : a.wpk_hi[0]);
    bf16x8 wh[WN], wl[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
      wh[nt] = wfrag(wrh, 0, nt);
      if constexpr (WLO) wl[nt] = wfrag(wrl, 0, nt);
    }
    int wp = WSTEP;

    {  // both components -> LDS (planar); the action vector is added in float32.  All loads first, in
       // straight-line code: a load inside `if (av)` is waited for where the branch rejoins, once per round
      bf16x8 vh[72 / NW], vl[72 / NW];
#pragma unroll
      for (int j = 0; j < 72 / NW; ++j) {
        const int it = wave + NW * j;
        const int pix = (it >> 3) * 16 + pl, q = (it & 7) * 4 + qq;
        vh[j] = *reinterpret_cast<const bf16x8*>(src + pix * kCh + q * 8);
        vl[j] = *reinterpret_cast<const bf16x8*>(src + IMG + pix * kCh + q * 8);
      }
      if (av) {
#pragma unroll
        for (int j = 0; j < 72 / NW; ++j) {
          const int q = ((wave + NW * j) & 7) * 4 + qq;
          const float4 a0 = *reinterpret_cast<const float4*>(av + q * 8);
          const float4 a1 = *reinterpret_cast<const float4*>(av + q * 8 + 4);
          const float add[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = ((float)vh[j][e] + (float)vl[j][e]) + add[e];
            vh[j][e] = to_op<T>(v);
            vl[j][e] = (T)(v - (float)vh[j][e]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 72 / NW; ++j) {
        const int it = wave + NW * j;
        const int pix = (it >> 3) * 16 + pl, q = (it & 7) * 4 + qq;
        *reinterpret_cast<bf16x8*>(lds + (plane_of(q) * 2) * kPlaneB + pix * 16) = vh[j];
        *reinterpret_cast<bf16x8*>(lds + (plane_of(q) * 2 + 1) * kPlaneB + pix * 16) = vl[j];
      }
    }
    __syncthreads();

    f32x4 acc[kMT][WN];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {
      const bool last = l == a.n_layers - 1;
      const int ln = last ? l : l + 1;
      const __amdgpu_buffer_rsrc_t wrh_next = wdesc(a.wpk_hi[ln]), wrl_next = wdesc(WLO ? a.wpk_lo[ln] : a.wpk_hi[ln]);

      int aaddr[kMT];
      tap_addresses(0, aaddr);
      bf16x8 af[kMT];
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) af[mt] = *(lds_frag_ptr)(aaddr[mt]);

#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int kc = 0; kc < kKC; ++kc) {
          bf16x8 w2h[WN], w2l[WN];
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) {
            w2h[nt] = wfrag(wrh, wp, nt);
            if constexpr (WLO) w2l[nt] = wfrag(wrl, wp, nt);
          }
          wp += WSTEP;
          // w_hi * x_hi
          if constexpr (WLO) {
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
              for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = mfma_16x16x32(wh[nt], af[mt], acc[mt][nt]);
          }
          // w_lo * x_hi (three products) or w_hi * x_hi (two); each operand register is then refilled with x_lo
          // of this k-step
#pragma unroll
          for (int mt = 0; mt < kMT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = mfma_16x16x32(WLO ? wl[nt] : wh[nt], af[mt], acc[mt][nt]);
            af[mt] = *(lds_frag_ptr)(aaddr[mt] + (kPlaneB + kc * KOFF));
          }
          __builtin_amdgcn_sched_group_barrier(0x020, (WLO ? 2 : 1) * WN, 0);
          if constexpr (WLO) __builtin_amdgcn_sched_group_barrier(0x008, kMT * WN, 0);
#pragma unroll
          for (int i = 0; i < kMT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, WN, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (kc == kKC - 1) tap_addresses(tap < 8 ? tap + 1 : 8, aaddr);
          // w_hi * x_lo; refill with x_hi of the next k-step
#pragma unroll
          for (int mt = 0; mt < kMT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = mfma_16x16x32(wh[nt], af[mt], acc[mt][nt]);
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
            wh[nt] = w2h[nt];
            if constexpr (WLO) wl[nt] = w2l[nt];
          }
        }
      }

      float4 bv[WN];
#pragma unroll
      for (int nt = 0; nt < WN; ++nt)
        bv[nt] = *reinterpret_cast<const float4*>(a.bias[l] + (wave * WN + nt) * 16 + kq * 4);
      __syncthreads();  // every wavefront has read the layer's input: it may be overwritten
      const bool relu = (a.relu_mask >> l) & 1u;
      const bool keep_input = !last && ((a.res_mask >> (l + 1)) & 1u);  // the next layer adds this layer's input
      {  // not guarded by `!last` (the last layer requests its own first fragments again, unused): loads in a
         // branch cannot be counted, and the epilogue's first use of the bias would wait for them too
        wrh = wrh_next;
        wrl = wrl_next;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
          wh[nt] = wfrag(wrh, 0, nt);
          if constexpr (WLO) wl[nt] = wfrag(wrl, 0, nt);
        }
        wp = WSTEP;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) asm volatile("" ::"v"(bv[nt].x));
      }
      const float floor_v = relu ? 0.f : -__builtin_inff();
      const bool want_mean = last && a.mean_out;
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) {
        const int cbase = (wave * WN + nt) * 16 + kq * 4;
        char* plane = lds + (plane_of(cbase >> 3) * 2) * kPlaneB + (cbase & 4) * 2;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt) {
          char* slot = plane + (mt * 16 + r16) * 16;
          const float v0 = fmaxf(acc[mt][nt][0] + bv[nt].x, floor_v), v1 = fmaxf(acc[mt][nt][1] + bv[nt].y, floor_v);
          const float v2 = fmaxf(acc[mt][nt][2] + bv[nt].z, floor_v), v3 = fmaxf(acc[mt][nt][3] + bv[nt].w, floor_v);
          // the next layer's starting value: this layer's input (hi + lo, exact in float32) where the
          // next layer adds it, else zero
          if (keep_input) {
            const bf16x4 rh = *reinterpret_cast<const bf16x4*>(slot);
            const bf16x4 rl = *reinterpret_cast<const bf16x4*>(slot + kPlaneB);
            acc[mt][nt] = f32x4{(float)rh[0] + (float)rl[0], (float)rh[1] + (float)rl[1], (float)rh[2] + (float)rl[2],
                                (float)rh[3] + (float)rl[3]};
          } else {
            acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          s0 += v0; s1 += v1; s2 += v2; s3 += v3;
          bf16x4 hi, lo;
          split4<T>(v0, v1, v2, v3, hi, lo);
          *reinterpret_cast<bf16x4*>(slot) = hi;
          *reinterpret_cast<bf16x4*>(slot + kPlaneB) = lo;
        }
        if (want_mean) {
          s0 = row16_sum(s0); s1 = row16_sum(s1); s2 = row16_sum(s2); s3 = row16_sum(s3);
          if (r16 == 0) {
            const float inv = 1.0f / (float)kPix;
            *reinterpret_cast<float4*>(a.mean_out + (long)b * kCh + cbase) =
                make_float4(s0 * inv, s1 * inv, s2 * inv, s3 * inv);
          }
        }
      }
      __syncthreads();
    }

    T* dst = a.out + (long)b * 2 * IMG;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 72 / NW; ++j) {
        const int it = wave + NW * j;
        const int pix = (it >> 3) * 16 + pl, q = (it & 7) * 4 + qq;
        *reinterpret_cast<bf16x8*>(dst + i * IMG + pix * kCh + q * 8) =
            *reinterpret_cast<const bf16x8*>(lds + (plane_of(q) * 2 + i) * kPlaneB + pix * 16);
      }
    __syncthreads();  // the next game overwrites these
  }
}

// float32 [n] <-> split bf16 pairs laid out [n / per_image][2][per_image]
template <typename T>
__global__ void split_kernel(const float* __restrict__ x, T* __restrict__ out, long n, long per_image) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long img = i / per_image, r = i - img * per_image;
  const float v = x[i];
  const T hi = to_op<T>(v);
  out[img * 2 * per_image + r] = hi;
  out[img * 2 * per_image + per_image + r] = (T)(v - (float)hi);
}
template <typename T>
__global__ void join_kernel(const T* __restrict__ x, float* __restrict__ out, long n, long per_image) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long img = i / per_image, r = i - img * per_image;
  out[i] = (float)x[img * 2 * per_image + r] + (float)x[img * 2 * per_image + per_image + r];
}

template <typename T, bool WLO>
int conv3x3_split_any(const char* what, const void* in, const void* wpk_hi, const void* wpk_lo, const float* bias,
                      const void* residual, void* out, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                      int32_t stride, int32_t relu, mz_stream_t stream) {
  MZ_REQUIRE(in && wpk_hi && (wpk_lo || !WLO) && bias && out && B > 0, "%s: null tensor or empty batch", what);
  X3Args<T> a{static_cast<const T*>(in), static_cast<const T*>(wpk_hi), static_cast<const T*>(wpk_lo), bias,
              static_cast<const T*>(residual), static_cast<T*>(out), relu};
  hipStream_t s = mz::as_stream(stream);
  if (matches<ConvStem>(H, W, Cin, Cout, stride)) return launch_x3<T, WLO, ConvStem>(a, B, s);
  if (matches<ConvDown0>(H, W, Cin, Cout, stride)) return launch_x3<T, WLO, ConvDown0>(a, B, s);
  if (matches<ConvMid>(H, W, Cin, Cout, stride)) return launch_x3<T, WLO, ConvMid>(a, B, s);
  if (matches<ConvDown1>(H, W, Cin, Cout, stride)) return launch_x3<T, WLO, ConvDown1>(a, B, s);
  return mz::fail(MZ_E_ARG, "%s: no kernel for %dx%d, %d->%d channels, stride %d (the 12x12 256->256 layers go "
                  "through the chain entry point)", what, H, W, Cin, Cout, stride);
}

// layers: mz_chain_layer_x3 (WLO) or mz_chain_layer (two products: one weight array per layer)
template <typename T, bool WLO, typename Layer>
int conv_chain_split_any(const char* what, const void* in, const int32_t* in_slot, int32_t slot_stride, const Layer* layers,
                         int32_t n_layers, void* out, const float* action_vec, const int32_t* actions, float* mean_out,
                         int32_t B, mz_stream_t stream) {
  MZ_REQUIRE(in && layers && out && B > 0, "%s: null tensor or empty batch", what);
  MZ_REQUIRE(n_layers >= 1 && n_layers <= kMaxChain, "%s: n_layers must be 1..20 (got %d)", what, n_layers);
  MZ_REQUIRE((action_vec == nullptr) == (actions == nullptr), "%s: action_vec and actions go together", what);
  MZ_REQUIRE(!layers[0].add_residual, "%s: the first layer has no residual source", what);
  ChainX3Args<T> a{};
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
    if constexpr (WLO) {
      MZ_REQUIRE(layers[l].wpk_hi && layers[l].wpk_lo && layers[l].bias, "%s: layer %d is null", what, l);
      a.wpk_hi[l] = static_cast<const T*>(layers[l].wpk_hi);
      a.wpk_lo[l] = static_cast<const T*>(layers[l].wpk_lo);
    } else {
      MZ_REQUIRE(layers[l].wpk && layers[l].bias, "%s: layer %d is null", what, l);
      a.wpk_hi[l] = static_cast<const T*>(layers[l].wpk);
    }
    a.bias[l] = layers[l].bias;
    if (layers[l].relu) a.relu_mask |= 1u << l;
    if (layers[l].add_residual) a.res_mask |= 1u << l;
  }
  constexpr int lds = 2 * kLdsBytes;
  static bool configured[mz::kMaxDevices] = {false};
  auto kern = conv_chain_x3_kernel<T, WLO>;
  if (int rc = mz::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, configured, "conv_chain_x3")) return rc;
  const int n_cu = mz::compute_units();
  hipLaunchKernelGGL(kern, dim3(B < n_cu ? B : n_cu), dim3(512), lds, mz::as_stream(stream), a);
  return mz::check_launch(what);
}

template <typename T>
int split_any(const char* what, const float* x, void* out, int64_t n_images, int64_t per_image, mz_stream_t stream) {
  MZ_REQUIRE(x && out && n_images > 0 && per_image > 0, "%s: bad arguments", what);
  const long n = n_images * per_image;
  hipLaunchKernelGGL(split_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mz::as_stream(stream), x,
                     static_cast<T*>(out), n, (long)per_image);
  return mz::check_launch(what);
}

template <typename T>
int join_any(const char* what, const void* x, float* out, int64_t n_images, int64_t per_image, mz_stream_t stream) {
  MZ_REQUIRE(x && out && n_images > 0 && per_image > 0, "%s: bad arguments", what);
  const long n = n_images * per_image;
  hipLaunchKernelGGL(join_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mz::as_stream(stream),
                     static_cast<const T*>(x), out, n, (long)per_image);
  return mz::check_launch(what);
}

}  // namespace

extern "C" {

int mz_conv3x3_bf16x3(const void* in, const void* wpk_hi, const void* wpk_lo, const float* bias, const void* residual,
                      void* out, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t stride,
                      int32_t relu, mz_stream_t stream) {
  return conv3x3_split_any<__bf16, true>("mz_conv3x3_bf16x3", in, wpk_hi, wpk_lo, bias, residual, out, B, H, W, Cin, Cout,
                                         stride, relu, stream);
}

int mz_conv3x3_f16x2(const void* in, const void* wpk, const float* bias, const void* residual, void* out, int32_t B,
                     int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t stride, int32_t relu, mz_stream_t stream) {
  return conv3x3_split_any<_Float16, false>("mz_conv3x3_f16x2", in, wpk, nullptr, bias, residual, out, B, H, W, Cin, Cout,
                                            stride, relu, stream);
}

int mz_conv_chain_bf16x3(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer_x3* layers,
                         int32_t n_layers, void* out, const float* action_vec, const int32_t* actions,
                         float* mean_out, int32_t B, mz_stream_t stream) {
  return conv_chain_split_any<__bf16, true>("mz_conv_chain_bf16x3", in, in_slot, slot_stride, layers, n_layers, out,
                                            action_vec, actions, mean_out, B, stream);
}

int mz_conv_chain_f16x2(const void* in, const int32_t* in_slot, int32_t slot_stride, const mz_chain_layer* layers,
                        int32_t n_layers, void* out, const float* action_vec, const int32_t* actions, float* mean_out,
                        int32_t B, mz_stream_t stream) {
  return conv_chain_split_any<_Float16, false>("mz_conv_chain_f16x2", in, in_slot, slot_stride, layers, n_layers, out,
                                               action_vec, actions, mean_out, B, stream);
}

int mz_split_f32(const float* x, void* out, int64_t n_images, int64_t per_image, mz_stream_t stream) {
  return split_any<__bf16>("mz_split_f32", x, out, n_images, per_image, stream);
}
int mz_join_f32(const void* x, float* out, int64_t n_images, int64_t per_image, mz_stream_t stream) {
  return join_any<__bf16>("mz_join_f32", x, out, n_images, per_image, stream);
}
int mz_split_f32_half(const float* x, void* out, int64_t n_images, int64_t per_image, mz_stream_t stream) {
  return split_any<_Float16>("mz_split_f32_half", x, out, n_images, per_image, stream);
}
int mz_join_f32_half(const void* x, float* out, int64_t n_images, int64_t per_image, mz_stream_t stream) {
  return join_any<_Float16>("mz_join_f32_half", x, out, n_images, per_image, stream);
}

}  // extern "C"
