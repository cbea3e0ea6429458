// Data gradient of the two stride-2 3x3 convolutions of the representation network (DownsampleBlock.conv,
// model.py:165; 48x48x256 -> 24x24x128 and 24x24x128 -> 12x12x256) for the unrolled training step
// (compute_losses' backward pass, train.py:303-308):
//     dx[n][iy][ix][ci] = sum over (ky, kx) with iy + 1 - ky and ix + 1 - kx even, and over co, of
//                         gout[n][(iy + 1 - ky) / 2][(ix + 1 - kx) / 2][co] * w[co][ci][ky][kx]   (zero outside gout)
// bf16 operands, float32 accumulation on the matrix cores, dx rounded to bf16.
//
// Which taps reach an input pixel depends only on the parity of its coordinates, so dx splits into four classes
// (iy % 2, ix % 2), each a dense stride-1 convolution of gout with 1, 2, 2 or 4 of the nine taps:
//     dx[2a + py][2b + px] = sum over r in R(py), c in R(px) of gout[a + r - 1][b + c - 1] * w[..][py + 3 - 2r][px + 3 - 2c]
// with R(0) = {1}, R(1) = {1, 2} - no multiplication by the zeros a transposed convolution would insert.
// One workgroup computes one class of one patch of (a, b) with the scheme of the forward kernels
// (conv_bf16.hip): gout's window staged once in LDS and read per tap as MFMA operands, the wavefronts
// splitting dx's channels and streaming their weight fragments (packed per class by
// mz_weights_pack_dgrad_s2) from L2 into registers, the result parked in LDS and written as whole
// pixels to every second pixel of every second row of dx.
#include "common.h"
#include "conv_common.h"

namespace {

// (py, px) class c = 2 py + px uses taps[c] = |R(py)| * |R(px)| taps; its fragments start tap_base[c] taps into the array
__host__ __device__ constexpr int class_taps(int c) { return ((c >> 1) + 1) * ((c & 1) + 1); }
__host__ __device__ constexpr int class_tap_base(int c) { return c == 0 ? 0 : c == 1 ? 1 : c == 2 ? 3 : 5; }
// tap i of class c in window coordinates (r, col), rows first
__host__ __device__ constexpr int class_tap_r(int c, int i) { return 1 + ((c & 1) ? i / 2 : i); }
__host__ __device__ constexpr int class_tap_c(int c, int i) { return 1 + ((c & 1) ? i % 2 : 0); }

constexpr int kDgChunk = 128;   // channels of gout per staged window (and per run of the fragment order)

//                     CIN  CINC     COUT S  TH  TW HIN WIN waves/SIMD PF NW     (CIN = the layer's Cout, COUT = its Cin)
using DgDown0 = Cfg<128, kDgChunk, 256, 1, 8, 12, 24, 24, 2, 1, 4>;    // dx 48x48x256 from gout 24x24x128
using DgDown1 = Cfg<256, kDgChunk, 128, 1, 12, 12, 12, 12, 2, 3, 4>;   // dx 24x24x128 from gout 12x12x256

template <class C>
constexpr int dg_lds_bytes() {
  constexpr int out_tile = C::M * (C::COUT * 2 + 16);
  return C::LDS_BYTES > out_tile ? C::LDS_BYTES : out_tile;
}

// gout's window of channel chunk cc: every load in flight before the first LDS write waits (pixels outside
// gout load a clamped address and are zeroed afterwards - no load inside a branch)
template <class C>
__device__ __forceinline__ void stage_window(char* lds, const __bf16* src, int cc, int iy0, int ix0, int tid) {
  constexpr int UPP = C::CINC / 8;
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
      if (!(iy >= 0 && iy < C::HIN && ix >= 0 && ix < C::WIN)) w = zero8<__bf16>();
      *reinterpret_cast<bf16x8*>(lds + py * C::ROWP + px * C::ROWB + cu * 16) = w;
    }
  }
}

// the multiply-accumulate of one class over all channel chunks; CLS is a compile-time constant, so every LDS
// offset rides in the ds_read immediate and the fragment ring has fixed indices
template <class C, int CLS>
__device__ __forceinline__ void class_body(char* lds, const __bf16* src, const __bf16* wpk, int iy0, int ix0, int tid,
                                           f32x4 (&acc)[C::MT][C::WN]) {
  constexpr int MT = C::MT, WN = C::WN, KC = C::KC, NT = class_taps(CLS);
  constexpr int NS = NT * KC;  // k-steps per channel chunk
  constexpr int RING = C::PF + 1;
  static_assert(NS % RING == 0, "k-steps per channel chunk must be a multiple of the prefetch ring");
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  constexpr int WSTEP = C::NTILES * 1024;  // bytes per k-step
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(wpk) + (long)class_tap_base(CLS) * C::CIN * C::COUT, 0, (C::NCC * NS + C::PF) * WSTEP, 0x00020000);
  const int wlane = wave * (WN * 1024) + lane * 16;
  auto wfrag = [&](int step_off, int nt) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, wlane + nt * 1024, step_off, 0));
  };
  bf16x8 wq[RING][WN];
#pragma unroll
  for (int pf = 0; pf < C::PF; ++pf)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) wq[pf][nt] = wfrag(pf * WSTEP, nt);
  int wp = C::PF * WSTEP;

  typedef const __attribute__((address_space(3))) bf16x8* lds_frag_ptr;
  const int lds_base = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = mt * 16 + r16;
    if (m >= C::M) m = C::M - 1;
    abase[mt] = lds_base + (m / C::TW) * C::ROWP + (m % C::TW) * C::ROWB + kq * 16;
  }
  auto step_offset = [](int st) {
    const int tap = st / KC, kc = st % KC;
    return class_tap_r(CLS, tap) * C::ROWP + class_tap_c(CLS, tap) * C::ROWB + kc * 64;
  };
  for (int cc = 0; cc < C::NCC; ++cc) {
    if (cc > 0) __syncthreads();
    stage_window<C>(lds, src, cc, iy0, ix0, tid);
    __syncthreads();
    bf16x8 af[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = *(lds_frag_ptr)(abase[mt] + step_offset(0));
#pragma unroll
    for (int st = 0; st < NS; ++st) {
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) wq[(st + C::PF) % RING][nt] = wfrag(wp, nt);  // PF k-steps ahead
      wp += WSTEP;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = mfma_16x16x32(wq[st % RING][nt], af[mt], acc[mt][nt]);
        if (st + 1 < NS) af[mt] = *(lds_frag_ptr)(abase[mt] + step_offset(st + 1));
      }
    }
  }
}

template <class C>
__global__ __launch_bounds__(C::THREADS, C::MINW) void conv_dgrad_s2_kernel(const __bf16* __restrict__ gout,
                                                                          const __bf16* __restrict__ wpk,
                                                                          __bf16* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int MT = C::MT, WN = C::WN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the heaviest class first: blocks are handed out in order, and the four-tap class takes four times the one-tap class
  const int cls = 3 - (int)(blockIdx.x / (gridDim.x / 4));
  const int rest = blockIdx.x % (gridDim.x / 4);
  const int b = rest / (C::PY * C::PX), patch = rest % (C::PY * C::PX);
  const int oy0 = (patch / C::PX) * C::TH, ox0 = (patch % C::PX) * C::TW;
  const __bf16* src = gout + (long)b * C::HIN * C::WIN * C::CIN;
  const int r16 = lane & 15, kq = lane >> 4;

  f32x4 acc[MT][WN];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  switch (cls) {  // uniform in the workgroup
    case 0: class_body<C, 0>(lds, src, wpk, oy0 - 1, ox0 - 1, tid, acc); break;
    case 1: class_body<C, 1>(lds, src, wpk, oy0 - 1, ox0 - 1, tid, acc); break;
    case 2: class_body<C, 2>(lds, src, wpk, oy0 - 1, ox0 - 1, tid, acc); break;
    default: class_body<C, 3>(lds, src, wpk, oy0 - 1, ox0 - 1, tid, acc); break;
  }

  // a lane holds channels cbase .. cbase + 3 of pixel mt * 16 + r16: round to bf16, park the tile in LDS (the window
  // is dead by now), then write whole pixels of dx - class pixel (a, b) is dx pixel (2a + py, 2b + px)
  constexpr int OROW = C::COUT * 2 + 16;
  __syncthreads();
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int cbase = (wave * WN + nt) * 16 + kq * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 16 + r16;
      if (m < C::M)
        *reinterpret_cast<bf16x4*>(lds + m * OROW + cbase * 2) =
            bf16x4{(__bf16)acc[mt][nt][0], (__bf16)acc[mt][nt][1], (__bf16)acc[mt][nt][2], (__bf16)acc[mt][nt][3]};
    }
  }
  __syncthreads();
  constexpr int UPR = C::COUT / 8;
  const int py = cls >> 1, px = cls & 1;
  for (int u = tid; u < C::M * UPR; u += C::THREADS) {
    const int m = u / UPR, c16 = u % UPR;
    const int a = oy0 + m / C::TW, bb = ox0 + m % C::TW;
    const long o = (((long)b * (2 * C::HIN) + 2 * a + py) * (2 * C::WIN) + 2 * bb + px) * C::COUT + c16 * 8;
    *reinterpret_cast<bf16x8*>(dx + o) = *reinterpret_cast<const bf16x8*>(lds + m * OROW + c16 * 16);
  }
}

template <class C>
int dg_launch(const void* gout, const void* wpk, void* dx, int B, hipStream_t s) {
  static bool configured[mz::kMaxDevices] = {false};
  auto kern = conv_dgrad_s2_kernel<C>;
  if (int rc = mz::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), dg_lds_bytes<C>(), configured, "conv_dgrad_s2"))
    return rc;
  hipLaunchKernelGGL(kern, dim3(4 * B * C::PY * C::PX), dim3(C::THREADS), dg_lds_bytes<C>(), s,
                     static_cast<const __bf16*>(gout), static_cast<const __bf16*>(wpk), static_cast<__bf16*>(dx));
  return mz::check_launch("mz_conv_dgrad_s2_bf16");
}

// fragments of class c, chunk cc, tap i, k-step kc, n-tile nt: [lane][j] = w[co = cc*128 + kc*32 + 8*(lane/16) + j]
//                                                                        [ci = nt*16 + lane%16][ky][kx]
__global__ void pack_dgrad_s2_kernel(const float* __restrict__ w, int Cout, int Cin, __bf16* __restrict__ out) {
  const long unit = (long)blockIdx.x * blockDim.x + threadIdx.x;  // 8 consecutive packed elements
  const long units = (long)Cout * Cin * 9 / 8;
  const int NT = Cin / 16, KC = kDgChunk / 32, NCC = Cout / kDgChunk;
  if (unit >= units) {  // the k-steps of zeros behind the fragments (prefetch overrun)
    const long z = unit - units;
    if (z < 4L * NT * 64) *reinterpret_cast<bf16x8*>(out + (units + z) * 8) = zero8<__bf16>();
    return;
  }
  const int lane = (int)(unit & 63);
  long r = unit >> 6;
  const int nt = (int)(r % NT); r /= NT;
  const int kc = (int)(r % KC); r /= KC;
  // r = class_tap_base(c) * NCC + cc * taps(c) + i   (classes back to back, each [cc][tap])
  int c = 0;
  while (c < 3 && r >= (long)class_tap_base(c + 1) * NCC) ++c;
  r -= (long)class_tap_base(c) * NCC;
  const int i = (int)(r % class_taps(c)), cc = (int)(r / class_taps(c));
  const int py = c >> 1, px = c & 1;
  const int ky = py + 3 - 2 * class_tap_r(c, i), kx = px + 3 - 2 * class_tap_c(c, i);
  const int ci = nt * 16 + (lane & 15);
  const int cb = cc * kDgChunk + kc * 32 + 8 * (lane >> 4);
#pragma unroll
  for (int j = 0; j < 8; ++j) out[unit * 8 + j] = (__bf16)w[((long)(cb + j) * Cin + ci) * 9 + ky * 3 + kx];
}

}  // namespace

extern "C" {

int64_t mz_packed_dgrad_s2_elems(int32_t Cout, int32_t Cin) {
  if (Cout <= 0 || Cin <= 0) return 0;
  return (int64_t)Cout * Cin * 9 + 4 * (int64_t)(Cin / 16) * 512;
}

int mz_weights_pack_dgrad_s2(const float* w, int32_t Cout, int32_t Cin, void* out, mz_stream_t stream) {
  MZ_REQUIRE(w && out, "mz_weights_pack_dgrad_s2: null tensor");
  MZ_REQUIRE(Cout > 0 && Cout % kDgChunk == 0 && Cin > 0 && Cin % 16 == 0,
             "mz_weights_pack_dgrad_s2: Cout must be a multiple of %d and Cin of 16 (got %d, %d)", kDgChunk, Cout, Cin);
  const long threads = (long)Cout * Cin * 9 / 8 + 4L * (Cin / 16) * 64;
  hipLaunchKernelGGL(pack_dgrad_s2_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, mz::as_stream(stream), w,
                     Cout, Cin, static_cast<__bf16*>(out));
  return mz::check_launch("mz_weights_pack_dgrad_s2");
}

int mz_conv_dgrad_s2_bf16(const void* grad_out, const void* wpk, void* dx, int32_t B, int32_t H, int32_t W, int32_t Cin,
                          int32_t Cout, mz_stream_t stream) {
  MZ_REQUIRE(grad_out && wpk && dx && B > 0, "mz_conv_dgrad_s2_bf16: null tensor or empty batch");
  hipStream_t s = mz::as_stream(stream);
  if (H == 48 && W == 48 && Cin == 256 && Cout == 128) return dg_launch<DgDown0>(grad_out, wpk, dx, B, s);
  if (H == 24 && W == 24 && Cin == 128 && Cout == 256) return dg_launch<DgDown1>(grad_out, wpk, dx, B, s);
  return mz::fail(MZ_E_ARG, "mz_conv_dgrad_s2_bf16: no kernel for a %dx%d input, %d->%d channels (the reference "
                  "architecture's two stride-2 layers with an input gradient have one)", H, W, Cin, Cout);
}

}  // extern "C"
