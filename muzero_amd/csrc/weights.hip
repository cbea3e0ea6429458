// Weight hand-over on the device: from the float32 parameters of a state_dict, as they arrive in the
// flat broadcast buffer, to what the kernels read - BatchNorm folded into the convolution (inference
// form, running statistics: model.py:143-175 in eval mode), weights rounded to bf16 and laid out in
// MFMA fragment order (or as float32 [tap][Cin][Cout] for the validation kernels), optionally the
// residue w - bf16(w) for the bf16x3 precision, the 32 input-channel rotations of the first
// convolution, and the transposed / flipped form a data-gradient convolution needs.  Also the
// linear layers of the heads in their fragment orders.  HBM-bound (one pass over the parameters).
//
// The arithmetic equals the host's: the fold runs in double with the same operations
// (scale = gamma / sqrt(var + eps); w * scale; (b - mean) * scale + beta), the results are rounded to
// float32 and then to bf16 (nearest even), so device-packed weights equal host-packed ones bit for bit.
#pragma clang fp contract(off)

#include "common.h"

namespace {

struct PackArgs {
  const float* w;      // [Cout][Cin][3][3]
  const float* cbias;  // [Cout] or null
  const float* gamma;  // BatchNorm weight, bias, running mean, running variance: [Cout] each, or all null
  const float* beta;
  const float* mean;
  const float* var;
  double eps;
  int Cout, Cin, chunk;
  int mode;       // 0 bf16 fragments of w; 1 bf16 fragments of w - bf16(w); 2 float32 [tap][Cin][Cout];
                  // 3 IEEE-half fragments of w (same order as 0)
  int transpose;  // data-gradient form: out channel <-> in channel, taps mirrored
  int rot_depth;  // > 0: n_copies = rot_depth copies with the input channels rotated by frame slot
  void* out;
  long copy_elems;  // elements between consecutive copies
  float* bias_out;  // [Cout] folded bias (written by copy 0), or null
};

__device__ __forceinline__ float folded_weight(const PackArgs& a, int co, int ci, int tap) {
  // logical convolution weight W[co][ci][tap] of the layer this packing is for
  float raw;
  int bn_ch;
  if (a.transpose) {  // W'[co][ci][tap] = w[ci][co][8 - tap]; no BatchNorm in this form
    raw = a.w[((long)ci * a.Cin + co) * 9 + (8 - tap)];   // a.Cin == a.Cout is checked by the host
    bn_ch = -1;
  } else {
    raw = a.w[((long)co * a.Cin + ci) * 9 + tap];
    bn_ch = co;
  }
  if (a.gamma && bn_ch >= 0) {
    const double scale = (double)a.gamma[bn_ch] / sqrt((double)a.var[bn_ch] + a.eps);
    return (float)((double)raw * scale);
  }
  return raw;
}

// bf16 fragments of copy `copy`: thread `unit` writes 8 consecutive packed elements (or zeros of the padding)
__device__ __forceinline__ void pack_fragments(const PackArgs& a, int copy, long unit) {
  const int NT = a.Cout / 16, KC = a.chunk / 32;
  const long per_copy = (long)a.Cout * a.Cin * 9;
  const long units = per_copy / 8;
  __bf16* out = reinterpret_cast<__bf16*>(a.out) + copy * a.copy_elems;
  if (unit >= units) {  // the two k-steps of zeros behind the fragments (prefetch overrun)
    const long z = unit - units;
    if (z < 2L * NT * 64) {
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      *reinterpret_cast<bf16x8*>(out + per_copy + z * 8) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    return;
  }
  // unit = (((cc*9 + tap)*KC + kc)*NT + nt)*64 + lane
  const int lane = (int)(unit & 63);
  long r = unit >> 6;
  const int nt = (int)(r % NT); r /= NT;
  const int kc = (int)(r % KC); r /= KC;
  const int tap = (int)(r % 9);
  const int cc = (int)(r / 9);
  const int co = nt * 16 + (lane & 15);
  const int cbase = cc * a.chunk + kc * 32 + 8 * (lane >> 4);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int cj = cbase + j;
    int ci = cj;
    if (a.rot_depth > 0) ci = ((cj / 3 - copy - 1 + 2 * a.rot_depth) % a.rot_depth) * 3 + cj % 3;
    const float f = folded_weight(a, co, ci, tap);
    if (a.mode == 3) {   // round to nearest even; beyond the largest finite half: saturate
      reinterpret_cast<_Float16*>(out)[unit * 8 + j] = (_Float16)fminf(fmaxf(f, -65504.f), 65504.f);
    } else {
      const __bf16 hi = (__bf16)f;
      out[unit * 8 + j] = a.mode == 0 ? hi : (__bf16)(f - (float)hi);
    }
  }
}

// one thread per 8 consecutive packed elements (bf16 modes) or per element (float32 mode)
__global__ void pack_conv_kernel(PackArgs a) {
  const int copy = blockIdx.y;
  const long per_copy = (long)a.Cout * a.Cin * 9;
  if (a.mode == 2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_copy) return;
    const int co = (int)(i % a.Cout);
    const long r = i / a.Cout;
    const int cj = (int)(r % a.Cin), tap = (int)(r / a.Cin);
    int ci = cj;
    if (a.rot_depth > 0) ci = ((cj / 3 - copy - 1 + 2 * a.rot_depth) % a.rot_depth) * 3 + cj % 3;
    reinterpret_cast<float*>(a.out)[copy * a.copy_elems + i] = folded_weight(a, co, ci, tap);
  } else {
    pack_fragments(a, copy, (long)blockIdx.x * blockDim.x + threadIdx.x);
  }
  if (a.bias_out && copy == 0 && blockIdx.x == 0) {
    for (int c = threadIdx.x; c < a.Cout; c += blockDim.x) {
      float b = a.cbias ? a.cbias[c] : 0.f;
      if (a.gamma) {
        const double scale = (double)a.gamma[c] / sqrt((double)a.var[c] + a.eps);
        b = (float)(((double)b - (double)a.mean[c]) * scale + (double)a.beta[c]);
      }
      a.bias_out[c] = b;
    }
  }
}

// Several weights of one shape in one launch (plain bf16 fragments, optionally the transposed form): the
// training step repacks every native layer twice per update, and a launch each is all overhead
constexpr int kMaxPackMany = 64;
struct PackMany {
  const float* w[kMaxPackMany];
  void* out[kMaxPackMany];
  unsigned long long transposed;  // bit i: item i in the data-gradient form
  int Cout, Cin, chunk;
};

__global__ void pack_many_kernel(PackMany m) {
  const int item = blockIdx.y;
  PackArgs a{};
  a.w = m.w[item];
  a.Cout = m.Cout, a.Cin = m.Cin, a.chunk = m.chunk;
  a.transpose = (int)((m.transposed >> item) & 1ull);
  a.out = m.out[item];
  pack_fragments(a, 0, (long)blockIdx.x * blockDim.x + threadIdx.x);
}

// torch Linear weight [O][K] -> MFMA fragments, rows beyond O zero (cf. mz_heads)
__global__ void pack_linear_kernel(const float* __restrict__ w, const float* __restrict__ b, int O, int K, int bf16,
                                   void* __restrict__ wp, float* __restrict__ bp) {
  const int tiles = (O + 15) / 16;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)tiles * 16 * K;
  if (i < total) {
    if (bf16) {  // [(t*(K/32) + ks)*64 + lane][j] = W[t*16 + lane%16][32*ks + 8*(lane/16) + j]; bf16 == 2: IEEE half
      const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
      const long r = i >> 9;
      const int ks = (int)(r % (K / 32)), t = (int)(r / (K / 32));
      const int row = t * 16 + (lane & 15), col = 32 * ks + 8 * (lane >> 4) + j;
      const float x = row < O ? w[(long)row * K + col] : 0.f;
      if (bf16 == 2) reinterpret_cast<_Float16*>(wp)[i] = (_Float16)fminf(fmaxf(x, -65504.f), 65504.f);
      else reinterpret_cast<__bf16*>(wp)[i] = (__bf16)x;
    } else {     // [(t*(K/4) + ks)*64 + lane] = W[t*16 + lane%16][4*ks + lane/16]
      const int lane = (int)(i & 63);
      const long r = i >> 6;
      const int ks = (int)(r % (K / 4)), t = (int)(r / (K / 4));
      const int row = t * 16 + (lane & 15), col = 4 * ks + (lane >> 4);
      reinterpret_cast<float*>(wp)[i] = row < O ? w[(long)row * K + col] : 0.f;
    }
  }
  if (bp && i < tiles * 16) bp[i] = i < O ? b[i] : 0.f;
}

// dynamics.conv_action as a per-action vector: out[a][c] = W[c][a] + bias[c] (model.py:231-236)
__global__ void action_vec_kernel(const float* __restrict__ w, const float* __restrict__ b, int C, int A,
                                  float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A * C) return;
  const int a = i / C, c = i - a * C;
  out[i] = w[(long)c * A + a] + b[c];
}

}  // namespace

extern "C" {

int64_t mz_packed_conv_elems(int32_t Cout, int32_t Cin, int32_t mode) {
  if (Cout <= 0 || Cin <= 0) return 0;
  const int64_t n = (int64_t)Cout * Cin * 9;
  return mode == 2 ? n : n + 2 * (int64_t)(Cout / 16) * 512;
}

int mz_weights_fold_pack(const float* w, const float* conv_bias, const float* bn_weight, const float* bn_bias,
                         const float* bn_mean, const float* bn_var, double eps, int32_t Cout, int32_t Cin,
                         int32_t cin_chunk, int32_t mode, int32_t transpose, int32_t rot_depth, void* out,
                         float* bias_out, mz_stream_t stream) {
  MZ_REQUIRE(w && out && Cout > 0 && Cin > 0, "mz_weights_fold_pack: null tensor or empty shape");
  MZ_REQUIRE(mode >= 0 && mode <= 3,
             "mz_weights_fold_pack: mode must be 0 (bf16), 1 (bf16 residue), 2 (float32) or 3 (half)");
  const bool bn = bn_weight || bn_bias || bn_mean || bn_var;
  MZ_REQUIRE(!bn || (bn_weight && bn_bias && bn_mean && bn_var), "mz_weights_fold_pack: BatchNorm needs all four vectors");
  MZ_REQUIRE(!(transpose && (bn || Cin != Cout || rot_depth)),
             "mz_weights_fold_pack: the transposed form is for square layers without BatchNorm or rotation");
  MZ_REQUIRE(rot_depth == 0 || Cin == 3 * rot_depth, "mz_weights_fold_pack: rotation needs Cin == 3 * rot_depth");
  if (mode != 2) {
    MZ_REQUIRE(Cout % 16 == 0 && cin_chunk > 0 && cin_chunk % 32 == 0 && Cin % cin_chunk == 0,
               "mz_weights_fold_pack: fragments need Cout %% 16 == 0 and Cin a multiple of the chunk (%d, %d, %d)", Cout,
               Cin, cin_chunk);
  }
  PackArgs a{w, conv_bias, bn_weight, bn_bias, bn_mean, bn_var, eps, Cout, Cin, cin_chunk, mode, transpose, rot_depth,
             out, (long)mz_packed_conv_elems(Cout, Cin, mode), bias_out};
  const long n = (long)Cout * Cin * 9;
  const long threads = mode == 2 ? n : n / 8 + 2L * (Cout / 16) * 64;
  hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)((threads + 255) / 256), rot_depth > 0 ? rot_depth : 1), dim3(256),
                     0, mz::as_stream(stream), a);
  return mz::check_launch("mz_weights_fold_pack");
}

int mz_weights_pack_many(const float* const* w, void* const* out, const int32_t* transpose, int32_t n, int32_t Cout,
                         int32_t Cin, int32_t cin_chunk, mz_stream_t stream) {
  MZ_REQUIRE(w && out && n > 0 && Cout > 0 && Cin > 0, "mz_weights_pack_many: null table or empty shape");
  MZ_REQUIRE(Cout % 16 == 0 && cin_chunk > 0 && cin_chunk % 32 == 0 && Cin % cin_chunk == 0,
             "mz_weights_pack_many: fragments need Cout %% 16 == 0 and Cin a multiple of the chunk (%d, %d, %d)", Cout, Cin,
             cin_chunk);
  const long n_el = (long)Cout * Cin * 9;
  const long threads = n_el / 8 + 2L * (Cout / 16) * 64;
  for (int i0 = 0; i0 < n; i0 += kMaxPackMany) {
    const int cnt = n - i0 < kMaxPackMany ? n - i0 : kMaxPackMany;
    PackMany m{};
    m.Cout = Cout, m.Cin = Cin, m.chunk = cin_chunk;
    for (int i = 0; i < cnt; ++i) {
      MZ_REQUIRE(w[i0 + i] && out[i0 + i], "mz_weights_pack_many: item %d has a null pointer", i0 + i);
      MZ_REQUIRE(!(transpose && transpose[i0 + i]) || Cin == Cout, "mz_weights_pack_many: the transposed form is for square layers");
      m.w[i] = w[i0 + i];
      m.out[i] = out[i0 + i];
      if (transpose && transpose[i0 + i]) m.transposed |= 1ull << i;
    }
    hipLaunchKernelGGL(pack_many_kernel, dim3((unsigned)((threads + 255) / 256), cnt), dim3(256), 0, mz::as_stream(stream), m);
  }
  return mz::check_launch("mz_weights_pack_many");
}

int mz_pack_linear(const float* w, const float* bias, int32_t O, int32_t K, int32_t bf16, void* w_packed,
                   float* bias_padded, mz_stream_t stream) {
  MZ_REQUIRE(w && w_packed && O > 0 && K > 0, "mz_pack_linear: null tensor or empty shape");
  MZ_REQUIRE(bf16 >= 0 && bf16 <= 2, "mz_pack_linear: format must be 0 (float32), 1 (bf16) or 2 (half)");
  MZ_REQUIRE(K % (bf16 ? 32 : 4) == 0, "mz_pack_linear: K = %d must be a multiple of %d", K, bf16 ? 32 : 4);
  MZ_REQUIRE((bias != nullptr) == (bias_padded != nullptr), "mz_pack_linear: bias and bias_padded go together");
  const long total = (long)((O + 15) / 16) * 16 * K;
  hipLaunchKernelGGL(pack_linear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mz::as_stream(stream), w,
                     bias, O, K, bf16, w_packed, bias_padded);
  return mz::check_launch("mz_pack_linear");
}

int mz_action_vectors(const float* w, const float* bias, int32_t C, int32_t A, float* out, mz_stream_t stream) {
  MZ_REQUIRE(w && bias && out && C > 0 && A > 0, "mz_action_vectors: bad arguments");
  hipLaunchKernelGGL(action_vec_kernel, dim3((A * C + 255) / 256), dim3(256), 0, mz::as_stream(stream), w, bias, C, A,
                     out);
  return mz::check_launch("mz_action_vectors");
}

}  // extern "C"
