// Frame-stack preprocessing on the device: a ring of the last D frames per
// game (CustomFrameStack, train.py:22-63) and the normalised network input
// (prepare_observation, train.py:529-540).  Only the newest 96x96x3 frame of
// each game has to cross PCIe per environment step.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
  unsigned int u = __float_as_uint(f);  // finite, non-negative inputs only
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
// a pixel value in [0, 1] as a 16-bit pattern: bf16, or IEEE half when HALF (both round to nearest even)
template <bool HALF>
__device__ __forceinline__ unsigned short f32_to_16(float f) {
  if constexpr (HALF) return __builtin_bit_cast(unsigned short, (_Float16)f);
  else return f32_to_bf16_rne(f);
}
template <bool HALF>
__device__ __forceinline__ float f32_from_16(unsigned short h) {
  if constexpr (HALF) return (float)__builtin_bit_cast(_Float16, h);
  else return __uint_as_float((unsigned)h << 16);
}

// ring [B][D][frame_bytes]; frames [B][frame_bytes].  fill_all != 0 copies the
// frame into every slot (the reset rule, train.py:49-50).
__global__ void ring_push_kernel(unsigned char* __restrict__ ring, const unsigned char* __restrict__ frames,
                                 int D, long frame_bytes, int slot, int fill_all,
                                 const int* __restrict__ fill_mask) {
  const int b = blockIdx.y;
  if (fill_mask && fill_mask[b]) fill_all = 1;  // this game starts a new episode
  const uint4* src = reinterpret_cast<const uint4*>(frames + (long)b * frame_bytes);
  const long n16 = frame_bytes / 16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    if (fill_all) {
      for (int d = 0; d < D; ++d) reinterpret_cast<uint4*>(ring + ((long)b * D + d) * frame_bytes)[i] = v;
    } else {
      reinterpret_cast<uint4*>(ring + ((long)b * D + slot) * frame_bytes)[i] = v;
    }
  }
}

// One block = 64 pixels of one game: the 64*3 bytes of every frame are staged in
// LDS, then written as [pixel][3*D] with channel = age_rank*3 + rgb, oldest frame
// first (train.py:62-63), normalised by /255 in float32 (bit-identical to the
// reference's float64 divide + float32 cast for every byte value).
constexpr int kPix = 64;
template <typename OutT, bool HALF = false>
__global__ __launch_bounds__(256) void ring_to_obs_kernel(const unsigned char* __restrict__ ring,
                                                         OutT* __restrict__ out, int D, int HW, int newest) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned char* st = reinterpret_cast<unsigned char*>(smem);  // [D][kPix*3]
  const int b = blockIdx.y, p0 = blockIdx.x * kPix;
  const int np = HW - p0 < kPix ? HW - p0 : kPix;
  const int row = kPix * 3;
  for (int i = threadIdx.x; i < D * row; i += blockDim.x) {
    const int f = i / row, o = i % row;
    const int slot = (newest + 1 + f) % D;
    st[i] = o < np * 3 ? ring[(((long)b * D + slot) * HW + p0) * 3 + o] : 0;
  }
  __syncthreads();
  const int C = 3 * D;
  for (int i = threadIdx.x; i < np * C; i += blockDim.x) {
    const int px = i / C, c = i % C;
    const float v = (float)st[(c / 3) * row + px * 3 + (c % 3)] / 255.0f;
    const long o = ((long)b * HW + p0 + px) * C + c;
    if constexpr (sizeof(OutT) == 2) out[o] = f32_to_16<HALF>(v);
    else out[o] = v;
  }
}

// Fast path of the same conversion for 3*D a multiple of 8 and HW a multiple of 64: every
// global access is 16 bytes per lane (the byte-at-a-time version above is bound by the
// number of memory instructions, not by bytes).
// SPLIT: OutT is 16-bit and the output is the (hi, lo) pair layout [B][2][HW][C] of the float32-accurate
// precisions: hi = round16(v), lo = round16(v - hi), in bf16 or (HALF) IEEE half.
template <typename OutT, bool SPLIT = false, bool HALF = false>
__global__ __launch_bounds__(256) void ring_to_obs_vec_kernel(const unsigned char* __restrict__ ring,
                                                             OutT* __restrict__ out, int D, int HW, int newest) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned char* st = reinterpret_cast<unsigned char*>(smem);  // [D][192]
  const int b = blockIdx.y, p0 = blockIdx.x * kPix;
  constexpr int row = kPix * 3;  // 192 bytes = 12 x 16
  for (int u = threadIdx.x; u < D * 12; u += 256) {
    const int f = u / 12, part = u - f * 12;
    const int slot = (newest + 1 + f) % D;
    *reinterpret_cast<uint4*>(st + f * row + part * 16) =
        *reinterpret_cast<const uint4*>(ring + (((long)b * D + slot) * HW + p0) * 3 + part * 16);
  }
  __syncthreads();
  const int C = 3 * D, cpp = C / 8;  // 16-byte (bf16) output chunks per pixel
  for (int ch = threadIdx.x; ch < kPix * cpp; ch += 256) {
    const int px = ch / cpp, c0 = (ch - px * cpp) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j, f = c / 3, r = c - f * 3;
      v[j] = (float)st[f * row + px * 3 + r] / 255.0f;
    }
    const long o = ((long)b * HW + p0 + px) * C + c0;
    if constexpr (SPLIT) {
      unsigned short hi[8], lo[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        hi[j] = f32_to_16<HALF>(v[j]);
        const float rest = v[j] - f32_from_16<HALF>(hi[j]);      // may be negative: the generic conversion
        if constexpr (HALF) lo[j] = __builtin_bit_cast(unsigned short, (_Float16)rest);
        else lo[j] = __builtin_bit_cast(unsigned short, (__bf16)rest);
      }
      const long oh = ((long)b * 2 * HW + p0 + px) * C + c0, ol = oh + (long)HW * C;
      *reinterpret_cast<uint4*>(out + oh) = make_uint4(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16),
                                                        hi[4] | ((unsigned)hi[5] << 16), hi[6] | ((unsigned)hi[7] << 16));
      *reinterpret_cast<uint4*>(out + ol) = make_uint4(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16),
                                                        lo[4] | ((unsigned)lo[5] << 16), lo[6] | ((unsigned)lo[7] << 16));
    } else if constexpr (sizeof(OutT) == 2) {
      uint4 pk;
      pk.x = f32_to_16<HALF>(v[0]) | ((unsigned)f32_to_16<HALF>(v[1]) << 16);
      pk.y = f32_to_16<HALF>(v[2]) | ((unsigned)f32_to_16<HALF>(v[3]) << 16);
      pk.z = f32_to_16<HALF>(v[4]) | ((unsigned)f32_to_16<HALF>(v[5]) << 16);
      pk.w = f32_to_16<HALF>(v[6]) | ((unsigned)f32_to_16<HALF>(v[7]) << 16);
      *reinterpret_cast<uint4*>(out + o) = pk;
    } else {
      *reinterpret_cast<float4*>(out + o) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(out + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

// Rotating observation: obs [B][H][W/16][D][56] keeps, for every row y and 16-pixel tile of x, frame slot s as
// one run of 17 pixels x 3 values + 5 values of padding (112 bytes in bf16): the tile's 16 pixels preceded by
// the LAST pixel of the tile to its left (zero for the first tile of a row).  A stride-2 convolution's 8-output
// patch needs exactly these 17 columns, so its input window is D whole runs per row and no pixel has to be
// fetched from a neighbouring tile.  A new frame costs 3 (+ 3/16) values per pixel instead of rebuilding the
// stack (the consumer gathers a pixel's 3*D values from the D runs of its tile and undoes the rotation by
// permuting its input-channel weights).  fill_all writes the frame into every slot (the reset rule,
// train.py:49-50).
// One thread per 16 bytes of a run (7 threads per (row, tile) in bf16, 14 in float32): it converts the 8 (4)
// values of its unit - halo pixel, tile pixels or zero padding - and writes them with one 16-byte store, so
// the lanes of a wavefront cover whole runs with consecutive addresses.
constexpr int kRotRun = 56;
template <typename OutT, bool HALF = false>
__global__ __launch_bounds__(256) void obs_slot_write_kernel(const unsigned char* __restrict__ frames,
                                                            OutT* __restrict__ obs, int D, int HW, int W, int slot,
                                                            int fill_all, const int* __restrict__ fill_mask) {
  constexpr int V = 16 / sizeof(OutT);      // values per 16-byte unit
  constexpr int UPR = kRotRun / V;          // units per run
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = i / UPR, u = i - t * UPR;   // tile index (y, x/16) in memory order; unit within the run
  if (t >= HW / 16) return;
  if (fill_mask && fill_mask[b]) fill_all = 1;  // this game starts a new episode
  const unsigned char* f = frames + ((long)b * HW + (long)t * 16) * 3 - 3;  // byte e of the run's 51 values
  const bool first_in_row = (t % (W / 16)) == 0;
  // the unit's bytes are loaded unconditionally (halo of a row's first tile and padding read a clamped, valid
  // byte and are zeroed afterwards): a load inside a branch is waited for before the next one is requested
  unsigned char raw[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    int e = u * V + j;
    e = e < 50 ? e : 50;
    e = first_in_row && e < 3 ? 3 : e;
    raw[j] = f[e];
  }
  OutT v[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int e = u * V + j;
    const float x = (e < 51 && !(first_in_row && e < 3)) ? (float)raw[j] / 255.0f : 0.f;
    if constexpr (sizeof(OutT) == 2) v[j] = f32_to_16<HALF>(x);
    else v[j] = x;
  }
  uint4 pk;
  if constexpr (sizeof(OutT) == 2) {
    pk.x = v[0] | ((unsigned)v[1] << 16); pk.y = v[2] | ((unsigned)v[3] << 16);
    pk.z = v[4] | ((unsigned)v[5] << 16); pk.w = v[6] | ((unsigned)v[7] << 16);
  } else {
    pk.x = __float_as_uint(v[0]); pk.y = __float_as_uint(v[1]); pk.z = __float_as_uint(v[2]); pk.w = __float_as_uint(v[3]);
  }
  OutT* o = obs + (((long)b * (HW / 16) + t) * D) * kRotRun + u * V;
  if (fill_all) {
    for (int d = 0; d < D; ++d) *reinterpret_cast<uint4*>(o + d * kRotRun) = pk;
  } else {
    *reinterpret_cast<uint4*>(o + slot * kRotRun) = pk;
  }
}

// uint8 NHWC [B][HW][C] -> float32 NCHW [B][C][HW], x / 255 (train.py:534-540)
__global__ void prepare_observation_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, int C,
                                           int HW) {
  __shared__ unsigned char t[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    t[r][tx] = (c < C && p < HW) ? in[((long)b * HW + p) * C + c] : 0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    if (c < C && p < HW) out[((long)b * C + c) * HW + p] = (float)t[tx][r] / 255.0f;
  }
}


// Stacked, prepared observations of n replay positions from device-resident episode frames: sample b is
// position q_pos[b] of game q_game[b]; its stack holds the D frames ending at that position, the first frame
// of the episode repeated in front (CustomFrameStack's reset rule, train.py:49-50), oldest first, as
// prepare_observation leaves it: float32 NCHW, x / 255.  One workgroup = 64 pixels of one frame of one sample.
__global__ __launch_bounds__(192) void gather_stacked_obs_kernel(const unsigned char* __restrict__ frames,
                                                                const long* __restrict__ frame_off,
                                                                const int* __restrict__ q_game,
                                                                const int* __restrict__ q_pos, int D, int HW,
                                                                float* __restrict__ out) {
  const int b = blockIdx.z, k = blockIdx.y, p0 = blockIdx.x * 64;
  const int c = threadIdx.x / 64, px = threadIdx.x % 64;
  if (p0 + px >= HW) return;
  int pos = q_pos[b] - D + 1 + k;
  pos = pos < 0 ? 0 : pos;
  const unsigned char* f = frames + (frame_off[q_game[b]] + pos) * (long)HW * 3;
  out[((long)b * 3 * D + k * 3 + c) * HW + p0 + px] = (float)f[(p0 + px) * 3 + c] / 255.0f;
}

}  // namespace

extern "C" {

int mz_ring_push(uint8_t* ring, const uint8_t* frames, int32_t B, int32_t D, int64_t frame_bytes, int32_t slot,
                 int32_t fill_all, const int32_t* fill_mask, mz_stream_t stream) {
  MZ_REQUIRE(ring && frames && B > 0 && D > 0 && B <= 65535, "mz_ring_push: bad arguments");
  MZ_REQUIRE(frame_bytes > 0 && frame_bytes % 16 == 0, "mz_ring_push: frame_bytes must be a multiple of 16");
  MZ_REQUIRE(slot >= 0 && slot < D, "mz_ring_push: slot %d outside ring of depth %d", slot, D);
  const int blocks = (int)((frame_bytes / 16 + 255) / 256);
  hipLaunchKernelGGL(ring_push_kernel, dim3(blocks, B), dim3(256), 0, mz::as_stream(stream), ring, frames, D,
                     (long)frame_bytes, slot, fill_all, fill_mask);
  return mz::check_launch("mz_ring_push");
}

int mz_ring_to_obs(const uint8_t* ring, void* out, int32_t out_is_bf16, int32_t B, int32_t D, int32_t HW,
                   int32_t newest_slot, mz_stream_t stream) {
  MZ_REQUIRE(ring && out && B > 0 && D > 0 && HW > 0 && B <= 65535, "mz_ring_to_obs: bad arguments");
  MZ_REQUIRE(newest_slot >= 0 && newest_slot < D, "mz_ring_to_obs: newest_slot outside the ring");
  const size_t lds = (size_t)D * kPix * 3;
  MZ_REQUIRE(lds <= 64 * 1024, "mz_ring_to_obs: ring depth %d too large", D);
  const dim3 grid((HW + kPix - 1) / kPix, B);
  MZ_REQUIRE(out_is_bf16 >= 0 && out_is_bf16 <= 4,
             "mz_ring_to_obs: output format must be 0 (float32), 1 (bf16), 2 (bf16 pairs), 3 (half) or 4 (half pairs)");
  if (out_is_bf16 == 4) {
    MZ_REQUIRE((3 * D) % 8 == 0 && HW % kPix == 0, "mz_ring_to_obs: the pair format needs 3*D %% 8 == 0 and HW %% 64 == 0");
    hipLaunchKernelGGL((ring_to_obs_vec_kernel<unsigned short, true, true>), grid, dim3(256), lds, mz::as_stream(stream), ring,
                       static_cast<unsigned short*>(out), D, HW, newest_slot);
    return mz::check_launch("mz_ring_to_obs");
  }
  if (out_is_bf16 == 3) {
    if ((3 * D) % 8 == 0 && HW % kPix == 0)
      hipLaunchKernelGGL((ring_to_obs_vec_kernel<unsigned short, false, true>), grid, dim3(256), lds, mz::as_stream(stream),
                         ring, static_cast<unsigned short*>(out), D, HW, newest_slot);
    else
      hipLaunchKernelGGL((ring_to_obs_kernel<unsigned short, true>), grid, dim3(256), lds, mz::as_stream(stream), ring,
                         static_cast<unsigned short*>(out), D, HW, newest_slot);
    return mz::check_launch("mz_ring_to_obs");
  }
  if (out_is_bf16 == 2) {
    MZ_REQUIRE((3 * D) % 8 == 0 && HW % kPix == 0, "mz_ring_to_obs: the pair format needs 3*D %% 8 == 0 and HW %% 64 == 0");
    hipLaunchKernelGGL((ring_to_obs_vec_kernel<unsigned short, true>), grid, dim3(256), lds, mz::as_stream(stream), ring,
                       static_cast<unsigned short*>(out), D, HW, newest_slot);
    return mz::check_launch("mz_ring_to_obs");
  }
  if ((3 * D) % 8 == 0 && HW % kPix == 0) {
    if (out_is_bf16)
      hipLaunchKernelGGL(ring_to_obs_vec_kernel<unsigned short>, grid, dim3(256), lds, mz::as_stream(stream), ring,
                         static_cast<unsigned short*>(out), D, HW, newest_slot);
    else
      hipLaunchKernelGGL(ring_to_obs_vec_kernel<float>, grid, dim3(256), lds, mz::as_stream(stream), ring,
                         static_cast<float*>(out), D, HW, newest_slot);
    return mz::check_launch("mz_ring_to_obs");
  }
  if (out_is_bf16)
    hipLaunchKernelGGL(ring_to_obs_kernel<unsigned short>, grid, dim3(256), lds, mz::as_stream(stream), ring,
                       static_cast<unsigned short*>(out), D, HW, newest_slot);
  else
    hipLaunchKernelGGL(ring_to_obs_kernel<float>, grid, dim3(256), lds, mz::as_stream(stream), ring,
                       static_cast<float*>(out), D, HW, newest_slot);
  return mz::check_launch("mz_ring_to_obs");
}

int mz_obs_slot_write(const uint8_t* frames, void* obs, int32_t out_is_bf16, int32_t B, int32_t D, int32_t H,
                      int32_t W, int32_t slot, int32_t fill_all, const int32_t* fill_mask, mz_stream_t stream) {
  MZ_REQUIRE(frames && obs && B > 0 && D > 0 && H > 0 && W > 0 && B <= 65535, "mz_obs_slot_write: bad arguments");
  MZ_REQUIRE(W % 16 == 0, "mz_obs_slot_write: the image width must be a multiple of 16 (W = %d)", W);
  MZ_REQUIRE(slot >= 0 && slot < D, "mz_obs_slot_write: slot %d outside a stack of depth %d", slot, D);
  const int HW = H * W;
  MZ_REQUIRE((uintptr_t)obs % 16 == 0, "mz_obs_slot_write: obs must be 16-byte aligned");
  MZ_REQUIRE(out_is_bf16 == 0 || out_is_bf16 == 1 || out_is_bf16 == 3,
             "mz_obs_slot_write: output format must be 0 (float32), 1 (bf16) or 3 (half)");
  const int units = (HW / 16) * (out_is_bf16 ? 7 : 14);   // 16-byte units of one slot's runs per game
  const dim3 grid((units + 255) / 256, B);
  if (out_is_bf16 == 3)
    hipLaunchKernelGGL((obs_slot_write_kernel<unsigned short, true>), grid, dim3(256), 0, mz::as_stream(stream), frames,
                       static_cast<unsigned short*>(obs), D, HW, W, slot, fill_all, fill_mask);
  else if (out_is_bf16)
    hipLaunchKernelGGL(obs_slot_write_kernel<unsigned short>, grid, dim3(256), 0, mz::as_stream(stream), frames,
                       static_cast<unsigned short*>(obs), D, HW, W, slot, fill_all, fill_mask);
  else
    hipLaunchKernelGGL(obs_slot_write_kernel<float>, grid, dim3(256), 0, mz::as_stream(stream), frames,
                       static_cast<float*>(obs), D, HW, W, slot, fill_all, fill_mask);
  return mz::check_launch("mz_obs_slot_write");
}

int mz_gather_stacked_obs(const uint8_t* frames, const int64_t* frame_off, const int32_t* q_game, const int32_t* q_pos,
                          int32_t n, int32_t D, int32_t HW, float* out_nchw, mz_stream_t stream) {
  MZ_REQUIRE(frames && frame_off && q_game && q_pos && out_nchw && n > 0 && D > 0 && HW > 0 && n <= 65535 && D <= 65535,
             "mz_gather_stacked_obs: bad arguments");
  hipLaunchKernelGGL(gather_stacked_obs_kernel, dim3((HW + 63) / 64, D, n), dim3(192), 0, mz::as_stream(stream), frames,
                     reinterpret_cast<const long*>(frame_off), q_game, q_pos, D, HW, out_nchw);
  return mz::check_launch("mz_gather_stacked_obs");
}

int mz_prepare_observation(const uint8_t* obs_nhwc, float* out_nchw, int32_t B, int32_t C, int32_t HW,
                           mz_stream_t stream) {
  MZ_REQUIRE(obs_nhwc && out_nchw && B > 0 && C > 0 && HW > 0 && B <= 65535, "mz_prepare_observation: bad arguments");
  hipLaunchKernelGGL(prepare_observation_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0,
                     mz::as_stream(stream), obs_nhwc, out_nchw, C, HW);
  return mz::check_launch("mz_prepare_observation");
}

}  // extern "C"
