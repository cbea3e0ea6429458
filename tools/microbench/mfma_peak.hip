// Sustained MFMA rate of the chip under its own clock management: bare loops of
// v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 on register operands, and the chain
// kernel's instruction mix (one ds_read_b128 per two 16x16x32 MFMAs), on random or zero data.
// Prints TFLOP/s per variant; the number to hold a real kernel's rate against when the nominal
// peak (2.4 GHz x 1024 SIMDs) is out of reach because the chip lowers its clock under MFMA load.
//   hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak && ./mfma_peak [ms_per_variant [threads_per_CU]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// 18 accumulators of 16x16 (the chain kernel's 9 pixel tiles x 2 channel tiles)
template <bool LDS>
__global__ __launch_bounds__(512, 2) void loop16(const bf16x8* __restrict__ src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  bf16x8 w[2], a[9];
  for (int i = 0; i < 2; ++i) w[i] = src[(tid * 11 + i) & 4095];
  for (int i = 0; i < 9; ++i) a[i] = src[(tid * 11 + 2 + i) & 4095];
  if (LDS) {
    for (int i = tid; i < 9 * 1024 * 8 / 16; i += blockDim.x) *reinterpret_cast<bf16x8*>(lds + i * 16) = src[i & 4095];
    __syncthreads();
  }
  f32x4 acc[9][2];
  for (int m = 0; m < 9; ++m) for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  int off = lane * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 9; ++m) {
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[n], a[m], acc[m][n], 0, 0, 0);
      if (LDS) a[m] = *reinterpret_cast<const bf16x8*>(lds + off + m * 1024 + (it & 7) * 9216);
    }
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
  }
  float s = 0;
  for (int m = 0; m < 9; ++m) for (int n = 0; n < 2; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * 512 + tid] = s;
}

// the chain kernel's full k-step: 18 MFMAs, 9 LDS reads one step ahead, 2 weight-fragment buffer
// loads two steps ahead (fragments stream from a 1.2 MB array as one layer's do)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MT>
__global__ __launch_bounds__(512, 2) void loop16w(const bf16x8* __restrict__ src, const void* wts, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 9 * 1024 * 8 / 16; i += blockDim.x) *reinterpret_cast<bf16x8*>(lds + i * 16) = src[i & 4095];
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wts), 0, 74 * 16384, 0x00020000);
  const int wl = wave * 2048 + lane * 16;
  auto wf = [&](int off, int nt) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, wl + nt * 1024, off, 0)); };
  bf16x8 a[MT], wq[2][2];
  for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(lds + lane * 16 + (i % 9) * 1024);
  for (int p = 0; p < 2; ++p) for (int n = 0; n < 2; ++n) wq[p][n] = wf(p * 16384, n);
  f32x4 acc[MT][2];
  for (int m = 0; m < MT; ++m) for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  const int off = lane * 16;
  for (int it = 0; it < iters; it += 8) {
    int wo = 2 * 16384 + (it % 64) * 16384;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      bf16x8 w2[2];
      for (int n = 0; n < 2; ++n) w2[n] = wf(wo, n);
      wo += 16384;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[0][n], a[m], acc[m][n], 0, 0, 0);
        a[m] = *reinterpret_cast<const bf16x8*>(lds + off + (m % 9) * 1024 + k * 9216);
      }
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      for (int n = 0; n < 2; ++n) { wq[0][n] = wq[1][n]; wq[1][n] = w2[n]; }
    }
  }
  float s = 0;
  for (int m = 0; m < MT; ++m) for (int n = 0; n < 2; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * 512 + tid] = s;
}

// the same k-step with 64 output channels per wavefront: the two wavefronts of a SIMD share the
// weight fragments (4 loads each) and split the 9 pixel tiles MA / 9-MA: half the LDS reads per MFMA
template <int MA>
__global__ __launch_bounds__(512, 2) void loop16x(const bf16x8* __restrict__ src, const void* wts, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 9 * 1024 * 8 / 16; i += blockDim.x) *reinterpret_cast<bf16x8*>(lds + i * 16) = src[i & 4095];
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wts), 0, 74 * 16384, 0x00020000);
  const int wl = (wave & 3) * 4096 + lane * 16;
  auto wf = [&](int off, int nt) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, wl + nt * 1024, off, 0)); };
  auto body = [&](auto mtiles) {
    constexpr int M = decltype(mtiles)::value;
    const int m0 = M == MA ? 0 : MA;
    bf16x8 a[M], wq[2][4];
    for (int i = 0; i < M; ++i) a[i] = *reinterpret_cast<const bf16x8*>(lds + lane * 16 + (m0 + i) * 1024);
    for (int p = 0; p < 2; ++p) for (int n = 0; n < 4; ++n) wq[p][n] = wf(p * 16384, n);
    f32x4 acc[M][4];
    for (int m = 0; m < M; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
    const int off = lane * 16 + m0 * 1024;
    for (int it = 0; it < iters; it += 8) {
      int wo = 2 * 16384 + (it % 64) * 16384;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        bf16x8 w2[4];
        for (int n = 0; n < 4; ++n) w2[n] = wf(wo, n);
        wo += 16384;
#pragma unroll
        for (int m = 0; m < M; ++m) {
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[0][n], a[m], acc[m][n], 0, 0, 0);
          a[m] = *reinterpret_cast<const bf16x8*>(lds + off + m * 1024 + k * 9216);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
#pragma unroll
        for (int i = 0; i < M; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        for (int n = 0; n < 4; ++n) { wq[0][n] = wq[1][n]; wq[1][n] = w2[n]; }
      }
    }
    float s = 0;
    for (int m = 0; m < M; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 512 + tid] = s;
  };
  if (wave < 4) body(std::integral_constant<int, MA>{}); else body(std::integral_constant<int, 9 - MA>{});
}

// one wavefront per SIMD, 64 channels x 18 pixel tiles (two images) per wavefront: 4 weight loads and
// 18 LDS reads per 72 MFMAs - the two-image k-step with half its LDS reads
__global__ __launch_bounds__(256, 1) void loop16y(const bf16x8* __restrict__ src, const void* wts, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 9 * 1024 * 8 / 16; i += blockDim.x) *reinterpret_cast<bf16x8*>(lds + i * 16) = src[i & 4095];
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wts), 0, 74 * 16384, 0x00020000);
  const int wl = wave * 4096 + lane * 16;
  auto wf = [&](int off, int nt) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, wl + nt * 1024, off, 0)); };
  constexpr int M = 18;
  bf16x8 a[9], wq[2][4];
  for (int i = 0; i < 9; ++i) a[i] = *reinterpret_cast<const bf16x8*>(lds + lane * 16 + i * 1024);
  for (int p = 0; p < 2; ++p) for (int n = 0; n < 4; ++n) wq[p][n] = wf(p * 16384, n);
  f32x4 acc[M][4];
  for (int m = 0; m < M; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  const int off = lane * 16;
  for (int it = 0; it < iters; it += 8) {
    int wo = 2 * 16384 + (it % 64) * 16384;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      bf16x8 w2[4];
      for (int n = 0; n < 4; ++n) w2[n] = wf(wo, n);
      wo += 16384;
#pragma unroll
      for (int m = 0; m < M; ++m) {
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[0][n], a[m % 9], acc[m][n], 0, 0, 0);
        a[m % 9] = *reinterpret_cast<const bf16x8*>(lds + off + (m % 9) * 1024 + ((k + m / 9) & 7) * 9216);
      }
      __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      for (int n = 0; n < 4; ++n) { wq[0][n] = wq[1][n]; wq[1][n] = w2[n]; }
    }
  }
  float s = 0;
  for (int m = 0; m < M; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * 512 + tid] = s;
}

// the same k-step with the accumulators pinned by hand: 64 of the 72 accumulator tiles in AGPRs (all 256), 8 in
// VGPRs, every MFMA an inline-asm statement with the register class in its constraint, so that the register
// allocator cannot shuttle accumulators between the two files
__device__ __forceinline__ void mfma_a(f32x4& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
__device__ __forceinline__ void mfma_v(f32x4& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}
__global__ __launch_bounds__(256, 1) void loop16z(const bf16x8* __restrict__ src, const void* wts, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 9 * 1024 * 8 / 16; i += blockDim.x) *reinterpret_cast<bf16x8*>(lds + i * 16) = src[i & 4095];
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wts), 0, 74 * 16384, 0x00020000);
  const int wl = wave * 4096 + lane * 16;
  auto wf = [&](int off, int nt) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, wl + nt * 1024, off, 0)); };
  constexpr int M = 18;
  bf16x8 a[9], wq[2][4];
  for (int i = 0; i < 9; ++i) a[i] = *reinterpret_cast<const bf16x8*>(lds + lane * 16 + i * 1024);
  for (int p = 0; p < 2; ++p) for (int n = 0; n < 4; ++n) wq[p][n] = wf(p * 16384, n);
  f32x4 acc[M][4];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  const int off = lane * 16;
  for (int it = 0; it < iters; it += 8) {
    int wo = 2 * 16384 + (it % 64) * 16384;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      bf16x8 w2[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) w2[n] = wf(wo, n);
      wo += 16384;
#pragma unroll
      for (int m = 0; m < M; ++m) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if (m < 16) mfma_a(acc[m][n], wq[0][n], a[m % 9]); else mfma_v(acc[m][n], wq[0][n], a[m % 9]);
        }
        a[m % 9] = *reinterpret_cast<const bf16x8*>(lds + off + (m % 9) * 1024 + ((k + m / 9) & 7) * 9216);
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) { wq[0][n] = wq[1][n]; wq[1][n] = w2[n]; }
    }
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");  // the asm MFMAs' results are read below: no hazard tracking for inline asm
  float s = 0;
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * 512 + tid] = s;
}

// 4 accumulators of 32x32 (same accumulator registers per wave as 16 of 16x16)
__global__ __launch_bounds__(512, 2) void loop32(const bf16x8* __restrict__ src, float* out, int iters) {
  const int tid = threadIdx.x;
  bf16x8 w[2], a[2];
  for (int i = 0; i < 2; ++i) { w[i] = src[(tid * 7 + i) & 4095]; a[i] = src[(tid * 7 + 2 + i) & 4095]; }
  f32x16 acc[2][2];
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int j = 0; j < 16; ++j) acc[m][n][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[n], a[m], acc[m][n], 0, 0, 0);
  }
  float s = 0;
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int j = 0; j < 16; ++j) s += acc[m][n][j];
  out[blockIdx.x * 512 + tid] = s;
}

// the two-image k-step on 32x32x16 MFMAs: 32 channels x 288 pixels per wavefront as 9 accumulator tiles of
// 32 x 32; per 32 input channels the same 18 LDS reads and 2 weight loads, but 18 MFMA instructions instead of 36
__global__ __launch_bounds__(512, 2) void loop32w(const bf16x8* __restrict__ src, const void* wts, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 9 * 1024 * 8 / 16; i += blockDim.x) *reinterpret_cast<bf16x8*>(lds + i * 16) = src[i & 4095];
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wts), 0, 74 * 16384, 0x00020000);
  const int wl = wave * 2048 + lane * 16;
  auto wf = [&](int off, int nt) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, wl + nt * 1024, off, 0)); };
  constexpr int M = 9;
  bf16x8 a[M], wq[2][2];
  for (int i = 0; i < M; ++i) a[i] = *reinterpret_cast<const bf16x8*>(lds + lane * 16 + i * 1024);
  for (int p = 0; p < 2; ++p) for (int n = 0; n < 2; ++n) wq[p][n] = wf(p * 16384, n);
  f32x16 acc[M];
  for (int m = 0; m < M; ++m) for (int j = 0; j < 16; ++j) acc[m][j] = 0;
  const int off = lane * 16;
  for (int it = 0; it < iters; it += 8) {
    int wo = 2 * 16384 + (it % 64) * 16384;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      bf16x8 w2[2];
      for (int n = 0; n < 2; ++n) w2[n] = wf(wo, n);
      wo += 16384;
#pragma unroll
      for (int h = 0; h < 2; ++h)   // the two 16-channel halves of the k-step
#pragma unroll
        for (int m = 0; m < M; ++m) {
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[0][h], a[m], acc[m], 0, 0, 0);
          a[m] = *reinterpret_cast<const bf16x8*>(lds + off + m * 1024 + ((2 * k + h) & 7) * 9216);
        }
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
#pragma unroll
      for (int i = 0; i < 2 * M; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      for (int n = 0; n < 2; ++n) { wq[0][n] = wq[1][n]; wq[1][n] = w2[n]; }
    }
  }
  float s = 0;
  for (int m = 0; m < M; ++m) for (int j = 0; j < 16; ++j) s += acc[m][j];
  out[blockIdx.x * 512 + tid] = s;
}

int main(int argc, char** argv) {
  const double target_ms = argc > 1 ? atof(argv[1]) : 300.0;
  const int threads = argc > 2 ? atoi(argv[2]) : 512;  // 512: two wavefronts per SIMD, 256: one
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  bf16x8* src; float* out;
  CHECK(hipMalloc(&src, 4096 * 16));
  CHECK(hipMalloc(&out, (size_t)cus * 512 * 4));
  void* wts;
  CHECK(hipMalloc(&wts, 74 * 16384));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop16w<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 1024 * 8));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop16w<18>), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 1024 * 8));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop16y), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 1024 * 8));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop32w), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 1024 * 8));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop16z), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 1024 * 8));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop16x<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 1024 * 8));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop16x<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 1024 * 8));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 1024 * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int zero = 0; zero < 2; ++zero) {
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto& v : h) v = zero ? 0 : (unsigned short)(0x3c00 + (rand() & 0x83ff));  // |x| in [0.5, 2) with random sign / mantissa
    CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (int i = 0; i < 74 * 16384; i += 4096 * 16) CHECK(hipMemcpy((char*)wts + i, h.data(), 74 * 16384 - i < 4096 * 16 ? 74 * 16384 - i : 4096 * 16, hipMemcpyHostToDevice));
    for (int variant = 0; variant < 10; ++variant) {
      if (variant >= 4 && threads != 512) continue;
      const double flop_per_iter = variant == 2 ? 16.0 * 2 * 32 * 32 * 16 : (variant == 6 || variant == 9 ? 36.0 : variant >= 7 ? 72.0 : 18.0) * 2 * 16 * 16 * 32;  // per wave
      auto launch = [&](int iters) {
        if (variant == 0) hipLaunchKernelGGL(loop16<false>, dim3(cus), dim3(threads), 0, 0, src, out, iters);
        if (variant == 1) hipLaunchKernelGGL(loop16<true>, dim3(cus), dim3(threads), 9 * 1024 * 8, 0, src, out, iters);
        if (variant == 2) hipLaunchKernelGGL(loop32, dim3(cus), dim3(threads), 0, 0, src, out, iters);
        if (variant == 4) hipLaunchKernelGGL(loop16x<5>, dim3(cus), dim3(threads), 9 * 1024 * 8, 0, src, wts, out, iters & ~7);
        if (variant == 5) hipLaunchKernelGGL(loop16x<6>, dim3(cus), dim3(threads), 9 * 1024 * 8, 0, src, wts, out, iters & ~7);
        if (variant == 7) hipLaunchKernelGGL(loop16y, dim3(cus), dim3(256), 9 * 1024 * 8, 0, src, wts, out, iters & ~7);
        if (variant == 8) hipLaunchKernelGGL(loop16z, dim3(cus), dim3(256), 9 * 1024 * 8, 0, src, wts, out, iters & ~7);
        if (variant == 6) hipLaunchKernelGGL(loop16w<18>, dim3(cus), dim3(threads), 9 * 1024 * 8, 0, src, wts, out, iters & ~7);
        if (variant == 9) hipLaunchKernelGGL(loop32w, dim3(cus), dim3(threads), 9 * 1024 * 8, 0, src, wts, out, iters & ~7);
        if (variant == 3) hipLaunchKernelGGL(loop16w<9>, dim3(cus), dim3(threads), 9 * 1024 * 8, 0, src, wts, out, iters & ~7);
      };
      int iters = 20000;
      launch(iters);
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {  // second pass sized to target_ms so the clock has settled
        CHECK(hipEventRecord(e0));
        launch(iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) iters = (int)(iters * target_ms / ms);
      }
      const double tf = flop_per_iter * iters * (variant == 7 || variant == 8 ? 4.0 : threads / 64.0) * cus / (ms * 1e-3) / 1e12;
      const char* names[10] = {"16x16x32 bf16, register operands", "16x16x32 bf16 + 1 ds_read_b128 per 2 MFMAs", "32x32x16 bf16, register operands",
                              "16x16x32 + LDS reads + 2 weight loads per 18",
                              "  64 channels per wave, tiles split 5 / 4", "  64 channels per wave, tiles split 6 / 3",
                              "  two images per pass: 18 tiles per 2 weight loads",
                              "  two images, 64 channels, one wave per SIMD",
                              "  the same, accumulators pinned in AGPRs (asm)",
                              "  two images per pass on 32x32x16 MFMAs"};
      printf("%-46s %-6s data: %7.1f TFLOP/s  (%.1f ms, %d CUs x %d waves)\n", names[variant], zero ? "zero" : "random", tf, ms, cus, threads / 64);
    }
  }
  return 0;
}
