// MFMA-only throughput under the package power cap (gfx950): registers only, RANDOM bf16 operands, every CU busy.
// Compares v_mfma_f32_32x32x16_bf16 with v_mfma_f32_16x16x32_bf16 (same peak flops per cycle) and the fp16 forms.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o tools/build/mfma_power_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// fp8 (e4m3) scaled MFMA: 32x32x64 vs 16x16x128 (block scales 2^0)
template <int KIND>
__global__ __launch_bounds__(512, 2) void probe8(const unsigned short* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 512 + threadIdx.x;
  i32x8 a[2], b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a[i] = *reinterpret_cast<const i32x8*>(src + ((size_t)(t * 4 + i) * 16) % (1 << 24));
    b[i] = *reinterpret_cast<const i32x8*>(src + ((size_t)(t * 4 + 2 + i) * 16) % (1 << 24));
  }
  constexpr int S1 = 0x7f7f7f7f;
  float sum = 0.f;
  if constexpr (KIND == 0) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(i + k) & 1], b[k], acc[i], 0, 0, 0, S1, 0, S1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[i][r];
  } else {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i & 1], b[(i >> 1) & 1], acc[i], 0, 0, 0, S1, 0, S1);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += acc[i][r];
  }
  out[t] = sum;
}

template <int KIND>
__global__ __launch_bounds__(512, 2) void probe(const unsigned short* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 512 + threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const bf16x8*>(src + ((size_t)(t * 8 + i) * 8) % (1 << 24));
    b[i] = *reinterpret_cast<const bf16x8*>(src + ((size_t)(t * 8 + 4 + i) * 8) % (1 << 24));
  }
  float sum = 0.f;
  if constexpr (KIND == 0 || KIND == 2) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + k) & 3], b[k], acc[i], 0, 0, 0);
          else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(i + k) & 3]), __builtin_bit_cast(f16x8, b[k]), acc[i], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[i][r];
  } else {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if constexpr (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + k) & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
          else if constexpr (KIND == 3) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(i + k) & 3]), __builtin_bit_cast(f16x8, b[(i >> 2) & 3]), acc[i], 0, 0, 0);
          else if (k == 0) {   // KIND 4: the two k-halves of one accumulator BACK TO BACK (dependent pair), 16 accumulators
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + 1) & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
          }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += acc[i][r];
  }
  out[t] = sum;
}


// Duty-cycle experiment (round 5): ONE wave per SIMD issues 32 independent v_mfma_f32_16x16x32_bf16 (512 matrix-pipe cycles) and then
// sleeps 64 * SLEEP cycles, every CU busy.  If the chip is energy-limited, the idle share is returned as clock: the rate stays flat until the duty
// cycle falls below (sustained clock / 2.4 GHz) — i.e. REMOVING idle matrix-pipe cycles from a power-capped kernel buys nothing.
template <int SLEEP, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void probe_duty(const unsigned short* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * THREADS + threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const bf16x8*>(src + ((size_t)(t * 8 + i) * 8) % (1 << 24));
    b[i] = *reinterpret_cast<const bf16x8*>(src + ((size_t)(t * 8 + 4 + i) * 8) % (1 << 24));
  }
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + k) & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    if constexpr (THREADS > 256) __builtin_amdgcn_s_barrier();     // two waves per SIMD: they go idle together
    if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) sum += acc[i][r];
  out[t] = sum;
}

template <int SLEEP, int THREADS>
static void run_duty(const unsigned short* src, float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256;                      // one workgroup per CU: 4 waves = one wave per SIMD, 8 waves = two
  probe_duty<SLEEP, THREADS><<<grid, THREADS>>>(src, out, iters / 10);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) probe_duty<SLEEP, THREADS><<<grid, THREADS>>>(src, out, iters);
  hipEventRecord(e0);
  for (int rep = 0; rep < 3; ++rep) probe_duty<SLEEP, THREADS><<<grid, THREADS>>>(src, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = 3.0 * (double)grid * (THREADS / 64) * iters * 32.0 * 2 * 16 * 16 * 32;
  // matrix-pipe cycles per iteration and SIMD: 512 per wave; nominal duty if nothing else interfered
  const double busy = 512.0 * (THREADS / 256);
  printf("%d wave(s)/SIMD  sleep %2d x 64 cyc  nominal duty %5.1f %%  %8.1f TFLOP/s   %7.1f ns per iteration  (%.1f ms per launch)\n", THREADS / 256, SLEEP,
         100.0 * busy / (busy + 64.0 * SLEEP), fl / (ms * 1e-3) / 1e12, ms * 1e6 / (3.0 * iters), ms / 3);
}

template <int KIND>
static void run8(const char* name, const unsigned short* src, float* out, int grid, int iters, double flops_per_iter_per_wave) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe8<KIND><<<grid, 512>>>(src, out, iters / 10);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) probe8<KIND><<<grid, 512>>>(src, out, iters);
  hipEventRecord(e0);
  for (int rep = 0; rep < 3; ++rep) probe8<KIND><<<grid, 512>>>(src, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = 3.0 * (double)grid * 8 * iters * flops_per_iter_per_wave;
  printf("%-28s %8.1f TFLOP/s  (%.1f ms per launch)\n", name, fl / (ms * 1e-3) / 1e12, ms / 3);
}

template <int KIND>
static void run(const char* name, const unsigned short* src, float* out, int grid, int iters, double flops_per_iter_per_wave) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<KIND><<<grid, 512>>>(src, out, iters / 10);
  hipDeviceSynchronize();
  // long enough for the power management to settle (~1 s), timed over the last launches
  for (int rep = 0; rep < 3; ++rep) probe<KIND><<<grid, 512>>>(src, out, iters);
  hipEventRecord(e0);
  for (int rep = 0; rep < 3; ++rep) probe<KIND><<<grid, 512>>>(src, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = 3.0 * (double)grid * 8 * iters * flops_per_iter_per_wave;
  printf("%-28s %8.1f TFLOP/s  (%.1f ms per launch)\n", name, fl / (ms * 1e-3) / 1e12, ms / 3);
}

int main(int argc, char** argv) {
  const int zero = argc > 1 && atoi(argv[1]) == 1;
  const size_t n = 1 << 24;
  std::vector<unsigned short> h(n);
  srand(1);
  for (size_t i = 0; i < n; ++i) {
    // random bf16 / fp16-safe values in (-2, 2): sign, exponent 0x3f..0x3f+-1, random mantissa
    const unsigned short mant = rand() & 0x7f, sign = (rand() & 1) << 15, ex = (0x7e + (rand() & 1)) << 7;
    h[i] = zero ? 0 : (unsigned short)(sign | ex | mant);
  }
  unsigned short* d; float* out;
  hipMalloc(&d, n * 2); hipMalloc(&out, 512 * 512 * 4);
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
  const int grid = 512, iters = 40000;       // 2 workgroups of 8 waves per CU = 4 waves per SIMD
  printf("operands: %s\n", zero ? "zeros" : "random");
  run<0>("mfma_f32_32x32x16_bf16", d, out, grid, iters, 16.0 * 2 * 32 * 32 * 16);
  run<1>("mfma_f32_16x16x32_bf16", d, out, grid, iters, 32.0 * 2 * 16 * 16 * 32);
  run<2>("mfma_f32_32x32x16_f16", d, out, grid, iters, 16.0 * 2 * 32 * 32 * 16);
  run<3>("mfma_f32_16x16x32_f16", d, out, grid, iters, 32.0 * 2 * 16 * 16 * 32);
  run<4>("16x16x32_bf16 dependent pairs", d, out, grid, iters, 32.0 * 2 * 16 * 16 * 32);
  // fp8: every byte a random e4m3 value in (-2, 2)-ish (the same bit patterns read as bytes)
  run8<0>("mfma_scale_32x32x64_fp8", d, out, grid, iters, 8.0 * 2 * 32 * 32 * 64);
  run8<1>("mfma_scale_16x16x128_fp8", d, out, grid, iters, 16.0 * 2 * 16 * 16 * 128);
  printf("duty-cycle experiment: 32 x mfma_f32_16x16x32_bf16 per wave, then (barrier +) s_sleep; every CU busy\n");
  run_duty<0, 256>(d, out, 16 * iters); run_duty<2, 256>(d, out, 16 * iters); run_duty<4, 256>(d, out, 16 * iters); run_duty<8, 256>(d, out, 12 * iters);
  run_duty<0, 512>(d, out, 8 * iters); run_duty<1, 512>(d, out, 8 * iters); run_duty<2, 512>(d, out, 8 * iters); run_duty<4, 512>(d, out, 8 * iters);
  run_duty<6, 512>(d, out, 8 * iters); run_duty<8, 512>(d, out, 6 * iters); run_duty<12, 512>(d, out, 6 * iters); run_duty<16, 512>(d, out, 4 * iters);
  run_duty<32, 512>(d, out, 3 * iters);
  return 0;
}
