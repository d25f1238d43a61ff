// Do the vector ALU and the matrix core of a gfx950 SIMD work at the same time?  (Round-3 question behind the attention
// kernels: profiles/r02p_attention.txt shows "58 % VALU + 27 % MFMA" on the SIMD that bounds the CLIP kernel, i.e. the two
// never seem to overlap.)  One workgroup, registers only, s_memtime around every wave's loop:
//   mode 0  4 waves (one per SIMD), MFMA only:  16 independent v_mfma_f32_32x32x16_f16 per iteration           -> T_m
//   mode 1  4 waves, VALU only: per iteration 32 v_exp_f32 (quarter rate) + 96 full-rate ops (the softmax mix)  -> T_v
//   mode 2  8 waves (two per SIMD): waves 0-3 run the MFMA loop, waves 4-7 the VALU loop, same SIMDs              -> T_mix
//   mode 3  4 waves, ONE instruction stream holding both (MFMAs interleaved with the VALU ops by the compiler)  -> T_one
//   mode 4  8 waves, both waves of a SIMD run the interleaved stream (the attention kernels' situation)         -> T_two
// If T_mix ~ max(T_m, T_v) the units overlap across waves; if ~ T_m + T_v the issue port serialises them.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_mfma_probe.hip -o tools/build/valu_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void mfma_block(f32x16 (&acc)[4], const f16x8 (&a)[4], const f16x8 (&b)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + k) & 3], b[k], acc[i], 0, 0, 0);
}
// the same 16 MFMAs with the issuing wave put to sleep between them (s_nop: scalar-side wait states), so that it is never
// parked in the vector issue stage waiting for the matrix core: PAD wait states after each MFMA (the 32x32x16 MFMA occupies
// the matrix core for 32 cycles, 4 of them are its own issue)
template <int PAD>
__device__ __forceinline__ void mfma_block_padded(f32x16 (&acc)[4], const f16x8 (&a)[4], const f16x8 (&b)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + k) & 3], b[k], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PAD == 16) asm volatile("s_nop 15");
      if constexpr (PAD == 2) asm volatile("s_nop 1");
      if constexpr (PAD == 4) asm volatile("s_nop 3");
      if constexpr (PAD == 6) asm volatile("s_nop 5");
      if constexpr (PAD == 8) asm volatile("s_nop 7");
      if constexpr (PAD == 10) asm volatile("s_nop 9");
      __builtin_amdgcn_sched_barrier(0);
    }
}
// the softmax VALU mix of one 64-key x 32-query step: 32 scores per lane -> fma, max, exp2, add, cvt
__device__ __forceinline__ void valu_block(float (&s)[32], float& m, float& sum, float scale, unsigned (&pk)[16]) {
  float mx = s[0];
#pragma unroll
  for (int r = 1; r < 32; ++r) mx = fmaxf(mx, s[r]);
  m = fmaxf(m, mx * scale);
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale, -m));
    sum += p;
    s[r] = p;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 v = {(_Float16)s[2 * r], (_Float16)s[2 * r + 1]};
    pk[r] = __builtin_bit_cast(unsigned, v);
    s[2 * r] = s[2 * r] * 0.5f + 0.25f;                      // keep the chain alive and bounded for the next iteration
    s[2 * r + 1] = s[2 * r + 1] * 0.5f - 0.25f;
  }
}

__global__ __launch_bounds__(512) void probe(const _Float16* __restrict__ src, float* __restrict__ out, long long* __restrict__ cyc,
                                             int iters, int mode) {
  const int t = threadIdx.x, w = t >> 6;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const f16x8*>(src + (size_t)(t * 8 + i) * 8);
    b[i] = *reinterpret_cast<const f16x8*>(src + (size_t)(t * 8 + 4 + i) * 8);
  }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float s[32], m = -1e30f, sum = 0.f;
  unsigned pk[16];
#pragma unroll
  for (int r = 0; r < 32; ++r) s[r] = (float)src[(t * 32 + r) & 4095];
#pragma unroll
  for (int r = 0; r < 16; ++r) pk[r] = 0;
  const bool split = mode == 2 || mode == 6 || mode == 7 || mode >= 9;
  const bool do_m = mode == 0 || (split && w < 4) || (mode >= 3 && !split);
  const bool do_v = mode == 1 || (split && w >= 4) || (mode >= 3 && !split);
  if (mode == 6 && w >= 4) __builtin_amdgcn_s_setprio(3);      // the VALU wave outranks the MFMA wave
  if (mode == 7 && w < 4) __builtin_amdgcn_s_setprio(3);       // the MFMA wave outranks the VALU wave
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (mode == 5 || mode == 8) {
    // one stream, interleave FORCED: one MFMA, then a dozen VALU instructions, sixteen times per iteration
    for (int it = 0; it < iters; ++it) {
      mfma_block(acc, a, b);
      valu_block(s, m, sum, 0.125f, pk);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
      }
    }
  } else if (do_m && do_v) {
    for (int it = 0; it < iters; ++it) {
      mfma_block(acc, a, b);
      valu_block(s, m, sum, 0.125f, pk);
    }
  } else if (do_m && mode == 9) {
    for (int it = 0; it < iters; ++it) mfma_block_padded<2>(acc, a, b);
  } else if (do_m && mode == 10) {
    for (int it = 0; it < iters; ++it) mfma_block_padded<4>(acc, a, b);
  } else if (do_m && mode == 11) {
    for (int it = 0; it < iters; ++it) mfma_block_padded<6>(acc, a, b);
  } else if (do_m && mode == 12) {
    for (int it = 0; it < iters; ++it) mfma_block_padded<8>(acc, a, b);
  } else if (do_m) {
    for (int it = 0; it < iters; ++it) mfma_block(acc, a, b);
  } else if (do_v) {
    for (int it = 0; it < iters; ++it) valu_block(s, m, sum, 0.125f, pk);
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = sum + m;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) r += acc[i][q];
#pragma unroll
  for (int q = 0; q < 16; ++q) r += (float)pk[q];
  out[blockIdx.x * 512 + t] = r;
  if ((t & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

int main() {
  const int iters = 2000;
  _Float16* src; float* out; long long* cyc;
  hipMalloc(&src, 512 * 64 * 2 + 4096 * 2 + (1 << 20)); hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 8 * 8);
  std::vector<_Float16> h((512 * 64 + 4096 + (1 << 19)));
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const char* names[13] = {"MFMA only, 1 wave/SIMD", "VALU only, 1 wave/SIMD", "MFMA wave + VALU wave per SIMD", "one wave/SIMD, both in one stream",
                          "two waves/SIMD, both in one stream each", "one wave/SIMD, one stream, 1 MFMA : 12 VALU forced",
                          "MFMA wave + VALU wave, VALU wave at s_setprio 3", "MFMA wave + VALU wave, MFMA wave at s_setprio 3",
                          "two waves/SIMD, forced interleave each", "MFMA wave + s_nop 1 per MFMA | VALU wave", "MFMA wave + s_nop 3 per MFMA | VALU wave",
                          "MFMA wave + s_nop 5 per MFMA | VALU wave", "MFMA wave + s_nop 7 per MFMA | VALU wave"};
  double base[13];
  for (int mode = 0; mode < 13; ++mode) {
    const int threads = (mode == 2 || mode == 4 || mode >= 6) ? 512 : 256;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(threads), 0, 0, src, out, cyc, iters, mode);
      hipDeviceSynchronize();
    }
    long long c[8] = {0};
    hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int i = 0; i < threads / 64; ++i) mx = c[i] > mx ? c[i] : mx;
    base[mode] = (double)mx / iters;
    printf("mode %d  %-42s  %8.1f cycles per iteration (slowest wave; s_memtime ticks = 100 MHz x ... see note)  waves:", mode, names[mode], base[mode]);
    for (int i = 0; i < threads / 64; ++i) printf(" %.1f", (double)c[i] / iters);
    printf("\n");
  }
  printf("per iteration: 16 MFMA 32x32x16 (= 512 matrix-core cycles at 1 MFMA / 32 clk) and the softmax VALU mix of a 64-key step\n");
  printf("forced interleave: T_one' / max(T_m, T_v) = %.2f   two waves: T_two' / (2 max) = %.2f   prio VALU: %.2f x (T_m + T_v)   prio MFMA: %.2f x\n",
         base[5] / (base[0] > base[1] ? base[0] : base[1]), base[8] / (2 * (base[0] > base[1] ? base[0] : base[1])),
         base[6] / (base[0] + base[1]), base[7] / (base[0] + base[1]));
  printf("T_mix / (T_m + T_v) = %.2f   T_mix / max(T_m, T_v) = %.2f   T_one / (T_m + T_v) = %.2f   T_two / (2 (T_m + T_v)) = %.2f\n",
         base[2] / (base[0] + base[1]), base[2] / (base[0] > base[1] ? base[0] : base[1]), base[3] / (base[0] + base[1]),
         base[4] / (2 * (base[0] + base[1])));
  return 0;
}
