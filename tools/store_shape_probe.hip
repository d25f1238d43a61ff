// Per-CU vector-memory rate of an epilogue-like burst: every workgroup (8 waves, one per CU) issues NI 16-byte-per-lane stores (or loads)
// of one of four shapes and the wave-0 clock brackets the burst:  shape 0: 64 rows x 16 B (row per lane)   1: 16 rows x 64 B
// 2: 8 rows x 128 B (full lines)   3: 4 rows x 256 B   4: 1 KB contiguous.  Row stride = ld bytes.  Each workgroup works in its own
// region (like one output tile), all workgroups at once or only `active` of them (arg 3) to separate the CU path from the chip's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool LOAD>
__global__ __launch_bounds__(512) void probe(char* base, size_t region, int ld, int ni, long long* out, int active) {
  if ((int)blockIdx.x >= active) return;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int LPR = SHAPE == 0 ? 1 : SHAPE == 1 ? 4 : SHAPE == 2 ? 8 : SHAPE == 3 ? 16 : 64;   // lanes per row
  constexpr int ROWS = 64 / LPR;
  char* p = base + (size_t)blockIdx.x * region + (size_t)(l / LPR) * ld + (l % LPR) * 16 + (size_t)w * (LPR * 16);
  u32x4 v = {(unsigned)l, 1u, 2u, 3u};
  u32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ni; ++i) {
    char* q = p + (size_t)i * ROWS * ld;
    if constexpr (LOAD) {
      u32x4 r;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(q) : "memory");
      if (i % 8 == 7) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += r; }
    } else {
      asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(q), "v"(v) : "memory");
    }
  }
  const long long t1 = __builtin_readcyclecounter();      // issue time of the burst (stores: the wave is done with them here)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = __builtin_readcyclecounter();
  __syncthreads();
  const long long t3 = __builtin_readcyclecounter();
  if (l == 0) { out[(blockIdx.x * 8 + w) * 4 + 0] = t1 - t0; out[(blockIdx.x * 8 + w) * 4 + 1] = t2 - t0; out[(blockIdx.x * 8 + w) * 4 + 2] = t3 - t0; out[(blockIdx.x * 8 + w) * 4 + 3] = acc[0]; }
}

template <int SHAPE, bool LOAD>
void run(char* buf, size_t region, int ld, int ni, long long* dout, int active, const char* name) {
  long long h[256 * 8 * 4];
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((probe<SHAPE, LOAD>), dim3(256), dim3(512), 0, 0, buf, region, ld, ni, dout, active);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  double issue = 0, done = 0, all = 0;
  for (int b = 0; b < active; ++b) {
    long long mi = 0, md = 0;
    for (int w = 0; w < 8; ++w) { if (h[(b * 8 + w) * 4] > mi) mi = h[(b * 8 + w) * 4]; if (h[(b * 8 + w) * 4 + 1] > md) md = h[(b * 8 + w) * 4 + 1]; }
    issue += mi; done += md; all += h[(b * 8) * 4 + 2];
  }
  issue /= active; done /= active; all /= active;
  const double bytes = 8.0 * ni * 1024;
  printf("%-5s shape %d (%s)  active CUs %3d: last wave issued after %8.0f cyc = %5.1f cyc per instr per CU, %5.1f B/cyc/CU; all landed %8.0f cyc = %5.1f B/cyc/CU\n",
         LOAD ? "load" : "store", SHAPE, name, active, issue, issue / (8.0 * ni), bytes / issue, done, bytes / done);
}

int main(int argc, char** argv) {
  const int ld = argc > 1 ? atoi(argv[1]) : 8192;
  const int ni = argc > 2 ? atoi(argv[2]) : 16;
  const size_t region = (size_t)ld * 64 * ni + 65536;
  char* buf; long long* dout;
  hipMalloc(&buf, region * 256);
  hipMemset(buf, 0, region * 256);
  hipMalloc(&dout, 256 * 8 * 4 * 8);
  for (int active : {256, 32, 1}) {
    run<0, false>(buf, region, ld, ni, dout, active, "64 rows x 16 B");
    run<1, false>(buf, region, ld, ni, dout, active, "16 rows x 64 B");
    run<2, false>(buf, region, ld, ni, dout, active, "8 rows x 128 B");
    run<3, false>(buf, region, ld, ni, dout, active, "4 rows x 256 B");
    run<4, false>(buf, region, ld, ni, dout, active, "1 KB contiguous");
    run<1, true>(buf, region, ld, ni, dout, active, "16 rows x 64 B");
    run<2, true>(buf, region, ld, ni, dout, active, "8 rows x 128 B");
    run<4, true>(buf, region, ld, ni, dout, active, "1 KB contiguous");
  }
  return 0;
}
