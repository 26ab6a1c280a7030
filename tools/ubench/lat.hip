// Instruction latencies on one wave of gfx950 (cycles of s_memtime per dependent instruction): the numbers
// the single-wave factorisation chain of chol_dag_kernel is designed against.
// build: hipcc --offload-arch=gfx950 -O3 -o lat lat.hip ; run: ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
#define REP(x) x x x x x x x x x x x x x x x x
__global__ void k(double* out, long long* cyc, double seed) {
  double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
  long long t[12];
  int s = 0;
  __shared__ double lds[256];
  lds[threadIdx.x] = a;
  __syncthreads();
#define T0 t[s++] = __builtin_readcyclecounter();
  T0
  for (int i = 0; i < N / 16; i++) { REP(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
  T0
  for (int i = 0; i < N / 16; i++) { REP(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
  T0
  for (int i = 0; i < N / 16; i++) { REP(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));) }
  T0
  for (int i = 0; i < N / 16; i++) { REP(asm volatile("v_rsq_f64 %0, %0" : "+v"(a));) }
  T0
  {  // readlane -> VALU use
    for (int i = 0; i < N / 16; i++) { REP(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_fma_f64 %2, s[20:21], %3, %2" : : "v"(((int*)&a)[0]), "v"(((int*)&a)[1]), "v"(a), "v"(b) : "s20", "s21");) }
  }
  T0
  {  // independent fma (throughput)
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < N / 16; i++) { REP(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));) }
    a += x0 + x1 + x2 + x3;
  }
  T0
  {  // LDS write -> read round trip (same wave)
    for (int i = 0; i < N / 16; i++) { REP(lds[threadIdx.x] = a; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a += lds[(threadIdx.x + 1) & 63]; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");) }
  }
  T0
  {  // fp32 fma chain
    float f = (float)a, g = 1.0001f, h = 1e-6f;
    for (int i = 0; i < N / 16; i++) { REP(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(g), "v"(h));) }
    a += f;
  }
  T0
  {  // dependent MFMA f64 16x16x4
    typedef double v4 __attribute__((ext_vector_type(4)));
    v4 acc = {a, a, a, a};
    for (int i = 0; i < N / 16; i++) { REP(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);) }
    a += acc[0];
  }
  T0
  {  // independent MFMA f64 (4 accumulators)
    typedef double v4 __attribute__((ext_vector_type(4)));
    v4 a0 = {a, a, a, a}, a1 = a0, a2 = a0, a3 = a0;
    for (int i = 0; i < N / 16; i++) { REP(a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, a3, 0, 0, 0);) }
    a += a0[0] + a1[0] + a2[0] + a3[0];
  }
  T0
  {  // s_memtime against the 100 MHz wall clock over ~1e5 dependent fmas
    const long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < 8192; i++) { REP(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    const long long w1 = wall_clock64(), c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cyc[12] = w1 - w0; cyc[13] = c1 - c0; }
  }
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) for (int i = 0; i < s; i++) cyc[i] = t[i];
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
  for (int r = 0; r < 2; r++) k<<<1, 64>>>(out, cyc, 1.5);
  long long h[16];
  hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
  const char* nm[] = {"v_fma_f64 dependent", "v_mul_f64 dependent", "v_rcp_f64 dependent", "v_rsq_f64 dependent", "readlane x2 + fma (dependent)", "v_fma_f64 x4 independent (per group of 4)",
                      "LDS store -> load round trip (+add)", "v_fma_f32 dependent", "mfma f64 16x16x4 dependent", "mfma f64 16x16x4 x4 independent (per group of 4)"};
  const int cnt[] = {N, N, N, N, N, N, N, N, N, N};
  int wall = 0;
  hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("s_memtime counts; clockRate attr %d kHz, wallClockRate %d kHz\n", clk, wall);
  for (int i = 0; i < 10; i++) printf("%-50s %8.1f counts per instruction (group)\n", nm[i], (double)(h[i + 1] - h[i]) / cnt[i]);
  printf("131072 dependent v_fma_f64: %lld wall ticks (10 ns), %lld s_memtime counts -> s_memtime runs at %.1f MHz; one fma = %.2f ns\n", h[12], h[13],
         (double)h[13] / ((double)h[12] * 1e-2), (double)h[12] * 10.0 / 131072);
  return 0;
}
