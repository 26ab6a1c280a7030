// One-way latency of a cross-workgroup hand-over on gfx950, the floor under every hop of the persistent
// kernels (chol_dag / trsv_dag / lu_panel): workgroup A publishes a 16-byte self-validating pair
// {seq, seq ^ MAGIC} with one write-through store, workgroup B polls it with agent-scope loads and answers
// the same way; N round trips, timed on the 100 MHz wall clock.  Second variant: data + release fence +
// flag (the chol_dag hand-over).
// build: hipcc --offload-arch=gfx950 -O3 -o hop hop.hip ; run: ./hop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef u64 u2 __attribute__((ext_vector_type(2)));
constexpr u64 MAGIC = 0x7ff4c0de5ea1ed01ull;
__device__ __forceinline__ void publish(u64* slot, u64 v) {
  u2 pr = {v, v ^ MAGIC};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" : : "v"((u2*)slot), "v"(pr) : "memory");
}
__device__ __forceinline__ bool poll(const u64* slot, u64& v) {
  v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u64 b = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (v ^ b) == MAGIC;
}
__global__ void pingpong_pair(u64* box, int n, long long* ticks, int stride_wg) {
  // workgroups 0 and stride_wg play; the others exit (stride_wg picks a partner on another XCD or the same)
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == stride_wg ? 1 : -1);
  if (me < 0 || threadIdx.x != 0) return;
  u64* mine = box + (me ? 2 : 0) * 8;   // separate cache lines
  u64* theirs = box + (me ? 0 : 2) * 8;
  const long long t0 = wall_clock64();
  for (int i = 1; i <= n; i++) {
    u64 v;
    if (me == 0) { publish(mine, (u64)i); while (!(poll(theirs, v) && v == (u64)i)) {} }
    else { while (!(poll(theirs, v) && v == (u64)i)) {} publish(mine, (u64)i); }
  }
  if (me == 0) ticks[0] = wall_clock64() - t0;
}
__global__ void pingpong_flag(u64* data, int* flag, int n, long long* ticks, int stride_wg) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == stride_wg ? 1 : -1);
  if (me < 0 || threadIdx.x != 0) return;
  u64* dmine = data + (me ? 64 : 0);
  u64* dtheirs = data + (me ? 0 : 64);
  int* fmine = flag + (me ? 64 : 0);
  int* ftheirs = flag + (me ? 0 : 64);
  const long long t0 = wall_clock64();
  u64 sink = 0;
  for (int i = 1; i <= n; i++) {
    if (me == 0) {
      __hip_atomic_store(dmine, (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(fmine, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ftheirs, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != i) {}
      sink += *dtheirs;
    } else {
      while (__hip_atomic_load(ftheirs, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != i) {}
      sink += *dtheirs;
      __hip_atomic_store(dmine, (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(fmine, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (me == 0) { ticks[0] = wall_clock64() - t0; ticks[1] = (long long)sink; }
}
// Third variant: the SAME hand-over with workgroup-scope cache policy only (store sc0: write through the CU's L1
// into L2; load sc0: miss L1, read L2).  Two workgroups on the same XCD share that L2, so the pair never has to
// reach memory; on different XCDs the reader keeps seeing its own L2's stale line — bounded spins, a timeout is
// the expected answer there.  Also reports which XCD each player ran on (HW_REG_XCC_ID).
__device__ __forceinline__ void publish_l2(u64* slot, u64 v) {
  u2 pr = {v, v ^ MAGIC};
  asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 2" : : "v"((u2*)slot), "v"(pr) : "memory");
}
__device__ __forceinline__ bool poll_l2(const u64* slot, u64& v) {
  u2 pr;
  asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(pr) : "v"((const u2*)slot) : "memory");
  v = pr.x;
  return (pr.x ^ pr.y) == MAGIC;
}
__global__ void pingpong_pair_l2(u64* box, int n, long long* ticks, int stride_wg, int* xcc) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == stride_wg ? 1 : -1);
  if (me < 0 || threadIdx.x != 0) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  xcc[me] = (int)(id & 0xf);
  u64* mine = box + (me ? 2 : 0) * 8;
  u64* theirs = box + (me ? 0 : 2) * 8;
  const long long t0 = wall_clock64();
  int done = n;
  for (int i = 1; i <= n; i++) {
    u64 v;
    long long spins = 0;
    if (me == 0) publish_l2(mine, (u64)i);
    while (!(poll_l2(theirs, v) && v == (u64)i)) { if (++spins > 200000) { done = i - 1; break; } }
    if (done != n) break;
    if (me == 1) publish_l2(mine, (u64)i);
  }
  if (me == 0) { ticks[0] = wall_clock64() - t0; ticks[1] = done; }
}
// Fourth variant (found with tools/ubench/gru_persist.hip, where it is worth 3.3 us per step): the store without write-through
// (no scope bits: it stops in the XCD's L2) or with sc1 only, the load at agent scope (sc1).  Inside one XCD the reader is served
// from that L2; across XCDs the plain store is never seen.
template <int ST> __device__ __forceinline__ void publish_mix(u64* slot, u64 v) {
  u2 pr = {v, v ^ MAGIC};
  if constexpr (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" : : "v"((u2*)slot), "v"(pr) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" : : "v"((u2*)slot), "v"(pr) : "memory");
}
__device__ __forceinline__ bool poll_sc1(const u64* slot, u64& v) {
  u2 pr;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(pr) : "v"((const u2*)slot) : "memory");
  v = pr.x;
  return (pr.x ^ pr.y) == MAGIC;
}
template <int ST> __global__ void pingpong_pair_mix(u64* box, int n, long long* ticks, int stride_wg) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == stride_wg ? 1 : -1);
  if (me < 0 || threadIdx.x != 0) return;
  u64* mine = box + (me ? 2 : 0) * 8;
  u64* theirs = box + (me ? 0 : 2) * 8;
  const long long t0 = wall_clock64();
  int done = n;
  for (int i = 1; i <= n; i++) {
    u64 v;
    long long spins = 0;
    if (me == 0) publish_mix<ST>(mine, (u64)i);
    while (!(poll_sc1(theirs, v) && v == (u64)i)) { if (++spins > 200000) { done = i - 1; break; } }
    if (done != n) break;
    if (me == 1) publish_mix<ST>(mine, (u64)i);
  }
  if (me == 0) { ticks[0] = wall_clock64() - t0; ticks[1] = done; }
}
int main() {
  u64* box; int* flag; long long* ticks;
  (void)hipMalloc(&box, 4096); (void)hipMalloc(&flag, 4096); (void)hipMalloc(&ticks, 64);
  const int n = 2000;
  for (int partner : {1, 8, 9, 16, 248, 255}) {
    long long h[2];
    (void)hipMemset(box, 0, 4096);
    pingpong_pair<<<256, 64>>>(box, n, ticks, partner);
    (void)hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
    const double pair_ns = h[0] * 10.0 / (2.0 * n);
    (void)hipMemset(box, 0, 4096); (void)hipMemset(flag, 0, 4096);
    pingpong_flag<<<256, 64>>>(box, flag, n, ticks, partner);
    (void)hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
    printf("workgroups 0 <-> %3d: self-validating pair %.0f ns one way; data + release + flag + acquire %.0f ns one way\n", partner, pair_ns,
           h[0] * 10.0 / (2.0 * n));
    int* xcc; int hx[2] = {-1, -1};
    (void)hipMalloc(&xcc, 8);
    (void)hipMemset(box, 0, 4096); (void)hipMemset(xcc, 0xff, 8);
    pingpong_pair_l2<<<256, 64>>>(box, n, ticks, partner, xcc);
    (void)hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost);
    if (h[1] == n) printf("      L2-scope pair (sc0 only), XCDs %d / %d: %.0f ns one way\n", hx[0], hx[1], h[0] * 10.0 / (2.0 * n));
    else printf("      L2-scope pair (sc0 only), XCDs %d / %d: timed out after %lld of %d round trips (not coherent at this scope)\n", hx[0], hx[1], h[1], n);
    for (int stk = 0; stk < 2; stk++) {
      (void)hipMemset(box, 0, 4096);
      if (stk == 0) pingpong_pair_mix<0><<<256, 64>>>(box, n, ticks, partner);
      else pingpong_pair_mix<1><<<256, 64>>>(box, n, ticks, partner);
      (void)hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
      if (h[1] == n) printf("      %s store + sc1 16-byte load: %.0f ns one way\n", stk ? "sc1  " : "plain", h[0] * 10.0 / (2.0 * n));
      else printf("      %s store + sc1 16-byte load: timed out after %lld of %d round trips\n", stk ? "sc1  " : "plain", h[1], n);
    }
    (void)hipFree(xcc);
  }
  return 0;
}
