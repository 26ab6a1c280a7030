// Fixed cost of one small host -> device -> host round trip on MI355X, the floor under a frozen plan's
// replay (pytensor_amd/plan.py: parameters in, a few launches, results out, wait).  Which upload path and
// which completion wait are cheapest?
//   upload   U0  hipMemcpyAsync(pinned -> device) then the consumer kernel reads device memory
//            U1  the consumer reads the pinned host block directly (every workgroup, 1 KB each)
//            U2  the parameters travel BY VALUE in the kernarg block (3.5 KB struct)
//   wait     W0  hipStreamSynchronize
//            W1  host polls a sequence word the last kernel stores into pinned memory (system-scope release)
//            W2  hipEventRecord + spin on hipEventQuery
// The consumer is `work`: G workgroups, each sums its 1 KB of parameters and spins `spin` clock ticks; the
// producer of results is `finish` (one workgroup) writing 2 KB + the sequence word into pinned memory.
// build: hipcc --offload-arch=gfx950 -O3 -o call_lat call_lat.hip ; run: ./call_lat
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int NP = 448;  // 3.5 KB of doubles
struct Params { double v[NP]; };

__global__ void work_ptr(const double* __restrict__ p, double* __restrict__ partial, int spin) {
  __shared__ double red[256];
  double s = 0;
  for (int i = threadIdx.x; i < 128; i += blockDim.x) s += p[(blockIdx.x * 7 + i) % NP];
  red[threadIdx.x] = s;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < 128; i++) t += red[i]; partial[blockIdx.x] = t; }
}
__global__ void work_val(const Params pr, double* __restrict__ partial, int spin) {
  __shared__ double red[256];
  double s = 0;
  for (int i = threadIdx.x; i < 128; i += blockDim.x) s += pr.v[(blockIdx.x * 7 + i) % NP];
  red[threadIdx.x] = s;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < 128; i++) t += red[i]; partial[blockIdx.x] = t; }
}
__global__ void finish(const double* __restrict__ partial, int n, double* out_pinned, volatile unsigned long long* seq_word,
                       unsigned long long seq) {
  __shared__ double red[256];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 256) out_pinned[threadIdx.x] = red[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0 && seq_word) {
    __hip_atomic_store((unsigned long long*)seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double *hp, *dp, *partial, *hout;
  unsigned long long* hseq;
  CK(hipHostMalloc(&hp, sizeof(Params), hipHostMallocDefault));
  CK(hipHostMalloc(&hout, 4096, hipHostMallocDefault));
  CK(hipHostMalloc(&hseq, 64, hipHostMallocDefault));
  CK(hipMalloc(&dp, sizeof(Params)));
  CK(hipMalloc(&partial, 8192 * 8));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  Params pv;
  for (int i = 0; i < NP; i++) { hp[i] = i * 0.5; pv.v[i] = i * 0.5; }
  *hseq = 0;
  const int iters = 3000;
  const char* un[] = {"U0 memcpyAsync H2D", "U1 read pinned host", "U2 kernarg by value"};
  const char* wn[] = {"W0 hipStreamSynchronize", "W1 poll pinned seq word", "W2 spin hipEventQuery"};
  for (int G : {256, 2048})
    for (int spin : {100, 2000}) {  // 100 MHz wall clock: 1 us, 20 us of "work" per workgroup
      printf("-- consumer: %d workgroups x %d us spin\n", G, spin / 100);
      for (int u = 0; u < 3; u++)
        for (int w = 0; w < 3; w++) {
          unsigned long long seq = *hseq;
          double best = 1e30, tot = 0;
          for (int rep = 0; rep < 3; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < iters; it++) {
              hp[it % NP] = it;  // (the caller's fresh parameters)
              seq++;
              if (u == 0) {
                CK(hipMemcpyAsync(dp, hp, sizeof(Params), hipMemcpyHostToDevice, st));
                work_ptr<<<G, 128, 0, st>>>(dp, partial, spin);
              } else if (u == 1) {
                work_ptr<<<G, 128, 0, st>>>(hp, partial, spin);
              } else {
                pv.v[it % NP] = it;
                work_val<<<G, 128, 0, st>>>(pv, partial, spin);
              }
              finish<<<1, 256, 0, st>>>(partial, G, hout, w == 1 ? hseq : nullptr, seq);
              if (w == 0) CK(hipStreamSynchronize(st));
              else if (w == 1) { while (__atomic_load_n(hseq, __ATOMIC_ACQUIRE) != seq) {} }
              else { CK(hipEventRecord(ev, st)); while (hipEventQuery(ev) == hipErrorNotReady) {} }
            }
            CK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            best = us < best ? us : best;
            tot += us;
          }
          printf("   %-24s %-26s  %7.2f us / round trip (best of 3; mean %.2f)\n", un[u], wn[w], best, tot / 3);
        }
    }
  return 0;
}
