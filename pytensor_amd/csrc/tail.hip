// tail.hip — one launch that shrinks every per-workgroup partial slab at the end of an evaluation.
//
// After the streaming kernels of a logp+grad graph, what is left is a handful of tiny dependent
// launches: the fixed-order sums of per-workgroup partials (the [nparts, M] slabs of a split
// Gemv — pytensor/tensor/blas/gemv.py:64-108 — or of a scatter-add; the CAReduce second stage of
// pytensor/tensor/elemwise.py:1233).  Each is a few microseconds of work behind ~4.5 us of
// dependent-launch latency.  `pthip_multi_finish` takes up to 16 slabs of different shapes and
// reduces each from [nparts, M] to [S, M] (S <= 16 row chunks) in ONE launch; the generated tail
// kernel (codegen.tail_chain_source) adds the S rows in order while it applies the epilogues and
// the scalar graph behind them.  Blocks map to (task, column tile, row chunk) through a prefix
// table; a tile is 16 adjacent columns (128-byte row segments), 16 row lanes walk their chunk with
// 4 independent accumulators (the slab is latency-bound: 2 MB must be in flight at once, hence
// one 16 KB piece per block and ~256 blocks) and are combined through LDS in a fixed order:
// deterministic, independent of scheduling.
#include "common.h"
#include "reduce_device.h"
#include "tail_device.h"

using namespace pthip_dev;

namespace {

constexpr int BLOCK = TAIL_SHRINK_BLOCK;
constexpr int TILE = TAIL_SHRINK_TILE;
constexpr int MAX_TASKS = 16;
using Tasks = TailTasksT<MAX_TASKS>;

// (the piece itself — task / column tile / row chunk of a block, four accumulators per row lane, fixed-order
//  LDS combine — lives in tail_device.h: the generated tail kernels run the same code as their prologue)
template <class T>
__global__ __launch_bounds__(BLOCK) void multi_finish_kernel(Tasks t) {
  __shared__ T smem[BLOCK];
  tail_shrink_block<T, MAX_TASKS>(t, (int)blockIdx.x, smem);
}

__global__ void join_signal_kernel(int* word) { __hip_atomic_store(word, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// ---- the probe behind the device-side join (pthip_join_probe) ----
__global__ __launch_bounds__(256) void join_probe_write_kernel(unsigned* __restrict__ buf, int n, unsigned val) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) buf[i] = val ^ (unsigned)i;
}
// the consumer side exactly as the generated tail kernel has it (codegen._tail_prologue): a relaxed agent-scope spin
// on the word, NO fence behind it, plain loads of what the other stream wrote
__global__ __launch_bounds__(256) void join_probe_wait_kernel(int* word, const unsigned* __restrict__ buf, int n, unsigned val,
                                                             int* bad) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();  // 100 MHz; 2 ms, the order of the join's own bound
    bool seen;
    while (!(seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) && wall_clock64() - t0 < 200000) {}
    ok = seen;
  }
  __syncthreads();
  if (!ok) {
    if (threadIdx.x == 0) atomicAdd(bad, 1 << 20);  // (the streams did not overlap: reported apart from mismatches)
    return;
  }
  int wrong = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) wrong += buf[i] != (val ^ (unsigned)i);
  if (wrong) atomicAdd(bad, wrong);
}
__global__ void join_probe_reset_kernel(int* word) { __hip_atomic_store(word, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace

// Does what the device-side join relies on hold on THIS device, driver and partition mode?  `iters` times: stream 0
// launches the waiting kernel (every CU's worth of workgroups), stream 1 overwrites a 1 MB buffer IN PLACE with a new
// pattern and signals; the waiters read the buffer with plain loads, no fence behind the wait — stale lines kept from
// the previous iteration in any XCD's L2 show up as mismatches.  *bad: mismatching words summed over the iterations
// (+ 2^20 per waiter that never saw the signal: the two streams did not run side by side).  The host side
// (plan.py) runs this once per process before the first plan that would use the join and falls back to the event
// between the streams when *bad != 0.  Not capturable; synchronises both streams.
extern "C" int pthip_join_probe(int iters, int* bad_out) {
  PTHIP_REQUIRE_INIT();
  if (!bad_out || iters <= 0) return pthip::set_error("pthip_join_probe: bad arguments");
  if (pthip::ctx().capturing || pthip::ctx().recorder) return pthip::set_error("pthip_join_probe: not inside a capture / recording");
  int r = pthip_stream_wait(1, 0);  // (creates stream 1; orders it behind what the caller has enqueued)
  if (r) return r;
  hipStream_t s0 = pthip::ctx().streams[0], s1 = pthip::ctx().streams[1];
  constexpr int N = 1 << 18;  // 1 MB: every L2 channel of every XCD
  void* mem = nullptr;
  if ((r = pthip_alloc((size_t)N * 4 + 256, &mem))) return r;
  unsigned* buf = (unsigned*)mem;
  int* word = (int*)((char*)mem + (size_t)N * 4);
  int* bad = word + 16;
  hipEvent_t ev = nullptr;
  // every exit path gives the buffer and the event back (the early returns of PTHIP_CHECK leaked both)
  auto run = [&]() -> int {
    PTHIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    PTHIP_CHECK(hipMemsetAsync(word, 0, 128, s0));
    PTHIP_CHECK(hipEventRecord(ev, s0));
    PTHIP_CHECK(hipStreamWaitEvent(s1, ev, 0));
    for (int it = 0; it < iters; it++) {
      const unsigned val = 0x9E3779B9u * (unsigned)(it + 1);
      hipLaunchKernelGGL(join_probe_wait_kernel, dim3(pthip::kNumCU), dim3(256), 0, s0, word, (const unsigned*)buf, N, val, bad);
      hipLaunchKernelGGL(join_probe_write_kernel, dim3(pthip::kNumCU), dim3(256), 0, s1, buf, N, val);
      hipLaunchKernelGGL(join_signal_kernel, dim3(1), dim3(1), 0, s1, word);
      hipLaunchKernelGGL(join_probe_reset_kernel, dim3(1), dim3(1), 0, s0, word);
      PTHIP_CHECK(hipEventRecord(ev, s0));  // the next overwrite waits for this iteration's readers
      PTHIP_CHECK(hipStreamWaitEvent(s1, ev, 0));
    }
    PTHIP_CHECK(hipGetLastError());
    int h = 0;
    PTHIP_CHECK(hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, s0));
    PTHIP_CHECK(hipStreamSynchronize(s0));
    PTHIP_CHECK(hipStreamSynchronize(s1));
    *bad_out = h;
    return 0;
  };
  r = run();
  if (r) {  // whatever was enqueued must have drained before the buffer goes back to the pool
    (void)hipStreamSynchronize(s0);
    (void)hipStreamSynchronize(s1);
  }
  if (ev) (void)hipEventDestroy(ev);
  const int rf = pthip_free(mem);
  return r ? r : rf;
}

// the last launch of a plan's latency-chain segment (include/pthip.h)
extern "C" int pthip_join_signal(void* word) {
  PTHIP_REQUIRE_INIT();
  if (!word) return pthip::set_error("pthip_join_signal: null word");
  PTHIP_KLAUNCH(join_signal_kernel, dim3(1), dim3(1), 0, pthip::ctx().stream, (int*)word);
  return pthip::post_launch("join_signal");
}

extern "C" int pthip_multi_finish(int dtype, int n_tasks, const int* ops, const void* const* parts,
                                  const int64_t* nparts, const int64_t* M, const int* S, void* const* outs) {
  PTHIP_REQUIRE_INIT();
  if (n_tasks <= 0) return 0;
  if (n_tasks > MAX_TASKS) return pthip::set_error("pthip_multi_finish: at most %d tasks per call", MAX_TASKS);
  Tasks t{};
  t.n = n_tasks;
  int nb = 0;
  for (int k = 0; k < n_tasks; k++) {
    if (ops[k] < PTHIP_RED_ADD || ops[k] > PTHIP_RED_MIN)
      return pthip::set_error("pthip_multi_finish: op %d is not ADD/MUL/MAX/MIN", ops[k]);
    if (nparts[k] < 0 || M[k] <= 0 || S[k] <= 0 || S[k] > 64) return pthip::set_error("pthip_multi_finish: bad extents");
    t.op[k] = ops[k];
    t.part[k] = parts[k];
    t.nparts[k] = nparts[k];
    t.M[k] = M[k];
    t.S[k] = S[k];
    t.out[k] = outs[k];
    t.blk0[k] = nb;
    nb += (int)((M[k] + TILE - 1) / TILE) * S[k];
  }
  t.blk0[n_tasks] = nb;
  hipStream_t st = pthip::ctx().stream;
  switch (dtype) {
    case PTHIP_F64: PTHIP_KLAUNCH(multi_finish_kernel<double>, dim3(nb), dim3(BLOCK), 0, st, t); break;
    case PTHIP_F32: PTHIP_KLAUNCH(multi_finish_kernel<float>, dim3(nb), dim3(BLOCK), 0, st, t); break;
    case PTHIP_I64: PTHIP_KLAUNCH(multi_finish_kernel<long long>, dim3(nb), dim3(BLOCK), 0, st, t); break;
    default: return pthip::set_error("pthip_multi_finish: unsupported accumulator dtype %d", dtype);
  }
  return pthip::post_launch("multi_finish");
}
