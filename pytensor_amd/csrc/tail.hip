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

}  // namespace

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
