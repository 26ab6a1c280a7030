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

using namespace pthip_dev;

namespace {

constexpr int BLOCK = 256;
constexpr int TILE = 16;          // columns per block
constexpr int LANES = BLOCK / TILE;  // row lanes per column
constexpr int MAX_TASKS = 16;

struct Tasks {
  int n;
  int op[MAX_TASKS];              // pthip_reduce_op (ADD / MUL / MAX / MIN)
  const void* part[MAX_TASKS];    // [nparts, M] row-major
  long long nparts[MAX_TASKS];
  long long M[MAX_TASKS];
  int S[MAX_TASKS];               // row chunks
  void* out[MAX_TASKS];           // [S, M] (accumulator dtype)
  int blk0[MAX_TASKS + 1];        // first block of each task; a task owns tiles(M) * S blocks
};

template <class Op, class T>
__device__ __forceinline__ void shrink_tile(const T* __restrict__ part, long long p0, long long p1, long long M,
                                            T* __restrict__ out_row, long long col0, T* smem) {
  const int c = threadIdx.x % TILE, r = threadIdx.x / TILE;
  const long long col = col0 + c;
  T a0 = Op::template identity<T>(), a1 = a0, a2 = a0, a3 = a0;
  if (col < M) {
    long long p = p0 + r;
    for (; p + 3 * LANES < p1; p += 4 * LANES) {
      const T v0 = part[p * M + col], v1 = part[(p + LANES) * M + col];
      const T v2 = part[(p + 2 * LANES) * M + col], v3 = part[(p + 3 * LANES) * M + col];
      a0 = Op::apply(a0, v0);
      a1 = Op::apply(a1, v1);
      a2 = Op::apply(a2, v2);
      a3 = Op::apply(a3, v3);
    }
    for (; p < p1; p += LANES) a0 = Op::apply(a0, part[p * M + col]);
  }
  smem[r * TILE + c] = Op::apply(Op::apply(a0, a1), Op::apply(a2, a3));
  __syncthreads();
  if (r == 0 && col < M) {
    T acc = smem[c];
#pragma unroll
    for (int j = 1; j < LANES; j++) acc = Op::apply(acc, smem[j * TILE + c]);
    out_row[col] = acc;
  }
}

template <class T>
__global__ __launch_bounds__(BLOCK) void multi_finish_kernel(Tasks t) {
  __shared__ T smem[BLOCK];
  int k = 0;
  while (k + 1 < t.n && (int)blockIdx.x >= t.blk0[k + 1]) k++;
  const int lb = (int)blockIdx.x - t.blk0[k];
  const int S = t.S[k];
  const int s = lb % S;
  const long long col0 = (long long)(lb / S) * TILE;
  const long long np = t.nparts[k], M = t.M[k];
  const long long chunk = (np + S - 1) / S;
  long long p0 = s * chunk, p1 = p0 + chunk;
  if (p1 > np) p1 = np;
  if (p0 > np) p0 = np;
  const T* part = (const T*)t.part[k];
  T* out_row = (T*)t.out[k] + (long long)s * M;
  switch (t.op[k]) {
    case PTHIP_RED_ADD: shrink_tile<OpAdd, T>(part, p0, p1, M, out_row, col0, smem); break;
    case PTHIP_RED_MUL: shrink_tile<OpMul, T>(part, p0, p1, M, out_row, col0, smem); break;
    case PTHIP_RED_MAX: shrink_tile<OpMax, T>(part, p0, p1, M, out_row, col0, smem); break;
    default: shrink_tile<OpMin, T>(part, p0, p1, M, out_row, col0, smem); break;
  }
}

}  // namespace

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
