#pragma once
// tail_device.h — the slab-shrinking piece shared by csrc/tail.hip (pthip_multi_finish: a launch of its
// own) and by the generated tail kernels (codegen.tail_chain_source with `shrink`: the same pieces as the
// PROLOGUE of the tail launch — every workgroup shrinks one piece, takes a ticket, and the last one to
// finish runs the single-workgroup chain; one dependent launch less per evaluation).
// Requires reduce_device.h (pthip_dev::OpAdd ...).  Block size 256.
namespace pthip_dev {

constexpr int TAIL_SHRINK_BLOCK = 256;
constexpr int TAIL_SHRINK_TILE = 16;                                    // columns per block
constexpr int TAIL_SHRINK_LANES = TAIL_SHRINK_BLOCK / TAIL_SHRINK_TILE;  // row lanes per column

// by-value kernel argument; MAXT = 16: 720 bytes, MAXT = 4: 192 bytes (codegen packs the same layout)
template <int MAXT>
struct TailTasksT {
  int n;
  int op[MAXT];            // pthip_reduce_op (0 ADD / 1 MUL / 2 MAX / 3 MIN, include/pthip.h)
  const void* part[MAXT];  // [nparts, M] row-major
  long long nparts[MAXT];
  long long M[MAXT];
  int S[MAXT];             // row chunks
  void* out[MAXT];         // [S, M] (accumulator dtype)
  int blk0[MAXT + 1];      // first block of each task; a task owns tiles(M) * S blocks
};

// Device-side join of a plan's two streams (pthip_plan_replay4, descriptor flag bit 1): the latency-chain segment ends
// with pthip_join_signal (a one-thread kernel storing 1 into a word), the generated tail kernel that opens the closing
// segment waits for the word in its prologue (codegen._tail_prologue) and puts it back to 0 — instead of the streaming
// segment's stream waiting for an event (5.6 us between `gchain` and the next launch, profiles/r4z_c4_timeline.md).
// No acquire fence behind the wait: at agent scope it invalidates the L2 and whatever the launch reads next comes back
// from HBM (measured on the slab launch: 4.8 -> 11 us, profiles/r4_c4_device_join.txt).  None is needed: the caches
// were invalidated when the waiting kernel started, and nothing on the device reads the other stream's result buffers
// between that point and the signal, so no stale line of them can have been fetched in between.

template <class Op, class T>
__device__ __forceinline__ void tail_shrink_tile(const T* __restrict__ part, long long p0, long long p1, long long M,
                                                 T* __restrict__ out_row, long long col0, T* smem) {
  constexpr int TILE = TAIL_SHRINK_TILE, LANES = TAIL_SHRINK_LANES;
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

// the piece of block `block` (of t.blk0[t.n] blocks in all): task k, column tile, row chunk s
template <class T, int MAXT>
__device__ __forceinline__ void tail_shrink_block(const TailTasksT<MAXT>& t, int block, T* smem) {
  int k = 0;
  while (k + 1 < t.n && block >= t.blk0[k + 1]) k++;
  const int lb = block - t.blk0[k];
  const int S = t.S[k];
  const int s = lb % S;
  const long long col0 = (long long)(lb / S) * TAIL_SHRINK_TILE;
  const long long np = t.nparts[k], M = t.M[k];
  const long long chunk = (np + S - 1) / S;
  long long p0 = s * chunk, p1 = p0 + chunk;
  if (p1 > np) p1 = np;
  if (p0 > np) p0 = np;
  const T* part = (const T*)t.part[k];
  T* out_row = (T*)t.out[k] + (long long)s * M;
  switch (t.op[k]) {
    case 0: tail_shrink_tile<OpAdd, T>(part, p0, p1, M, out_row, col0, smem); break;
    case 1: tail_shrink_tile<OpMul, T>(part, p0, p1, M, out_row, col0, smem); break;
    case 2: tail_shrink_tile<OpMax, T>(part, p0, p1, M, out_row, col0, smem); break;
    default: tail_shrink_tile<OpMin, T>(part, p0, p1, M, out_row, col0, smem); break;
  }
}

}  // namespace pthip_dev
