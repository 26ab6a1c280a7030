// reduce.hip — CAReduce kernels (full / row / column / general (A,R,B) reductions).
//
// Replaces the reference's generated C loops for CAReduce
// (pytensor/tensor/elemwise.py:1520-1678, elemwise_cgen.py:467-761): sequential
// NpyIter accumulation there; here wave64 butterfly + LDS-staged workgroup reduction,
// two deterministic stages, accumulating in acc_dtype (elemwise.py:1383-1417).
//
// View of the problem: out[a,b] = reduce_{r<R} x[a*sA + r*sR + b*sB].
//   * "contig" kernel  : threads of a group cooperate along r (used when sR is the
//                        fast stride: full reductions, row sums);
//   * "strided" kernel : one thread per output, adjacent threads = adjacent b
//                        (coalesced when sB == 1: column sums), loop over r.
// When there are too few outputs to fill 256 CUs, R is split across workgroups and a
// second launch combines the partials (HBM-bound: bytes = R*A*B*itemsize read once).
#include "common.h"
#include "reduce_device.h"

using namespace pthip_dev;

namespace {

constexpr int BLOCK = 256;

// ---- contig: GROUP threads cooperate on one (output, split) --------------------------
// grid.x = n_out * nsplit groups (GROUP==256: one per block; GROUP==64: 4 per block)
template <class Op, class Tin, class Tacc, class Tout, int GROUP>
__global__ __launch_bounds__(BLOCK) void reduce_contig_kernel(
    const Tin* __restrict__ x, Tout* __restrict__ out, long long n_out, long long R, long long B,
    long long sA, long long sR, long long sB, long long nsplit, long long chunk) {
  __shared__ Tacc smem[BLOCK / 64];
  constexpr int GPB = BLOCK / GROUP;  // groups per block
  const long long g = (long long)blockIdx.x * GPB + (GROUP == BLOCK ? 0 : (threadIdx.x / GROUP));
  const int t = threadIdx.x % GROUP;
  const bool active = g < n_out * nsplit;
  const long long o = active ? g / nsplit : 0, s = active ? g % nsplit : 0;
  const long long a = o / B, b = o % B;
  const Tin* base = x + a * sA + b * sB;
  long long r0 = s * chunk, r1 = r0 + chunk;
  if (r1 > R) r1 = R;
  if (!active) r1 = r0;
  Tacc acc0 = Op::template identity<Tacc>(), acc1 = acc0, acc2 = acc0, acc3 = acc0;
  long long r = r0 + t;
  for (; r + 3 * GROUP < r1; r += 4 * GROUP) {
    Tacc v0 = (Tacc)base[r * sR];
    Tacc v1 = (Tacc)base[(r + GROUP) * sR];
    Tacc v2 = (Tacc)base[(r + 2 * GROUP) * sR];
    Tacc v3 = (Tacc)base[(r + 3 * GROUP) * sR];
    acc0 = Op::apply(acc0, v0);
    acc1 = Op::apply(acc1, v1);
    acc2 = Op::apply(acc2, v2);
    acc3 = Op::apply(acc3, v3);
  }
  for (; r < r1; r += GROUP) acc0 = Op::apply(acc0, (Tacc)base[r * sR]);
  Tacc acc = Op::apply(Op::apply(acc0, acc1), Op::apply(acc2, acc3));
  if constexpr (GROUP == BLOCK) {
    acc = block_reduce<Op, Tacc, BLOCK>(acc, smem);
    if (threadIdx.x == 0 && active) out[o * nsplit + s] = (Tout)acc;
  } else {
    acc = wave_reduce<Op>(acc);
    if (t == 0 && active) out[o * nsplit + s] = (Tout)acc;
  }
}

// ---- strided: one thread per output, loop over r ------------------------------------
// grid = (ceil(n_out/BLOCK), nsplit); partials laid out [split][n_out]
template <class Op, class Tin, class Tacc, class Tout>
__global__ __launch_bounds__(BLOCK) void reduce_strided_kernel(
    const Tin* __restrict__ x, Tout* __restrict__ out, long long n_out, long long R, long long B,
    long long sA, long long sR, long long sB, long long chunk) {
  const long long o = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (o >= n_out) return;
  const long long s = blockIdx.y;
  const long long a = o / B, b = o % B;
  const Tin* base = x + a * sA + b * sB;
  long long r0 = s * chunk, r1 = r0 + chunk;
  if (r1 > R) r1 = R;
  Tacc acc0 = Op::template identity<Tacc>(), acc1 = acc0, acc2 = acc0, acc3 = acc0;
  long long r = r0;
  for (; r + 3 < r1; r += 4) {
    Tacc v0 = (Tacc)base[r * sR];
    Tacc v1 = (Tacc)base[(r + 1) * sR];
    Tacc v2 = (Tacc)base[(r + 2) * sR];
    Tacc v3 = (Tacc)base[(r + 3) * sR];
    acc0 = Op::apply(acc0, v0);
    acc1 = Op::apply(acc1, v1);
    acc2 = Op::apply(acc2, v2);
    acc3 = Op::apply(acc3, v3);
  }
  for (; r < r1; r++) acc0 = Op::apply(acc0, (Tacc)base[r * sR]);
  out[s * n_out + o] = (Tout)Op::apply(Op::apply(acc0, acc1), Op::apply(acc2, acc3));
}

struct Plan {
  bool contig;          // which stage-1 kernel
  int group;            // 64 or 256 (contig)
  long long nsplit;     // R splits
  long long chunk;      // elements of R per split
};

Plan make_plan(long long A, long long R, long long B, long long sR, long long sB) {
  Plan p{};
  const long long n_out = A * B;
  const long long target_groups = (long long)pthip::kNumCU * 8;  // ≫256 workgroups
  // contiguous reduce axis, or a single output column: cooperate along r
  p.contig = (sR == 1 || sR == -1 || B == 1) && !(sB == 1 && B >= 64 && sR != 1);
  if (R <= 32 && n_out >= 256) p.contig = false;  // tiny rows: thread per output
  if (p.contig) {
    p.group = (R >= 2048) ? 256 : 64;
    const long long min_chunk = (long long)p.group * 8;
    long long want = (target_groups + n_out - 1) / (n_out ? n_out : 1);
    long long max_split = (R + min_chunk - 1) / min_chunk;
    p.nsplit = want < 1 ? 1 : want;
    if (p.nsplit > max_split) p.nsplit = max_split;
    if (p.nsplit < 1) p.nsplit = 1;
    if (p.nsplit > 2048) p.nsplit = 2048;
  } else {
    p.group = 1;
    long long blocks = (n_out + BLOCK - 1) / BLOCK;
    long long want = (target_groups / 4 + blocks - 1) / (blocks ? blocks : 1);
    long long max_split = (R + 63) / 64;
    p.nsplit = want < 1 ? 1 : want;
    if (p.nsplit > max_split) p.nsplit = max_split;
    if (p.nsplit < 1) p.nsplit = 1;
    if (p.nsplit > 1024) p.nsplit = 1024;
  }
  p.chunk = (R + p.nsplit - 1) / p.nsplit;
  if (p.chunk < 1) p.chunk = 1;
  return p;
}

template <class Op, class Tin, class Tacc, class Tout>
int launch_contig(hipStream_t st, int group, const Tin* x, Tout* out, long long n_out, long long R,
                  long long B, long long sA, long long sR, long long sB, long long nsplit,
                  long long chunk) {
  long long groups = n_out * nsplit;
  if (groups == 0) return 0;
  if (group == 256) {
    PTHIP_KLAUNCH((reduce_contig_kernel<Op, Tin, Tacc, Tout, 256>), dim3((unsigned)groups),
                       dim3(BLOCK), 0, st, x, out, n_out, R, B, sA, sR, sB, nsplit, chunk);
  } else {
    PTHIP_KLAUNCH((reduce_contig_kernel<Op, Tin, Tacc, Tout, 64>),
                       dim3((unsigned)((groups + 3) / 4)), dim3(BLOCK), 0, st, x, out, n_out, R, B,
                       sA, sR, sB, nsplit, chunk);
  }
  return pthip::post_launch("reduce_contig");
}

template <class Op, class Tin, class Tacc, class Tout>
int launch_strided(hipStream_t st, const Tin* x, Tout* out, long long n_out, long long R,
                   long long B, long long sA, long long sR, long long sB, long long nsplit,
                   long long chunk) {
  if (n_out == 0) return 0;
  PTHIP_KLAUNCH((reduce_strided_kernel<Op, Tin, Tacc, Tout>),
                     dim3((unsigned)((n_out + BLOCK - 1) / BLOCK), (unsigned)nsplit), dim3(BLOCK), 0,
                     st, x, out, n_out, R, B, sA, sR, sB, chunk);
  return pthip::post_launch("reduce_strided");
}

template <class Op, class Tin, class Tacc, class Tout>
int run(const void* xv, void* outv, long long A, long long R, long long B, long long sA,
        long long sR, long long sB, void* ws) {
  hipStream_t st = pthip::ctx().stream;
  const Tin* x = (const Tin*)xv;
  Tout* out = (Tout*)outv;
  const long long n_out = A * B;
  if (n_out == 0) return 0;
  Plan p = make_plan(A, R, B, sR, sB);
  if (p.nsplit == 1) {
    if (p.contig) return launch_contig<Op, Tin, Tacc, Tout>(st, p.group, x, out, n_out, R, B, sA, sR, sB, 1, p.chunk);
    return launch_strided<Op, Tin, Tacc, Tout>(st, x, out, n_out, R, B, sA, sR, sB, 1, p.chunk);
  }
  Tacc* part = (Tacc*)ws;
  if (p.contig) {
    // partials [n_out][nsplit]; stage 2: contiguous reduce over nsplit
    int r = launch_contig<Op, Tin, Tacc, Tacc>(st, p.group, x, part, n_out, R, B, sA, sR, sB, p.nsplit, p.chunk);
    if (r) return r;
    int g2 = p.nsplit >= 512 ? 256 : 64;
    return launch_contig<Op, Tacc, Tacc, Tout>(st, g2, part, out, n_out, p.nsplit, 1, p.nsplit, 1, 0, 1, p.nsplit);
  }
  // partials [nsplit][n_out]; stage 2: strided reduce over nsplit
  int r = launch_strided<Op, Tin, Tacc, Tacc>(st, x, part, n_out, R, B, sA, sR, sB, p.nsplit, p.chunk);
  if (r) return r;
  return launch_strided<Op, Tacc, Tacc, Tout>(st, part, out, n_out, p.nsplit, n_out, 0, n_out, 1, 1, p.nsplit);
}

// (input, accumulator, output) dtype triples the reference can produce (elemwise.py:1383-1417
// `_acc_dtype`: signed -> int64, unsigned -> uint64, bool -> int64, float16 -> float32,
// float32 -> float64; output = the accumulator dtype for Sum/Prod, the input dtype for a raw
// CAReduce, or whatever `dtype=` asked for among the float types).
#define PTHIP_RUN(TIN, TACC, TOUT) return run<Op, TIN, TACC, TOUT>(x, out, A, R, B, sA, sR, sB, ws)
#define PTHIP_CASE(IN, ACC, OUT, TIN, TACC, TOUT) \
  if (in == IN && acc == ACC && outd == OUT) PTHIP_RUN(TIN, TACC, TOUT);

// Add / Mul: accumulate wide, store wide or narrow (wrap-around cast, like the C backend's
// final `(out_dtype)acc`, elemwise.py:1668-1676)
template <class Op>
int dispatch_types(int in, int acc, int outd, const void* x, void* out, long long A, long long R,
                   long long B, long long sA, long long sR, long long sB, void* ws) {
  PTHIP_CASE(PTHIP_F64, PTHIP_F64, PTHIP_F64, double, double, double)
  PTHIP_CASE(PTHIP_F32, PTHIP_F64, PTHIP_F32, float, double, float)
  PTHIP_CASE(PTHIP_F64, PTHIP_F64, PTHIP_F32, double, double, float)
  PTHIP_CASE(PTHIP_F32, PTHIP_F64, PTHIP_F64, float, double, double)
  PTHIP_CASE(PTHIP_F32, PTHIP_F32, PTHIP_F32, float, float, float)
  PTHIP_CASE(PTHIP_F16, PTHIP_F32, PTHIP_F16, _Float16, float, _Float16)
  PTHIP_CASE(PTHIP_F16, PTHIP_F32, PTHIP_F32, _Float16, float, float)
  PTHIP_CASE(PTHIP_I64, PTHIP_I64, PTHIP_I64, long long, long long, long long)
  PTHIP_CASE(PTHIP_I32, PTHIP_I64, PTHIP_I64, int, long long, long long)
  PTHIP_CASE(PTHIP_I16, PTHIP_I64, PTHIP_I64, short, long long, long long)
  PTHIP_CASE(PTHIP_I8, PTHIP_I64, PTHIP_I64, signed char, long long, long long)
  PTHIP_CASE(PTHIP_BOOL, PTHIP_I64, PTHIP_I64, unsigned char, long long, long long)
  PTHIP_CASE(PTHIP_U8, PTHIP_I64, PTHIP_I64, unsigned char, long long, long long)
  PTHIP_CASE(PTHIP_I32, PTHIP_I64, PTHIP_I32, int, long long, int)
  PTHIP_CASE(PTHIP_I16, PTHIP_I64, PTHIP_I16, short, long long, short)
  PTHIP_CASE(PTHIP_I8, PTHIP_I64, PTHIP_I8, signed char, long long, signed char)
  PTHIP_CASE(PTHIP_U64, PTHIP_U64, PTHIP_U64, unsigned long long, unsigned long long, unsigned long long)
  PTHIP_CASE(PTHIP_U32, PTHIP_U64, PTHIP_U64, unsigned int, unsigned long long, unsigned long long)
  PTHIP_CASE(PTHIP_U16, PTHIP_U64, PTHIP_U64, unsigned short, unsigned long long, unsigned long long)
  PTHIP_CASE(PTHIP_U8, PTHIP_U64, PTHIP_U64, unsigned char, unsigned long long, unsigned long long)
  PTHIP_CASE(PTHIP_U32, PTHIP_U64, PTHIP_U32, unsigned int, unsigned long long, unsigned int)
  PTHIP_CASE(PTHIP_U16, PTHIP_U64, PTHIP_U16, unsigned short, unsigned long long, unsigned short)
  PTHIP_CASE(PTHIP_U8, PTHIP_U64, PTHIP_U8, unsigned char, unsigned long long, unsigned char)
  PTHIP_CASE(PTHIP_I32, PTHIP_I32, PTHIP_I32, int, int, int)
  PTHIP_CASE(PTHIP_I16, PTHIP_I16, PTHIP_I16, short, short, short)
  PTHIP_CASE(PTHIP_I8, PTHIP_I8, PTHIP_I8, signed char, signed char, signed char)
  PTHIP_CASE(PTHIP_U8, PTHIP_U8, PTHIP_U8, unsigned char, unsigned char, unsigned char)
  // second stage over accumulator-typed partials of a fused Elemwise+reduce kernel
  PTHIP_CASE(PTHIP_I64, PTHIP_I64, PTHIP_I32, long long, long long, int)
  PTHIP_CASE(PTHIP_I64, PTHIP_I64, PTHIP_I16, long long, long long, short)
  PTHIP_CASE(PTHIP_I64, PTHIP_I64, PTHIP_I8, long long, long long, signed char)
  PTHIP_CASE(PTHIP_U64, PTHIP_U64, PTHIP_U32, unsigned long long, unsigned long long, unsigned int)
  PTHIP_CASE(PTHIP_U64, PTHIP_U64, PTHIP_U16, unsigned long long, unsigned long long, unsigned short)
  PTHIP_CASE(PTHIP_U64, PTHIP_U64, PTHIP_U8, unsigned long long, unsigned long long, unsigned char)
  PTHIP_CASE(PTHIP_F32, PTHIP_F32, PTHIP_F16, float, float, _Float16)
  return pthip::set_error("pthip_reduce: unsupported dtype combination in=%d acc=%d out=%d", in, acc, outd);
}

// Maximum / Minimum: no widening is ever needed; the accumulator dtype the graph names is
// honoured for the float pairs, every other type reduces in itself
template <class Op>
int dispatch_minmax(int in, int acc, int outd, const void* x, void* out, long long A, long long R,
                    long long B, long long sA, long long sR, long long sB, void* ws) {
  PTHIP_CASE(PTHIP_F64, PTHIP_F64, PTHIP_F64, double, double, double)
  PTHIP_CASE(PTHIP_F32, PTHIP_F32, PTHIP_F32, float, float, float)
  PTHIP_CASE(PTHIP_F32, PTHIP_F64, PTHIP_F32, float, float, float)
  PTHIP_CASE(PTHIP_F16, PTHIP_F16, PTHIP_F16, _Float16, _Float16, _Float16)
  PTHIP_CASE(PTHIP_F16, PTHIP_F32, PTHIP_F16, _Float16, _Float16, _Float16)
  PTHIP_CASE(PTHIP_F64, PTHIP_F64, PTHIP_F32, double, double, float)
  // second stage over accumulator-typed partials of a fused Elemwise+reduce kernel
  PTHIP_CASE(PTHIP_I64, PTHIP_I64, PTHIP_I32, long long, long long, int)
  PTHIP_CASE(PTHIP_I64, PTHIP_I64, PTHIP_I16, long long, long long, short)
  PTHIP_CASE(PTHIP_I64, PTHIP_I64, PTHIP_I8, long long, long long, signed char)
  PTHIP_CASE(PTHIP_U64, PTHIP_U64, PTHIP_U32, unsigned long long, unsigned long long, unsigned int)
  PTHIP_CASE(PTHIP_U64, PTHIP_U64, PTHIP_U16, unsigned long long, unsigned long long, unsigned short)
  PTHIP_CASE(PTHIP_U64, PTHIP_U64, PTHIP_U8, unsigned long long, unsigned long long, unsigned char)
  PTHIP_CASE(PTHIP_F32, PTHIP_F32, PTHIP_F16, float, float, _Float16)
  if (in == outd) {
    switch (in) {
      case PTHIP_I64: PTHIP_RUN(long long, long long, long long);
      case PTHIP_I32: PTHIP_RUN(int, int, int);
      case PTHIP_I16: PTHIP_RUN(short, short, short);
      case PTHIP_I8: PTHIP_RUN(signed char, signed char, signed char);
      case PTHIP_U64: PTHIP_RUN(unsigned long long, unsigned long long, unsigned long long);
      case PTHIP_U32: PTHIP_RUN(unsigned int, unsigned int, unsigned int);
      case PTHIP_U16: PTHIP_RUN(unsigned short, unsigned short, unsigned short);
      case PTHIP_U8: PTHIP_RUN(unsigned char, unsigned char, unsigned char);
      default: break;
    }
  }
  return pthip::set_error("pthip_reduce: unsupported dtype combination in=%d acc=%d out=%d", in, acc, outd);
}
#undef PTHIP_CASE

template <class Op>
int dispatch_bool(int in, int acc, int outd, const void* x, void* out, long long A, long long R,
                  long long B, long long sA, long long sR, long long sB, void* ws) {
  if (in == PTHIP_BOOL && acc == PTHIP_BOOL && outd == PTHIP_BOOL)
    return run<Op, bool, bool, bool>(x, out, A, R, B, sA, sR, sB, ws);
#define CASE(DT, T) \
  if (in == DT && acc == DT && outd == DT) return run<Op, T, T, T>(x, out, A, R, B, sA, sR, sB, ws);
  CASE(PTHIP_I64, long long)
  CASE(PTHIP_I32, int)
  CASE(PTHIP_I16, short)
  CASE(PTHIP_I8, signed char)
  CASE(PTHIP_U8, unsigned char)
  CASE(PTHIP_U16, unsigned short)
  CASE(PTHIP_U32, unsigned int)
  CASE(PTHIP_U64, unsigned long long)
#undef CASE
  return pthip::set_error("pthip_reduce: unsupported dtype combination in=%d acc=%d out=%d", in, acc, outd);
}
#undef PTHIP_RUN

}  // namespace

extern "C" {

size_t pthip_reduce_workspace(int acc_dtype, int64_t A, int64_t R, int64_t B) {
  // upper bound independent of strides: both plans are bounded by 2048 splits
  long long n_out = A * B;
  if (n_out == 0 || R == 0) return 0;
  long long ns = 2048;
  long long max_split = (R + 63) / 64;
  if (ns > max_split) ns = max_split;
  if (ns <= 1) return 0;
  return (size_t)n_out * (size_t)ns * (size_t)pthip::dtype_size(acc_dtype);
}

int pthip_reduce(int op, int in_dtype, int acc_dtype, int out_dtype, const void* x, void* out,
                 int64_t A, int64_t R, int64_t B, int64_t sA, int64_t sR, int64_t sB, void* ws,
                 size_t ws_bytes) {
  PTHIP_REQUIRE_INIT();
  if (A < 0 || R < 0 || B < 0) return pthip::set_error("pthip_reduce: negative extent");
  if (ws_bytes < pthip_reduce_workspace(acc_dtype, A, R, B))
    return pthip::set_error("pthip_reduce: workspace too small");
  switch (op) {
    case PTHIP_RED_ADD: return dispatch_types<OpAdd>(in_dtype, acc_dtype, out_dtype, x, out, A, R, B, sA, sR, sB, ws);
    case PTHIP_RED_MUL: return dispatch_types<OpMul>(in_dtype, acc_dtype, out_dtype, x, out, A, R, B, sA, sR, sB, ws);
    case PTHIP_RED_MAX:
      if (R == 0) return pthip::set_error("zero-size array to reduction operation maximum which has no identity");
      if (in_dtype == PTHIP_BOOL) return dispatch_bool<OpOr>(in_dtype, acc_dtype, out_dtype, x, out, A, R, B, sA, sR, sB, ws);
      return dispatch_minmax<OpMax>(in_dtype, acc_dtype, out_dtype, x, out, A, R, B, sA, sR, sB, ws);
    case PTHIP_RED_MIN:
      if (R == 0) return pthip::set_error("zero-size array to reduction operation minimum which has no identity");
      if (in_dtype == PTHIP_BOOL) return dispatch_bool<OpAnd>(in_dtype, acc_dtype, out_dtype, x, out, A, R, B, sA, sR, sB, ws);
      return dispatch_minmax<OpMin>(in_dtype, acc_dtype, out_dtype, x, out, A, R, B, sA, sR, sB, ws);
    case PTHIP_RED_AND: return dispatch_bool<OpAnd>(in_dtype, acc_dtype, out_dtype, x, out, A, R, B, sA, sR, sB, ws);
    case PTHIP_RED_OR: return dispatch_bool<OpOr>(in_dtype, acc_dtype, out_dtype, x, out, A, R, B, sA, sR, sB, ws);
    case PTHIP_RED_XOR: return dispatch_bool<OpXor>(in_dtype, acc_dtype, out_dtype, x, out, A, R, B, sA, sR, sB, ws);
  }
  return pthip::set_error("pthip_reduce: unknown op %d", op);
}

}  // extern "C"
