// gemm.hip — Gemm / Dot22 / Dot22Scalar / BatchedDot on MFMA tiles (gfx950).
//
// Reference: Gemm.perform z <- beta*z + alpha*x@y (pytensor/tensor/blas/gemm.py:183-216),
// Dot22 (248-285), Dot22Scalar (298+), BatchedDot (batched.py:18-79); the C glue picks
// N/T flags from strides (c_code/codegen.py:159-250) — here the same four layout cases
// are template instances.
//
// fp64: v_mfma_f64_16x16x4_f64   (A: lane l -> A[l&15][l>>4]; B: B[l>>4][l&15];
//                                 D reg r -> row (l>>4)+4r, col l&15)
// fp32: v_mfma_f32_32x32x2_f32   (A: A[l&31][l>>5]; B: B[l>>5][l&31];
//                                 D reg r -> row (r&3)+8(r>>2)+4(l>>5), col l&31)
// Exact IEEE fma chains in both cases (no reduced-precision path exists on gfx950).
//
// Tiling: 256 threads = 4 waves (2x2), block tile 128x128, wave tile 64x64, BK = 16,
// double-buffered LDS, next tile prefetched global->registers while the current one
// is multiplied (one barrier per K step).  LDS images are chosen per operand layout so
// that both the 16-byte staging writes and the fragment reads are bank-conflict free:
//   K-contiguous operand  -> image [row][k], leading dimension 18 elements
//   M/N-contiguous operand-> image [k][row], leading dimension 144 (f64) / 128 (f32)
// Tile ids are remapped so that the 8 XCDs (block b runs on XCD b%8) each walk a
// contiguous band of tile rows (L2 reuse of the A panel).
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int BM = 128, BN = 128, BK = 16;

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------
// staging: global -> registers -> LDS
// ---------------------------------------------------------------------------------
// An operand tile is ROWS(=128) x BK(=16) elements, `row` along M (for A) or N (for B).
// KC (K-contiguous in global memory): element (row, k) at base + row*ld + k.
// else (row-contiguous):               element (row, k) at base + k*ld + row.
template <class T, bool KC, int BKT = 16> struct Stage;  // BKT: K extent of a staged tile

// ROWS = 128 (the 128 x 128 tile) or 64 (the 64 x 64 tile of mid-size products, see dgemm_kernel)
template <bool KC, int ROWS> struct StageD {
  static constexpr int NV = ROWS * 16 / (2 * BLOCK);  // 16-byte vectors per thread per tile (4 / 2)
  static constexpr int LD = KC ? 18 : ROWS + 16;      // LDS leading dimension (elements)
  static constexpr int SIZE = KC ? ROWS * 18 : 16 * (ROWS + 16);
  static constexpr int RP = ROWS / 2;                 // 16-byte pairs along a row-contiguous k-row
  double2 v[NV];
  __device__ __forceinline__ void load(const double* __restrict__ base, long long ld,
                                       long long row0, long long k0, long long rows,
                                       long long K, bool vec_ok) {
    if (vec_ok && row0 + ROWS <= rows && k0 + BK <= K) {
      // interior tile (workgroup-uniform test): straight 16-byte loads, no per-vector branches
#pragma unroll
      for (int p = 0; p < NV; p++) {
        const int id = threadIdx.x + p * BLOCK;
        if constexpr (KC) v[p] = *(const double2*)(base + (row0 + (id >> 3)) * ld + k0 + (id & 7) * 2);
        else v[p] = *(const double2*)(base + (k0 + id / RP) * ld + row0 + (id % RP) * 2);
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < NV; p++) {
      const int id = threadIdx.x + p * BLOCK;
      if constexpr (KC) {
        const int r = id >> 3, kv = (id & 7) * 2;
        const long long row = row0 + r, k = k0 + kv;
        const double* g = base + row * ld + k;
        if (row < rows && k + 1 < K && vec_ok) v[p] = *(const double2*)g;
        else {
          v[p].x = (row < rows && k < K) ? g[0] : 0.0;
          v[p].y = (row < rows && k + 1 < K) ? g[1] : 0.0;
        }
      } else {
        const int kk = id / RP, rv = (id % RP) * 2;
        const long long row = row0 + rv, k = k0 + kk;
        const double* g = base + k * ld + row;
        if (k < K && row + 1 < rows && vec_ok) v[p] = *(const double2*)g;
        else {
          v[p].x = (k < K && row < rows) ? g[0] : 0.0;
          v[p].y = (k < K && row + 1 < rows) ? g[1] : 0.0;
        }
      }
    }
  }
  __device__ __forceinline__ void store(double* __restrict__ s) const {
#pragma unroll
    for (int p = 0; p < NV; p++) {
      const int id = threadIdx.x + p * BLOCK;
      if constexpr (KC) {
        const int r = id >> 3, kv = (id & 7) * 2;
        *(double2*)(s + r * LD + kv) = v[p];
      } else {
        const int kk = id / RP, rv = (id % RP) * 2;
        *(double2*)(s + kk * LD + rv) = v[p];
      }
    }
  }
  // fragment element for MFMA 16x16x4: row = r0 + (l&15), k = kk*4 + (l>>4)
  static __device__ __forceinline__ double frag(const double* __restrict__ s, int r0, int kk,
                                                int lane) {
    if constexpr (KC) return s[(r0 + (lane & 15)) * LD + kk * 4 + (lane >> 4)];
    else return s[(kk * 4 + (lane >> 4)) * LD + r0 + (lane & 15)];
  }
};
template <bool KC> struct Stage<double, KC, 16> : StageD<KC, 128> {};

template <bool KC, int BKT> struct Stage<float, KC, BKT> {
  static constexpr int VPR = BKT / 4;          // 16-byte vectors per tile row (K-contiguous image)
  static constexpr int NV = BKT / 8;
  static constexpr int LD = KC ? BKT + 2 : 128;  // 18 / 34: float2 fragment reads hit 32 distinct banks per 16 lanes
  static constexpr int SIZE = KC ? 128 * (BKT + 2) : BKT * 128;
  float4 v[NV];
  __device__ __forceinline__ void load(const float* __restrict__ base, long long ld,
                                       long long row0, long long k0, long long rows,
                                       long long K, bool vec_ok) {
    if (vec_ok && row0 + 128 <= rows && k0 + BKT <= K) {
      // interior tile (workgroup-uniform test): straight 16-byte loads, no per-vector branches
#pragma unroll
      for (int p = 0; p < NV; p++) {
        const int id = threadIdx.x + p * BLOCK;
        if constexpr (KC) v[p] = *(const float4*)(base + (row0 + (id / VPR)) * ld + k0 + (id % VPR) * 4);
        else v[p] = *(const float4*)(base + (k0 + (id >> 5)) * ld + row0 + (id & 31) * 4);
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < NV; p++) {
      const int id = threadIdx.x + p * BLOCK;
      long long row, k, se;  // se: element stride between the 4 vector components
      const float* g;
      bool full;
      if constexpr (KC) {
        const int r = id / VPR, kv = (id % VPR) * 4;
        row = row0 + r; k = k0 + kv;
        g = base + row * ld + k;
        full = row < rows && k + 3 < K;
        if (full && vec_ok) { v[p] = *(const float4*)g; continue; }
        v[p].x = (row < rows && k < K) ? g[0] : 0.f;
        v[p].y = (row < rows && k + 1 < K) ? g[1] : 0.f;
        v[p].z = (row < rows && k + 2 < K) ? g[2] : 0.f;
        v[p].w = (row < rows && k + 3 < K) ? g[3] : 0.f;
      } else {
        const int kk = id >> 5, rv = (id & 31) * 4;
        row = row0 + rv; k = k0 + kk;
        g = base + k * ld + row;
        full = k < K && row + 3 < rows;
        if (full && vec_ok) { v[p] = *(const float4*)g; continue; }
        v[p].x = (k < K && row < rows) ? g[0] : 0.f;
        v[p].y = (k < K && row + 1 < rows) ? g[1] : 0.f;
        v[p].z = (k < K && row + 2 < rows) ? g[2] : 0.f;
        v[p].w = (k < K && row + 3 < rows) ? g[3] : 0.f;
      }
      (void)se;
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ s) const {
#pragma unroll
    for (int p = 0; p < NV; p++) {
      const int id = threadIdx.x + p * BLOCK;
      if constexpr (KC) {
        const int r = id / VPR, kv = (id % VPR) * 4;
        float2* d = (float2*)(s + r * LD + kv);  // LD=18: 8-byte aligned rows
        d[0] = make_float2(v[p].x, v[p].y);
        d[1] = make_float2(v[p].z, v[p].w);
      } else {
        const int kk = id >> 5, rv = (id & 31) * 4;
        *(float4*)(s + kk * LD + rv) = v[p];
      }
    }
  }
  // two fragment elements for MFMA 32x32x2 pair s=0,1: row = r0 + (l&31),
  // k = kk*4 + 2*(l>>5) + s   (same k assignment for A and B)
  static __device__ __forceinline__ float2 frag2(const float* __restrict__ s, int r0, int kk,
                                                 int lane) {
    const int h = lane >> 5, i = lane & 31;
    if constexpr (KC) return *(const float2*)(s + (r0 + i) * LD + kk * 4 + 2 * h);
    else return make_float2(s[(kk * 4 + 2 * h) * LD + r0 + i], s[(kk * 4 + 2 * h + 1) * LD + r0 + i]);
  }
};

__device__ __forceinline__ void tile_coords(long long tiles_m, long long tiles_n, long long& tm,
                                            long long& tn, long long& bz) {
  // XCD-aware remap over the whole (batch x tile) space: workgroups are dispatched x-fastest and
  // block L runs on XCD L % 8, so XCD x gets the contiguous range [x*total/8, (x+1)*total/8) of
  // (batch, tile) pairs — the tiles of one matrix, and the tiles of one band of a big matrix,
  // share an L2.  (Round 1 remapped the tiles of one matrix only: the four 128x128 tiles of a
  // 256x256 batch item landed on four XCDs and every operand was fetched from HBM twice.)
  const long long nt = tiles_m * tiles_n;
  long long pid = blockIdx.x + nt * (long long)blockIdx.z;
  const long long total = nt * (long long)gridDim.z;
  if (total % 8 == 0) {
    const long long per = total / 8;
    pid = (pid % 8) * per + pid / 8;
  }
  bz = pid / nt;
  pid -= bz * nt;
  tm = pid / tiles_n;
  tn = pid % tiles_n;
}

// ---------------------------------------------------------------------------------
// fp64 kernel
// ---------------------------------------------------------------------------------
// SKINNY (M <= 64): the four waves split N (wave tile 64 x 32) so no MFMA is spent on
// padding rows.  gridDim.y > 1 = split-K: block y covers K range [y*kchunk, (y+1)*kchunk) and
// writes its raw accumulators to partial slab y of `out` (combined by splitk_finish_kernel).
// TILE = 64: 64 x 64 output tiles (wave tile 32 x 32) for products whose 128 x 128 tiling leaves most of the 256 CUs
// without a workgroup (the M = 512 solve steps of the blocked triangular solve, 1024^2 products: 64 tiles) — four times
// the workgroups, full K each, instead of split-K slabs and a finishing pass over them.
template <bool AKC, bool BKC, bool SKINNY, int TILE = 128>
__global__ __launch_bounds__(BLOCK, 2) void dgemm_kernel(
    double* __restrict__ out, const double* __restrict__ A, const double* __restrict__ B,
    const double* __restrict__ C, long long M, long long N, long long K, long long lda,
    long long ldb, long long sAb, long long sBb, long long sCb, long long sC0, long long sC1,
    double alpha, double beta, long long tiles_m, long long tiles_n, int vecA, int vecB,
    long long kchunk, long long ldo) {
  using SA = StageD<AKC, TILE>;
  using SB = StageD<BKC, TILE>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* As = (double*)smem_raw;                // [2][SA::SIZE]
  double* Bs = As + 2 * SA::SIZE;                // [2][SB::SIZE]
  long long tm, tn;
  long long bz;
  tile_coords(tiles_m, tiles_n, tm, tn, bz);
  const long long m0 = tm * TILE, n0 = tn * TILE;
  A += bz * sAb;
  B += bz * sBb;
  const bool split = gridDim.y > 1;
  out += ((long long)blockIdx.y * gridDim.z + bz) * M * N;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  static_assert(TILE == 128 || (TILE == 64 && !SKINNY), "tile shapes: 128 x 128 (skinny: 64 x 128) or 64 x 64");
  constexpr int NI = TILE == 128 ? 4 : 2;
  constexpr int NJ = TILE == 128 ? (SKINNY ? 2 : 4) : 2;
  constexpr int WT = TILE / 2;  // wave tile edge of the 2 x 2 wave layout
  const int wm0 = SKINNY ? 0 : (w >> 1) * WT, wn0 = SKINNY ? w * 32 : (w & 1) * WT;
  double4_t acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  SA sa;
  SB sb;
  const long long kb = (long long)blockIdx.y * kchunk;
  const long long Kend = (kb + kchunk < K) ? kb + kchunk : K;
  const long long nk = (Kend - kb + BK - 1) / BK;
  sa.load(A, lda, m0, kb, M, Kend, vecA);
  sb.load(B, ldb, n0, kb, N, Kend, vecB);
  sa.store(As);
  sb.store(Bs);
  __syncthreads();
  for (long long kt = 0; kt < nk; kt++) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      sa.load(A, lda, m0, kb + (kt + 1) * BK, M, Kend, vecA);
      sb.load(B, ldb, n0, kb + (kt + 1) * BK, N, Kend, vecB);
    }
    const double* as = As + cur * SA::SIZE;
    const double* bs = Bs + cur * SB::SIZE;
#pragma unroll
    for (int kk = 0; kk < BK / 4; kk++) {
      double af[NI], bf[NJ];
#pragma unroll
      for (int i = 0; i < NI; i++) af[i] = SA::frag(as, wm0 + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < NJ; j++) bf[j] = SB::frag(bs, wn0 + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      sa.store(As + (cur ^ 1) * SA::SIZE);
      sb.store(Bs + (cur ^ 1) * SB::SIZE);
    }
    __syncthreads();
  }
  // epilogue: D reg r -> row (l>>4) + 4r, col l&15
  const bool has_c = !split && (beta != 0.0) && C != nullptr;
  if (has_c) C += bz * sCb;
  if (split) alpha = 1.0;
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const long long row = m0 + wm0 + i * 16 + (lane >> 4) + 4 * r;
        const long long col = n0 + wn0 + j * 16 + (lane & 15);
        if (row < M && col < N) {
          double v = alpha * acc[i][j][r];
          if (has_c) v += beta * C[row * sC0 + col * sC1];
          out[row * ldo + col] = v;
        }
      }
}

// ---------------------------------------------------------------------------------
// fp32 kernel
// ---------------------------------------------------------------------------------
template <bool AKC, bool BKC, bool SKINNY, int BKT>
// (PMC, 4096^3: waves parked at barriers / s_waitcnt 37 % of their cycles vs 11 % in the fp64
//  kernel — an fp32 step has half the MFMA time to hide the same latencies.  Tried, measured,
//  rejected: three workgroups per CU (same 101 TFLOP/s); two BK steps per barrier interval
//  with four LDS stages (88 TFLOP/s: the extra staging registers cost more than the barriers).)
__global__ __launch_bounds__(BLOCK, 2) void sgemm_kernel(
    float* __restrict__ out, const float* __restrict__ A, const float* __restrict__ B,
    const float* __restrict__ C, long long M, long long N, long long K, long long lda,
    long long ldb, long long sAb, long long sBb, long long sCb, long long sC0, long long sC1,
    float alpha, float beta, long long tiles_m, long long tiles_n, int vecA, int vecB,
    long long kchunk, long long ldo) {
  using SA = Stage<float, AKC, BKT>;
  using SB = Stage<float, BKC, BKT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* As = (float*)smem_raw;
  float* Bs = As + 2 * SA::SIZE;
  long long tm, tn;
  long long bz;
  tile_coords(tiles_m, tiles_n, tm, tn, bz);
  const long long m0 = tm * BM, n0 = tn * BN;
  A += bz * sAb;
  B += bz * sBb;
  const bool split = gridDim.y > 1;
  out += ((long long)blockIdx.y * gridDim.z + bz) * M * N;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int NJ = SKINNY ? 1 : 2;
  const int wm0 = SKINNY ? 0 : (w >> 1) * 64, wn0 = SKINNY ? w * 32 : (w & 1) * 64;
  float16_t acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  SA sa;
  SB sb;
  const long long kb = (long long)blockIdx.y * kchunk;
  const long long Kend = (kb + kchunk < K) ? kb + kchunk : K;
  const long long nk = (Kend - kb + BKT - 1) / BKT;
  auto compute = [&](const float* as, const float* bs) {
#pragma unroll
    for (int kk = 0; kk < BKT / 4; kk++) {
      float2 af[2], bf[NJ];
#pragma unroll
      for (int i = 0; i < 2; i++) af[i] = SA::frag2(as, wm0 + i * 32, kk, lane);
#pragma unroll
      for (int j = 0; j < NJ; j++) bf[j] = SB::frag2(bs, wn0 + j * 32, kk, lane);
      // two passes so that consecutive MFMAs never share an accumulator (a dependent
      // back-to-back pair stalls the matrix pipe: 98 -> 102 TFLOP/s at 4096^3)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
    }
  };
  constexpr int PRE = 4;
  if (SKINNY && nk <= PRE) {
    // a split-K slice of a skinny product is a few BK steps long: every step's global loads
    // are issued up front (one HBM/L2 latency for the whole workgroup instead of one per
    // step — there is almost no MFMA work per step to hide them behind)
    SA sap[PRE];
    SB sbp[PRE];
#pragma unroll
    for (int t = 0; t < PRE; t++)
      if (t < nk) {
        sap[t].load(A, lda, m0, kb + t * BKT, M, Kend, vecA);
        sbp[t].load(B, ldb, n0, kb + t * BKT, N, Kend, vecB);
      }
    sap[0].store(As);
    sbp[0].store(Bs);
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < PRE; kt++)
      if (kt < nk) {
        const int cur = kt & 1;
        compute(As + cur * SA::SIZE, Bs + cur * SB::SIZE);
        if (kt + 1 < PRE && kt + 1 < nk) {
          sap[kt + 1 < PRE ? kt + 1 : 0].store(As + (cur ^ 1) * SA::SIZE);
          sbp[kt + 1 < PRE ? kt + 1 : 0].store(Bs + (cur ^ 1) * SB::SIZE);
        }
        __syncthreads();
      }
  } else {
    sa.load(A, lda, m0, kb, M, Kend, vecA);
    sb.load(B, ldb, n0, kb, N, Kend, vecB);
    sa.store(As);
    sb.store(Bs);
    __syncthreads();
    for (long long kt = 0; kt < nk; kt++) {
      const int cur = kt & 1;
      if (kt + 1 < nk) {
        sa.load(A, lda, m0, kb + (kt + 1) * BKT, M, Kend, vecA);
        sb.load(B, ldb, n0, kb + (kt + 1) * BKT, N, Kend, vecB);
      }
      compute(As + cur * SA::SIZE, Bs + cur * SB::SIZE);
      if (kt + 1 < nk) {
        sa.store(As + (cur ^ 1) * SA::SIZE);
        sb.store(Bs + (cur ^ 1) * SB::SIZE);
      }
      __syncthreads();
    }
  }
  const bool has_c = !split && (beta != 0.f) && C != nullptr;
  if (has_c) C += bz * sCb;
  if (split) alpha = 1.f;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const long long row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const long long col = n0 + wn0 + j * 32 + (lane & 31);
        if (row < M && col < N) {
          float v = alpha * acc[i][j][r];
          if (has_c) v += beta * C[row * sC0 + col * sC1];
          out[row * ldo + col] = v;
        }
      }
}

// ---------------------------------------------------------------------------------
// fp32, persistent tiles
// ---------------------------------------------------------------------------------
// Two workgroups per CU walk the (batch x tile) space; the first K-tile of a workgroup's NEXT
// output tile is requested during the last K-step of the current one, and the epilogue stores of a
// tile overlap the first MFMA steps of the next.  With one launch-wide wave of identical
// workgroups every prologue (two global-load latencies before the first MFMA) and every epilogue
// (64 KB of stores) of the co-resident workgroups coincide — nothing to overlap them with; at
// K = 256 (8 K-steps per tile) that is a quarter of a tile's life (config #3 BatchedDot:
// 512 x 256^3).  No split-K, M > 64 only (the one-tile-per-workgroup kernel above keeps those).
// Tried and rejected: global loads two K-steps ahead with a second staging register set — 256 VGPRs
// with spills, 5-15 % slower on every shape (profiles/r2w_sgemm_pf2.txt).
// DIAG (timing experiments only, results invalid when != 0; PTHIP_SGEMM_DIAG): 1 = no global loads
// after the first K-tile, 2 = also no LDS staging stores, 3 = also no LDS fragment reads (operands
// from registers) — what the MFMA loop sustains with each feeder removed.
template <bool AKC, bool BKC, int DIAG = 0>
__global__ __launch_bounds__(BLOCK, 2) void sgemm_persistent_kernel(
    float* __restrict__ out, const float* __restrict__ A, const float* __restrict__ B,
    const float* __restrict__ C, long long M, long long N, long long K, long long lda,
    long long ldb, long long sAb, long long sBb, long long sCb, long long sC0, long long sC1,
    float alpha, float beta, long long tiles_m, long long tiles_n, int vecA, int vecB,
    long long batch, long long ldo) {
  constexpr int BKT = 32;
  using SA = Stage<float, AKC, BKT>;
  using SB = Stage<float, BKC, BKT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* As = (float*)smem_raw;
  float* Bs = As + 2 * SA::SIZE;
  const long long nt = tiles_m * tiles_n, total = nt * batch;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm0 = (w >> 1) * 64, wn0 = (w & 1) * 64;
  const long long nk = (K + BKT - 1) / BKT;
  auto coords = [&](long long L, long long& m0, long long& n0, long long& bz) {
    long long pid = L;
    if (total % 8 == 0) {  // XCD x walks the contiguous range [x*total/8, (x+1)*total/8)
      const long long per = total / 8;
      pid = (L % 8) * per + L / 8;
    }
    bz = pid / nt;
    pid -= bz * nt;
    m0 = (pid / tiles_n) * BM;
    n0 = (pid % tiles_n) * BN;
  };
  long long L = blockIdx.x;
  if (L >= total) return;
  long long m0, n0, bz;
  coords(L, m0, n0, bz);
  SA sa;
  SB sb;
  sa.load(A + bz * sAb, lda, m0, 0, M, K, vecA);
  sb.load(B + bz * sBb, ldb, n0, 0, N, K, vecB);
  sa.store(As);
  sb.store(Bs);
  __syncthreads();
  int buf = 0;
  const bool has_c = (beta != 0.f) && C != nullptr;
  for (;;) {
    const long long Ln = L + gridDim.x;
    const bool has_next = Ln < total;
    long long m1 = 0, n1 = 0, b1 = 0;
    if (has_next) coords(Ln, m1, n1, b1);
    float16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const float* Ab = A + bz * sAb;
    const float* Bb = B + bz * sBb;
    for (long long kt = 0; kt < nk; kt++) {
      const bool more = kt + 1 < nk;
      if constexpr (DIAG == 0) {
        if (more) {
          sa.load(Ab, lda, m0, (kt + 1) * BKT, M, K, vecA);
          sb.load(Bb, ldb, n0, (kt + 1) * BKT, N, K, vecB);
        } else if (has_next) {  // the next tile's first K-tile, behind this tile's last MFMA step
          sa.load(A + b1 * sAb, lda, m1, 0, M, K, vecA);
          sb.load(B + b1 * sBb, ldb, n1, 0, N, K, vecB);
        }
      }
      const float* as = As + buf * SA::SIZE;
      const float* bs = Bs + buf * SB::SIZE;
      // fragment reads one kk step AHEAD of the MFMAs that use them (two register sets): the
      // straightforward loop compiled to ds_read -> s_waitcnt lgkmcnt(0) -> MFMA for every group,
      // ~100 exposed LDS-latency cycles per 2-4 MFMAs (r2t diagnostic: operands from registers
      // instead of LDS ran 138 vs 107 TFLOP/s at 4096^3)
      float2 af[2][2], bf[2][2];
      if constexpr (DIAG >= 3) {
#pragma unroll
        for (int i = 0; i < 2; i++) { af[0][i] = make_float2(sa.v[0].x + i, sa.v[0].y); bf[0][i] = make_float2(sb.v[0].x + i, sb.v[0].y); }
      } else {
#pragma unroll
        for (int i = 0; i < 2; i++) af[0][i] = SA::frag2(as, wm0 + i * 32, 0, lane);
#pragma unroll
        for (int j = 0; j < 2; j++) bf[0][j] = SB::frag2(bs, wn0 + j * 32, 0, lane);
      }
#pragma unroll
      for (int kk = 0; kk < BKT / 4; kk++) {
        const int cs = kk & 1, ns = cs ^ 1;
        if (kk + 1 < BKT / 4) {
          if constexpr (DIAG >= 3) {
#pragma unroll
            for (int i = 0; i < 2; i++) { af[ns][i] = make_float2(af[cs][i].y, af[cs][i].x + kk); bf[ns][i] = make_float2(bf[cs][i].y, bf[cs][i].x - kk); }
          } else {
#pragma unroll
            for (int i = 0; i < 2; i++) af[ns][i] = SA::frag2(as, wm0 + i * 32, kk + 1, lane);
#pragma unroll
            for (int j = 0; j < 2; j++) bf[ns][j] = SB::frag2(bs, wn0 + j * 32, kk + 1, lane);
          }
        }
        // (without the fences the machine scheduler sinks each read back next to its use)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cs][i].x, bf[cs][j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cs][i].y, bf[cs][j].y, acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if ((more || has_next) && DIAG < 2) {
        sa.store(As + (buf ^ 1) * SA::SIZE);
        sb.store(Bs + (buf ^ 1) * SB::SIZE);
      }
      __syncthreads();
      buf ^= 1;
    }
    float* ob = out + bz * M * N;
    const float* Cb = has_c ? C + bz * sCb : nullptr;
    if (!has_c && m0 + BM <= M && n0 + BN <= N) {
      // interior tile, no C operand (workgroup-uniform test): 64 unconditional stores per lane
      // instead of 64 exec-masked branches
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const long long row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const long long col = n0 + wn0 + j * 32 + (lane & 31);
            ob[row * ldo + col] = alpha * acc[i][j][r];
          }
    } else {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const long long row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const long long col = n0 + wn0 + j * 32 + (lane & 31);
            if (row < M && col < N) {
              float v = alpha * acc[i][j][r];
              if (has_c) v += beta * Cb[row * sC0 + col * sC1];
              ob[row * ldo + col] = v;
            }
          }
    }
    if (!has_next) break;
    L = Ln;
    m0 = m1;
    n0 = n1;
    bz = b1;
  }
}

// ---------------------------------------------------------------------------------
// fp32, 256 x 256 output tile per workgroup (round 3): row-major A (K-contiguous) and row-major B
// (N-contiguous), M and N multiples of 256, K a multiple of 16.  The short-K batched shape of
// BASELINE config #3 (512 x 256^3) is ONE tile per matrix: A and B cross L2->LDS once instead of
// twice, each wave owns 128 x 128 = 4 x 4 MFMA tiles so that a k-step is 8 LDS fragment reads for
// 32 MFMAs (the 128 x 128 kernel: 4 reads for 8), and the 64 stores per lane of the epilogue are
// amortised over four times the MFMA work.  One workgroup per CU (512 registers per lane: 256
// accumulators), persistent over (batch, tile) pairs; the next tile's first K-step is loaded behind
// the last MFMA step of the current one and is already in LDS while the epilogue stores.
// ---------------------------------------------------------------------------------
constexpr int T2 = 256, T2K = 16, T2ALD = T2K + 2;
// Round 4: every operand orientation (the reference's stride -> transpose-flag dispatch,
// pytensor/tensor/blas/c_code/codegen.py:159-250).  An operand whose K axis is contiguous (AKC: row-major A;
// BKC: B stored N x K) is fetched as float4 along K into the padded [256][18] LDS image and read back as float2
// fragments; one whose 256-wide axis is contiguous (A stored K x M; row-major B) is fetched as float4 along that
// axis into the [16][256] image and read back as two scalars per fragment — the two layouts round 3 used for A and B
// respectively, now chosen per operand.
template <bool PF, bool AKC, bool BKC>  // PF: LDS fragments read one k-step ahead (two register sets)
__global__ __launch_bounds__(BLOCK, 1) void sgemm256_kernel(
    float* __restrict__ out, const float* __restrict__ A, const float* __restrict__ B,
    const float* __restrict__ C, long long M, long long N, long long K, long long lda,
    long long ldb, long long sAb, long long sBb, long long sCb, long long sC0, long long sC1,
    float alpha, float beta, long long tiles_m, long long tiles_n, long long batch, long long ldo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int ASZ = AKC ? T2 * T2ALD : T2K * T2;  // one buffer of the A image
  constexpr int BSZ = BKC ? T2 * T2ALD : T2K * T2;
  float* As = (float*)smem_raw;  // [2][ASZ]
  float* Bs = As + 2 * ASZ;      // [2][BSZ]
  const long long nt = tiles_m * tiles_n, total = nt * batch;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm0 = (w >> 1) * 128, wn0 = (w & 1) * 128, h = lane >> 5, i32 = lane & 31;
  const long long nk = K / T2K;
  auto coords = [&](long long L, long long& m0, long long& n0, long long& bz) {
    // (32-bit arithmetic: the tile count of anything that fits in HBM is far below 2^31, and a 64-bit
    //  division is a few hundred instructions per tile in front of the first MFMA)
    const unsigned tot = (unsigned)total, ntu = (unsigned)nt, tnu = (unsigned)tiles_n;
    unsigned pid = (unsigned)L;
    if (tot % 8 == 0) {  // XCD x walks the contiguous range [x*total/8, (x+1)*total/8)
      const unsigned per = tot / 8;
      pid = (pid % 8) * per + pid / 8;
    }
    const unsigned b = pid / ntu;
    pid -= b * ntu;
    bz = b;
    m0 = (long long)(pid / tnu) * T2;
    n0 = (long long)(pid % tnu) * T2;
  };
  // staging registers as eight named float4 (arrays assigned on two branches ended up in scratch)
  float4 ga0, ga1, ga2, ga3, gb0, gb1, gb2, gb3;
  // K-contiguous operand: thread -> row tid/4 (+64 p), k = 4 (tid%4); otherwise: k row tid/64 (+4 p), element 4 (tid%64)
  const int kc_off = (tid >> 2) * T2ALD + (tid & 3) * 4;
  const int wc_off = (tid >> 6) * T2 + (tid & 63) * 4;
  // tile origins and per-k-step advances in elements
  auto a_tile = [&](long long bz, long long m0) { return A + bz * sAb + (AKC ? m0 * lda : m0); };
  auto b_tile = [&](long long bz, long long n0) { return B + bz * sBb + (BKC ? n0 * ldb : n0); };
  const long long a_kstep = AKC ? (long long)T2K : (long long)T2K * lda;
  const long long b_kstep = BKC ? (long long)T2K : (long long)T2K * ldb;
#define T2_GLOAD(PA, PB)                                                                            \
  do {                                                                                              \
    if constexpr (AKC) {                                                                            \
      const float* pa_ = (PA) + (long long)(tid >> 2) * lda + (tid & 3) * 4;                        \
      ga0 = *(const float4*)pa_; ga1 = *(const float4*)(pa_ + 64 * lda);                            \
      ga2 = *(const float4*)(pa_ + 128 * lda); ga3 = *(const float4*)(pa_ + 192 * lda);             \
    } else {                                                                                        \
      const float* pa_ = (PA) + (long long)(tid >> 6) * lda + (tid & 63) * 4;                       \
      ga0 = *(const float4*)pa_; ga1 = *(const float4*)(pa_ + 4 * lda);                             \
      ga2 = *(const float4*)(pa_ + 8 * lda); ga3 = *(const float4*)(pa_ + 12 * lda);                \
    }                                                                                               \
    if constexpr (BKC) {                                                                            \
      const float* pb_ = (PB) + (long long)(tid >> 2) * ldb + (tid & 3) * 4;                        \
      gb0 = *(const float4*)pb_; gb1 = *(const float4*)(pb_ + 64 * ldb);                            \
      gb2 = *(const float4*)(pb_ + 128 * ldb); gb3 = *(const float4*)(pb_ + 192 * ldb);             \
    } else {                                                                                        \
      const float* pb_ = (PB) + (long long)(tid >> 6) * ldb + (tid & 63) * 4;                       \
      gb0 = *(const float4*)pb_; gb1 = *(const float4*)(pb_ + 4 * ldb);                             \
      gb2 = *(const float4*)(pb_ + 8 * ldb); gb3 = *(const float4*)(pb_ + 12 * ldb);                \
    }                                                                                               \
  } while (0)
#define T2_ST_KC(DST, V)                                                                            \
  do {                                                                                              \
    float2* d_ = (float2*)(DST);                                                                    \
    d_[0] = make_float2((V).x, (V).y);                                                              \
    d_[1] = make_float2((V).z, (V).w);                                                              \
  } while (0)
#define T2_SSTORE(AS, BS)                                                                           \
  do {                                                                                              \
    if constexpr (AKC) {                                                                            \
      float* as_ = (AS) + kc_off;                                                                   \
      T2_ST_KC(as_, ga0); T2_ST_KC(as_ + 64 * T2ALD, ga1); T2_ST_KC(as_ + 128 * T2ALD, ga2); T2_ST_KC(as_ + 192 * T2ALD, ga3); \
    } else {                                                                                        \
      float* as_ = (AS) + wc_off;                                                                   \
      *(float4*)as_ = ga0; *(float4*)(as_ + 4 * T2) = ga1; *(float4*)(as_ + 8 * T2) = ga2; *(float4*)(as_ + 12 * T2) = ga3; \
    }                                                                                               \
    if constexpr (BKC) {                                                                            \
      float* bs_ = (BS) + kc_off;                                                                   \
      T2_ST_KC(bs_, gb0); T2_ST_KC(bs_ + 64 * T2ALD, gb1); T2_ST_KC(bs_ + 128 * T2ALD, gb2); T2_ST_KC(bs_ + 192 * T2ALD, gb3); \
    } else {                                                                                        \
      float* bs_ = (BS) + wc_off;                                                                   \
      *(float4*)bs_ = gb0; *(float4*)(bs_ + 4 * T2) = gb1; *(float4*)(bs_ + 8 * T2) = gb2; *(float4*)(bs_ + 12 * T2) = gb3; \
    }                                                                                               \
  } while (0)
  long long L = blockIdx.x;
  if (L >= total) return;
  long long m0, n0, bz;
  coords(L, m0, n0, bz);
  T2_GLOAD(a_tile(bz, m0), b_tile(bz, n0));
  T2_SSTORE(As, Bs);
  __syncthreads();
  int buf = 0;
  const bool has_c = (beta != 0.f) && C != nullptr;
  for (;;) {
    const long long Ln = L + gridDim.x;
    const bool has_next = Ln < total;
    long long m1 = 0, n1 = 0, b1 = 0;
    if (has_next) coords(Ln, m1, n1, b1);
    float16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const float* Ab = a_tile(bz, m0);
    const float* Bb = b_tile(bz, n0);
    for (long long kt = 0; kt < nk; kt++) {
      const bool more = kt + 1 < nk;
      if (more) T2_GLOAD(Ab + (kt + 1) * a_kstep, Bb + (kt + 1) * b_kstep);
      else if (has_next) T2_GLOAD(a_tile(b1, m1), b_tile(b1, n1));
      const float* as = As + buf * ASZ;
      const float* bs = Bs + buf * BSZ;
      // fragment bases: K-contiguous image -> row (w?0 + i32), k = 2h; otherwise -> k row 2h, column (w?0 + i32)
      const float* abase = AKC ? as + (wm0 + i32) * T2ALD + 2 * h : as + (2 * h) * T2 + wm0 + i32;
      const float* bbase = BKC ? bs + (wn0 + i32) * T2ALD + 2 * h : bs + (2 * h) * T2 + wn0 + i32;
      auto afrag = [&](int i, int kk) -> float2 {
        if constexpr (AKC) return *(const float2*)(abase + i * 32 * T2ALD + kk * 4);
        else return make_float2(abase[kk * 4 * T2 + i * 32], abase[(kk * 4 + 1) * T2 + i * 32]);
      };
      auto bfrag = [&](int j, int kk) -> float2 {
        if constexpr (BKC) return *(const float2*)(bbase + j * 32 * T2ALD + kk * 4);
        else return make_float2(bbase[kk * 4 * T2 + j * 32], bbase[(kk * 4 + 1) * T2 + j * 32]);
      };
      if constexpr (PF) {
      // fragments one k-step AHEAD of the MFMAs that use them (two register sets): read just in time,
      // every group of 16 MFMAs starts behind an LDS round trip (~10 % of the loop at one wave per SIMD)
      float2 af[2][4], bf[2][4];
#pragma unroll
      for (int i = 0; i < 4; i++) af[0][i] = afrag(i, 0);
#pragma unroll
      for (int j = 0; j < 4; j++) bf[0][j] = bfrag(j, 0);
#pragma unroll
      for (int kk = 0; kk < T2K / 4; kk++) {
        const int cs = kk & 1, ns = cs ^ 1;
        if (kk + 1 < T2K / 4) {
#pragma unroll
          for (int i = 0; i < 4; i++) af[ns][i] = afrag(i, kk + 1);
#pragma unroll
          for (int j = 0; j < 4; j++) bf[ns][j] = bfrag(j, kk + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cs][i].x, bf[cs][j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cs][i].y, bf[cs][j].y, acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      } else {
#pragma unroll
      for (int kk = 0; kk < T2K / 4; kk++) {
        float2 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; i++) af[i] = afrag(i, kk);
#pragma unroll
        for (int j = 0; j < 4; j++) bf[j] = bfrag(j, kk);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
      }
      }
      if (more || has_next) T2_SSTORE(As + (buf ^ 1) * ASZ, Bs + (buf ^ 1) * BSZ);
      __syncthreads();
      buf ^= 1;
    }
    float* ob = out + bz * M * N + (m0 + wm0 + 4 * h) * ldo + n0 + wn0 + i32;
    if (!has_c) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          float* orow = ob + (long long)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo;
#pragma unroll
          for (int j = 0; j < 4; j++) orow[j * 32] = alpha * acc[i][j][r];
        }
    } else {
      const float* Cb = C + bz * sCb;
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const long long row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const long long col = n0 + wn0 + j * 32 + i32;
            ob[(long long)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo + j * 32] = alpha * acc[i][j][r] + beta * Cb[row * sC0 + col * sC1];
          }
    }
    if (!has_next) break;
    L = Ln;
    m0 = m1;
    n0 = n1;
    bz = b1;
  }
#undef T2_GLOAD
#undef T2_ST_KC
#undef T2_SSTORE
}

// out[b][m][n] = alpha * sum_s part[s][b][m][n] + beta * C[b][m][n]   (fixed order over s)
template <class T>
__global__ __launch_bounds__(BLOCK) void splitk_finish_kernel(
    T* __restrict__ out, const T* __restrict__ part, const T* __restrict__ C, long long M,
    long long N, long long total, int nsplit, long long sCb, long long sC0, long long sC1, T alpha,
    T beta, long long ldo) {
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < total;
       i += (long long)gridDim.x * BLOCK) {
    T v = part[i];
    for (int s = 1; s < nsplit; s++) v += part[(long long)s * total + i];
    v *= alpha;
    if (beta != T(0) && C != nullptr) {
      const long long b = i / (M * N), r = i - b * M * N;
      const long long m = r / N, n = r - m * N;
      v += beta * C[b * sCb + m * sC0 + n * sC1];
    }
    if (ldo == N) out[i] = v;
    else out[(i / N) * ldo + (i % N)] = v;  // (row stride of the destination: batch == 1)
  }
}

template <class T> struct KernelSel;
template <> struct KernelSel<double> {
  template <bool a, bool b, bool s, int bk> static auto get() { return dgemm_kernel<a, b, s>; }
};
template <> struct KernelSel<float> {
  template <bool a, bool b, bool s, int bk> static auto get() { return sgemm_kernel<a, b, s, bk>; }
};

// split-K when the tile grid cannot fill the chip (skinny / small GEMMs, e.g. the
// (B x H)@(H x H) products of a Scan step): aim for >= 256 workgroups, >= 64 k per split.
// A pure function of the shape: pthip_gemm_nslabs() promises the same count to callers that
// consume the raw partial slabs themselves (pthip_gemm_partials).
struct SplitPlan {
  long long nsplit, kchunk;
};
inline SplitPlan split_plan(long long batch, long long M, long long N, long long K) {
  const long long tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const long long tiles = tiles_m * tiles_n * batch;
  long long nsplit = 1;
  if (tiles < 128 && K >= 128) {
    static const long long cap = getenv("PTHIP_GEMM_MAXSPLIT") ? atoll(getenv("PTHIP_GEMM_MAXSPLIT")) : 64;
    static const long long mink = getenv("PTHIP_GEMM_MINK") ? atoll(getenv("PTHIP_GEMM_MINK")) : 64;
    static const long long wgs = getenv("PTHIP_GEMM_WGS") ? atoll(getenv("PTHIP_GEMM_WGS")) : 256;
    nsplit = (wgs + tiles - 1) / tiles;
    if (nsplit > K / mink) nsplit = K / mink;
    if (nsplit > cap) nsplit = cap;
    if (nsplit < 1) nsplit = 1;
  }
  long long kchunk = (K + nsplit - 1) / nsplit;
  kchunk = (kchunk + BK - 1) / BK * BK;
  if (kchunk < BK) kchunk = BK;
  nsplit = (K + kchunk - 1) / kchunk;
  if (nsplit < 1) nsplit = 1;
  return {nsplit, kchunk};
}

// `partials` != nullptr: write the raw products into partials[nsplit][batch][M][N] and stop
// (no finish launch: the consumer folds the fixed-order sum and the alpha/beta epilogue in)
template <class T, bool AKC, bool BKC, bool SKINNY, int BKT = 16>
int launch(long long batch, long long M, long long N, long long K, T alpha, const T* A,
           long long sAb, long long lda, const T* B, long long sBb, long long ldb, T beta,
           const T* C, long long sCb, long long sC0, long long sC1, T* out,
           T* partials = nullptr, long long ldo = 0) {
  hipStream_t st = pthip::ctx().stream;
  if (ldo == 0) ldo = N;  // row stride of `out` (!= N: batch == 1, the in-place update of a sub-block)
  using SA = Stage<T, AKC, BKT>;
  using SB = Stage<T, BKC, BKT>;
  const size_t shmem = (size_t)(2 * SA::SIZE + 2 * SB::SIZE) * sizeof(T);
  auto k = KernelSel<T>::template get<AKC, BKC, SKINNY, BKT>();
  static bool attr_set = false;
  if (!attr_set && shmem > 64 * 1024) {
    PTHIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr_set = true;
  }
  const long long tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  constexpr int VN = 16 / sizeof(T);
  const int vecA = (lda % VN == 0) && (((uintptr_t)A) % 16 == 0) && (sAb % VN == 0);
  const int vecB = (ldb % VN == 0) && (((uintptr_t)B) % 16 == 0) && (sBb % VN == 0);
  const SplitPlan sp = split_plan(batch, M, N, K);
  const long long nsplit = sp.nsplit, kchunk = sp.kchunk;
  dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)nsplit, (unsigned)batch);
  if (partials != nullptr) {
    // nsplit == 1: the epilogue with alpha = 1, beta = 0 stores the plain product in slab 0
    PTHIP_KLAUNCH(k, grid, dim3(BLOCK), shmem, st, partials, A, B, (const T*)nullptr, M, N, K,
                       lda, ldb, sAb, sBb, (long long)(M * N), N, 1LL, T(1), T(0), tiles_m, tiles_n,
                       vecA, vecB, kchunk, N);
    return pthip::post_launch("gemm(partials)");
  }
  if constexpr (sizeof(T) == 8 && !SKINNY) {
    // mid-size fp64 products (the 128 x 128 tiling gives fewer than 128 workgroups, so split_plan above cut K into slabs):
    // 64 x 64 tiles when THEY fill the chip without splitting — no slabs, no finishing pass.  (Only here: callers of
    // pthip_gemm_partials consume the 128-tile slabs split_plan promises.)
    static const bool t64 = !(getenv("PTHIP_DGEMM_T64") && atoi(getenv("PTHIP_DGEMM_T64")) == 0);
    const long long t64m = (M + 63) / 64, t64n = (N + 63) / 64;
    // (also when the 128-tiling has no split but fewer workgroups than CUs: 1024 x 2048 x 1024 ran on 128 of them, 23.7
    //  TFLOP/s — profiles/r5q_gemm_mid.txt)
    if (t64 && (nsplit > 1 || tiles_m * tiles_n * batch < pthip::kNumCU) && M > 64 && t64m * t64n * batch >= 128) {
      auto k64 = dgemm_kernel<AKC, BKC, false, 64>;
      const size_t sh64 = (size_t)(2 * StageD<AKC, 64>::SIZE + 2 * StageD<BKC, 64>::SIZE) * sizeof(double);
      dim3 g64((unsigned)(t64m * t64n), 1u, (unsigned)batch);
      const long long kc64 = (K + BK - 1) / BK * BK;
      PTHIP_KLAUNCH(k64, g64, dim3(BLOCK), sh64, st, (double*)out, (const double*)A, (const double*)B, (const double*)C, M, N, K, lda, ldb, sAb, sBb,
                    sCb, sC0, sC1, (double)alpha, (double)beta, t64m, t64n, vecA, vecB, kc64, ldo);
      return pthip::post_launch("gemm(64x64 tiles)");
    }
  }
  if (nsplit == 1) {
    if constexpr (sizeof(T) == 4 && !SKINNY && BKT == 32) {
      // 256-aligned M, N (any operand orientation since round 4): one 256 x 256 tile per workgroup
      static const bool big = !(getenv("PTHIP_SGEMM_256") && atoi(getenv("PTHIP_SGEMM_256")) == 0);
      if (big && M % T2 == 0 && N % T2 == 0 && K % T2K == 0 && K >= T2K && vecA && vecB && (ldo % 1 == 0)) {
        const size_t sh = (size_t)(2 * (AKC ? T2 * T2ALD : T2K * T2) + 2 * (BKC ? T2 * T2ALD : T2K * T2)) * sizeof(float);
        static const bool pf = !(getenv("PTHIP_SGEMM_256_PF") && atoi(getenv("PTHIP_SGEMM_256_PF")) == 0);
        auto k256 = pf ? sgemm256_kernel<true, AKC, BKC> : sgemm256_kernel<false, AKC, BKC>;
        static bool attr_b = false;
        if (!attr_b) {
          PTHIP_CHECK(hipFuncSetAttribute((const void*)k256, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pthip::kLdsPerCU));
          attr_b = true;
        }
        const long long t_m = M / T2, t_n = N / T2, tot = t_m * t_n * batch;
        // Experiment switches for a product enqueued on a side stream (the hoisted sequence product of a Scan, one chunk
        // ahead of the steps that read it: dispatch/blas.py LazySeq, PTHIP_SCAN_OVERLAP=1): PTHIP_SIDE_GEMM_WGS caps its
        // workgroups (each walks several tiles), PTHIP_SIDE_GEMM_WHOLE_CU=1 makes every workgroup ask for all of a CU's
        // LDS so that the other stream's workgroups are placed on the CUs this launch does not use.  Both OFF by
        // default: measured on config #5 (profiles/r6d_c5_overlap.md) a step kernel runs 8.6-9.0 us instead of 5.4-5.8
        // whenever one of these products is in flight — with shared CUs and with disjoint ones alike — so the 6.9 ms of
        // capped products hide 380 steps' worth of time and the evaluation ends where it started (13.8 ms).
        static const long long side_wgs = getenv("PTHIP_SIDE_GEMM_WGS") ? atoll(getenv("PTHIP_SIDE_GEMM_WGS")) : 0;
        static const bool side_whole = getenv("PTHIP_SIDE_GEMM_WHOLE_CU") && atoi(getenv("PTHIP_SIDE_GEMM_WHOLE_CU")) == 1;
        long long cap = pthip::kNumCU;
        if (pthip::ctx().current != 0 && side_wgs > 0 && side_wgs < cap) cap = side_wgs;
        const long long grid = tot < cap ? tot : cap;
        const size_t sh_launch = (cap != pthip::kNumCU && side_whole) ? (size_t)pthip::kLdsPerCU : sh;
        PTHIP_KLAUNCH(k256, dim3((unsigned)grid), dim3(BLOCK), sh_launch, st, (float*)out, (const float*)A, (const float*)B,
                      (const float*)C, M, N, K, lda, ldb, sAb, sBb, sCb, sC0, sC1, (float)alpha, (float)beta, t_m, t_n, batch, ldo);
        return pthip::post_launch("gemm(256x256)");
      }
    }
    if constexpr (sizeof(T) == 4 && !SKINNY && BKT == 32) {
      // more tiles than resident workgroups: the persistent kernel (prologues / epilogues of
      // consecutive tiles overlap).  PTHIP_SGEMM_PERSIST=0 keeps one tile per workgroup.
      static const bool persist = !(getenv("PTHIP_SGEMM_PERSIST") && atoi(getenv("PTHIP_SGEMM_PERSIST")) == 0);
      const long long total = tiles_m * tiles_n * batch, resident = (long long)pthip::kNumCU * 2;
      if (persist && total > resident) {
        static bool attr_p = false;
        static const int diag = getenv("PTHIP_SGEMM_DIAG") ? atoi(getenv("PTHIP_SGEMM_DIAG")) : 0;
        auto kp = diag == 1 ? sgemm_persistent_kernel<AKC, BKC, 1> : diag == 2 ? sgemm_persistent_kernel<AKC, BKC, 2>
                  : diag == 3 ? sgemm_persistent_kernel<AKC, BKC, 3> : sgemm_persistent_kernel<AKC, BKC, 0>;
        if (!attr_p && shmem > 64 * 1024) {
          PTHIP_CHECK(hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
          attr_p = true;
        }
        PTHIP_KLAUNCH(kp, dim3((unsigned)resident), dim3(BLOCK), shmem, st, (float*)out, (const float*)A, (const float*)B,
                      (const float*)C, M, N, K, lda, ldb, sAb, sBb, sCb, sC0, sC1, (float)alpha, (float)beta, tiles_m, tiles_n,
                      vecA, vecB, batch, ldo);
        return pthip::post_launch("gemm(persistent)");
      }
    }
    PTHIP_KLAUNCH(k, grid, dim3(BLOCK), shmem, st, out, A, B, C, M, N, K, lda, ldb, sAb, sBb,
                       sCb, sC0, sC1, alpha, beta, tiles_m, tiles_n, vecA, vecB, kchunk, ldo);
    return pthip::post_launch("gemm");
  }
  const long long total = batch * M * N;
  void* part = nullptr;
  int r = pthip_alloc((size_t)nsplit * total * sizeof(T), &part);
  if (r) return r;
  PTHIP_KLAUNCH(k, grid, dim3(BLOCK), shmem, st, (T*)part, A, B, C, M, N, K, lda, ldb, sAb, sBb,
                     sCb, sC0, sC1, alpha, beta, tiles_m, tiles_n, vecA, vecB, kchunk, N);
  r = pthip::post_launch("gemm(split-K)");
  if (!r) {
    long long blocks = (total + BLOCK - 1) / BLOCK;
    if (blocks > 2048) blocks = 2048;
    PTHIP_KLAUNCH((splitk_finish_kernel<T>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, out,
                       (const T*)part, C, M, N, total, (int)nsplit, sCb, sC0, sC1, alpha, beta, ldo);
    r = pthip::post_launch("gemm splitk_finish");
  }
  pthip_free(part);  // stream-ordered reuse keeps this safe
  return r;
}

// fp32, M > 64: K extent of the staged tiles.  32 = half as many barrier intervals per tile (the
// fp32 MFMA step is half as long as the fp64 one for the same latencies); PTHIP_SGEMM_BK=16
// selects the round-1 kernel for A/B measurements.
inline bool sgemm_bk32() {
  static const bool v = !(getenv("PTHIP_SGEMM_BK") && atoi(getenv("PTHIP_SGEMM_BK")) == 16);
  return v;
}

template <class T>
int gemm_typed(long long batch, long long M, long long N, long long K, double alpha, const void* A,
               long long sAb, long long sA0, long long sA1, const void* B, long long sBb,
               long long sB0, long long sB1, double beta, const void* C, long long sCb,
               long long sC0, long long sC1, void* out, void* partials = nullptr, long long ldo = 0) {
  if (batch == 0 || M == 0 || N == 0) return 0;
  // Normalise the strides of degenerate (length-1) dims, then classify:
  //   A (m,k) at m*sA0 + k*sA1 : K-contiguous iff sA1 == 1 (lda = sA0), else M-contiguous (lda = sA1)
  //   B (k,n) at k*sB0 + n*sB1 : N-contiguous iff sB1 == 1 (ldb = sB0), else K-contiguous (ldb = sB1)
  if (K == 1) sA1 = 1;
  if (M == 1) { if (sA1 == 1) sA0 = K; else sA0 = 1; }
  if (N == 1) sB1 = 1;
  if (K == 1) { if (sB1 == 1) sB0 = N; else sB0 = 1; }
  bool akc, bkc;
  long long lda, ldb;
  if (sA1 == 1) { akc = true; lda = sA0; }
  else if (sA0 == 1) { akc = false; lda = sA1; }
  else return pthip::set_error("pthip_gemm: A has no unit stride (%lld, %lld)", (long long)sA0, (long long)sA1);
  if (sB1 == 1) { bkc = false; ldb = sB0; }
  else if (sB0 == 1) { bkc = true; ldb = sB1; }
  else return pthip::set_error("pthip_gemm: B has no unit stride (%lld, %lld)", (long long)sB0, (long long)sB1);
  const T* a = (const T*)A;
  const T* b = (const T*)B;
  const T* c = (const T*)C;
  T* o = (T*)out;
  const bool skinny = M <= 64;
#define GO(X, Y)                                                                                   \
  do {                                                                                             \
    if (skinny)                                                                                    \
      return launch<T, X, Y, true>(batch, M, N, K, (T)alpha, a, sAb, lda, b, sBb, ldb, (T)beta, c, \
                                   sCb, sC0, sC1, o, (T*)partials, ldo);                                \
    if constexpr (sizeof(T) == 4) {                                                                \
      if (sgemm_bk32())                                                                            \
        return launch<T, X, Y, false, 32>(batch, M, N, K, (T)alpha, a, sAb, lda, b, sBb, ldb,      \
                                          (T)beta, c, sCb, sC0, sC1, o, (T*)partials, ldo);             \
    }                                                                                              \
    return launch<T, X, Y, false>(batch, M, N, K, (T)alpha, a, sAb, lda, b, sBb, ldb, (T)beta, c,  \
                                  sCb, sC0, sC1, o, (T*)partials, ldo);                                 \
  } while (0)
  if (akc && bkc) GO(true, true);
  if (akc && !bkc) GO(true, false);
  if (!akc && bkc) GO(false, true);
  GO(false, false);
#undef GO
}

}  // namespace

namespace pthip {
// C (M x N, row stride ldc) <- beta*C + alpha * A @ B, in place: the trailing update of the blocked
// Cholesky (linalg.hip).  Every element is read and written by the same lane of the same launch.
int gemm_inplace(int dtype, long long M, long long N, long long K, double alpha, const void* A, long long sA0,
                 long long sA1, const void* B, long long sB0, long long sB1, double beta, void* C, long long ldc) {
  if (dtype == PTHIP_F64)
    return gemm_typed<double>(1, M, N, K, alpha, A, 0, sA0, sA1, B, 0, sB0, sB1, beta, C, 0, ldc, 1, C, nullptr, ldc);
  return gemm_typed<float>(1, M, N, K, alpha, A, 0, sA0, sA1, B, 0, sB0, sB1, beta, C, 0, ldc, 1, C, nullptr, ldc);
}
}  // namespace pthip

extern "C" int pthip_gemm(int dtype, int64_t batch, int64_t M, int64_t N, int64_t K, double alpha,
                          const void* A, int64_t sAb, int64_t sA0, int64_t sA1, const void* B,
                          int64_t sBb, int64_t sB0, int64_t sB1, double beta, const void* C,
                          int64_t sCb, int64_t sC0, int64_t sC1, void* out) {
  PTHIP_REQUIRE_INIT();
  if (batch == 1 && M > 0 && N > 0 && K > 0) {
    bool handled = false;
    const int r = pthip::gemm_skinny(dtype, M, N, K, alpha, A, sA0, sA1, B, sB0, sB1, beta, C, sC0, sC1, out, &handled);
    if (r || handled) return r;
  }
  if (dtype == PTHIP_F64)
    return gemm_typed<double>(batch, M, N, K, alpha, A, sAb, sA0, sA1, B, sBb, sB0, sB1, beta, C, sCb, sC0, sC1, out);
  if (dtype == PTHIP_F32)
    return gemm_typed<float>(batch, M, N, K, alpha, A, sAb, sA0, sA1, B, sBb, sB0, sB1, beta, C, sCb, sC0, sC1, out);
  return pthip::set_error("pthip_gemm: dtype %d not supported (float32/float64 only)", dtype);
}

// Number of partial slabs pthip_gemm_partials writes for this shape (1 = no split-K).
extern "C" int64_t pthip_gemm_nslabs(int64_t batch, int64_t M, int64_t N, int64_t K) {
  if (batch <= 0 || M <= 0 || N <= 0) return 1;
  return split_plan(batch, M, N, K).nsplit;
}

// Raw products for a consumer that folds the split-K sum into its own kernel:
// part[s][b][m][n] = sum over the s-th K range of A[b][m][k] * B[b][k][n]; the caller adds the
// slabs in ascending s (the order splitk_finish_kernel uses) and applies alpha/beta itself.
// `nslabs` must equal pthip_gemm_nslabs(batch, M, N, K).
extern "C" int pthip_gemm_partials(int dtype, int64_t batch, int64_t M, int64_t N, int64_t K,
                                   const void* A, int64_t sAb, int64_t sA0, int64_t sA1,
                                   const void* B, int64_t sBb, int64_t sB0, int64_t sB1, void* part,
                                   int64_t nslabs) {
  PTHIP_REQUIRE_INIT();
  if (nslabs != pthip_gemm_nslabs(batch, M, N, K))
    return pthip::set_error("pthip_gemm_partials: slab count %lld does not match the plan", (long long)nslabs);
  if (K == 0) return pthip_memset(part, 0, (size_t)(batch * M * N) * (dtype == PTHIP_F64 ? 8 : 4));
  if (dtype == PTHIP_F64)
    return gemm_typed<double>(batch, M, N, K, 1.0, A, sAb, sA0, sA1, B, sBb, sB0, sB1, 0.0, nullptr, 0, 0, 0, part, part);
  if (dtype == PTHIP_F32)
    return gemm_typed<float>(batch, M, N, K, 1.0, A, sAb, sA0, sA1, B, sBb, sB0, sB1, 0.0, nullptr, 0, 0, 0, part, part);
  return pthip::set_error("pthip_gemm_partials: dtype %d not supported (float32/float64 only)", dtype);
}
