/* pthip.h — C ABI of the MI355X (gfx950) execution backend for PyTensor's `hip` linker.
 *
 * This is the drop-in boundary below `HipLinker` (pytensor_amd/linker.py): plain
 * pointers and sizes, no Python / torch types.  The precedent in the reference is
 * the C-thunk ABI `int (*fn)(void*)` with non-zero = failure
 * (pytensor/link/c/cutils.py:25-43, pytensor/link/c/basic.py:1736-1761); every
 * entry point here returns `int` (0 = ok) and parks a message retrievable with
 * `pthip_last_error()`.
 *
 * Conventions
 *   - all `const void*` / `void*` data arguments are DEVICE pointers unless named `host_*`;
 *   - strides are in ELEMENTS (not bytes); arrays are described NumPy-style
 *     (any sign/size of stride where noted, else contiguous row-major);
 *   - every call enqueues on the context's single in-order HIP stream and returns
 *     without synchronising (except where documented);
 *   - LAPACK-style numerical failure is NOT an error: results are NaN-filled
 *     (pytensor/tensor/linalg/decomposition/cholesky.py:78-80).
 *
 * Each entry point cites the reference interface it replaces.
 */
#ifndef PTHIP_H
#define PTHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dtype codes (subset of pytensor/tensor/type.py:300 `dtype_specs`) */
enum pthip_dtype {
  PTHIP_BOOL = 0,
  PTHIP_I8 = 1,
  PTHIP_I16 = 2,
  PTHIP_I32 = 3,
  PTHIP_I64 = 4,
  PTHIP_U8 = 5,
  PTHIP_F32 = 6,
  PTHIP_F64 = 7,
  PTHIP_U16 = 8,
  PTHIP_U32 = 9,
  PTHIP_U64 = 10,
  PTHIP_F16 = 11 /* storage type; arithmetic rounds to half after every scalar op, as NumPy does */
};

/* CAReduce scalar ops (pytensor/tensor/elemwise.py:1233; subclasses tensor/math.py:3498,3587,468,475,3438,3468) */
enum pthip_reduce_op {
  PTHIP_RED_ADD = 0,
  PTHIP_RED_MUL = 1,
  PTHIP_RED_MAX = 2,
  PTHIP_RED_MIN = 3,
  PTHIP_RED_AND = 4,
  PTHIP_RED_OR = 5,
  PTHIP_RED_XOR = 6
};

/* ---- context ------------------------------------------------------------------ */
/* Bind the calling process to `device` (one process per GPU) and create the stream. Idempotent. */
int pthip_init(int device);
int pthip_device_count(int* n);
int pthip_device_name(char* buf, size_t buflen);
const char* pthip_last_error(void);
int pthip_synchronize(void);
/* Multi-stream plans: the eager path uses stream 0 only.  A frozen plan may fork
 * independent branches of the graph (e.g. the single-CU Cholesky/solve chain next to the
 * HBM-streaming kernels) onto up to 4 streams; cross-stream dependencies become event
 * edges of the captured hipGraph.  `select` makes stream i the target of every
 * subsequent launch/copy; `wait(waiter, signaler)` orders waiter after everything
 * enqueued so far on signaler. */
int pthip_stream_select(int i);
int pthip_stream_wait(int waiter, int signaler);
/* raw hipStream_t of the context (for interop / event timing on the right stream) */
void* pthip_stream(void);

/* ---- memory: size-bucketed caching pool (replaces NumPy's allocator that backs
 *      every output of Elemwise.perform etc., pytensor/tensor/elemwise.py:935-961) ---- */
int pthip_alloc(size_t bytes, void** dptr);
int pthip_free(void* dptr);
int pthip_pool_stats(size_t* bytes_in_use, size_t* bytes_reserved, size_t* n_device_allocs);
int pthip_pool_trim(void);
int pthip_host_alloc(size_t bytes, void** hptr); /* pinned */
int pthip_host_free(void* hptr);
/* copies are stream-ordered; host memory may be pageable (then the copy is staged synchronously) */
int pthip_h2d(void* dst, const void* host_src, size_t bytes);
int pthip_d2h(void* host_dst, const void* src, size_t bytes);
int pthip_d2d(void* dst, const void* src, size_t bytes);
int pthip_memset(void* dst, int byte, size_t bytes);

/* Write tracking for a HOST array mirrored in HBM (a resident shared variable): the reference
 * reads the storage cell on every call (pytensor/compile/sharedvalue.py:97-130 — `get_value(borrow=True)`
 * hands out the storage, in-place edits must be seen).  `protect` makes the whole pages inside
 * [host_ptr, host_ptr+bytes) read-only (callers pass a page-aligned interior and hash the ragged
 * ends themselves); the first CPU store into them sets `*dirty_flag` (stable address, readable
 * without a call), restores write access and proceeds.  `release` restores write access and frees
 * the slot.  pthip_h2d reads a source that overlaps a protected range through pinned bounce buffers (the
 * protection stays; round 3 lifted it, which marked every overlapping slot dirty).  Ranges known to the HIP
 * runtime (pinned / registered host memory) are refused. */
int pthip_guard_protect(const void* host_ptr, size_t bytes, int* slot, const int** dirty_flag);
/* the ragged ends of the watched array (the partial pages outside the protected interior, at most one page
 * each) are snapshotted; pthip_guard_clean = "no store faulted since protect AND the ends are unchanged" */
int pthip_guard_set_edges(int slot, const void* edge0, size_t n0, const void* edge1, size_t n1);
int pthip_guard_clean(int slot);
int pthip_guard_release(int slot);
int pthip_guard_stats(int* slots_in_use, int* slots_active);

/* A private arena: between begin/end every pthip_alloc/pthip_free is served from a
 * private free list, so that a second, captured run of the same launch sequence sees
 * the same pointers.  Used to freeze a plan into a hipGraph. */
int pthip_arena_begin(void** arena); /* *arena == NULL → create; else re-enter and rewind */
int pthip_arena_end(void);
/* blocks freed inside the arena are not reused before the next rewind (multi-stream plans) */
int pthip_arena_set_no_reuse(void* arena, int no_reuse);
int pthip_arena_destroy(void* arena);

/* ---- hipGraph capture of the launch sequence of one Function call
 *      (the analogue of the CVM, pytensor/link/c/c_code/lazylinker_c.c:749: one native
 *      call per Function.__call__ instead of one Python call per node) ---- */
int pthip_capture_begin(void);
int pthip_capture_end(void** graph_exec);
int pthip_graph_launch(void* graph_exec);
/* launch into stream `stream` (segmented plans: independent segments on different streams) */
int pthip_graph_launch_on(void* graph_exec, int stream);
/* One native call per Function.__call__ (what CLazyLinker_call is to the CVM,
 * lazylinker_c.c:749): upload the staged parameters, replay the captured segment(s)
 * — ga (latency chain, stream 1) may be NULL for an unsegmented plan — and, if `sync`,
 * wait for the result. */
int pthip_plan_replay(void* ga, void* gb, void* gc, void* dev_in, const void* host_in,
                      size_t in_bytes, int sync);
/* Launch lists: the C++ launch-plan executor (the reference's analogue is the CVM's
 * pre-resolved thunk walk, link/c/c_code/lazylinker_c.c:749).  Between pthip_record_begin and
 * pthip_record_end every kernel launch and async copy the library issues on stream 0 is executed AND
 * kept with its arguments resolved; pthip_list_launch repeats the sequence as direct launches on a
 * stream.  For a few kernels this beats a hipGraph replay (no graph-launch floor, no graph-to-graph
 * boundary behind an event wait).  A sequence that contains a host-to-device copy cannot be
 * recorded (pthip_record_end fails, *list = NULL).  pthip_launch_count: kernel launches + async
 * copies issued so far (used to size plan segments).  pthip_plan_replay2 = pthip_plan_replay with
 * each segment given either as a graph (g*) or as a list (l*).  pthip_plan_replay3 additionally
 * copies the packed results (out_bytes at dev_out) into host_out after the last segment: the
 * destination is a parameter of the call, not of the plan, so every call can fill a pinned block
 * of its own that the caller hands out as the result arrays (JITLinker thunks return fresh
 * arrays, link/basic.py:670-684) without copying them. */
int pthip_record_begin(void);
int pthip_record_end(void** list, int64_t* n_ops);
int pthip_list_launch(void* list, int stream);
int pthip_list_destroy(void* list);
int64_t pthip_launch_count(void);
int pthip_plan_replay2(void* ga, void* la, void* gb, void* lb, void* gc, void* lc, void* dev_in,
                       const void* host_in, size_t in_bytes, int sync);
int pthip_plan_replay3(void* ga, void* la, void* gb, void* lb, void* gc, void* lc, void* dev_in,
                       const void* host_in, size_t in_bytes, const void* dev_out, void* host_out,
                       size_t out_bytes, int sync);
/* The same through a descriptor filled once per plan (one pointer instead of 13 arguments per call), with two
 * more ways to finish: sync = 2 polls `done_word` — an int32 in pinned host memory that the LAST kernel of the
 * plan sets to 1 behind its results; it is cleared here before anything is launched — instead of waiting for
 * the stream's completion signal (~5 us less per call, tools/ubench/call_lat.hip); flags bit 1: device-side join (pthip_join_signal); flags bit 0: segment A reads
 * its staged parameters straight from the pinned block (no event between the parameter upload and segment B).
 * Replaces the output loop + return of the JIT thunk, pytensor/link/basic.py:670-684. */
typedef struct pthip_replay_desc {
  void *ga, *la, *gb, *lb, *gc, *lc; /* segments: captured graph (g*) or launch list (l*), A may be absent */
  void* dev_in;
  const void* host_in;
  size_t in_bytes;
  const void* dev_out;
  size_t out_bytes;
  int flags;
} pthip_replay_desc;
int pthip_plan_replay4(const pthip_replay_desc* desc, void* host_out, volatile int* done_word, int sync);
/* a zero-initialised int32 device slot for a last-workgroup ticket (self-resetting; see csrc/tail_device.h) */
int pthip_ticket_slot(void** slot);
/* n consecutive such slots (one ticket per output tile of a one-pass N-d reduction: csrc/runtime.hip) */
int pthip_ticket_slots(int n, void** first);
/* Device-side join of a segmented plan's two streams (pthip_replay_desc.flags bit 1: pthip_plan_replay4 in poll mode
 * then issues no event between segment A's stream and the closing segment).  pthip_join_signal: a one-thread launch
 * on the current stream that stores 1 into `word` (a pthip_ticket_slot) — recorded as the last launch of segment A;
 * the generated tail kernel that opens the closing segment takes the word as an argument, waits for it and puts it
 * back to 0.  A wait that lasts 1 ms (kernels of the two streams cannot overlap: a counter-collecting profiler) stores
 * 2 into the done word and leaves; pthip_plan_replay4 then waits for segment A's stream and runs the closing segment
 * again, and after three such calls goes back to the event for that descriptor.  Replaces nothing in the reference:
 * its Loop / Stack VMs (pytensor/link/vm.py) run one thunk at a time, there is nothing to join. */
int pthip_join_signal(void* word);
/* The assumption behind that join — a kernel that has waited for the word reads the other stream's in-place results
 * with plain loads and finds them, no fence behind the wait — checked on this device: `iters` rounds of overwrite +
 * signal on stream 1 against waiting readers on stream 0; *bad = mismatching words (+ 2^20 per reader that never saw
 * the signal).  The host side runs it once per process and keeps the event between the streams when *bad != 0. */
int pthip_join_probe(int iters, int* bad);
int pthip_graph_destroy(void* graph_exec);

/* ---- events (HIP events on the context stream) ---- */
int pthip_event_create(void** ev);
int pthip_event_record(void* ev);
int pthip_event_synchronize(void* ev);
int pthip_event_elapsed_ms(void* start, void* stop, float* ms);
int pthip_event_destroy(void* ev);

/* ---- JIT of generated fused Elemwise/Composite kernels.
 *      Replaces the C linker's per-Op module build (pytensor/link/c/cmodule.py:2016
 *      GCC_compiler, 612 ModuleCache) with hiprtc for gfx950. ---- */
/* Compile HIP source to a gfx950 code object. *code is malloc'd (free with pthip_buffer_free).
 * Works without a GPU (used by build()). `log` receives the compiler log (may be NULL). */
int pthip_jit_compile(const char* src, const char* name, const char* const* opts, int n_opts,
                      void** code, size_t* code_size, char* log, size_t log_len);
void pthip_buffer_free(void* p);
int pthip_module_load(const void* code, size_t code_size, void** module);
int pthip_module_unload(void* module);
int pthip_module_get_function(void* module, const char* name, void** fn);
/* Launch with a packed argument buffer (kernel params laid out with natural alignment). */
int pthip_launch(void* fn, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx, uint32_t by,
                 uint32_t bz, uint32_t shmem_bytes, const void* host_argbuf, size_t argbuf_bytes);

/* ---- CAReduce (pytensor/tensor/elemwise.py:1233 CAReduce, perform 1493-1511,
 *      C loops elemwise_cgen.py:467-761).
 *      out[a,b] = reduce_{r<R} x[a*sA + r*sR + b*sB], accumulated in `acc_dtype`
 *      (elemwise.py:1383-1417) and cast to `out_dtype`; `out` is contiguous (A,B).
 *      `ws` must hold pthip_reduce_workspace(...) bytes (may be NULL if that is 0). ---- */
size_t pthip_reduce_workspace(int acc_dtype, int64_t A, int64_t R, int64_t B);
int pthip_reduce(int op, int in_dtype, int acc_dtype, int out_dtype, const void* x, void* out,
                 int64_t A, int64_t R, int64_t B, int64_t sA, int64_t sR, int64_t sB, void* ws,
                 size_t ws_bytes);

/* ---- BLAS (pytensor/tensor/blas/: Gemv gemv.py:16, Gemm gemm.py:76, Dot22 gemm.py:248,
 *      Dot22Scalar gemm.py:298, Ger ger.py:8, BatchedDot batched.py:18).  dtype F32 or F64.
 *      Operands are strided 2-D views (element strides), outputs contiguous row-major. ---- */
/* out[i] = beta*y[i*sy] + alpha * sum_j A[i*sA0 + j*sA1] * x[j*sx];  beta==0 => y not read
 * (gemv.py:79-86).  ws: pthip_gemv_workspace bytes. */
size_t pthip_gemv_workspace(int dtype, int64_t M, int64_t N, int64_t sA0, int64_t sA1);
int pthip_gemv(int dtype, int64_t M, int64_t N, double alpha, const void* A, int64_t sA0,
               int64_t sA1, const void* x, int64_t sx, double beta, const void* y, int64_t sy,
               void* out, void* ws, size_t ws_bytes);
/* out[i] = beta*y[i*sy] + alpha * sum_{p<nparts} part[p*M + i], summed in a fixed order:
 * second stage of the column-layout Gemv and of the fused one-pass GemvChain kernel
 * (generated by pytensor_amd/codegen.py:gemv_chain_source). */
int pthip_gemv_finish(int dtype, int64_t M, int64_t nparts, const void* part, double alpha,
                      double beta, const void* y, int64_t sy, void* out);
/* out[b] (M×N, contiguous) = beta*C[b] + alpha * A[b] (M×K) @ B[b] (K×N); batch stride 0 = shared.
 * beta==0 => C not read (gemm.py:183-216; codegen.py:159-250). MFMA tiles. */
int pthip_gemm(int dtype, int64_t batch, int64_t M, int64_t N, int64_t K, double alpha,
               const void* A, int64_t sAb, int64_t sA0, int64_t sA1, const void* B, int64_t sBb,
               int64_t sB0, int64_t sB1, double beta, const void* C, int64_t sCb, int64_t sC0,
               int64_t sC1, void* out);
/* Split-K products left unfinished for a consumer that folds the sum into its own kernel (the
 * Gemm -> Elemwise pairs of a Scan step: gemm.py:183-216 followed by elemwise.py:755):
 * part[s][b][m][n], s < pthip_gemm_nslabs(batch, M, N, K); the consumer adds the slabs in
 * ascending s and applies alpha/beta itself.  nslabs == 1: slab 0 is the plain product. */
int64_t pthip_gemm_nslabs(int64_t batch, int64_t M, int64_t N, int64_t K);
int pthip_gemm_partials(int dtype, int64_t batch, int64_t M, int64_t N, int64_t K, const void* A,
                        int64_t sAb, int64_t sA0, int64_t sA1, const void* B, int64_t sBb,
                        int64_t sB0, int64_t sB1, void* part, int64_t nslabs);
/* Loop-constant right operand of a skinny product (the recurrent weights of a Scan step:
 * Dot22/Gemm, blas/gemm.py:76,248, whose left operand has a few hundred rows at most), repacked
 * once per evaluation into the MFMA operand order the generated product+epilogue kernels stream:
 * Bp[ceil(N/16)][ceil(K/16)*4][16][4], Bp[ct][k4][j][q] = B[4*k4+q][16*ct+j], zero padded.
 * itemsize 4 or 8; B element strides sB0, sB1. */
int pthip_pack_b16(int itemsize, int64_t K, int64_t N, const void* B, int64_t sB0, int64_t sB1,
                   void* Bp);
/* out (M×N contiguous) = A + alpha * x y^T  (ger.py) */
int pthip_ger(int dtype, int64_t M, int64_t N, double alpha, const void* A, int64_t sA0,
              int64_t sA1, const void* x, int64_t sx, const void* y, int64_t sy, void* out);

/* ---- dense linear algebra (reference calls SciPy LAPACK: Cholesky potrf
 *      decomposition/cholesky.py:48-83; SolveTriangular trtrs solvers/triangular.py:32-71;
 *      CholeskySolve potrs solvers/psd.py:35-53).  Row-major n×n per batch item, contiguous.
 *      Failure (non-PD / singular) => the item's result is NaN-filled. ---- */
int pthip_potrf(int dtype, int lower, int64_t batch, int64_t n, const void* A, void* L);
/* Cholesky(lower) fused with its first triangular solve, x = L^-1 b (b, x: batch x n): the
 * factor stays in LDS between the two (Cholesky -> SolveTriangular(L, b) of a Gaussian logp).
 * Only the LDS-resident size class (n <= ~140 fp64 / ~200 fp32); callers fall back to
 * pthip_potrf + pthip_trsm otherwise. */
int pthip_potrf_trsv(int dtype, int64_t batch, int64_t n, const void* A, const void* b, void* L,
                     void* x);
/* LU with partial pivoting (LAPACK getrf behind scipy.linalg.solve(assume_a="gen"), np.linalg.det /
 * slogdet / inv: solvers/general.py Solve.perform, linalg/summary.py, linalg/inverse.py).
 * A, LU: batch x n x n row-major; perm (int64, batch x n): row i of P*A is row perm[i] of A;
 * sign, logabsdet (dtype, batch): of det(A) (0 and -inf when exactly singular; with
 * flag_singular that also raises bit 1 (value 2) of the device error word -> the LinAlgError of
 * np.linalg.inv at the caller's next sync; Solve NaN-fills and det returns 0 instead). */
int pthip_getrf(int dtype, int64_t batch, int64_t n, const void* A, void* LU, void* perm, void* sign,
                void* logabsdet, int flag_singular);
/* Eigh.perform (pytensor/tensor/linalg/decomposition/eigen.py:177-195, scipy.linalg.eigh of the
 * standard problem): batch x (n, n) symmetric matrices of which only the lower (or upper)
 * triangle is read -> eigenvalues W (batch, n) ascending and eigenvectors as the columns of V
 * (batch, n, n).  Parallel cyclic Jacobi; no convergence raises bit 3 of the device error word
 * (scipy: LinAlgError).  n <= 512. */
int pthip_eigh(int dtype, int64_t batch, int64_t n, int lower, const void* A, void* W, void* V);
/* ARange.perform (pytensor/tensor/basic.py: np.arange(start, stop, step, dtype)) for a length n
 * the host has computed: integers istart + i*istep, floating point fstart + i*delta with
 * delta = (fstart + fstep) - fstart in the output type (NumPy's fill loop). */
int pthip_arange(int dtype, int64_t n, double fstart, double fstep, int64_t istart, int64_t istep, void* out);
/* Eye.perform (np.eye(n, m, k, dtype)): out (n, m), ones on the k-th diagonal */
int pthip_eye(int dtype, int64_t n, int64_t m, int64_t k, void* out);
/* SortOp / ArgSortOp (pytensor/tensor/sort.py:31, 156: np.sort / np.argsort along an axis):
 * `rows` contiguous rows of n elements sorted ascending along the row, NaNs last, ties in
 * position order (stable).  out_vals (same dtype) and/or out_idx (int64 positions within the
 * row) — either may be NULL.  Bitonic network; rows up to 4096 elements in one LDS pass. */
int pthip_sort(int dtype, int64_t rows, int64_t n, const void* in, void* out_vals, void* out_idx);
/* Ordered stream compaction for boolean-mask indexing (AdvancedSubtensor / AdvancedIncSubtensor
 * with a bool index, pytensor/tensor/subtensor.py:1932, 2275: x[mask] == x[mask.nonzero()]) and
 * the Nonzero op (tensor/basic.py): ascending C-order flat indices of the non-zero bytes of the
 * contiguous `mask` (n bytes) into idx_out (room for n int64), their number into count_out
 * (device int64) — the caller reads it back, the output length is data dependent. */
int pthip_nonzero(int64_t n, const void* mask, void* idx_out, void* count_out);
/* RandomVariable.perform (pytensor/tensor/random/op.py; distributions: random/basic.py) — n draws
 * of distribution `dist` from the Philox4x64-10 stream (key[2], counter[4]) of a
 * numpy.random.Generator(Philox): uniform output i is word i%4 of block counter+1+i/4 (the
 * numbers Generator.random(n) returns), every other distribution gives element i the block
 * counter+1+i.  The caller advances the counter by ceil(n/4) (uniform) or n blocks.
 * dist: 0 uniform(low,high) 1 normal(loc,scale) 2 halfnormal 3 lognormal 4 exponential(scale)
 * 5 laplace 6 logistic 7 cauchy 8 halfcauchy 9 gumbel 10 weibull(shape) 11 pareto(b,scale)
 * 12 triangular(left,mode,right) 13 gamma(shape,scale) 14 beta(a,b) 15 invgamma(shape,scale)
 * 16 t(df,loc,scale) 17 bernoulli(p) 18 geometric(p) 19 poisson(lam) 20 integers(low,high)
 * 21 binomial(n,p) 22 negative_binomial(n,p) 23 wald(mean,scale) 24 truncexpon(b,loc,scale)
 * 25 gengamma(alpha,p,lambd) 26 beta_binomial(n,a,b) 27 vonmises(mu,kappa)
 * 28 hypergeometric(ngood,nbad,nsample).
 * params[j]: contiguous array broadcast to the output (stride 1) or one element (stride 0),
 * any numeric dtype.  out: float64 / float32 / int64, contiguous. */
int pthip_random(int dist, int out_dtype, int64_t n, const uint64_t* key, const uint64_t* counter,
                 int nparams, const void* const* params, const int* param_dtypes,
                 const int64_t* param_strides, void* out);
/* CategoricalRV (random/basic.py:1821): one draw per row of p (rows x k, row_stride elements
 * apart, 0 = the same vector for every row) by inversion of the running sum; int64 out */
int pthip_random_categorical(int p_dtype, int64_t rows, int64_t k, const uint64_t* key,
                             const uint64_t* counter, const void* p, int64_t row_stride, void* out);
/* MultinomialRV (random/basic.py:1748 signature "(),(p)->(p)"): rows x k int64 counts, row i from the
 * conditional binomials of n[i * n_stride] trials over p (row_stride as above); category j of a
 * row draws on substream j of that row's block */
int pthip_random_multinomial(int p_dtype, int64_t rows, int64_t k, const uint64_t* key,
                             const uint64_t* counter, const void* n, int n_dtype, int64_t n_stride,
                             const void* p, int64_t row_stride, void* out);
/* SearchsortedOp.perform (extra_ops.py:155-166: np.searchsorted): out[j] (int64) = insertion point of
 * v[j] in the ascending x[n] (read through `sorter` (int64) if not NULL); right = side "right";
 * any numeric dtypes (compared as int64 when both are integers, else as float64, NaN last) */
int pthip_searchsorted(int x_dtype, int64_t n, const void* x, const void* sorter, int v_dtype, int64_t m,
                       const void* v, int right, void* out);
/* Convolve1d.perform (signal/conv.py:124-128: np.convolve): out = a * b, "full" (na + nb - 1 values)
 * or "valid" (|na - nb| + 1); float32 / float64 / int64 */
int pthip_convolve1d(int dtype, int64_t na, const void* a, int64_t nb, const void* b, int full, void* out);
/* Convolve2d.perform (signal/conv.py:260-263: scipy.signal.convolve of two matrices, "full" or
 * "valid"): direct sums, a (ha x wa) and b (hb x wb) contiguous */
int pthip_convolve2d(int dtype, int64_t ha, int64_t wa, const void* a, int64_t hb, int64_t wb, const void* b, int full,
                     void* out);

/* ---- dense decompositions, correct-first tier (csrc/decomp.hip): one workgroup per matrix ---- */
/* QR.perform (linalg/decomposition/qr.py:153-221): LAPACK geqrf in place on `batch` contiguous
 * row-major m x n matrices — Householder vectors below the diagonal (v_k = 1 implicit), R on and
 * above it with dlarfg's signs, tau[batch][min(m,n)] */
int pthip_geqrf(int dtype, int64_t batch, int64_t m, int64_t n, void* A, void* tau);
/* orgqr: Q (m x ncols, contiguous) = H_0 ... H_{k-1} applied to the leading columns of the identity;
 * QR as pthip_geqrf left it (row stride ldqr, qr_stride elements between batch items) */
int pthip_orgqr(int dtype, int64_t batch, int64_t m, int64_t ncols, int64_t k, const void* QR, int64_t ldqr,
                int64_t qr_stride, const void* tau, void* Q);
/* SVD.perform (linalg/decomposition/svd.py:85-93; np.linalg.svd): one-sided Jacobi on the rows of X
 * (r x c contiguous, r <= c <= any, r <= 2048; destroyed): X = P diag(S) Wt, S descending.
 * vectors != 0: Wt (r x c, orthonormal rows; zero rows where S is 0) and Pt = P^T (r x r);
 * Pt_work: r x r scratch.  Non-convergence (60 sweeps) raises bit 2 of the device error word. */
int pthip_svd_rows(int dtype, int64_t batch, int64_t r, int64_t c, int vectors, void* X, void* Pt_work, void* S,
                   void* Wt, void* Pt);
/* rows i of Wt (r x c) with i >= r_valid or S[i] == 0 (S: r_valid per item) are replaced by column
 * i of Q (c x ldq): an orthonormal completion of a rank-deficient or economy-sized factor */
int pthip_fill_null_rows(int dtype, int64_t batch, int64_t r, int64_t c, void* Wt, const void* S, int64_t r_valid,
                         const void* Q, int64_t ldq);
/* dst (rows x cols, contiguous) = the upper (lower != 0: lower) triangle of the leading rows of src
 * (row stride ld, src_stride between batch items), zeros elsewhere, the diagonal 1 if unit_diag:
 * the R of QR.perform (qr.py:171-174), the L and U of LU.perform (lu.py:77-89) */
int pthip_triu(int dtype, int64_t batch, int64_t rows, int64_t cols, const void* src, int64_t ld,
               int64_t src_stride, void* dst, int lower, int unit_diag);
/* LUFactorTridiagonal / SolveLUFactorTridiagonal.perform (linalg/solvers/tridiagonal.py:70-90,
 * 170-180): LAPACK gttrf in place on (dl[n-1], d[n], du[n-1]) -> du2[n-2], ipiv[n] (int32, 1-based
 * as LAPACK returns it); gttrs on B (n x nrhs row-major, in place), trans = 0 | 1 */
int pthip_gttrf(int dtype, int64_t batch, int64_t n, void* dl, void* d, void* du, void* du2, void* ipiv);
int pthip_gttrs(int dtype, int64_t batch, int64_t n, int64_t nrhs, int trans, const void* dl, const void* d,
                const void* du, const void* du2, const void* ipiv, void* B);
/* out (n, n) = P * I for the gather vector perm of pthip_getrf (row i = unit vector e_perm[i]):
 * the right-hand side of MatrixInverse (pytensor/tensor/linalg/inverse.py:87), built on the device */
int pthip_permuted_identity(int dtype, int64_t n, const void* perm, void* out);
/* solve op(T) X = B, T triangular n×n (strided), B n×nrhs (contiguous row-major), out contiguous */
int pthip_trsm(int dtype, int lower, int trans, int unit_diag, int64_t batch, int64_t n,
               int64_t nrhs, const void* T, int64_t sTb, int64_t sT0, int64_t sT1, const void* B,
               int64_t sBb, void* out);

/* ---- indexing / data movement (bit-exact tier; pytensor/tensor/subtensor.py:868-2614,
 *      pytensor/tensor/basic.py:1545 Alloc, 2405 Join, compile/ops.py:121 DeepCopyOp) ---- */
/* N-d strided copy with broadcasting (src stride 0): dst[idx·dstr] = src[idx·sstr]; ndim<=6 */
int pthip_copy_strided(int itemsize, int ndim, const int64_t* shape, void* dst,
                       const int64_t* dst_strides, const void* src, const int64_t* src_strides);
/* out[i, :] = x[idx[i], :] rows of `inner` contiguous elements; x has n_rows rows with
 * element stride sx0; negative indices wrap; out-of-range => status flag (see pthip_check_status) */
int pthip_take_rows(int itemsize, int64_t n_idx, int64_t inner, const void* x, int64_t n_rows,
                    int64_t sx0, const int64_t* idx, void* out);
/* out = x (already copied by the caller); out[idx[i], :] (+)= y[i, :] — AdvancedIncSubtensor
 * (subtensor.py:2275).  inc: deterministic (index-sorted segmented sum), set: last write wins
 * in index order like NumPy. y_stride0 = 0 broadcasts a single row. */
size_t pthip_scatter_rows_workspace(int64_t n_idx, int64_t n_rows, int64_t inner);
int pthip_scatter_rows(int dtype, int inc, int64_t n_idx, int64_t inner, void* out, int64_t n_rows,
                       const int64_t* idx, const void* y, int64_t y_stride0, void* ws,
                       size_t ws_bytes);
/* Second stages in one launch: up to 16 row-major partial slabs parts[k] of shape [nparts[k], M[k]]
 * (per-workgroup partials of a split Gemv / scatter-add — pytensor/tensor/blas/gemv.py:64-108 —
 * or of a CAReduce, elemwise.py:1233) are reduced with ops[k] (pthip_reduce_op ADD/MUL/MAX/MIN)
 * to outs[k] of shape [S[k], M[k]]; row chunk s covers rows [s*ceil(nparts/S), ...).  Fixed
 * order, deterministic.  The tail kernel of the executor adds the S rows. */
int pthip_multi_finish(int dtype, int n_tasks, const int* ops, const void* const* parts,
                       const int64_t* nparts, const int64_t* M, const int* S, void* const* outs);
/* gather up to 16 small contiguous device buffers into one staging buffer (one launch instead of
 * one D2H copy per Function output; cf. the output loop of pytensor/link/basic.py:683-684) */
int pthip_pack(int n, const void* const* srcs, const int64_t* nbytes, const int64_t* dst_offsets,
               void* dst);
/* ---- Softmax / LogSoftmax over the last axis (pytensor/tensor/special.py:26,67; the reference
 *      inlines them into Max + Composite + Sum + Composite before fusing, rewriting/ofg.py:46-70;
 *      here one kernel, one HBM read + one write).  x, out: contiguous rows x cols; log_ = 1 for
 *      LogSoftmax.  float32 sums accumulate in double like the reference's Sum. ---- */
int pthip_softmax(int dtype, int log_, int64_t rows, int64_t cols, const void* x, void* out);
/* ---- round 6: log-sum-exp as one reduction, and Softmax / LogSoftmax over a NON-trailing axis.  The reference has no
 *      op for the former: pytensor/tensor/math.py logsumexp is rewritten into Max -> Composite -> Sum -> Composite
 *      (its benchmark: tests/benchmarks/test_logsumexp.py:9-37); the latter is Softmax(axis=0) (special.py:26, perform =
 *      scipy.special.softmax).  Semantics of the stabilised graph: shift = isinf(max) ? 0 : max; NaN propagates.
 *      float32: exp in float32, the sum in double (the reference's Sum accumulator, elemwise.py:1383-1417). ---- */
/* out[r] = log(sum_j exp(x[r, j])), x contiguous rows x cols (a row per thread up to 32 columns, else per wave, streamed:
 * meant for many rows; pthip_logsumexp_rows_max = the longest row accepted) */
int pthip_logsumexp_rows(int dtype, int64_t rows, int64_t cols, const void* x, void* out);
int64_t pthip_logsumexp_rows_max(int dtype);
/* x contiguous (batch, R, C), the statistic runs over R: out[b, c] = log(sum_r exp(x[b, r, c])) /
 * out[b, r, c] = softmax or log-softmax of x[b, :, c].  ws: pthip_colstat_workspace(dtype, batch, R, C) bytes of device
 * memory (per-split (max, sum) pairs, folded in a fixed order: deterministic). */
size_t pthip_colstat_workspace(int dtype, int64_t batch, int64_t R, int64_t C);
int pthip_logsumexp_cols(int dtype, int64_t batch, int64_t R, int64_t C, const void* x, void* out, void* ws, size_t ws_bytes);
int pthip_softmax_cols(int dtype, int log_, int64_t batch, int64_t R, int64_t C, const void* x, void* out, void* ws, size_t ws_bytes);
/* ---- order-defined scans (pytensor/tensor/extra_ops.py CumOp.perform = np.cumsum/np.cumprod;
 *      pytensor/tensor/math.py Argmax.perform = np.argmax over the flattened trailing axes) ---- */
/* dst[o, k, i] = fold_{j<=k} src[o, j, i] (mul = 0: +, 1: *), strictly left to right; src and
 * dst contiguous (outer, n, inner) */
int pthip_cumulative(int dtype, int mul, int64_t outer, int64_t n, int64_t inner, const void* src,
                     void* dst);
/* integer Dot (pytensor/tensor/math.py Dot.perform = np.dot without BLAS): out (M x N, contiguous)
 * = A (M x K, strided) @ B (K x N, strided) with wrap-around arithmetic in `dtype` */
int pthip_imatmul(int dtype, int64_t M, int64_t N, int64_t K, const void* A, int64_t sA0,
                  int64_t sA1, const void* B, int64_t sB0, int64_t sB1, void* out);
/* out[row] (int64) = index of the first maximum of src[row, 0:R] (a NaN is the maximum) */
int pthip_argmax(int dtype, int64_t rows, int64_t R, const void* src, void* out);
/* device-side error flag raised by kernels (index out of bounds ...); sync + read + clear */
int pthip_check_status(int* status);
/* device address of that flag, for generated (JIT) kernels that bounds-check indices */
void* pthip_status_ptr(void);
/* Launch-per-step forms instead of the persistent / cooperative linear-algebra kernels (Cholesky task graph -> blocked
 * steps, vector triangular solve -> blocked solve, LU panel -> one-workgroup sweep), process-wide, until switched off
 * again.  What the host side turns on — and then evaluates the graph once more — when status bit 4 reports that a
 * bounded dependency wait expired: a cooperative LU panel whose workgroups were not all resident because another
 * process's kernels held the device (the reference has no such failure mode: LAPACK on the host,
 * pytensor/tensor/linalg/decomposition/lu.py:239-300).  Returns the previous setting. */
int pthip_set_safe_mode(int on);

/* out[b] (n x n contiguous) = the symmetric matrix defined by the lower (or upper) triangle of A[b]:
 * the operand convention of scipy.linalg.eigh(a, b, lower=...) (Eigh.perform, eigen.py:177-186),
 * used by the device reduction of the generalised problem to a standard one. */
int pthip_symmetrize(int dtype, int64_t batch, int64_t n, int lower, const void* A, void* out);
/* LUFactor (pytensor/tensor/linalg/decomposition/lu.py:239-300: scipy getrf -> (LU, pivots), LU
 * NaN-filled when a pivot is exactly zero) = pthip_getrf + this finish: the 0-based LAPACK
 * interchange vector piv[batch][n] (int32) from the row order perm[batch][n] (int64), and the NaN
 * fill.  PivotToPermutations (lu.py:206-231): out[batch][n] (int64) from pivots (int32/int64). */
int pthip_lu_factor_finish(int dtype, int64_t batch, int64_t n, void* LU, const void* perm, void* piv);
int pthip_pivots_to_perm(int itemsize, int inverse, int64_t batch, int64_t n, const void* piv, void* out);

/* ---- the one data-path collective -----------------------------------------------------------
 * north_star: "RCCL over xGMI only for the rare explicit all-reduce Op" — the Op is
 * pytensor_amd/collective.py (the reference has no distributed layer: SURVEY.md §5 last row, §8e;
 * closest reference interface: Op.perform of an ordinary Apply, pytensor/graph/op.py).  One process
 * per GPU.  Rank 0 calls pthip_comm_unique_id and ships the 128 bytes to its peers over the control
 * plane; every rank then calls pthip_comm_init.  pthip_all_reduce reduces `n` elements in place over
 * all ranks on the context stream (stream-ordered, no device synchronisation): op 0 sum, 1 prod,
 * 2 max, 3 min.  Without a communicator it is the identity.  librccl.so is loaded on first use. */
int pthip_comm_unique_id(void* id128);
int pthip_comm_init(int nranks, int rank, const void* id128);
int pthip_comm_size(int* nranks, int* rank);
int pthip_comm_destroy(void);
int pthip_all_reduce(int dtype, int op, int64_t n, void* buf);

#ifdef __cplusplus
}
#endif
#endif /* PTHIP_H */
