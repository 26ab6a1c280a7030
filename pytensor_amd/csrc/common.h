// common.h — shared host-side helpers for libpthip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/pthip.h"

namespace pthip {

constexpr int kMaxStreams = 4;

// A recorded launch sequence (the C++ launch-plan executor of SURVEY §8f row 2, the analogue of
// the reference's lazylinker_c.c thunk walk): every kernel launch / async copy issued while a
// recorder is attached is kept as a closure with its arguments resolved, and replayed as DIRECT
// launches.  For a handful of kernels that beats a hipGraph: no graph-launch floor on the host
// (≈8 µs per hipGraphLaunch) and no graph-to-graph boundary on the device (≈14 µs behind an event
// wait, profiles/r2n_c4_timeline.md) — a direct launch behind a stream wait starts ≈2 µs after its
// predecessor ends.
struct LaunchList {
  std::vector<std::function<hipError_t(hipStream_t)>> ops;
  bool ok = true;        // false: something was issued that a replay cannot repeat
  std::string why;
};

struct Context {
  int device = -1;
  hipStream_t stream = nullptr;               // CURRENT stream: every launch/copy goes here
  hipStream_t streams[kMaxStreams] = {};      // streams[0] = origin (the eager path never leaves it)
  int current = 0;
  int* status_dev = nullptr;  // device-side error flag
  bool capturing = false;
  LaunchList* recorder = nullptr;  // attached between pthip_record_begin / pthip_record_end
  long long launch_count = 0;      // kernel launches + async copies issued so far (plan sizing)
  bool safe_mode = false;          // pthip_set_safe_mode: no persistent / cooperative linear-algebra kernels
};

Context& ctx();
bool guard_overlaps_active(const void* p, size_t bytes);  // guard.hip: the range is write-protected right now
// gemm.hip: C (M x N, row stride ldc) <- beta*C + alpha * A @ B in place (element strides)
int gemm_inplace(int dtype, long long M, long long N, long long K, double alpha, const void* A, long long sA0,
                 long long sA1, const void* B, long long sB0, long long sB1, double beta, void* C, long long ldc);
// gemm_skinny.hip: tall-and-skinny products (one side ~1e6 long, the other <= 16 wide) on their own HBM-streaming kernels;
// *handled = false leaves the product to the MFMA tiles of gemm.hip
int gemm_skinny(int dtype, long long M, long long N, long long K, double alpha, const void* A, long long sA0, long long sA1, const void* B,
                long long sB0, long long sB1, double beta, const void* C, long long sC0, long long sC1, void* out, bool* handled);
int set_error(const char* fmt, ...);
int check(hipError_t e, const char* what);

#define PTHIP_CHECK(expr)                             \
  do {                                                \
    hipError_t _e = (expr);                           \
    if (_e != hipSuccess) return ::pthip::check(_e, #expr); \
  } while (0)

#define PTHIP_REQUIRE_INIT()                                              \
  do {                                                                    \
    if (::pthip::ctx().device < 0) {                                      \
      int _r = pthip_init(0);                                             \
      if (_r) return _r;                                                  \
    }                                                                     \
  } while (0)

inline int dtype_size(int dt) {
  switch (dt) {
    case PTHIP_BOOL: case PTHIP_I8: case PTHIP_U8: return 1;
    case PTHIP_I16: case PTHIP_U16: case PTHIP_F16: return 2;
    case PTHIP_I32: case PTHIP_U32: case PTHIP_F32: return 4;
    case PTHIP_I64: case PTHIP_U64: case PTHIP_F64: return 8;
  }
  return 0;
}

inline int post_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return check(e, what);
  return 0;
}

constexpr int kNumCU = 256;  // MI355X
constexpr int kLdsPerCU = 160 * 1024;  // bytes of LDS per CU (and the most one workgroup can ask for)

// async copies / fills issued by the library: performed now and, while a recorder is attached,
// kept for replay (host memory involved must outlive the list: the plan's pinned blocks do)
inline hipError_t memcpy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
  Context& c = ctx();
  c.launch_count++;
  if (c.recorder)
    c.recorder->ops.emplace_back([=](hipStream_t s) { return hipMemcpyAsync(dst, src, bytes, kind, s); });
  return hipMemcpyAsync(dst, src, bytes, kind, st);
}
inline hipError_t memset_async(void* dst, int byte, size_t bytes, hipStream_t st) {
  Context& c = ctx();
  c.launch_count++;
  if (c.recorder)
    c.recorder->ops.emplace_back([=](hipStream_t s) { return hipMemsetAsync(dst, byte, bytes, s); });
  return hipMemsetAsync(dst, byte, bytes, st);
}

}  // namespace pthip

// Every kernel launch of the library goes through this macro: launch now, and — while a recorder is
// attached — keep a closure that repeats the launch on a given stream (arguments captured by value).
#define PTHIP_KLAUNCH(kernel, grid, block, shmem, stream, ...)                                    \
  do {                                                                                            \
    ::pthip::Context& pk_ctx_ = ::pthip::ctx();                                                   \
    pk_ctx_.launch_count++;                                                                       \
    if (pk_ctx_.recorder)                                                                         \
      pk_ctx_.recorder->ops.emplace_back([=](hipStream_t pk_s_) -> hipError_t {                   \
        hipLaunchKernelGGL(kernel, grid, block, shmem, pk_s_, __VA_ARGS__);                       \
        return hipGetLastError();                                                                 \
      });                                                                                         \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                          \
  } while (0)
