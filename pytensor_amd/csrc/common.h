// common.h — shared host-side helpers for libpthip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/pthip.h"

namespace pthip {

constexpr int kMaxStreams = 4;

struct Context {
  int device = -1;
  hipStream_t stream = nullptr;               // CURRENT stream: every launch/copy goes here
  hipStream_t streams[kMaxStreams] = {};      // streams[0] = origin (the eager path never leaves it)
  int current = 0;
  int* status_dev = nullptr;  // device-side error flag
  bool capturing = false;
};

Context& ctx();
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

}  // namespace pthip
