// runtime.hip — context, caching device pool, copies, hipGraph capture, events, hiprtc JIT.
// MI355X-native: one process per GPU, one in-order stream per process, everything
// stream-ordered; memory is pooled in power-of-two-ish buckets and never returned to the
// driver on the hot path (288 GB of HBM3E: reuse, don't free).
#include "common.h"

#include <hip/hiprtc.h>
#include <time.h>

#include <atomic>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace pthip {

static thread_local std::string g_err;
static Context g_ctx;
Context& ctx() { return g_ctx; }

int set_error(const char* fmt, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}

int check(hipError_t e, const char* what) {
  return set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
}

// ---------------------------------------------------------------------------------
// caching pool
// ---------------------------------------------------------------------------------
struct Block {
  void* ptr;
  size_t size;
};

static size_t round_size(size_t n) {
  if (n == 0) n = 1;
  if (n <= 4096) return (n + 255) & ~size_t(255);
  // buckets: 1/8-octave steps above 4 KiB keep internal fragmentation <= 12.5 %
  size_t p = size_t(1) << (63 - __builtin_clzll(n));
  size_t step = p >> 3;
  return (n + step - 1) / step * step;
}

struct Arena {
  std::multimap<size_t, void*> free_list;     // private free blocks
  std::unordered_map<void*, size_t> live;     // blocks handed out
  std::vector<Block> owned;                   // everything the arena owns
  // no_reuse: a freed block is not handed out again before the next rewind.  Needed when a
  // plan runs on several streams: stream order no longer serialises all users of a block.
  bool no_reuse = false;
};

struct Pool {
  std::mutex mu;
  std::multimap<size_t, void*> free_list;
  std::unordered_map<void*, size_t> live;  // ptr -> rounded size
  std::unordered_map<void*, Arena*> arena_of;
  Arena* active = nullptr;
  size_t in_use = 0, reserved = 0, n_allocs = 0;
};
static Pool g_pool;

static int pool_alloc(size_t bytes, void** out) {
  size_t sz = round_size(bytes);
  std::lock_guard<std::mutex> lk(g_pool.mu);
  if (Arena* a = g_pool.active) {
    auto it = a->free_list.lower_bound(sz);
    if (it != a->free_list.end() && it->first == sz) {
      *out = it->second;
      a->free_list.erase(it);
      a->live[*out] = sz;
      return 0;
    }
    if (g_ctx.capturing)
      return set_error("pool: allocation of %zu bytes during graph capture missed the arena "
                       "(launch sequence differs from the warm-up run)", bytes);
    void* p = nullptr;
    PTHIP_CHECK(hipMalloc(&p, sz));
    g_pool.reserved += sz;
    g_pool.n_allocs++;
    a->owned.push_back({p, sz});
    a->live[p] = sz;
    g_pool.arena_of[p] = a;
    *out = p;
    return 0;
  }
  auto it = g_pool.free_list.lower_bound(sz);
  if (it != g_pool.free_list.end() && it->first == sz) {
    *out = it->second;
    g_pool.free_list.erase(it);
  } else {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
      // release cached blocks and retry once
      for (auto& kv : g_pool.free_list) {
        (void)hipFree(kv.second);
        g_pool.reserved -= kv.first;
      }
      g_pool.free_list.clear();
      (void)hipGetLastError();
      e = hipMalloc(&p, sz);
      if (e != hipSuccess) return check(e, "hipMalloc");
    }
    g_pool.reserved += sz;
    g_pool.n_allocs++;
    *out = p;
  }
  g_pool.live[*out] = sz;
  g_pool.in_use += sz;
  return 0;
}

static int pool_free(void* p) {
  if (!p) return 0;
  std::lock_guard<std::mutex> lk(g_pool.mu);
  auto ao = g_pool.arena_of.find(p);
  if (ao != g_pool.arena_of.end()) {
    Arena* a = ao->second;
    auto it = a->live.find(p);
    if (it == a->live.end()) return set_error("pool: double free of arena block %p", p);
    if (!a->no_reuse) a->free_list.emplace(it->second, p);
    a->live.erase(it);
    return 0;
  }
  auto it = g_pool.live.find(p);
  if (it == g_pool.live.end()) return set_error("pool: free of unknown pointer %p", p);
  // Stream-ordered reuse: the single in-order stream guarantees that any kernel
  // enqueued later runs after every earlier user of this block.
  g_pool.free_list.emplace(it->second, p);
  g_pool.in_use -= it->second;
  g_pool.live.erase(it);
  return 0;
}

}  // namespace pthip

using namespace pthip;

extern "C" {

int pthip_init(int device) {
  if (g_ctx.device == device && g_ctx.stream) return 0;
  int n = 0;
  PTHIP_CHECK(hipGetDeviceCount(&n));
  if (n <= 0) return set_error("pthip_init: no HIP device visible");
  if (device < 0 || device >= n) return set_error("pthip_init: device %d out of range (%d visible)", device, n);
  PTHIP_CHECK(hipSetDevice(device));
  // One process per GPU and a latency-critical host loop (one synchronisation per Function
  // call): spin on completion instead of sleeping on an interrupt.  Best effort — the flag
  // is rejected once the primary context is active (e.g. torch initialised HIP first).
  if (getenv("PTHIP_NO_SPIN") == nullptr) {
    (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
    (void)hipGetLastError();
  }
  if (!g_ctx.streams[0]) {
    PTHIP_CHECK(hipStreamCreateWithFlags(&g_ctx.streams[0], hipStreamNonBlocking));
    g_ctx.stream = g_ctx.streams[0];
    g_ctx.current = 0;
  }
  if (!g_ctx.status_dev) {
    PTHIP_CHECK(hipMalloc((void**)&g_ctx.status_dev, 256));
    PTHIP_CHECK(hipMemset(g_ctx.status_dev, 0, 256));
  }
  g_ctx.device = device;
  return 0;
}

// Stream 1 carries the latency-bound chain of a segmented plan (single-workgroup Cholesky /
// triangular solves): create it at the highest queue priority so that its dispatches are not
// queued behind the streaming kernels of stream 0.
static hipError_t create_stream(int i) {
  int lo = 0, hi = 0;
  if (i == 1 && getenv("PTHIP_NO_STREAM_PRIORITY") == nullptr &&
      hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
    return hipStreamCreateWithPriority(&g_ctx.streams[i], hipStreamNonBlocking, hi);
  return hipStreamCreateWithFlags(&g_ctx.streams[i], hipStreamNonBlocking);
}

int pthip_device_count(int* n) {
  hipError_t e = hipGetDeviceCount(n);
  if (e != hipSuccess) {
    *n = 0;
    (void)hipGetLastError();
  }
  return 0;
}

int pthip_device_name(char* buf, size_t buflen) {
  PTHIP_REQUIRE_INIT();
  hipDeviceProp_t prop;
  PTHIP_CHECK(hipGetDeviceProperties(&prop, g_ctx.device));
  snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return 0;
}

const char* pthip_last_error(void) { return g_err.c_str(); }

int pthip_synchronize(void) {
  PTHIP_REQUIRE_INIT();
  for (int i = 0; i < kMaxStreams; i++)
    if (g_ctx.streams[i]) PTHIP_CHECK(hipStreamSynchronize(g_ctx.streams[i]));
  return 0;
}

int pthip_stream_select(int i) {
  PTHIP_REQUIRE_INIT();
  if (i < 0 || i >= kMaxStreams) return set_error("pthip_stream_select: stream %d out of range", i);
  if (!g_ctx.streams[i]) PTHIP_CHECK(create_stream(i));
  g_ctx.stream = g_ctx.streams[i];
  g_ctx.current = i;
  return 0;
}

int pthip_stream_wait(int waiter, int signaler) {
  PTHIP_REQUIRE_INIT();
  if (waiter < 0 || waiter >= kMaxStreams || signaler < 0 || signaler >= kMaxStreams)
    return set_error("pthip_stream_wait: stream out of range");
  if (waiter == signaler) return 0;
  for (int i : {waiter, signaler})
    if (!g_ctx.streams[i]) PTHIP_CHECK(create_stream(i));
  hipEvent_t ev;
  PTHIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  PTHIP_CHECK(hipEventRecord(ev, g_ctx.streams[signaler]));
  PTHIP_CHECK(hipStreamWaitEvent(g_ctx.streams[waiter], ev, 0));
  PTHIP_CHECK(hipEventDestroy(ev));  // released once the recorded work completes / captured as an edge
  return 0;
}

void* pthip_stream(void) { return (void*)g_ctx.stream; }

int pthip_alloc(size_t bytes, void** dptr) {
  PTHIP_REQUIRE_INIT();
  int r = pool_alloc(bytes, dptr);
  // debugging aid: PTHIP_POISON=1 fills every block handed out with 0xFF bytes (NaN as a
  // float, -1 as an integer) on the current stream, so that a read of never-written memory
  // shows up in the results instead of depending on what the pool happened to hold
  static const bool poison = getenv("PTHIP_POISON") != nullptr;
  if (r == 0 && poison && bytes) PTHIP_CHECK(hipMemsetAsync(*dptr, 0xFF, bytes, g_ctx.stream));
  return r;
}

int pthip_free(void* dptr) { return pool_free(dptr); }

int pthip_pool_stats(size_t* in_use, size_t* reserved, size_t* n_allocs) {
  std::lock_guard<std::mutex> lk(g_pool.mu);
  if (in_use) *in_use = g_pool.in_use;
  if (reserved) *reserved = g_pool.reserved;
  if (n_allocs) *n_allocs = g_pool.n_allocs;
  return 0;
}

int pthip_pool_trim(void) {
  PTHIP_REQUIRE_INIT();
  PTHIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  std::lock_guard<std::mutex> lk(g_pool.mu);
  for (auto& kv : g_pool.free_list) {
    (void)hipFree(kv.second);
    g_pool.reserved -= kv.first;
  }
  g_pool.free_list.clear();
  return 0;
}

int pthip_host_alloc(size_t bytes, void** hptr) {
  PTHIP_REQUIRE_INIT();
  PTHIP_CHECK(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
  return 0;
}

int pthip_host_free(void* hptr) {
  if (hptr) PTHIP_CHECK(hipHostFree(hptr));
  return 0;
}

int pthip_h2d(void* dst, const void* src, size_t bytes) {
  PTHIP_REQUIRE_INIT();
  if (!bytes) return 0;
  // (a pageable source would be snapshotted at call time by the runtime, not at replay time)
  if (g_ctx.recorder) { g_ctx.recorder->ok = false; g_ctx.recorder->why = "host-to-device copy"; }
  if (::pthip::guard_overlaps_active(src, bytes)) {
    // the source is write-protected (guard.hip) on behalf of some executable: the runtime would want to pin
    // its pages writable.  Stage through two pinned bounce buffers instead — the CPU copy reads the
    // protected pages without faulting, the protection (and every slot's clean flag) stays as it is.
    constexpr size_t CH = 8u << 20;
    static void* bounce[2] = {nullptr, nullptr};
    static hipEvent_t done[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; b++)
      if (!bounce[b]) {
        PTHIP_CHECK(hipHostMalloc(&bounce[b], CH, hipHostMallocDefault));
        PTHIP_CHECK(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
      }
    size_t off = 0;
    for (int b = 0; off < bytes; b ^= 1) {
      const size_t nb = bytes - off < CH ? bytes - off : CH;
      PTHIP_CHECK(hipEventSynchronize(done[b]));  // (the DMA that last read this buffer)
      memcpy(bounce[b], (const char*)src + off, nb);
      PTHIP_CHECK(hipMemcpyAsync((char*)dst + off, bounce[b], nb, hipMemcpyHostToDevice, g_ctx.stream));
      PTHIP_CHECK(hipEventRecord(done[b], g_ctx.stream));
      off += nb;
    }
    return 0;
  }
  PTHIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g_ctx.stream));
  return 0;
}

int pthip_d2h(void* dst, const void* src, size_t bytes) {
  PTHIP_REQUIRE_INIT();
  if (!bytes) return 0;
  PTHIP_CHECK(memcpy_async(dst, src, bytes, hipMemcpyDeviceToHost, g_ctx.stream));
  return 0;
}

int pthip_d2d(void* dst, const void* src, size_t bytes) {
  PTHIP_REQUIRE_INIT();
  if (!bytes) return 0;
  PTHIP_CHECK(memcpy_async(dst, src, bytes, hipMemcpyDeviceToDevice, g_ctx.stream));
  return 0;
}

int pthip_memset(void* dst, int byte, size_t bytes) {
  PTHIP_REQUIRE_INIT();
  if (!bytes) return 0;
  PTHIP_CHECK(memset_async(dst, byte, bytes, g_ctx.stream));
  return 0;
}

// ---- arena --------------------------------------------------------------------
int pthip_arena_set_no_reuse(void* arena, int no_reuse) {
  if (!arena) return set_error("pthip_arena_set_no_reuse: null arena");
  std::lock_guard<std::mutex> lk(g_pool.mu);
  ((Arena*)arena)->no_reuse = no_reuse != 0;
  return 0;
}

int pthip_arena_begin(void** arena) {
  PTHIP_REQUIRE_INIT();
  std::lock_guard<std::mutex> lk(g_pool.mu);
  if (g_pool.active) return set_error("pthip_arena_begin: an arena is already active");
  Arena* a = (Arena*)*arena;
  if (!a) {
    a = new Arena();
    *arena = a;
  } else {
    if (!a->live.empty())
      return set_error("pthip_arena_begin: %zu blocks of the arena are still live", a->live.size());
    // rewind: every owned block is free again, in deterministic (size, address) order
    a->free_list.clear();
    for (auto& b : a->owned) a->free_list.emplace(b.size, b.ptr);
  }
  g_pool.active = a;
  return 0;
}

int pthip_arena_end(void) {
  std::lock_guard<std::mutex> lk(g_pool.mu);
  g_pool.active = nullptr;
  return 0;
}

int pthip_arena_destroy(void* arena) {
  if (!arena) return 0;
  PTHIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  std::lock_guard<std::mutex> lk(g_pool.mu);
  Arena* a = (Arena*)arena;
  if (g_pool.active == a) g_pool.active = nullptr;
  for (auto& b : a->owned) {
    g_pool.arena_of.erase(b.ptr);
    if (a->live.count(b.ptr)) {
      // Still referenced by the caller — e.g. arrays in frames that a stored exception
      // traceback keeps alive after the plan that owned the arena was closed.  Releasing the
      // memory here would leave those owners with a dangling pointer whose late pthip_free hits
      // whatever the address has become by then (observed: two live arrays sharing one block).
      // The block is adopted by the general pool instead; its owner frees it as usual.
      g_pool.live[b.ptr] = b.size;
      g_pool.in_use += b.size;
      continue;
    }
    (void)hipFree(b.ptr);
    g_pool.reserved -= b.size;
  }
  delete a;
  return 0;
}

// ---- graph capture -----------------------------------------------------------
int pthip_capture_begin(void) {
  PTHIP_REQUIRE_INIT();
  if (g_ctx.capturing) return set_error("pthip_capture_begin: already capturing");
  if (g_ctx.current != 0) return set_error("pthip_capture_begin: select stream 0 first");
  PTHIP_CHECK(hipStreamBeginCapture(g_ctx.streams[0], hipStreamCaptureModeThreadLocal));
  g_ctx.capturing = true;
  return 0;
}

int pthip_capture_end(void** graph_exec) {
  if (!g_ctx.capturing) return set_error("pthip_capture_end: not capturing");
  g_ctx.capturing = false;
  hipGraph_t graph = nullptr;
  g_ctx.stream = g_ctx.streams[0];
  g_ctx.current = 0;
  PTHIP_CHECK(hipStreamEndCapture(g_ctx.streams[0], &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return check(e, "hipGraphInstantiate");
  *graph_exec = (void*)exec;
  return 0;
}

int pthip_graph_launch(void* graph_exec) {
  PTHIP_CHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, g_ctx.streams[0]));
  return 0;
}

int pthip_graph_launch_on(void* graph_exec, int stream) {
  if (stream < 0 || stream >= kMaxStreams) return set_error("pthip_graph_launch_on: bad stream");
  if (!g_ctx.streams[stream]) PTHIP_CHECK(create_stream(stream));
  PTHIP_CHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, g_ctx.streams[stream]));
  return 0;
}

// One native call per Function call (the CVM analogue): parameter upload, the captured
// segments on their streams, and the single synchronisation of the call.
//   segmented (ga, gb, gc != NULL): H2D -> [gb on stream 0 || ga on stream 1] -> gc on stream 0
//   single    (only gb != NULL)   : H2D -> gb on stream 0
// The streaming segment gb is enqueued first (each hipGraphLaunch costs ~6-11 us of host time
// and gb is the critical path: 4.03k vs 3.95k evals/s on C4); PTHIP_PLAN_A_FIRST=1 swaps.
// PTHIP_PLAN_TRACE=1: host-side cost of each step of a replay, averaged over 100 calls (stderr)
struct ReplayTrace {
  bool on = getenv("PTHIP_PLAN_TRACE") != nullptr;
  double acc[8] = {0};
  int n = 0;
  timespec t{};
  void start() { if (on) clock_gettime(CLOCK_MONOTONIC, &t); }
  void lap(int k) {
    if (!on) return;
    timespec u;
    clock_gettime(CLOCK_MONOTONIC, &u);
    acc[k] += (u.tv_sec - t.tv_sec) * 1e6 + (u.tv_nsec - t.tv_nsec) * 1e-3;
    t = u;
  }
  void done() {
    if (!on || ++n < 100) return;
    fprintf(stderr, "[pthip plan] us/call: h2d %.1f  A %.1f  B %.1f  evA+C %.1f  sync %.1f\n",
            acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n);
    for (double& a : acc) a = 0;
    n = 0;
  }
};

// A segment is a captured hipGraph (g*) or a recorded launch list (l*), never both.
static int run_segment(void* g, void* l, hipStream_t st) {
  if (g) PTHIP_CHECK(hipGraphLaunch((hipGraphExec_t)g, st));
  else if (l)
    for (auto& op : ((LaunchList*)l)->ops) PTHIP_CHECK(op(st));
  return 0;
}

int pthip_plan_replay3(void* ga, void* la, void* gb, void* lb, void* gc, void* lc, void* dev_in,
                       const void* host_in, size_t in_bytes, const void* dev_out, void* host_out,
                       size_t out_bytes, int sync) {
  PTHIP_REQUIRE_INIT();
  static hipEvent_t ev_in = nullptr, ev_a = nullptr;
  static ReplayTrace tr;
  hipStream_t s0 = g_ctx.streams[0];
  tr.start();
  // (measured and rejected: a one-workgroup kernel fetching the parameters from the pinned block
  //  instead of the copy engine — 212.6 vs 211.9 us per evaluation of config #4, no gain)
  if (in_bytes) PTHIP_CHECK(hipMemcpyAsync(dev_in, host_in, in_bytes, hipMemcpyHostToDevice, s0));
  tr.lap(0);
  if ((ga || la) && (gc || lc)) {
    if (!g_ctx.streams[1]) PTHIP_CHECK(create_stream(1));
    hipStream_t s1 = g_ctx.streams[1];
    if (!ev_in) {
      PTHIP_CHECK(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
      PTHIP_CHECK(hipEventCreateWithFlags(&ev_a, hipEventDisableTiming));
    }
    static const bool b_first = getenv("PTHIP_PLAN_A_FIRST") == nullptr;
    PTHIP_CHECK(hipEventRecord(ev_in, s0));
    if (b_first) { if (int r = run_segment(gb, lb, s0)) return r; }
    PTHIP_CHECK(hipStreamWaitEvent(s1, ev_in, 0));
    if (int r = run_segment(ga, la, s1)) return r;
    tr.lap(1);
    if (!b_first) { if (int r = run_segment(gb, lb, s0)) return r; }
    tr.lap(2);
    PTHIP_CHECK(hipEventRecord(ev_a, s1));
    PTHIP_CHECK(hipStreamWaitEvent(s0, ev_a, 0));
    if (int r = run_segment(gc, lc, s0)) return r;
    tr.lap(3);
  } else {
    if (int r = run_segment(gb, lb, s0)) return r;
    tr.lap(2);
  }
  // results packed by the last segment: one D2H into the pinned block of THIS call (the caller
  // hands the block to its user without a copy, so the destination changes from call to call)
  if (out_bytes) PTHIP_CHECK(hipMemcpyAsync(host_out, dev_out, out_bytes, hipMemcpyDeviceToHost, s0));
  if (sync) PTHIP_CHECK(hipStreamSynchronize(s0));
  tr.lap(4);
  tr.done();
  return 0;
}

// ---- replay through a descriptor, completion by polling --------------------------------------
// What a replay costs beyond its kernels is fixed cost (tools/ubench/call_lat.hip, MI355X): waiting for the
// stream's completion signal is ~5 us slower than polling a word the LAST kernel of the plan stores into
// pinned host memory behind its results (system-scope release).  `done_word` (pinned, inside the plan's
// result block) is cleared here before anything is launched and polled afterwards; the stream itself is
// synchronised every 256th poll-mode call so that the runtime retires its completion signals.
//   flags bit 0  the latency-chain segment (A, stream 1) reads its staged parameters straight from the pinned
//                block, so it does not depend on the parameter upload: no event between the upload and B
int pthip_plan_replay4(const pthip_replay_desc* d, void* host_out, volatile int* done_word, int sync) {
  PTHIP_REQUIRE_INIT();
  if (sync != 2 && !(d->flags & 1))
    return pthip_plan_replay3(d->ga, d->la, d->gb, d->lb, d->gc, d->lc, d->dev_in, d->host_in, d->in_bytes, d->dev_out,
                              host_out, d->out_bytes, sync);
  static hipEvent_t ev_in = nullptr, ev_a = nullptr;
  static unsigned long long calls = 0;
  hipStream_t s0 = g_ctx.streams[0];
  // flags bit 1 (poll mode only: the tail kernel reports a wait it gave up through the done word).  A descriptor whose
  // join gave up three times — kernels of the two streams do not overlap here, e.g. under rocprofv3 --pmc — goes back
  // to the event for good (the tail kernel's wait is then always satisfied on arrival).
  static std::unordered_map<const void*, int> join_gave_up;
  bool dev_join = (d->flags & 2) && sync == 2;
  if (dev_join) {
    auto it = join_gave_up.find((const void*)d);
    if (it != join_gave_up.end() && it->second >= 3) dev_join = false;
  }
  if (sync == 2) {
    if (!done_word) return set_error("pthip_plan_replay4: poll mode without a done word");
    *done_word = 0;
    std::atomic_thread_fence(std::memory_order_seq_cst);
  }
  if (d->in_bytes) PTHIP_CHECK(hipMemcpyAsync(d->dev_in, d->host_in, d->in_bytes, hipMemcpyHostToDevice, s0));
  if ((d->ga || d->la) && (d->gc || d->lc)) {
    if (!g_ctx.streams[1]) PTHIP_CHECK(create_stream(1));
    hipStream_t s1 = g_ctx.streams[1];
    if (!ev_in) {
      PTHIP_CHECK(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
      PTHIP_CHECK(hipEventCreateWithFlags(&ev_a, hipEventDisableTiming));
    }
    // (only in poll mode: such a call returns after its closing segment finished, so segment A of the NEXT call
    // cannot overtake it.  An asynchronous replay, or a waiting one issued behind asynchronous ones, keeps the
    // event: without it A_{n+1} on stream 1 could overwrite arena buffers the closing segment of call n, still
    // in flight on stream 0, is reading — ADVICE r4)
    const bool a_free = (d->flags & 1) != 0 && sync == 2;
    if (!a_free) PTHIP_CHECK(hipEventRecord(ev_in, s0));
    if (int r = run_segment(d->gb, d->lb, s0)) return r;
    if (!a_free) PTHIP_CHECK(hipStreamWaitEvent(s1, ev_in, 0));
    if (int r = run_segment(d->ga, d->la, s1)) return r;
    if (!dev_join) {  // (device join: the closing segment's tail kernel waits for segment A's signal word itself)
      PTHIP_CHECK(hipEventRecord(ev_a, s1));
      PTHIP_CHECK(hipStreamWaitEvent(s0, ev_a, 0));
    }
    if (int r = run_segment(d->gc, d->lc, s0)) return r;
  } else {
    if (int r = run_segment(d->gb, d->lb, s0)) return r;
  }
  if (d->out_bytes) PTHIP_CHECK(hipMemcpyAsync(host_out, d->dev_out, d->out_bytes, hipMemcpyDeviceToHost, s0));
  if (sync == 1) PTHIP_CHECK(hipStreamSynchronize(s0));
  if (sync == 2) {
    timespec t0{};
    unsigned long long spins = 0;
  poll_again:
    while (__atomic_load_n((const int*)done_word, __ATOMIC_ACQUIRE) == 0) {
      __builtin_ia32_pause();
      if ((++spins & 0x3fff) == 0) {
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        if (!t0.tv_sec) t0 = t;
        else if (t.tv_sec - t0.tv_sec >= 10) {
          // the word never arrived: let the stream say what happened (an asynchronous fault surfaces here)
          PTHIP_CHECK(hipStreamSynchronize(s0));
          if (__atomic_load_n((const int*)done_word, __ATOMIC_ACQUIRE) == 0)
            return set_error("pthip_plan_replay4: the plan finished without storing its done word");
          break;
        }
      }
    }
    if (__atomic_load_n((const int*)done_word, __ATOMIC_ACQUIRE) == 2) {
      // the tail kernel gave up waiting for segment A's signal and did nothing: let A finish, run the closing segment again
      if (!dev_join || !g_ctx.streams[1]) return set_error("pthip_plan_replay4: the tail kernel reported a join it was not asked for");
      join_gave_up[(const void*)d]++;
      PTHIP_CHECK(hipStreamSynchronize(g_ctx.streams[1]));
      PTHIP_CHECK(hipStreamSynchronize(s0));
      *done_word = 0;
      std::atomic_thread_fence(std::memory_order_seq_cst);
      if (int r = run_segment(d->gc, d->lc, s0)) return r;
      t0 = timespec{};
      goto poll_again;
    }
    if ((++calls & 255) == 0) {
      PTHIP_CHECK(hipStreamSynchronize(s0));
      if (g_ctx.streams[1]) PTHIP_CHECK(hipStreamSynchronize(g_ctx.streams[1]));
    }
  }
  return 0;
}

// A zero-initialised int32 slot in device memory for a "last workgroup continues" ticket (the generated
// tail kernel: every workgroup shrinks one piece of a partial slab, the last one to finish runs the chain
// and puts the slot back to zero).  Slots come from one 256 KiB block, handed out round-robin.
static int ticket_block(int** base_out) {
  constexpr size_t N = 65536;
  static int* base = nullptr;
  if (!base) {
    PTHIP_CHECK(hipMalloc((void**)&base, N * sizeof(int)));
    PTHIP_CHECK(hipMemset(base, 0, N * sizeof(int)));
    PTHIP_CHECK(hipDeviceSynchronize());
  }
  *base_out = base;
  return 0;
}

int pthip_ticket_slot(void** slot) {
  PTHIP_REQUIRE_INIT();
  constexpr size_t N = 32768;  // single slots: the first half of the block, handed out round-robin
  static size_t next = 0;
  int* base = nullptr;
  int r = ticket_block(&base);
  if (r) return r;
  *slot = (void*)(base + (next++ % N));
  return 0;
}

// n CONSECUTIVE zero-initialised int32 slots (the per-group tickets of a one-pass N-d reduction — one per output tile,
// each self-resetting like a single slot): the second half of the block, round-robin; a request that would run over
// the end starts again at the beginning of that half.
int pthip_ticket_slots(int n, void** first) {
  PTHIP_REQUIRE_INIT();
  constexpr size_t HALF = 32768;
  if (n <= 0 || (size_t)n > HALF / 4 || !first) return pthip::set_error("pthip_ticket_slots: 1 <= n <= %zu", HALF / 4);
  static size_t next = 0;
  int* base = nullptr;
  int r = ticket_block(&base);
  if (r) return r;
  if (next + (size_t)n > HALF) next = 0;
  *first = (void*)(base + HALF + next);
  next += (size_t)n;
  return 0;
}

int pthip_plan_replay2(void* ga, void* la, void* gb, void* lb, void* gc, void* lc, void* dev_in,
                       const void* host_in, size_t in_bytes, int sync) {
  return pthip_plan_replay3(ga, la, gb, lb, gc, lc, dev_in, host_in, in_bytes, nullptr, nullptr, 0, sync);
}

int pthip_plan_replay(void* ga, void* gb, void* gc, void* dev_in, const void* host_in,
                      size_t in_bytes, int sync) {
  return pthip_plan_replay2(ga, nullptr, gb, nullptr, gc, nullptr, dev_in, host_in, in_bytes, sync);
}

int pthip_graph_destroy(void* graph_exec) {
  if (graph_exec) PTHIP_CHECK(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return 0;
}

// ---- events -----------------------------------------------------------------
int pthip_event_create(void** ev) {
  PTHIP_REQUIRE_INIT();
  hipEvent_t e;
  PTHIP_CHECK(hipEventCreate(&e));
  *ev = (void*)e;
  return 0;
}
int pthip_event_record(void* ev) {
  PTHIP_CHECK(hipEventRecord((hipEvent_t)ev, g_ctx.stream));
  return 0;
}
int pthip_event_synchronize(void* ev) {
  PTHIP_CHECK(hipEventSynchronize((hipEvent_t)ev));
  return 0;
}
int pthip_event_elapsed_ms(void* a, void* b, float* ms) {
  PTHIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return 0;
}
int pthip_event_destroy(void* ev) {
  if (ev) PTHIP_CHECK(hipEventDestroy((hipEvent_t)ev));
  return 0;
}

// ---- JIT ----------------------------------------------------------------------
int pthip_jit_compile(const char* src, const char* name, const char* const* opts, int n_opts,
                      void** code, size_t* code_size, char* log, size_t log_len) {
  hiprtcProgram prog;
  hiprtcResult r = hiprtcCreateProgram(&prog, src, name, 0, nullptr, nullptr);
  if (r != HIPRTC_SUCCESS) return set_error("hiprtcCreateProgram: %s", hiprtcGetErrorString(r));
  std::vector<const char*> o;
  o.push_back("--offload-arch=gfx950");
  o.push_back("-O3");
  o.push_back("-std=c++17");
  for (int i = 0; i < n_opts; i++) o.push_back(opts[i]);
  r = hiprtcCompileProgram(prog, (int)o.size(), o.data());
  size_t ls = 0;
  hiprtcGetProgramLogSize(prog, &ls);
  std::string lg(ls + 1, '\0');
  if (ls) hiprtcGetProgramLog(prog, lg.data());
  if (log && log_len) {
    strncpy(log, lg.c_str(), log_len - 1);
    log[log_len - 1] = 0;
  }
  if (r != HIPRTC_SUCCESS) {
    hiprtcDestroyProgram(&prog);
    return set_error("hiprtc compile of %s failed: %s\n%s", name, hiprtcGetErrorString(r), lg.c_str());
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  void* buf = malloc(cs);
  hiprtcGetCode(prog, (char*)buf);
  hiprtcDestroyProgram(&prog);
  *code = buf;
  *code_size = cs;
  return 0;
}

void pthip_buffer_free(void* p) { free(p); }

int pthip_module_load(const void* code, size_t code_size, void** module) {
  PTHIP_REQUIRE_INIT();
  (void)code_size;
  hipModule_t m;
  PTHIP_CHECK(hipModuleLoadData(&m, code));
  *module = (void*)m;
  return 0;
}

int pthip_module_unload(void* module) {
  if (module) PTHIP_CHECK(hipModuleUnload((hipModule_t)module));
  return 0;
}

int pthip_module_get_function(void* module, const char* name, void** fn) {
  hipFunction_t f;
  PTHIP_CHECK(hipModuleGetFunction(&f, (hipModule_t)module, name));
  *fn = (void*)f;
  return 0;
}

int pthip_launch(void* fn, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx, uint32_t by,
                 uint32_t bz, uint32_t shmem, const void* argbuf, size_t argbuf_bytes) {
  PTHIP_REQUIRE_INIT();
  size_t sz = argbuf_bytes;
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, (void*)argbuf, HIP_LAUNCH_PARAM_BUFFER_SIZE,
                    &sz, HIP_LAUNCH_PARAM_END};
  g_ctx.launch_count++;
  if (g_ctx.recorder) {
    // the argument block is copied: the caller's buffer does not outlive this call
    std::vector<char> args((const char*)argbuf, (const char*)argbuf + argbuf_bytes);
    g_ctx.recorder->ops.emplace_back([=](hipStream_t s) mutable {
      size_t n = args.size();
      void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, (void*)args.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE, &n, HIP_LAUNCH_PARAM_END};
      return hipModuleLaunchKernel((hipFunction_t)fn, gx, gy, gz, bx, by, bz, shmem, s, nullptr, cfg);
    });
  }
  PTHIP_CHECK(hipModuleLaunchKernel((hipFunction_t)fn, gx, gy, gz, bx, by, bz, shmem, g_ctx.stream,
                                    nullptr, config));
  return 0;
}

// ---- launch lists (the C++ launch-plan executor; common.h LaunchList) ------------------------
int pthip_record_begin(void) {
  PTHIP_REQUIRE_INIT();
  if (g_ctx.capturing || g_ctx.recorder) return set_error("pthip_record_begin: already capturing / recording");
  if (g_ctx.current != 0) return set_error("pthip_record_begin: select stream 0 first");
  g_ctx.recorder = new LaunchList();
  g_ctx.capturing = true;  // same discipline as a hipGraph capture: no host reads, no syncs
  return 0;
}

int pthip_record_end(void** list, int64_t* n_ops) {
  if (!g_ctx.recorder) return set_error("pthip_record_end: not recording");
  LaunchList* l = g_ctx.recorder;
  g_ctx.recorder = nullptr;
  g_ctx.capturing = false;
  g_ctx.stream = g_ctx.streams[0];
  g_ctx.current = 0;
  if (n_ops) *n_ops = (int64_t)l->ops.size();
  if (!l->ok) {
    std::string why = l->why;
    delete l;
    *list = nullptr;
    return set_error("pthip_record_end: the sequence cannot be replayed (%s)", why.c_str());
  }
  *list = (void*)l;
  return 0;
}

int pthip_list_launch(void* list, int stream) {
  if (!list) return set_error("pthip_list_launch: null list");
  if (stream < 0 || stream >= kMaxStreams) return set_error("pthip_list_launch: bad stream");
  if (!g_ctx.streams[stream]) PTHIP_CHECK(create_stream(stream));
  for (auto& op : ((LaunchList*)list)->ops) PTHIP_CHECK(op(g_ctx.streams[stream]));
  return 0;
}

int pthip_list_destroy(void* list) {
  delete (LaunchList*)list;
  return 0;
}

int64_t pthip_launch_count(void) { return (int64_t)g_ctx.launch_count; }

int pthip_set_safe_mode(int on) {
  const int was = g_ctx.safe_mode ? 1 : 0;
  g_ctx.safe_mode = on != 0;
  return was;
}

void* pthip_status_ptr(void) {
  if (g_ctx.device < 0 && pthip_init(0)) return nullptr;
  return (void*)g_ctx.status_dev;
}

int pthip_check_status(int* status) {
  PTHIP_REQUIRE_INIT();
  int h = 0;
  PTHIP_CHECK(hipMemcpyAsync(&h, g_ctx.status_dev, sizeof(int), hipMemcpyDeviceToHost, g_ctx.stream));
  PTHIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  if (h) PTHIP_CHECK(hipMemsetAsync(g_ctx.status_dev, 0, sizeof(int), g_ctx.stream));
  *status = h;
  return 0;
}

}  // extern "C"
