// guard.hip — write tracking for host arrays that are mirrored in HBM.
//
// A shared variable's value lives in a host ndarray the reference reads on EVERY call
// (pytensor/compile/sharedvalue.py:97-130: `get_value(borrow=True)` hands out the storage itself, an
// in-place edit is seen by the next call).  The hip linker keeps such arrays resident in HBM, so it
// has to learn about CPU stores into them without re-reading gigabytes per call.  This is the
// page-protection scheme a software DSM uses: after the upload the pages holding the array are made
// read-only; the first store faults, the SIGSEGV handler marks the slot dirty, makes the pages
// writable again and returns — the store is re-executed and succeeds.  A clean array costs the call
// one load of the slot's flag.  Sound for every CPU store (any view, any thread, C extensions);
// the one thing it cannot see is the kernel writing into the pages on the process's behalf
// (`read(2)` / `readinto` / `recv_into` into the buffer fails with EFAULT instead) — documented in
// INTEGRATION.md and in the help text of the `hip__resident` flag.  Ranges the HIP runtime knows (pinned or
// registered host memory) are refused: the caller hashes those.  Our own uploads read a protected source
// through pinned bounce buffers (pthip_h2d, runtime.hip) instead of lifting the protection.
//
// Host code only (no kernels); lives in libpthip.so because the handler must be native.
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>

#include "common.h"

namespace {

constexpr int kSlots = 128;

struct Slot {
  std::atomic<int> active{0};  // pages are currently read-only on behalf of this slot
  int used = 0;                // handed out (protect .. release)
  uintptr_t lo = 0, hi = 0;    // page-rounded [lo, hi)
  volatile int dirty = 0;      // a store (or a conservative event) happened since protect
  // the ragged ends of the array (the partial pages either side of the protected interior), kept as copies:
  // pthip_guard_clean compares them natively (one call per resident and evaluation instead of two hashes in Python)
  const unsigned char* edge[2] = {nullptr, nullptr};
  size_t edge_n[2] = {0, 0};
  unsigned char edge_copy[2][4096];
};

Slot g_slots[kSlots];
std::atomic_flag g_lock = ATOMIC_FLAG_INIT;
struct sigaction g_old;
bool g_installed = false;
long g_page = 4096;

struct Lock {
  Lock() { while (g_lock.test_and_set(std::memory_order_acquire)) {} }
  ~Lock() { g_lock.clear(std::memory_order_release); }
};

// (lock held) slot k stops protecting: pages writable again, dirty; every other active slot that
// shares a page with it can no longer trust its protection either.
void open_slot(int k) {
  int stack[kSlots], n = 0;
  bool queued[kSlots] = {};
  stack[n++] = k;
  queued[k] = true;
  while (n) {
    Slot& s = g_slots[stack[--n]];
    if (!s.active.load(std::memory_order_relaxed)) continue;
    s.active.store(0, std::memory_order_relaxed);
    s.dirty = 1;
    mprotect(reinterpret_cast<void*>(s.lo), s.hi - s.lo, PROT_READ | PROT_WRITE);
    for (int j = 0; j < kSlots; ++j) {
      Slot& o = g_slots[j];
      if (!queued[j] && o.active.load(std::memory_order_relaxed) && o.lo < s.hi && s.lo < o.hi) {
        queued[j] = true;
        stack[n++] = j;
      }
    }
  }
}

void on_segv(int sig, siginfo_t* info, void* uctx) {
  uintptr_t a = reinterpret_cast<uintptr_t>(info->si_addr);
  bool mine = false;
  {
    Lock l;
    for (int k = 0; k < kSlots; ++k) {
      Slot& s = g_slots[k];
      if (s.active.load(std::memory_order_relaxed) && a >= s.lo && a < s.hi) {
        open_slot(k);
        mine = true;
      }
    }
  }
  if (mine) return;  // the faulting store is re-executed
  // not ours: whoever was there before us (faulthandler, a runtime's own handler, the default)
  if (g_old.sa_flags & SA_SIGINFO) {
    if (g_old.sa_sigaction) { g_old.sa_sigaction(sig, info, uctx); return; }
  } else if (g_old.sa_handler != SIG_DFL && g_old.sa_handler != SIG_IGN) {
    g_old.sa_handler(sig);
    return;
  }
  struct sigaction dfl;
  memset(&dfl, 0, sizeof dfl);
  dfl.sa_handler = SIG_DFL;
  sigaction(SIGSEGV, &dfl, nullptr);  // returning re-faults into the default action
}

int ensure_handler() {
  struct sigaction cur;
  if (sigaction(SIGSEGV, nullptr, &cur) != 0) return ::pthip::set_error("guard: sigaction query failed");
  if (g_installed && (cur.sa_flags & SA_SIGINFO) && cur.sa_sigaction == on_segv) return 0;
  // first use, or somebody (faulthandler.enable()) replaced us: go (back) in front and chain to them
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_segv;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
  sigemptyset(&sa.sa_mask);
  if (sigaction(SIGSEGV, &sa, &cur) != 0) return ::pthip::set_error("guard: cannot install the SIGSEGV handler");
  g_old = cur;
  g_installed = true;
  g_page = sysconf(_SC_PAGESIZE);
  return 0;
}

}  // namespace

namespace pthip {
// A host range is about to be handed to the HIP runtime as a copy source.  The runtime may pin the pages
// for DMA, which wants them writable — but a READ of the source is not a write: opening the overlapping
// slots (the round-3 behaviour) marked them dirty, so two executables holding the same large shared value
// re-uploaded it on every alternating call.  Instead the caller is told to stage the copy
// through its own pinned bounce buffers (the CPU reads read-only pages without faulting): the protection
// stays in place, nobody's slot turns dirty, and a store that lands during the copy still faults and is seen.
bool guard_overlaps_active(const void* p, size_t bytes) {
  if (!g_installed || !bytes) return false;
  uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
  Lock l;
  for (int k = 0; k < kSlots; ++k) {
    Slot& s = g_slots[k];
    if (s.active.load(std::memory_order_relaxed) && s.lo < hi && lo < s.hi) return true;
  }
  return false;
}
}  // namespace pthip

extern "C" {

int pthip_guard_protect(const void* host_ptr, size_t bytes, int* slot, const int** dirty_flag) {
  if (!host_ptr || !bytes || !slot || !dirty_flag) return ::pthip::set_error("guard_protect: null argument");
  // memory the HIP runtime knows (hipHostMalloc / hipHostRegister — ours or anybody else's, e.g. torch pinned
  // memory behind set_value(borrow=True)) must never be write-protected: the device writes through its own
  // mapping and a CPU-side mprotect makes the driver drop it (profiles/r3b_guard_probe.txt).  The caller
  // falls back to a content hash (coherence.watch).
  {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    hipError_t e = hipPointerGetAttributes(&at, host_ptr);
    if (e == hipSuccess) {
      if (at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged)
        return ::pthip::set_error("guard_protect: the range is pinned / registered with the HIP runtime");
    } else {
      (void)hipGetLastError();  // an ordinary pageable pointer: "invalid value" is the expected answer
    }
  }
  if (int rc = ensure_handler()) return rc;
  uintptr_t lo = reinterpret_cast<uintptr_t>(host_ptr) & ~uintptr_t(g_page - 1);
  uintptr_t hi = (reinterpret_cast<uintptr_t>(host_ptr) + bytes + g_page - 1) & ~uintptr_t(g_page - 1);
  Lock l;
  int k = 0;
  while (k < kSlots && g_slots[k].used) ++k;
  if (k == kSlots) return ::pthip::set_error("guard_protect: all %d slots in use", kSlots);
  Slot& s = g_slots[k];
  s.lo = lo;
  s.hi = hi;
  s.dirty = 0;
  s.edge_n[0] = s.edge_n[1] = 0;
  if (mprotect(reinterpret_cast<void*>(lo), hi - lo, PROT_READ) != 0)
    return ::pthip::set_error("guard_protect: mprotect(PROT_READ) failed for %zu bytes", size_t(hi - lo));
  s.used = 1;
  s.active.store(1, std::memory_order_release);
  *slot = k;
  *dirty_flag = const_cast<const int*>(&s.dirty);
  return 0;
}

int pthip_guard_set_edges(int slot, const void* e0, size_t n0, const void* e1, size_t n1) {
  if (slot < 0 || slot >= kSlots || !g_slots[slot].used) return ::pthip::set_error("guard_set_edges: bad slot %d", slot);
  if (n0 > 4096 || n1 > 4096) return ::pthip::set_error("guard_set_edges: an edge is longer than a page");
  Slot& s = g_slots[slot];
  s.edge[0] = (const unsigned char*)e0;
  s.edge[1] = (const unsigned char*)e1;
  s.edge_n[0] = n0;
  s.edge_n[1] = n1;
  if (n0) memcpy(s.edge_copy[0], e0, n0);
  if (n1) memcpy(s.edge_copy[1], e1, n1);
  return 0;
}

// 1: no store reached the protected pages since `protect` and the ragged ends still hold what they held at
// pthip_guard_set_edges; 0: something changed (or the slot is not in use)
int pthip_guard_clean(int slot) {
  if (slot < 0 || slot >= kSlots) return 0;
  const Slot& s = g_slots[slot];
  if (!s.used || s.dirty) return 0;
  for (int e = 0; e < 2; e++)
    if (s.edge_n[e] && memcmp(s.edge_copy[e], s.edge[e], s.edge_n[e]) != 0) return 0;
  return 1;
}

int pthip_guard_release(int slot) {
  if (slot < 0 || slot >= kSlots) return ::pthip::set_error("guard_release: bad slot %d", slot);
  Lock l;
  if (!g_slots[slot].used) return 0;
  open_slot(slot);
  g_slots[slot].used = 0;
  return 0;
}

int pthip_guard_stats(int* slots_in_use, int* slots_active) {
  Lock l;
  int u = 0, a = 0;
  for (int k = 0; k < kSlots; ++k) {
    u += g_slots[k].used;
    a += g_slots[k].active.load(std::memory_order_relaxed);
  }
  if (slots_in_use) *slots_in_use = u;
  if (slots_active) *slots_active = a;
  return 0;
}

}  // extern "C"
