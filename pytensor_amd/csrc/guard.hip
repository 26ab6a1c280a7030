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
// (`read(2)` into the buffer fails with EFAULT instead) — documented in INTEGRATION.md.
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
// A host range is about to be handed to the HIP runtime as a copy source: the runtime may pin the
// pages for DMA, which wants them writable.  Overlapping slots are opened (conservatively dirty).
void guard_before_host_read(const void* p, size_t bytes) {
  if (!g_installed || !bytes) return;
  uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
  Lock l;
  for (int k = 0; k < kSlots; ++k) {
    Slot& s = g_slots[k];
    if (s.active.load(std::memory_order_relaxed) && s.lo < hi && lo < s.hi) open_slot(k);
  }
}
}  // namespace pthip

extern "C" {

int pthip_guard_protect(const void* host_ptr, size_t bytes, int* slot, const int** dirty_flag) {
  if (!host_ptr || !bytes || !slot || !dirty_flag) return ::pthip::set_error("guard_protect: null argument");
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
  if (mprotect(reinterpret_cast<void*>(lo), hi - lo, PROT_READ) != 0)
    return ::pthip::set_error("guard_protect: mprotect(PROT_READ) failed for %zu bytes", size_t(hi - lo));
  s.used = 1;
  s.active.store(1, std::memory_order_release);
  *slot = k;
  *dirty_flag = const_cast<const int*>(&s.dirty);
  return 0;
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
