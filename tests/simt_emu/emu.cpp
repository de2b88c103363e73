// tests/simt_emu/emu.cpp -- cooperative-fibre SIMT emulator (TEST INFRASTRUCTURE ONLY).
//
// Executes a HIP-style kernel body on the host: one ucontext fibre per thread of a
// workgroup, round-robin switched at __syncthreads().  Workgroups of one launch are
// distributed over a few OS threads.  This is deliberately simple and slow; it is used
// only by tests/ to run the engine's real kernel sources in a container without a GPU.
#define LPC_SIMT_EMU 1
#include "lpc_rt.h"

#include <ucontext.h>
#include <atomic>
#include <thread>
#include <vector>

// Sanitizer flavour (build_emu.sh --san: -fsanitize=address,undefined): AddressSanitizer has to be told about every
// switch between the scheduler's stack and a fibre's, and the LDS of a workgroup becomes an exact-size heap block per
// workgroup, so that an index one element past the tile is a report instead of a read of the slack behind it.
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#define LPC_EMU_ASAN 1
#endif

namespace lpc_emu {

namespace {
constexpr size_t kStack = 96 * 1024;

struct Worker {
  std::vector<ucontext_t> fibres;
  std::vector<char> stacks;
  std::vector<ThreadCtx> ctxs;
  std::vector<char> done;
  std::vector<char> smem;
  ucontext_t sched;
#if defined(LPC_EMU_ASAN)
  const void* sched_stack = nullptr;     // the scheduler's (= the OS thread's) stack, learnt at the first switch back
  size_t sched_size = 0;
  void* exact_smem = nullptr;
#endif
  int cur = -1;
  int nthreads = 0;
  const std::function<void()>* body = nullptr;
};

thread_local Worker* tl_worker = nullptr;

#if defined(LPC_EMU_ASAN)
// scheduler -> fibre t
static void switch_to_fibre(Worker& w, int t) {
  void* fake = nullptr;
  __sanitizer_start_switch_fiber(&fake, w.stacks.data() + (size_t)t * kStack, kStack);
  swapcontext(&w.sched, &w.fibres[t]);
  __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
}
// fibre -> scheduler (at a barrier: the fibre lives on; at its end: its fake stack is destroyed)
static void switch_to_sched(Worker& w, int me, bool dying) {
  void* fake = nullptr;
  __sanitizer_start_switch_fiber(dying ? nullptr : &fake, w.sched_stack, w.sched_size);
  swapcontext(&w.fibres[me], &w.sched);
  __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
}
#endif

void fibre_entry() {
  Worker* w = tl_worker;
#if defined(LPC_EMU_ASAN)
  __sanitizer_finish_switch_fiber(nullptr, &w->sched_stack, &w->sched_size);
#endif
  (*w->body)();
  w->done[w->cur] = 1;
#if defined(LPC_EMU_ASAN)
  switch_to_sched(*w, w->cur, true);     // (never resumed)
#endif
  // returning resumes uc_link == scheduler
}

void run_block(Worker& w, dim3 bid, dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
  const int nt = (int)(block.x * block.y * block.z);
  if ((int)w.fibres.size() < nt) {
    w.fibres.resize(nt);
    w.stacks.resize((size_t)nt * kStack);
    w.ctxs.resize(nt);
    w.done.resize(nt);
  }
#if defined(LPC_EMU_ASAN)
  std::free(w.exact_smem);
  w.exact_smem = std::malloc(smem_bytes ? smem_bytes : 1);     // 16-byte aligned, red zones on both sides
  char* smem = (char*)w.exact_smem;
#else
  if (w.smem.size() < smem_bytes + 64) w.smem.resize(smem_bytes + 64);
  char* smem = w.smem.data();
  smem += (64 - ((uintptr_t)smem & 63)) & 63;
#endif
  w.nthreads = nt;
  w.body = &body;
  for (int t = 0; t < nt; ++t) {
    ThreadCtx& c = w.ctxs[t];
    c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    c.bid = bid; c.bdim = block; c.gdim = grid; c.smem = smem;
    w.done[t] = 0;
    getcontext(&w.fibres[t]);
    w.fibres[t].uc_stack.ss_sp = w.stacks.data() + (size_t)t * kStack;
    w.fibres[t].uc_stack.ss_size = kStack;
    w.fibres[t].uc_link = &w.sched;
    makecontext(&w.fibres[t], (void (*)())fibre_entry, 0);
  }
  // round-robin until every fibre has finished
  int remaining = nt;
  while (remaining > 0) {
    for (int t = 0; t < nt; ++t) {
      if (w.done[t]) continue;
      w.cur = t;
#if defined(LPC_EMU_ASAN)
      switch_to_fibre(w, t);
#else
      swapcontext(&w.sched, &w.fibres[t]);
#endif
      if (w.done[t]) --remaining;
    }
  }
  w.cur = -1;
#if defined(LPC_EMU_ASAN)
  std::free(w.exact_smem);
  w.exact_smem = nullptr;
#endif
}
}  // namespace

ThreadCtx& ctx() { return tl_worker->ctxs[tl_worker->cur]; }

void barrier() {
  Worker* w = tl_worker;
  int me = w->cur;
#if defined(LPC_EMU_ASAN)
  switch_to_sched(*w, me, false);
#else
  swapcontext(&w->fibres[me], &w->sched);  // scheduler resumes the next fibre; we continue next round
#endif
}

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
  const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  if (nblocks == 0) return;
  unsigned nworkers = std::thread::hardware_concurrency();
  if (nworkers == 0) nworkers = 1;
  if (nworkers > 8) nworkers = 8;
  if (const char* e = std::getenv("LPC_EMU_THREADS")) nworkers = (unsigned)std::max(1, atoi(e));
  if (nblocks < nworkers) nworkers = (unsigned)nblocks;
  std::atomic<size_t> next{0};
  auto work = [&]() {
    Worker w;
    tl_worker = &w;
    for (;;) {
      size_t b = next.fetch_add(1);
      if (b >= nblocks) break;
      dim3 bid((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y)));
      run_block(w, bid, grid, block, smem_bytes, body);
    }
    tl_worker = nullptr;
  };
  if (nworkers == 1) { work(); return; }
  std::vector<std::thread> th;
  for (unsigned i = 0; i < nworkers; ++i) th.emplace_back(work);
  for (auto& t : th) t.join();
}

}  // namespace lpc_emu

// Self-test of the sanitizer flavour (tests/test_sanitizer.py): a two-lane workgroup reads one byte past its 64 bytes of
// LDS, across a barrier (= on a fibre stack, after a switch).  The plain build reads the slack behind the tile; the
// --san build must die here with a heap-buffer-overflow report.
extern "C" int lpc_emu_selftest_lds_overrun() {
  volatile int sink = 0;
  lpc_emu::launch(dim3(1), dim3(2), 64, [&]() {
    char* s = lpc_emu::ctx().smem;
    lpc_emu::barrier();
    sink = sink + s[64 + (int)lpc_emu::ctx().tid.x];
  });
  return sink;
}

