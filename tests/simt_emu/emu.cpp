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
  int cur = -1;
  int nthreads = 0;
  const std::function<void()>* body = nullptr;
};

thread_local Worker* tl_worker = nullptr;

void fibre_entry() {
  Worker* w = tl_worker;
  (*w->body)();
  w->done[w->cur] = 1;
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
  if (w.smem.size() < smem_bytes + 64) w.smem.resize(smem_bytes + 64);
  char* smem = w.smem.data();
  smem += (64 - ((uintptr_t)smem & 63)) & 63;
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
      swapcontext(&w.sched, &w.fibres[t]);
      if (w.done[t]) --remaining;
    }
  }
  w.cur = -1;
}
}  // namespace

ThreadCtx& ctx() { return tl_worker->ctxs[tl_worker->cur]; }

void barrier() {
  Worker* w = tl_worker;
  int me = w->cur;
  swapcontext(&w->fibres[me], &w->sched);  // scheduler resumes the next fibre; we continue next round
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
