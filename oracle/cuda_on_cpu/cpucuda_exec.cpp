// TEST INFRASTRUCTURE — block executor of the "CUDA on the CPU" shim (see cuda_runtime.h).
// Each thread of a block is a coroutine with its own small stack; the scheduler resumes the live threads of the
// block round robin.  __syncthreads() and __shfl_down_sync() are generation barriers over the block / the 32-lane
// warp, so divergent-but-legal code (warp 0 alone running a second shuffle tree, as reduce.cu:118-122 does) works.
#include "cuda_runtime.h"

#include <cstdio>
#include <vector>

#if !defined(__x86_64__)
#error "cpucuda_exec.cpp implements its context switch for x86-64 only"
#endif

uint3 threadIdx{0, 0, 0}, blockIdx{0, 0, 0};
dim3 blockDim, gridDim;

extern "C" void cpucuda_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl cpucuda_switch
.type cpucuda_switch,@function
cpucuda_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size cpucuda_switch,.-cpucuda_switch
)");

namespace cpucuda {
namespace {

constexpr size_t STACK_BYTES = 64 * 1024;

struct Warp {
  uint64_t buf[2][32];
  unsigned arrived = 0, live = 0, gen = 0;
};
struct Thread {
  void* sp = nullptr;
  bool done = false;
  uint3 tid{0, 0, 0};
  unsigned linear = 0;
};
struct Block {
  std::vector<Thread> threads;
  std::vector<Warp> warps;
  unsigned live = 0, arrived = 0, gen = 0;
  const std::function<void()>* body = nullptr;
  void* sched_sp = nullptr;
  int current = -1;
  char* stacks = nullptr;
  size_t stacks_cap = 0;
};
Block B;

void yield_to_scheduler() {
  Thread& t = B.threads[B.current];
  cpucuda_switch(&t.sp, B.sched_sp);
}
void release_block_barrier_if_complete() {
  if (B.live > 0 && B.arrived == B.live) { B.arrived = 0; B.gen++; }
}
void release_warp_if_complete(Warp& w) {
  if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; w.gen++; }
}
void entry() {
  (*B.body)();
  Thread& t = B.threads[B.current];
  t.done = true;
  B.live--;
  Warp& w = B.warps[t.linear / 32];
  w.live--;
  // a thread that exits no longer takes part in barriers: whoever is waiting may now be complete
  release_block_barrier_if_complete();
  release_warp_if_complete(w);
  void* dead = nullptr;
  cpucuda_switch(&dead, B.sched_sp);
  abort();  // a finished coroutine is never resumed
}

}  // namespace

void syncthreads() {
  const unsigned my = B.gen;
  B.arrived++;
  release_block_barrier_if_complete();
  while (B.gen == my) yield_to_scheduler();
}

uint64_t shfl_down(uint64_t bits, unsigned delta, unsigned width) {
  Thread& t = B.threads[B.current];
  Warp& w = B.warps[t.linear / 32];
  const unsigned lane = t.linear % 32;
  const unsigned my = w.gen;
  w.buf[my & 1][lane] = bits;
  w.arrived++;
  release_warp_if_complete(w);
  while (w.gen == my) yield_to_scheduler();
  // CUDA: source lane = lane + delta if it stays inside the lane's `width`-sized segment, else the lane's own value
  const unsigned src = lane + delta;
  if ((lane % width) + delta >= width || src >= 32) return bits;
  return w.buf[my & 1][src];
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  const unsigned n = block.x * block.y * block.z;
  if (n == 0 || grid.x * grid.y * grid.z == 0) return;
  if (B.current >= 0) { fprintf(stderr, "cpucuda: nested launch\n"); abort(); }
  gridDim = grid;
  blockDim = block;
  if (B.stacks_cap < (size_t)n * STACK_BYTES) {
    free(B.stacks);
    B.stacks_cap = (size_t)n * STACK_BYTES;
    if (posix_memalign((void**)&B.stacks, 4096, B.stacks_cap) != 0) abort();
  }
  B.body = &body;
  B.threads.assign(n, Thread{});
  const unsigned nwarps = (n + 31) / 32;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = uint3{bx, by, bz};
        B.warps.assign(nwarps, Warp{});
        B.live = n; B.arrived = 0; B.gen = 0;
        unsigned lin = 0;
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx, ++lin) {
              Thread& t = B.threads[lin];
              t.done = false;
              t.tid = uint3{tx, ty, tz};
              t.linear = lin;
              B.warps[lin / 32].live++;
              // fresh stack: [top-8] fake return address of entry(), [top-16] entry, then six callee-saved registers
              char* top = B.stacks + (size_t)(lin + 1) * STACK_BYTES;
              void** sp = (void**)top;
              *--sp = nullptr;
              *--sp = (void*)&entry;
              for (int r = 0; r < 6; ++r) *--sp = nullptr;
              t.sp = (void*)sp;
            }
        while (B.live > 0) {
          const unsigned before = B.live;
          unsigned progressed = 0;
          for (unsigned i = 0; i < n; ++i) {
            Thread& t = B.threads[i];
            if (t.done) continue;
            threadIdx = t.tid;
            B.current = (int)i;
            cpucuda_switch(&B.sched_sp, t.sp);
            ++progressed;
          }
          B.current = -1;
          (void)before; (void)progressed;
        }
      }
  B.body = nullptr;
}

}  // namespace cpucuda
