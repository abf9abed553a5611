// TEST INFRASTRUCTURE — the persistent tracker's grid barrier (PtSync, pt_arrive, pt_wait: cut out of ef_track_kernels.hip by
// tests/test_grid_barrier_emulation.py and included below as BARRIER_SOURCE) driven by host threads, one per workgroup leader.  The HIP
// atomics become GCC __atomic builtins on the same words, s_sleep a yield, drain_stores a full fence.  What this pins is the PROTOCOL
// (counter re-armed before the generation opens, monotonic generations across "launches", payload regions alternating by parity, the bounded
// spin and the sticky abort flag); memory-model questions of the real machine (agent scope, per-XCD L2) are the GPU suite's business.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#define __device__
#define __forceinline__ inline
#define __ATOMIC_RELAXED_HIP 0
#define __HIP_MEMORY_SCOPE_AGENT 0
template <typename T> static inline T __hip_atomic_load(const T* p, int, int) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
template <typename T> static inline void __hip_atomic_store(T* p, T v, int, int) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T __hip_atomic_fetch_add(T* p, T v, int, int) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline void __builtin_amdgcn_s_sleep(int) { std::this_thread::yield(); }
static inline void drain_stores() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

#ifndef PT_WGS
#error "PT_WGS comes from the command line (the product's value, read out of ef_track.hpp by the test)"
#endif
#ifndef PT_SPIN
#error "PT_SPIN comes from the command line"
#endif
#include BARRIER_SOURCE

// `launches` launches of `iters` iterations each; workgroup w publishes {launch, iteration, w} into its slot of region (iteration & 1), everyone
// meets at the barrier, then every workgroup checks EVERY slot of that region.  jitter_us > 0: random sleeps between the steps, so that fast
// workgroups run a full iteration ahead of slow ones.  missing >= 0: that workgroup never arrives at barrier `missing_at` of the first launch.
// Returns 0 when every check held; fills stats = {barriers completed by workgroup 0, workgroups that saw a time-out, abort flag, mismatches}.
extern "C" int run_barrier(int launches, int iters, int jitter_us, int missing, int missing_at, long long* stats) {
  static_assert(sizeof(unsigned) == 4, "");
  PtSync* Y = new PtSync();
  std::memset(Y, 0, sizeof(PtSync));
  std::vector<unsigned long long> payload((size_t)2 * PT_WGS, 0ull);
  std::atomic<long long> mismatches{0}, timeouts{0}, done0{0};
  for (int l = 0; l < launches; ++l) {
    // a launch reads the generation counter once at its start (all workgroups see the value the previous launch left)
    const unsigned gen0 = __hip_atomic_load(&Y->gen, 0, 0);
    const bool dead_launch = __hip_atomic_load(&Y->abort, 0, 0) != 0;   // later launches on an aborted instance return at once
    if (dead_launch) break;
    std::vector<std::thread> th;
    for (int w = 0; w < PT_WGS; ++w)
      th.emplace_back([&, w, l] {
        std::mt19937 rng((unsigned)(w * 7919 + l));
        bool dead = false;
        unsigned gen_next = gen0;
        for (int it = 0; it < iters; ++it) {
          if (jitter_us > 0 && (rng() & 3) == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % (unsigned)jitter_us));
          const unsigned long long tag = ((unsigned long long)(l + 1) << 40) | ((unsigned long long)(it + 1) << 16) | (unsigned)w;
          __atomic_store_n(&payload[(size_t)(it & 1) * PT_WGS + w], tag, __ATOMIC_RELAXED);
          drain_stores();
          ++gen_next;
          if (l == 0 && w == missing && it == missing_at) return;   // this workgroup is never scheduled again
          pt_arrive(Y);
          const bool gone = pt_wait(Y, gen_next, dead);
          if (gone && !dead) { dead = true; timeouts++; }
          if (!dead) {
            for (int o = 0; o < PT_WGS; ++o) {
              const unsigned long long want = ((unsigned long long)(l + 1) << 40) | ((unsigned long long)(it + 1) << 16) | (unsigned)o;
              if (__atomic_load_n(&payload[(size_t)(it & 1) * PT_WGS + o], __ATOMIC_RELAXED) != want) mismatches++;
            }
            if (w == 0) done0++;
          }
        }
      });
    for (auto& t : th) t.join();
  }
  stats[0] = done0.load();
  stats[1] = timeouts.load();
  stats[2] = (long long)__hip_atomic_load(&Y->abort, 0, 0);
  stats[3] = mismatches.load();
  stats[4] = (long long)__hip_atomic_load(&Y->gen, 0, 0);
  stats[5] = (long long)__hip_atomic_load(&Y->count, 0, 0);
  delete Y;
  return mismatches.load() == 0 ? 0 : 1;
}
