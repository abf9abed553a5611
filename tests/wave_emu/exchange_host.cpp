// TEST INFRASTRUCTURE — the persistent tracker's granule exchanges (ef_track_fast_persistent.inc), driven by host threads, one per workgroup.
// The product's own column mapping (fast_plan / fast_groups_per_xcd / fast_group_of / ft_column: cut out of the product's sources by
// tests/test_granule_exchange_emulation.py and included below as EXCHANGE_SOURCE) decides who writes where; the exchanges themselves are
// restated at workgroup level — exchange A: every workgroup publishes one record {value, epoch}, everybody waits for all FT_WGS records of this
// epoch; exchange B: every workgroup publishes NA granules G1[a][column], reducer a sweeps G1[a][*], publishes the total G2[a], everybody
// polls the NA totals — with ONE set of slots (no double buffering), a fresh epoch per exchange, `tag == epoch` waits and bounded spins, as in
// the kernel.  What this pins is the PROTOCOL: that no slot is rewritten before every reader of its previous value has moved on — which holds
// when EVERY workgroup publishes into every exchange (zeros without pixels) and fails when pixel-less workgroups only listen (the first
// version of the kernel: a listener that falls behind is lapped and waits for an epoch that is gone).
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#define __host__
#define __device__
#define __forceinline__ inline
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
#ifndef FT_WGS
#error "FT_WGS comes from the command line (the product's value)"
#endif
#include EXCHANGE_SOURCE

typedef unsigned long long u64;
static inline u64 granule(unsigned data, unsigned epoch) { return ((u64)epoch << 32) | data; }
static inline void put(u64* p, unsigned data, unsigned epoch) { __atomic_store_n(p, granule(data, epoch), __ATOMIC_RELAXED); }
static inline u64 get(const u64* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }

// iters iterations; iteration i runs at a "level" of n_pixels[i % n_levels] pixels (NG groups own pixels, the other workgroups do not).
// all_publish = 1: the product's protocol; 0: pixel-less workgroups only listen and sweeps cover the NG columns (the lapping bug).
// sleeper >= 0: that workgroup sleeps sleep_ms before iteration sleeper_at (it must be pixel-less there for the listen-only case to show).
// stats = {iterations completed by workgroup 0, workgroups that saw a time-out, wrong totals seen}
extern "C" int run_exchange(int iters, const int* n_pixels, int n_levels, int na, int all_publish, int spin, int jitter_us, int sleeper, int sleeper_at,
                            int sleep_ms, long long* stats) {
  std::vector<u64> G1((size_t)na * FT_WGS, 0ull), G2((size_t)na, 0ull), GA((size_t)FT_WGS, 0ull);
  std::atomic<long long> timeouts{0}, wrong{0}, done0{0};
  std::atomic<int> abort_flag{0};
  std::vector<std::thread> th;
  for (int w = 0; w < FT_WGS; ++w)
    th.emplace_back([&, w] {
      std::mt19937 rng((unsigned)(w * 7919 + 17));
      unsigned epoch = 1;
      bool dead = false;
      auto wait_all = [&](const u64* base, int n, unsigned e, long long& sum) {   // every granule base[0..n) tagged e; sum of their data
        sum = 0;
        if (dead) return;
        std::vector<char> ok((size_t)n, 0);
        int left = n;
        for (int s = 0; s < spin && left > 0; ++s) {
          for (int i = 0; i < n; ++i)
            if (!ok[i]) {
              const u64 g = get(base + i);
              if ((unsigned)(g >> 32) == e) { ok[i] = 1; --left; sum += (unsigned)g; }
            }
          if (left > 0) {
            if (abort_flag.load(std::memory_order_relaxed)) break;
            std::this_thread::yield();
          }
        }
        if (left > 0) { dead = true; if (!abort_flag.exchange(1)) {} timeouts.fetch_add(1); }
      };
      for (int it = 0; it < iters; ++it) {
        if (w == sleeper && it == sleeper_at) std::this_thread::sleep_for(std::chrono::milliseconds(sleep_ms));
        if (jitter_us > 0 && (rng() & 3) == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % (unsigned)jitter_us));
        const FastPlan fp = fast_plan(n_pixels[it % n_levels]);
        const int g = fast_group_of(w, fp.NG);
        const int col = ft_column(w, g, fp.NG);
        const int ncols = all_publish ? FT_WGS : fp.NG;
        // ---- exchange A ----
        const unsigned ea = epoch++;
        const unsigned va = g >= 0 ? (unsigned)(it + 1) : 0u;                     // a pixel-less workgroup contributes zero
        if (all_publish || g >= 0) put(&GA[(size_t)(all_publish ? col : g)], va, ea);
        long long sa;
        wait_all(GA.data(), ncols, ea, sa);
        if (!dead && sa != (long long)fp.NG * (it + 1)) wrong.fetch_add(1);
        // ---- exchange B, hop 1 ----
        const unsigned eb = epoch++;
        if (all_publish || g >= 0)
          for (int a = 0; a < na; ++a) put(&G1[(size_t)a * FT_WGS + (all_publish ? col : g)], g >= 0 ? (unsigned)(a + 1) : 0u, eb);
        if (w < na) {   // reducer of accumulator w
          long long sb;
          wait_all(G1.data() + (size_t)w * FT_WGS, ncols, eb, sb);
          put(&G2[(size_t)w], (unsigned)sb, eb);
        }
        // ---- hop 2 ----
        long long tot;
        wait_all(G2.data(), na, eb, tot);
        if (!dead && tot != (long long)fp.NG * na * (na + 1) / 2) wrong.fetch_add(1);
        if (w == 0 && !dead) done0.fetch_add(1);
      }
    });
  for (auto& t : th) t.join();
  stats[0] = done0.load();
  stats[1] = timeouts.load();
  stats[2] = wrong.load();
  return (timeouts.load() == 0 && wrong.load() == 0) ? 0 : 1;
}

// the column map of a level: out[w] for w < FT_WGS, groups[w] = the workgroup's group or -1
extern "C" void exchange_columns(int n_pixels, int* columns, int* groups, int* ng) {
  const FastPlan fp = fast_plan(n_pixels);
  *ng = fp.NG;
  for (int w = 0; w < FT_WGS; ++w) {
    groups[w] = fast_group_of(w, fp.NG);
    columns[w] = ft_column(w, groups[w], fp.NG);
  }
}
