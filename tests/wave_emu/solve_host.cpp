// TEST INFRASTRUCTURE — the product's wave-parallel 6x6 LDL^T (efs::ldlt6_wave, elasticfusion_amd/csrc/ef_solve_dev.hpp: one matrix
// element per lane, pivot search and row/column swaps through cross-lane shuffles) executed on the host: 64 host threads are the 64
// lanes (tests/wave_emu/hip/hip_runtime.h).  Built by tests/test_wave_emulation.py.
#include <hip/hip_runtime.h>
#include <thread>
#include <vector>
#include "ef_solve_dev.hpp"

extern "C" void run_ldlt6_wave(const double* A36, const double* b6, double* x6) {
  static efs::SolveScratch S;
  for (int i = 0; i < 6; ++i) S.b[i] = b6[i];
  unsigned long long slots[64];
  std::barrier<> bar(64);
  std::vector<std::thread> th;
  for (int l = 0; l < 64; ++l)
    th.emplace_back([&, l] {
      emu::lane.tid = {(unsigned)l, 0, 0};
      emu::lane.bdim = {64, 1, 1};
      emu::wave.barrier = &bar; emu::wave.slots = slots; emu::wave.lane = l;
      efs::ldlt6_wave(l < 36 ? A36[l] : 0.0, S);
    });
  for (auto& t : th) t.join();
  for (int i = 0; i < 6; ++i) x6[i] = S.x[i];
}
// ... and the variant that keeps the whole (lower-triangular) matrix in every lane's registers (efs::ldlt6_every_lane): no cross-lane traffic
// but the LDS round trip that gathers the matrix, so four emulated lanes are as good as 64
extern "C" void run_ldlt6_every_lane(const double* A36, const double* b6, double* x6) {
  static efs::SolveScratch S;
  for (int i = 0; i < 6; ++i) S.b[i] = b6[i];
  unsigned long long slots[64];
  std::barrier<> bar(64);
  std::vector<std::thread> th;
  for (int l = 0; l < 64; ++l)
    th.emplace_back([&, l] {
      emu::lane.tid = {(unsigned)l, 0, 0};
      emu::lane.bdim = {64, 1, 1};
      emu::wave.barrier = &bar; emu::wave.slots = slots; emu::wave.lane = l;
      efs::ldlt6_every_lane(l < 36 ? A36[l] : 0.0, S);
    });
  for (auto& t : th) t.join();
  for (int i = 0; i < 6; ++i) x6[i] = S.x[i];
}
// the scalar statement of the same algorithm (efl::ldlt_solve<double, 6>: the Eigen::LDLT restatement the device evaluates on one lane)
extern "C" void run_ldlt6_scalar(const double* A36, const double* b6, double* x6) { efl::ldlt_solve<double, 6>(A36, b6, x6); }
