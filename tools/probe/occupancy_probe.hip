// Developer probe (GPU box): how many workgroups of a given shape does the dispatcher make RESIDENT at once?
// 512 workgroups; each stamps wall_clock64() (100 MHz) on entry and then spins ~6 us; the number of entries within the
// first microsecond is the number of co-resident workgroups.  Build: hipcc --offload-arch=gfx950 -O2 tools/probe/occupancy_probe.hip -o tools/probe/occupancy_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int VG>
__global__ void probe(unsigned long long* entry, int spin_ticks, float* sink) {
  extern __shared__ float dyn[];
  float keep[VG];
#pragma unroll
  for (int i = 0; i < VG; ++i) keep[i] = (float)(threadIdx.x * (i + 1));
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) entry[blockIdx.x] = t0;
  if (blockDim.x > 100000) dyn[threadIdx.x] = 1.f;
  while (wall_clock64() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
    for (int i = 0; i < VG; ++i) keep[i] = keep[i] * 1.0001f + 0.5f;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < VG; ++i) s += keep[i];
  if (s == 12345.678f) sink[0] = s;
  __syncthreads();
}

// SGPR-pressure variant: NS wave-uniform values loaded up front and all kept live across the spin
template <int VG, int NS>
__global__ void probe_sgpr(unsigned long long* entry, int spin_ticks, float* sink, const float* __restrict__ uni) {
  __shared__ float stat[9729];   // 38916 B static LDS like k_se3_accum
  float u[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) u[i] = uni[i];
  float keep[VG];
#pragma unroll
  for (int i = 0; i < VG; ++i) keep[i] = (float)(threadIdx.x * (i + 1));
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) entry[blockIdx.x] = t0;
  stat[threadIdx.x] = keep[0];
  while (wall_clock64() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
    for (int i = 0; i < VG; ++i) keep[i] = keep[i] * u[i % NS] + u[(i + 7) % NS];
  }
  float s = stat[(threadIdx.x + 1) % blockDim.x];
#pragma unroll
  for (int i = 0; i < VG; ++i) s += keep[i];
#pragma unroll
  for (int i = 0; i < NS; ++i) s += u[i];
  if (s == 12345.678f) sink[0] = s;
  __syncthreads();
}
template <int VG, int NS>
void run_sgpr(int block, unsigned long long* d_entry, float* d_sink, const float* d_uni) {
  const int grid = 512;
  std::vector<unsigned long long> h(grid);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe_sgpr<VG, NS>), dim3(grid), dim3(block), 0, 0, d_entry, 600, d_sink, d_uni);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d_entry, grid * 8, hipMemcpyDeviceToHost);
  const unsigned long long t0 = *std::min_element(h.begin(), h.end());
  int early = 0;
  unsigned long long last = 0;
  for (auto v : h) { if (v - t0 < 100) ++early; last = std::max(last, v - t0); }
  printf("static-lds vgpr~%3d uniforms %3d block %4d : %3d of 512 workgroups entered within 1 us; last entry at %.2f us\n", VG + 10, NS, block, early, last / 100.0);
}

template <int VG>
void run(int block, int lds, unsigned long long* d_entry, float* d_sink) {
  const int grid = 512;
  std::vector<unsigned long long> h(grid);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe<VG>, dim3(grid), dim3(block), lds, 0, d_entry, 600, d_sink);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d_entry, grid * 8, hipMemcpyDeviceToHost);
  const unsigned long long t0 = *std::min_element(h.begin(), h.end());
  int early = 0;
  unsigned long long last = 0;
  for (auto v : h) { if (v - t0 < 100) ++early; last = std::max(last, v - t0); }
  printf("vgpr~%3d block %4d lds %6d : %3d of 512 workgroups entered within 1 us; last entry at %.2f us\n", VG + 10, block, lds, early, last / 100.0);
}

int main() {
  unsigned long long* d_entry;
  float* d_sink;
  hipMalloc(&d_entry, 512 * 8);
  hipMalloc(&d_sink, 4);
  float* d_uni;
  hipMalloc(&d_uni, 128 * 4);
  std::vector<float> hu(128, 1.0001f);
  hipMemcpy(d_uni, hu.data(), 128 * 4, hipMemcpyHostToDevice);
  for (int block : {256, 320, 640}) run<64>(block, 38916, d_entry, d_sink);
  for (int block : {256, 320, 640}) run_sgpr<48, 8>(block, d_entry, d_sink, d_uni);
  for (int block : {256, 320, 640}) run_sgpr<48, 40>(block, d_entry, d_sink, d_uni);
  for (int block : {256, 320, 640}) run_sgpr<48, 70>(block, d_entry, d_sink, d_uni);
  for (int block : {256, 320, 640}) run_sgpr<48, 90>(block, d_entry, d_sink, d_uni);
  return 0;
}
