// Developer probe: what a kernel of k_se3_accum's SHAPE can reach at 640x480 on this box, measured the way the bench measures it
// (dispatch begin/end timestamps through hipExtLaunchKernelGGL events = the duration rocprofv3 --kernel-trace reports).
//   empty      256 workgroups x 512 threads that do nothing: the fixed cost a dispatch of this shape carries in that number
//   stream     the same grid reading the level-0 normal-equation kernel's 16.0 MB (four planar float3 maps + the packed correspondences)
//              as perfectly coalesced 16-byte loads, one burst, no arithmetic: the bandwidth ceiling for ONE memory phase
//   two-phase  the same bytes in the kernel's real dependence structure: 28 B/px addressed by the pixel, then 24 B/px gathered through an
//              index computed from the first phase (identity + a few pixels of shift): two dependent memory phases, no arithmetic
// Between timed launches another kernel rewrites 1.2 MB (as the correspondence search does between two accumulation launches).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/stream_probe.hip -o tools/probe/stream_probe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

constexpr int W = 640, H = 480, N = W * H;

__global__ void k_empty(float* sink) {
  if (sink == nullptr && threadIdx.x == 12345) sink[0] = 1.f;
}
__global__ void __launch_bounds__(512) k_stream(const float4* __restrict__ a, int n4, float* sink) {
  float s = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    const float4 v = a[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) sink[0] = s;
}
// planar maps as the tracker has them: curr v, n (6 planes) + corr; model v, n (6 planes)
__global__ void __launch_bounds__(512) k_two_phase(const float* __restrict__ curr, const unsigned* __restrict__ corr, const float* __restrict__ model, float* sink) {
  float s = 0.f;
  const int per = (N + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
  float c[6][5];
  unsigned cc[5];
  int idx[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int p = (blockIdx.x * blockDim.x + threadIdx.x) + u * gridDim.x * blockDim.x;
    const int q = p < N ? p : N - 1;
#pragma unroll
    for (int k = 0; k < 6; ++k) c[k][u] = curr[k * N + q];
    cc[u] = corr[q];
  }
  (void)per;
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int p = (blockIdx.x * blockDim.x + threadIdx.x) + u * gridDim.x * blockDim.x;
    int q = p + (int)(c[2][u] * 3.f) + (int)(cc[u] & 3u) * W;   // a few pixels / rows away, as the projective association lands
    q = q < 0 ? 0 : (q >= N ? N - 1 : q);
    idx[u] = q;
  }
#pragma unroll
  for (int u = 0; u < 5; ++u)
#pragma unroll
    for (int k = 0; k < 6; ++k) s += model[k * N + idx[u]] * c[k][u];
  if (s == 123.456f) sink[0] = s;
}
__global__ void k_between(unsigned* corr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) corr[i] = (unsigned)i * 2654435761u;
}

template <typename F>
static double timed(F launch, hipStream_t s, unsigned* corr, int reps) {
  std::vector<hipEvent_t> a(reps), b(reps);
  for (int i = 0; i < reps; ++i) { hipEventCreate(&a[i]); hipEventCreate(&b[i]); }
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(k_between, dim3((N + 255) / 256), dim3(256), 0, s, corr);
    launch(a[i], b[i]);
  }
  hipStreamSynchronize(s);
  double tot = 0;
  for (int i = 10; i < reps; ++i) { float ms = 0; hipEventElapsedTime(&ms, a[i], b[i]); tot += ms; }
  for (int i = 0; i < reps; ++i) { hipEventDestroy(a[i]); hipEventDestroy(b[i]); }
  return 1e3 * tot / (reps - 10);
}

int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  float *curr, *model, *sink;
  unsigned* corr;
  hipMalloc(&curr, sizeof(float) * 6 * N);
  hipMalloc(&model, sizeof(float) * 6 * N);
  hipMalloc(&corr, sizeof(unsigned) * N);
  hipMalloc(&sink, 64);
  std::vector<float> h(6 * N);
  for (int i = 0; i < 6 * N; ++i) h[i] = (float)((i * 7919) % 1000) / 1000.f;
  hipMemcpy(curr, h.data(), sizeof(float) * 6 * N, hipMemcpyHostToDevice);
  hipMemcpy(model, h.data(), sizeof(float) * 6 * N, hipMemcpyHostToDevice);
  // one contiguous 16.0 MB buffer for the pure stream (52 B per pixel)
  float4* flat;
  const int n4 = (int)((size_t)N * 52 / 16);
  hipMalloc(&flat, (size_t)n4 * 16);
  hipMemset(flat, 0, (size_t)n4 * 16);
  const int reps = 210;
  const dim3 grid(256), block(512);
  const double t_empty = timed([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, grid, block, 0, s, a, b, 0, sink); }, s, corr, reps);
  const double t_stream = timed([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_stream, grid, block, 0, s, a, b, 0, (const float4*)flat, n4, sink); }, s, corr, reps);
  const double t_stream8 = timed([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_stream, dim3(2048), block, 0, s, a, b, 0, (const float4*)flat, n4, sink); }, s, corr, reps);
  const double t_two = timed([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_two_phase, grid, dim3(256), 0, s, a, b, 0, (const float*)curr, (const unsigned*)corr, (const float*)model, sink); }, s, corr, reps);
  // the fixed cost of a dispatch by its shape (nothing executes): workgroups x threads
  const int shapes[][2] = {{1, 64}, {1, 256}, {64, 256}, {256, 256}, {600, 256}, {1200, 256}, {256, 512}, {512, 512}, {256, 1024}, {2048, 256}, {4800, 256}};
  printf("{\"empty_kernel_us_by_shape\": {");
  for (size_t i = 0; i < sizeof(shapes) / sizeof(shapes[0]); ++i) {
    const dim3 g(shapes[i][0]), b(shapes[i][1]);
    const double t = timed([&](hipEvent_t a, hipEvent_t bb) { hipExtLaunchKernelGGL(k_empty, g, b, 0, s, a, bb, 0, sink); }, s, corr, 110);
    printf("%s\"%dx%d\": %.3f", i ? ", " : "", shapes[i][0], shapes[i][1], t);
  }
  printf("}}\n");
  // host-paired wall time per DEPENDENT launch in a chain of 2000 (what a kernel boundary costs on this stream, no events, no profiler)
  {
    int printed = 0;
    auto chain = [&](const char* name, auto launch) {
      launch();
      hipStreamSynchronize(s);
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 2000; ++i) launch();
      hipStreamSynchronize(s);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000.0;
      printf("%s\"%s\": %.3f", printed++ ? ", " : "", name, us);
    };
    printf("{\"wall_us_per_dependent_launch\": {");
    chain("1x64 empty", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, sink); });
    chain("256x512 empty", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, s, sink); });
    chain("1200x256 writes 1.2 MB", [&] { hipLaunchKernelGGL(k_between, dim3((N + 255) / 256), dim3(256), 0, s, corr); });
    chain("256x512 streams 16 MB", [&] { hipLaunchKernelGGL(k_stream, dim3(256), dim3(512), 0, s, (const float4*)flat, n4, sink); });
    chain("256x256 two-phase 16 MB", [&] { hipLaunchKernelGGL(k_two_phase, dim3(256), dim3(256), 0, s, (const float*)curr, (const unsigned*)corr, (const float*)model, sink); });
    printf("}}\n");
  }
  const double bytes = (double)N * 52;
  printf("{\"shape\": \"256 WG x 512 thr, 640x480\", \"empty_us\": %.3f, \"stream_16MB_us\": %.3f, \"stream_16MB_2048wg_us\": %.3f, \"two_phase_16MB_us\": %.3f, "
         "\"stream_GBps\": %.1f, \"two_phase_GBps\": %.1f, \"stream_frac_of_8TBps\": %.4f, \"two_phase_frac_of_8TBps\": %.4f}\n",
         t_empty, t_stream, t_stream8, t_two, bytes / t_stream * 1e-3, bytes / t_two * 1e-3, bytes / t_stream * 1e-3 / 8000.0, bytes / t_two * 1e-3 / 8000.0);
  return 0;
}
