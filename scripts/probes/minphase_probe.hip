#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
// MODE 0: v_or_b32_dpp + v_min_f64 (k_boot's pair); 1: plain v_or_b32 + v_min_f64; 2: v_min_f64 only; 3: v_or_b32_dpp only
template <int MODE, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k(double *sink, int iters) {
  double m[8];
  for (int c = 0; c < 8; ++c) m[c] = 1e300;
  unsigned msk = sink[threadIdx.x & 15] != 0.0 ? 0xffffffffu : 0u;   // all zero at run time
  double d = 1.0 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const unsigned lo = (unsigned)__double_as_longlong(d), hi = (unsigned)(__double_as_longlong(d) >> 32);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      u2 cand[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        cand[c].x = lo;
        if (MODE == 0 || MODE == 3) asm volatile("v_or_b32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(cand[c].y) : "v"(msk), "v"(hi));
        else if (MODE == 1) asm volatile("v_or_b32 %0, %1, %2" : "=v"(cand[c].y) : "v"(msk), "v"(hi));
        else cand[c].y = hi;
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (MODE != 3) asm volatile("v_min_f64 %0, %0, %1" : "+v"(m[c]) : "v"(cand[c]));
        else asm volatile("" :: "v"(cand[c]));
      }
    }
    d += 1.0;
  }
  double t = 0;
  for (int c = 0; c < 8; ++c) t += m[c];
  if (t == 12345.678) sink[threadIdx.x] = t;
}
template <int MODE, int WPE>
void run(double *sink) {
  const int iters = 4000, blocks = 256 * WPE;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, WPE>), dim3(blocks), dim3(256), 0, 0, sink, iters);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WPE>), dim3(blocks), dim3(256), 0, 0, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double n = (double)iters * 32 * (MODE == 0 || MODE == 1 ? 2 : 1);
  printf("{\"mode\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"instr_per_simd_per_us\": %.1f}\n", MODE, WPE, best, n * WPE / (best * 1e3));
}
int main() {
  double *sink; hipMalloc(&sink, 1 << 20); hipMemset(sink, 0, 1 << 20);
  run<2, 2>(sink); run<0, 2>(sink); run<1, 2>(sink); run<2, 2>(sink); run<3, 2>(sink); run<0, 1>(sink); run<0, 4>(sink);
  return 0;
}
