// Probe: how often can ONE wave issue binary64 vector instructions on gfx950, and how many waves per SIMD does the
// FP64 vector pipe need to stay busy?  CH independent accumulator chains per lane (CH = 1: every instruction depends on
// the previous one), WPE waves per SIMD, v_fma_f64 or the v_add_f64 / v_mul_f64 mix of the reference's distance loop.
//   hipcc --offload-arch=gfx950 -O3 -o bin/fp64_issue_probe fp64_issue_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

template <int CH, int WPE, int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_chain(double *sink, int iters, double x, double y) {
  double acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = 1.0 + 1e-3 * (threadIdx.x + c);
  double bc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) bc[c] = sink[(threadIdx.x & 15) + c];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (KIND == 0) {
          acc[c] = __builtin_fma(acc[c], x, y);
        } else if (KIND == 4) {   // v_min_f64 alone
          asm volatile("v_min_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(bc[c]));
        } else if (KIND == 5) {   // v_max_f64 through the compiler's builtin
          acc[c] = __builtin_fmax(acc[c], bc[c]);
          asm volatile("" : "+v"(acc[c]));
        } else if (KIND == 7) {   // the same with every second candidate a quiet NaN (an unselected live point)
          asm volatile("" : "+v"(bc[c]));
          unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)__double2hiint(bc[c]), 0x150 + 7, 0xf, 0xf, true) | ((u & 1) ? 0xffffffffu : 0u);
          double cand = __hiloint2double((int)hi, __double2loint(bc[c]));
          asm volatile("v_min_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(cand));
        } else if (KIND == 6) {   // k_boot's masked minimum: v_or_b32 with a DPP operand on the high word, then v_min_f64
          asm volatile("" : "+v"(bc[c]));   // opaque: the DPP operand is not loop invariant
          unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)__double2hiint(bc[c]), 0x150 + 7, 0xf, 0xf, true) | (unsigned)u;
          double cand = __hiloint2double((int)hi, __double2loint(bc[c]));
          asm volatile("v_min_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(cand));
        } else if (KIND == 2) {   // DPP row broadcast
          asm volatile("" : "+v"(bc[c]));   // opaque: the broadcast is not loop invariant of a binary64 value (v_mov_b64_dpp row_newbcast) + one add
          acc[c] = acc[c] + __builtin_amdgcn_update_dpp(0.0, bc[c], 0x150 + 3, 0xf, 0xf, true);
        } else if (KIND == 3) {   // k_boot's four instructions per coordinate: broadcast, subtract, multiply, accumulate
          asm volatile("" : "+v"(bc[c]));
          const double dd = __builtin_amdgcn_update_dpp(0.0, bc[c], 0x150 + 5, 0xf, 0xf, true) - y;
          acc[c] = acc[c] + dd * dd;
        } else {   // sub, mul, add: the distance loop's three instructions per coordinate (the add carries the chain)
          const double dd = acc[c] - y;
          asm volatile("" : "+v"(acc[c]));
          acc[c] = acc[c] + dd * x;   // contracted or not, the compiler decides; KIND 1 is compiled with -ffp-contract=off
        }
      }
    }
  }
  double t = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) t += acc[c];
  if (t == 12345.678) sink[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int CH, int WPE, int KIND>
static void run(double *sink, int iters) {
  const int blocks = 256 * WPE;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_chain<CH, WPE, KIND>), dim3(blocks), dim3(256), 0, 0, sink, iters, 1.0000001, 1e-9);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chain<CH, WPE, KIND>), dim3(blocks), dim3(256), 0, 0, sink, iters, 1.0000001, 1e-9);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double instr_per_wave = (double)iters * 16 * CH * (KIND == 0 || KIND == 4 || KIND == 5 ? 1 : (KIND == 2 || KIND == 6 || KIND == 7 ? 2 : (KIND == 3 ? 4 : 3)));
  const double waves = (double)blocks * 4;
  const double per_simd_per_us = instr_per_wave * WPE / (best * 1e3);   // vector instructions per SIMD per microsecond
  printf("{\"kind\": \"%s\", \"chains_per_lane\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"instr_per_simd_per_us\": %.1f, \"ns_per_instr_per_wave\": %.3f, \"total_instr\": %.3g}\n",
         KIND == 0 ? "v_fma_f64" : (KIND == 2 ? "v_mov_b64_dpp + add" : (KIND == 3 ? "v_mov_b64_dpp/sub/mul/add" : (KIND == 4 ? "v_min_f64" : (KIND == 5 ? "fmax builtin" : (KIND == 6 ? "v_or_b32_dpp + v_min_f64" : KIND == 7 ? "v_or_b32_dpp + v_min_f64, half the candidates NaN" : "sub/mul/add"))))), CH, WPE, best, per_simd_per_us, best * 1e6 / instr_per_wave, instr_per_wave * waves);
  fflush(stdout);
}

int main() {
  double *sink;
  CK(hipMalloc(&sink, 256 * 8 * 256 * sizeof(double)));
  CK(hipMemset(sink, 0, 256 * 8 * 256 * sizeof(double)));
  const int iters = 2000;
  run<8, 1, 0>(sink, iters);   // clocks up
  run<1, 1, 0>(sink, iters);
  run<2, 1, 0>(sink, iters);
  run<4, 1, 0>(sink, iters);
  run<8, 1, 0>(sink, iters);
  run<1, 2, 0>(sink, iters);
  run<4, 2, 0>(sink, iters);
  run<8, 2, 0>(sink, iters);
  run<1, 4, 0>(sink, iters);
  run<4, 4, 0>(sink, iters);
  run<8, 4, 0>(sink, iters);
  run<1, 8, 0>(sink, iters);
  run<4, 8, 0>(sink, iters);
  run<1, 1, 1>(sink, iters);
  run<1, 2, 1>(sink, iters);
  run<1, 4, 1>(sink, iters);
  run<4, 2, 1>(sink, iters);
  run<4, 2, 2>(sink, iters);
  run<8, 4, 2>(sink, iters);
  run<1, 2, 3>(sink, iters);
  run<2, 2, 3>(sink, iters);
  run<4, 2, 3>(sink, iters);
  run<4, 4, 3>(sink, iters);
  run<1, 2, 4>(sink, iters);
  run<4, 2, 4>(sink, iters);
  run<8, 2, 4>(sink, iters);
  run<8, 2, 5>(sink, iters);
  run<4, 2, 6>(sink, iters);
  run<8, 2, 6>(sink, iters);
  run<8, 2, 7>(sink, iters);
  run<4, 2, 7>(sink, iters);
  return 0;
}
