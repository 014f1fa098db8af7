// Probe: what does v_mfma_f32_32x32x16_f16 sustain on this part, alone and with vector instructions pinned between
// the matrix instructions?  One wave = 4 accumulator chains x 4 k-steps per "tile" (k_sweep's shape), operands
// resident in registers; NV independent v_min3_i32 per matrix instruction (sched_group_barrier), reading either
// private registers (IND = 1: pure issue-port load) or the accumulators of the previous half (IND = 0: k_sweep's
// real dependency pattern).  1, 2 or 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o bin/mfma16_issue_probe mfma16_issue_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

__device__ __forceinline__ int min3i(int a, int b, int c) {
  const int m = a < b ? a : b;
  return m < c ? m : c;
}

template <int NV, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_issue(float *sink, int iters) {
  const int lane = threadIdx.x & 63;
  half8 bq[4][4], af[4];
  for (int g = 0; g < 4; ++g)
    for (int s = 0; s < 4; ++s)
      for (int j = 0; j < 8; ++j) bq[g][s][j] = (_Float16)(0.001f * ((lane * 7 + g * 3 + s + j) % 97) - 0.04f);
  for (int s = 0; s < 4; ++s)
    for (int j = 0; j < 8; ++j) af[s][j] = (_Float16)(0.002f * ((lane * 5 + s + j) % 89) - 0.08f);
  int v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane * (i + 3) + 1000;
  float tot = 0.f;
  for (int it = 0; it < iters; ++it) {
    float16v acc[4];
    const float16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], bq[g][s], s == 0 ? z : acc[g], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16 * NV; ++i) v[i & 7] = min3i(v[(i + 1) & 7], v[(i + 3) & 7], v[i & 7] + 0);
    if (NV > 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    tot += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    af[0][0] = (_Float16)(float)(it & 1);
  }
  int vs = 0;
  for (int i = 0; i < 8; ++i) vs += v[i];
  if (tot == 12345.f || vs == 77) sink[blockIdx.x * 256 + threadIdx.x] = tot;
}

typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// K = 56 as 3 x 32x32x16 + 1 x 32x32x8 per chain (the padded columns 56..63 of the production operand are zeros)
template <int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_mixed(float *sink, int iters) {
  const int lane = threadIdx.x & 63;
  half8 bq[4][3], af[3];
  half4 bq8[4], af8;
  for (int g = 0; g < 4; ++g) {
    for (int s = 0; s < 3; ++s)
      for (int j = 0; j < 8; ++j) bq[g][s][j] = (_Float16)(0.001f * ((lane * 7 + g * 3 + s + j) % 97) - 0.04f);
    for (int j = 0; j < 4; ++j) bq8[g][j] = (_Float16)(0.001f * ((lane * 3 + g + j) % 91) - 0.04f);
  }
  for (int s = 0; s < 3; ++s)
    for (int j = 0; j < 8; ++j) af[s][j] = (_Float16)(0.002f * ((lane * 5 + s + j) % 89) - 0.08f);
  for (int j = 0; j < 4; ++j) af8[j] = (_Float16)(0.002f * ((lane + j) % 83) - 0.08f);
  float tot = 0.f;
  for (int it = 0; it < iters; ++it) {
    float16v acc[4];
    const float16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], bq[g][s], s == 0 ? z : acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x8f16(af8, bq8[g], acc[g], 0, 0, 0);
    tot += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    af[0][0] = (_Float16)(float)(it & 1);
  }
  if (tot == 12345.f) sink[blockIdx.x * 256 + threadIdx.x] = tot;
}

template <int WPE>
static void run_mixed(float *sink, int iters) {
  const int blocks = 256 * WPE;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_mixed<WPE>), dim3(blocks), dim3(256), 0, 0, sink, iters / 8);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mixed<WPE>), dim3(blocks), dim3(256), 0, 0, sink, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  // compare TIME per tile with the 4 x (x16) kernel: tiles per ms
  printf("{\"variant\": \"3 x 32x32x16 + 1 x 32x32x8 per chain (K = 56)\", \"waves_per_simd\": %d, \"ms\": %.4f, \"tiles_per_us_per_wave\": %.4f}\n", WPE,
         best, (double)iters / (best * 1e3));
  fflush(stdout);
}

template <int NV, int WPE>
static void run(const char *name, float *sink, int iters) {
  const int blocks = 256 * WPE;   // 4 waves per workgroup: WPE workgroups per CU
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_issue<NV, WPE>), dim3(blocks), dim3(256), 0, 0, sink, iters / 8);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_issue<NV, WPE>), dim3(blocks), dim3(256), 0, 0, sink, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double flops = (double)blocks * 4 * iters * 16 * 32768.0;
  printf("{\"variant\": \"%s\", \"vector_per_matrix\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"TFLOPs\": %.0f, \"tiles_per_us_per_wave\": %.4f}\n", name, NV, WPE,
         best, flops / (best * 1e-3) / 1e12, (double)iters / (best * 1e3));
  fflush(stdout);
}

int main() {
  float *sink;
  CK(hipMalloc(&sink, 256 * 4 * 256 * sizeof(float)));
  const int iters = 4000;
  run_mixed<2>(sink, iters);
  run<0, 1>("matrix only", sink, iters);
  run<0, 2>("matrix only", sink, iters);
  run<0, 4>("matrix only", sink, iters);
  run<1, 2>("pinned independent v_min3", sink, iters);
  run<2, 2>("pinned independent v_min3", sink, iters);
  run<3, 2>("pinned independent v_min3", sink, iters);
  run<4, 2>("pinned independent v_min3", sink, iters);
  run<5, 2>("pinned independent v_min3", sink, iters);
  run<6, 2>("pinned independent v_min3", sink, iters);
  run<7, 2>("pinned independent v_min3", sink, iters);
  run<3, 1>("pinned independent v_min3", sink, iters);
  run<5, 1>("pinned independent v_min3", sink, iters);
  run<3, 4>("pinned independent v_min3", sink, iters / 2);
  return 0;
}
