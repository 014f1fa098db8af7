// Probe: ONE wave per SIMD holding 8 query groups (256 queries; accumulators in AGPRs) against the production shape
// (two waves per SIMD, 4 groups each).  The wide wave reads a live-point tile once per 32 matrix instructions instead
// of 16 and hides its own epilogue behind its own matrix instructions: the groups are processed in two halves, the
// min3 epilogue of one half is interleaved (sched_group_barrier) with the matrix instructions of the other.
// A fragments streamed from a 500 KB L2-resident buffer, pseudo-random binary16 operands.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__);      \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

__device__ __forceinline__ int min3i(int a, int b, int c) {
  const int m = a < b ? a : b;
  return m < c ? m : c;
}

__device__ __forceinline__ float lane_min(const float16v &c) {
  const int m0 = min3i(__float_as_int(c[0]), __float_as_int(c[1]), __float_as_int(c[2]));
  const int m1 = min3i(__float_as_int(c[3]), __float_as_int(c[4]), __float_as_int(c[5]));
  const int m2 = min3i(__float_as_int(c[6]), __float_as_int(c[7]), __float_as_int(c[8]));
  const int m3 = min3i(__float_as_int(c[9]), __float_as_int(c[10]), __float_as_int(c[11]));
  const int m4 = min3i(__float_as_int(c[12]), __float_as_int(c[13]), __float_as_int(c[14]));
  return __int_as_float(min3i(min3i(m0, m1, m2), min3i(m3, m4, __float_as_int(c[15])), m0));
}

#define ZERO16 ((float16v){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f})

// NG query groups per wave, processed in two halves of NG / 2
template <int NG, bool INTERLEAVE>
__global__ __launch_bounds__(256, 1) void k_wide(const half8 *qF, const half8 *refF, int ntiles, float *sink, int sweeps) {
  constexpr int H = NG / 2;
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  half8 bq[NG][4], af[4], an[4];
  for (int g = 0; g < NG; ++g)
    for (int s = 0; s < 4; ++s) bq[g][s] = qF[(((wave * NG + g) % 2048) * 4 + s) * 64 + lane];
  for (int s = 0; s < 4; ++s) af[s] = refF[s * 64 + lane];
  float16v acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) acc[g] = ZERO16;
  float thr[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) thr[g] = -1.0e30f;
  unsigned long long hits = 0;
  for (int sw = 0; sw < sweeps; ++sw)
    for (int t = 0; t < ntiles; ++t) {
      const int tn = t + 1 < ntiles ? t + 1 : 0;
#pragma unroll
      for (int s = 0; s < 4; ++s) an[s] = refF[((size_t)tn * 4 + s) * 64 + lane];
      __builtin_amdgcn_sched_barrier(0);
      // first half of the groups on tile t; meanwhile the epilogue of the second half on tile t - 1
      unsigned long long need = 0;
#pragma unroll
      for (int g = H; g < NG; ++g) need |= __ballot(lane_min(acc[g]) <= thr[g]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int g = 0; g < H; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], bq[g][s], s == 0 ? ZERO16 : acc[g], 0, 0, 0);
      if (INTERLEAVE) {
#pragma unroll
        for (int i = 0; i < 4 * H; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one matrix instruction
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // a few vector instructions of the epilogue
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // second half on tile t; meanwhile the epilogue of the first half on tile t
#pragma unroll
      for (int g = 0; g < H; ++g) need |= __ballot(lane_min(acc[g]) <= thr[g]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int g = H; g < NG; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], bq[g][s], s == 0 ? ZERO16 : acc[g], 0, 0, 0);
      if (INTERLEAVE) {
#pragma unroll
        for (int i = 0; i < 4 * H; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (need) hits += 1;
#pragma unroll
      for (int s = 0; s < 4; ++s) af[s] = an[s];
    }
  if (hits == 77ull) sink[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[NG - 1][3];
}

int main() {
  const int ntiles = 125, sweeps = 4;
  const size_t nq = (size_t)2048 * 4 * 64 * 8, nr = (size_t)ntiles * 4 * 64 * 8;
  std::vector<_Float16> hq(nq), hr(nr);
  unsigned st = 12345u;
  auto rnd = [&]() {
    st = st * 1664525u + 1013904223u;
    return (float)(st >> 8) * (1.0f / 8388608.0f) - 1.0f;
  };
  for (size_t i = 0; i < nq; ++i) hq[i] = (_Float16)(-0.25f * rnd());
  for (size_t i = 0; i < nr; ++i) hr[i] = (_Float16)(0.125f * rnd());
  _Float16 *dq, *dr;
  float *sink;
  CK(hipMalloc(&dq, nq * 2));
  CK(hipMalloc(&dr, nr * 2));
  CK(hipMalloc(&sink, 1024 * 256 * 4));
  CK(hipMemcpy(dq, hq.data(), nq * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dr, hr.data(), nr * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char *name, void (*launch)(const half8 *, const half8 *, int, float *, int, int), int blocks, int ng) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      launch((const half8 *)dq, (const half8 *)dr, ntiles, sink, sweeps, blocks);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double flops = (double)blocks * 4 * sweeps * ntiles * ng * 4.0 * 32768.0;
    printf("{\"variant\": \"%s\", \"workgroups\": %d, \"ms\": %.4f, \"TFLOPs\": %.0f}\n", name, blocks, ms, flops / ms * 1e-9);
  };
  run("8 groups, 1 wave/SIMD, interleaved", [](const half8 *q, const half8 *r, int n, float *s, int sw, int b) {
    hipLaunchKernelGGL((k_wide<8, true>), dim3(b), dim3(256), 0, 0, q, r, n, s, sw); }, 256, 8);
  run("8 groups, 1 wave/SIMD, compiler order", [](const half8 *q, const half8 *r, int n, float *s, int sw, int b) {
    hipLaunchKernelGGL((k_wide<8, false>), dim3(b), dim3(256), 0, 0, q, r, n, s, sw); }, 256, 8);
  run("6 groups, 1 wave/SIMD, interleaved", [](const half8 *q, const half8 *r, int n, float *s, int sw, int b) {
    hipLaunchKernelGGL((k_wide<6, true>), dim3(b), dim3(256), 0, 0, q, r, n, s, sw); }, 256, 6);
  run("4 groups, 2 waves/SIMD, halves interleaved", [](const half8 *q, const half8 *r, int n, float *s, int sw, int b) {
    hipLaunchKernelGGL((k_wide<4, true>), dim3(b), dim3(256), 0, 0, q, r, n, s, sw); }, 512, 4);
  return 0;
}
