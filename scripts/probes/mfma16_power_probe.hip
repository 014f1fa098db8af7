// Probe: does the sustained rate of v_mfma_f32_32x32x16_f16 depend on the DATA?  Same loop as k_filter's sweep
// (4 accumulator chains x 4 k-steps per tile, 2 waves per SIMD, min3 epilogue), operands either smooth functions
// of the lane index (what mfma16_probe.hip used) or pseudo-random binary16 values in [-1, 1]; A fragments either
// resident in registers or streamed from a 500 KB L2-resident buffer as in the real kernel.
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

template <int LOADS>
__global__ __launch_bounds__(256) void k_rate(const half8 *qF, const half8 *refF, int ntiles, float *sink, int sweeps) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  half8 bq[4][4], af[4];
  for (int g = 0; g < 4; ++g)
    for (int s = 0; s < 4; ++s) bq[g][s] = qF[((wave * 4 + g) * 4 + s) * 64 + lane];
  for (int s = 0; s < 4; ++s) af[s] = refF[s * 64 + lane];
  unsigned long long hits = 0;
  const float thr = -1.0e30f;
  for (int sw = 0; sw < sweeps; ++sw)
    for (int t = 0; t < ntiles; ++t) {
      half8 an[4];
      const int tn = t + 1 < ntiles ? t + 1 : 0;
      if (LOADS) {
#pragma unroll
        for (int s = 0; s < 4; ++s) an[s] = refF[((size_t)tn * 4 + s) * 64 + lane];
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) an[s] = af[s];
        an[0][0] = (_Float16)(float)(t & 1);
      }
      float16v acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bq[g][0], (float16v){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int s = 1; s < 4; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], bq[g][s], acc[g], 0, 0, 0);
      unsigned long long need = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float16v &c = acc[g];
        const int m0 = min3i(__float_as_int(c[0]), __float_as_int(c[1]), __float_as_int(c[2]));
        const int m1 = min3i(__float_as_int(c[3]), __float_as_int(c[4]), __float_as_int(c[5]));
        const int m2 = min3i(__float_as_int(c[6]), __float_as_int(c[7]), __float_as_int(c[8]));
        const int m3 = min3i(__float_as_int(c[9]), __float_as_int(c[10]), __float_as_int(c[11]));
        const int m4 = min3i(__float_as_int(c[12]), __float_as_int(c[13]), __float_as_int(c[14]));
        const float vmin = __int_as_float(min3i(min3i(m0, m1, m2), min3i(m3, m4, __float_as_int(c[15])), m0));
        need |= __ballot(vmin <= thr);
      }
      if (need) hits += 1;
#pragma unroll
      for (int s = 0; s < 4; ++s) af[s] = an[s];
    }
  if (hits == 77ull) sink[blockIdx.x * 256 + threadIdx.x] = 1.0f;
}

int main() {
  const int blocks = 512, ntiles = 125, sweeps = 4;
  const size_t nq = (size_t)blocks * 4 * 4 * 4 * 64 * 8, nr = (size_t)ntiles * 4 * 64 * 8;
  std::vector<_Float16> hq(nq), hr(nr);
  _Float16 *dq, *dr;
  float *sink;
  CK(hipMalloc(&dq, nq * 2));
  CK(hipMalloc(&dr, nr * 2));
  CK(hipMalloc(&sink, blocks * 256 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int data = 0; data < 3; ++data) {
    unsigned st = 12345u;
    auto rnd = [&]() {
      st = st * 1664525u + 1013904223u;
      return (float)(st >> 8) * (1.0f / 8388608.0f) - 1.0f;
    };
    // 0: smooth small positive values; 1: uniform in [-1, 1]; 2: like the filter's operands (|x| ~ 0.1, sign random, -2 folded in)
    for (size_t i = 0; i < nq; ++i) hq[i] = (_Float16)(data == 0 ? 0.001f * (float)(i % 71) : (data == 1 ? rnd() : -0.25f * rnd()));
    for (size_t i = 0; i < nr; ++i) hr[i] = (_Float16)(data == 0 ? 0.002f * (float)(i % 67) : (data == 1 ? rnd() : 0.125f * rnd()));
    CK(hipMemcpy(dq, hq.data(), nq * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dr, hr.data(), nr * 2, hipMemcpyHostToDevice));
    for (int loads = 0; loads < 2; ++loads) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (loads)
          hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, (const half8 *)dq, (const half8 *)dr, ntiles, sink, sweeps);
        else
          hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, (const half8 *)dq, (const half8 *)dr, ntiles, sink, sweeps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double flops = (double)blocks * 4 * sweeps * ntiles * 16.0 * 32768.0;
      printf("{\"data\": %d, \"a_from_l2\": %d, \"ms\": %.4f, \"TFLOPs\": %.0f}\n", data, loads, ms, flops / ms * 1e-9);
    }
  }
  return 0;
}
