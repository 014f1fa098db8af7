// Probe: v_mfma_f64_16x16x4_f64 on gfx950 -- operand/result layout, accumulation order (is it the
// k-ascending FMA chain?), throughput alone and next to independent FP64 VALU work.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double double4v __attribute__((ext_vector_type(4)));

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);        \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// C[16][16] = A[16][K] * B[K][16], one wave, K multiple of 4
__global__ void k_gemm(const double *A, const double *B, double *C, int K) {
  const int l = threadIdx.x;
  double4v acc = {0.0, 0.0, 0.0, 0.0};
  for (int ks = 0; ks < K / 4; ++ks) {
    const double a = A[(l & 15) * K + 4 * ks + (l >> 4)];
    const double b = B[(4 * ks + (l >> 4)) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

__global__ void k_rate(double *sink, int iters, int valu) {
  const int l = threadIdx.x & 63;
  double4v acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (double4v){0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  double x0 = l, x1 = l + 1, x2 = l + 2, x3 = l + 3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if (valu) {
        x0 = __builtin_fma(x0, 1.0000001, 1e-9);
        x1 = __builtin_fma(x1, 1.0000001, 1e-9);
        x2 = __builtin_fma(x2, 1.0000001, 1e-9);
        x3 = __builtin_fma(x3, 1.0000001, 1e-9);
        if (valu > 1) {
          x0 = __builtin_fma(x0, 0.9999999, 1e-9);
          x1 = __builtin_fma(x1, 0.9999999, 1e-9);
          x2 = __builtin_fma(x2, 0.9999999, 1e-9);
          x3 = __builtin_fma(x3, 0.9999999, 1e-9);
        }
      }
    }
  }
  double s = x0 + x1 + x2 + x3;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  const int K = 52;
  std::vector<double> A(16 * K), B(K * 16), C(256), R(256), Rrev(256);
  srand(1);
  for (auto &v : A) v = (rand() / (double)RAND_MAX - 0.5) * 3;
  for (auto &v : B) v = (rand() / (double)RAND_MAX - 0.5) * 3;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double acc = 0.0, rev = 0.0;
      for (int k = 0; k < K; ++k) acc = fma(A[i * K + k], B[k * 16 + j], acc);
      for (int k = K - 1; k >= 0; --k) rev = fma(A[i * K + k], B[k * 16 + j], rev);
      R[i * 16 + j] = acc;
      Rrev[i * 16 + j] = rev;
    }
  double *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 8));
  CK(hipMalloc(&dB, B.size() * 8));
  CK(hipMalloc(&dC, 256 * 8));
  CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_gemm, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
  CK(hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost));
  int same = 0, same_rev = 0;
  double maxrel = 0;
  for (int i = 0; i < 256; ++i) {
    same += C[i] == R[i];
    same_rev += C[i] == Rrev[i];
    maxrel = fmax(maxrel, fabs(C[i] - R[i]) / fabs(R[i]));
  }
  printf("layout+order: %d/256 equal to the k-ascending FMA chain, %d/256 equal to the descending chain, max rel diff %.3g\n",
         same, same_rev, maxrel);

  double *sink;
  CK(hipMalloc(&sink, 1024 * 256 * 8 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int wpb = 1; wpb <= 2; ++wpb)
    for (int valu = 0; valu <= 2; ++valu) {
      const int blocks = 1024 * 2, iters = 2000;
      hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256 * wpb), 0, 0, sink, 10, valu);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256 * wpb), 0, 0, sink, iters, valu);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double waves = (double)blocks * 4 * wpb;
      const double mfma_flops = waves * iters * 8.0 * 2048.0;
      const double valu_flops = waves * iters * 8.0 * 4.0 * valu * 64.0 * 2.0;
      printf("threads/block %d valu %d: %.3f ms  MFMA %.1f TFLOP/s  VALU-FMA %.1f TFLOP/s\n", 256 * wpb, valu, ms,
             mfma_flops / ms * 1e-9, valu_flops / ms * 1e-9);
    }
  return 0;
}
