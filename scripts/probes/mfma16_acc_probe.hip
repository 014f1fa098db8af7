// Probe: how does v_mfma_f32_32x32x16_f16 accumulate?  D = C + sum_k A[i][k] B[k][j] with exact binary16 products;
// the question is how many roundings the 16-term sum and the addition of C cost, and in which order.
//   test 1 (position sweep): C = -2^20, one product = +2^20 at position pos, the other 15 products = 2^-8.
//           exact result 15 * 2^-8; a binary32 sum that meets the big term late loses the small ones.
//   test 2 (random + cancellation): error of one instruction and of chains of 12 instructions against binary64,
//           in units of u * S  (u = 2^-24, S = |C| + sum |a b|) and in units of u * max|partial sum|.
// Fragment layout (32x32x16 f16): lane l holds A[i = l & 31][k = 8 (l >> 5) + j], B[k = 8 (l >> 5) + j][col = l & 31];
// D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31].
#include <hip/hip_runtime.h>

#include <cmath>
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

// nchain instructions: A [nchain][32][16], B [nchain][16][32] (row-major), C [32][32] -> D [32][32]
__global__ void k_mfma(const _Float16 *A, const _Float16 *B, const float *C, float *D, int nchain) {
  const int l = threadIdx.x, i = l & 31, kb = 8 * (l >> 5);
  float16v acc;
  for (int r = 0; r < 16; ++r) acc[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + i];
  for (int c = 0; c < nchain; ++c) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = A[(c * 32 + i) * 16 + kb + j];
      b[j] = B[(c * 16 + kb + j) * 32 + i];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + i] = acc[r];
}

static std::vector<float> run(const std::vector<_Float16> &A, const std::vector<_Float16> &B, const std::vector<float> &C, int nchain) {
  _Float16 *dA, *dB;
  float *dC, *dD;
  CK(hipMalloc(&dA, A.size() * 2));
  CK(hipMalloc(&dB, B.size() * 2));
  CK(hipMalloc(&dC, 4096));
  CK(hipMalloc(&dD, 4096));
  CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, nchain);
  std::vector<float> D(1024);
  CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
  CK(hipFree(dA));
  CK(hipFree(dB));
  CK(hipFree(dC));
  CK(hipFree(dD));
  return D;
}

static double urand() { return rand() / (RAND_MAX + 1.0); }

int main() {
  // ---- test 1
  printf("{\"position_sweep\": [");
  for (int pos = 0; pos < 16; ++pos) {
    std::vector<_Float16> A(32 * 16), B(16 * 32);
    std::vector<float> C(1024, 0.0f);
    for (int i = 0; i < 32; ++i)
      for (int k = 0; k < 16; ++k) {
        A[i * 16 + k] = (_Float16)(k == pos ? 1024.0f : 0.0625f);
        B[k * 32 + i] = (_Float16)(k == pos ? 1024.0f : 0.0625f);
      }
    for (auto &c : C) c = -1048576.0f;
    std::vector<float> D = run(A, B, C, 1);
    printf("%s%.10g", pos ? ", " : "", (double)D[0]);
  }
  printf("], \"position_sweep_exact\": %.10g,\n", 15.0 / 256.0);
  // ---- test 2
  srand(12345);
  for (int nchain : {1, 12}) {
    double worst_s = 0.0, worst_p = 0.0, mean_s = 0.0;
    long count = 0;
    for (int trial = 0; trial < 200; ++trial) {
      std::vector<_Float16> A((size_t)nchain * 32 * 16), B((size_t)nchain * 16 * 32);
      std::vector<float> C(1024);
      const int mode = trial % 4;   // 0 random signs, 1 all positive, 2 C cancels the sum, 3 wide dynamic range
      for (auto &v : A) v = (_Float16)(float)((mode == 1 ? urand() : 2 * urand() - 1) * (mode == 3 ? std::ldexp(1.0, (rand() % 16) - 8) : 1.0));
      for (auto &v : B) v = (_Float16)(float)((mode == 1 ? urand() : 2 * urand() - 1) * (mode == 3 ? std::ldexp(1.0, (rand() % 16) - 8) : 1.0));
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double s = 0.0;
          for (int c = 0; c < nchain; ++c)
            for (int k = 0; k < 16; ++k) s += (double)(float)A[(c * 32 + i) * 16 + k] * (double)(float)B[(c * 16 + k) * 32 + j];
          C[i * 32 + j] = mode == 2 ? (float)(-s * (1.0 + 1e-3 * (urand() - 0.5))) : (float)(2 * urand() - 1);
        }
      std::vector<float> D = run(A, B, C, nchain);
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double exact = C[i * 32 + j], S = std::fabs((double)C[i * 32 + j]), part = exact, pmax = std::fabs(exact);
          for (int c = 0; c < nchain; ++c)
            for (int k = 0; k < 16; ++k) {
              const double p = (double)(float)A[(c * 32 + i) * 16 + k] * (double)(float)B[(c * 16 + k) * 32 + j];
              exact += p;
              S += std::fabs(p);
              part += p;
              pmax = std::fmax(pmax, std::fabs(part));
            }
          const double err = std::fabs((double)D[i * 32 + j] - exact), u = std::ldexp(1.0, -24);
          worst_s = std::fmax(worst_s, err / (u * S));
          worst_p = std::fmax(worst_p, err / (u * std::fmax(pmax, 1e-300)));
          mean_s += err / (u * S);
          ++count;
        }
    }
    printf(" \"chain_%d\": {\"max_err_over_uS\": %.4f, \"max_err_over_u_maxpartial\": %.4f, \"mean_err_over_uS\": %.5f, \"outputs\": %ld}%s\n",
           nchain, worst_s, worst_p, mean_s / count, count, nchain == 1 ? "," : "}");
  }
  return 0;
}
