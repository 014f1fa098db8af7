// Probe: v_mfma_f64_4x4x4_4b_f64 on gfx950 -- which (A lane, B lane) pairs feed which result lane, is the
// accumulation the k-ascending FMA chain, and the issue rate compared with v_mfma_f64_16x16x4_f64.
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

// table[s][l] = result in lane l when A is one-hot at lane s and B[lane] = lane + 1
__global__ void k_layout(double *table) {
  const int l = threadIdx.x;
  for (int s = 0; s < 64; ++s) {
    const double a = (l == s) ? 1.0 : 0.0;
    const double b = (double)(l + 1);
    table[s * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  }
}

// one instruction on arbitrary operands (+ accumulator input)
__global__ void k_one(const double *a, const double *b, const double *c, double *d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}

__global__ void k_rate4(double *sink, int iters) {
  const int l = threadIdx.x & 63;
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
  const double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 123.456) sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_rate16(double *sink, int iters) {
  const int l = threadIdx.x & 63;
  double4v acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (double4v){0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  double *d_table;
  CK(hipMalloc(&d_table, 64 * 64 * sizeof(double)));
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d_table);
  std::vector<double> table(64 * 64);
  CK(hipMemcpy(table.data(), d_table, table.size() * sizeof(double), hipMemcpyDeviceToHost));
  // pairs[l] = list of (A lane, B lane) feeding result lane l
  std::vector<std::vector<std::pair<int, int>>> pairs(64);
  for (int s = 0; s < 64; ++s)
    for (int l = 0; l < 64; ++l)
      if (table[s * 64 + l] != 0.0) pairs[l].push_back({s, (int)table[s * 64 + l] - 1});
  printf("result lane : (A lane, B lane) contributions\n");
  for (int l = 0; l < 64; ++l) {
    if (!(l < 6 || l == 16 || l == 17 || l == 63)) continue;
    printf("  %2d :", l);
    for (auto &p : pairs[l]) printf(" (%d,%d)", p.first, p.second);
    printf("\n");
  }
  // hypothesis: block = l / 16; A lane of (block, row i, k) = 16 block + 4 k + i; B lane of (block, k, col j) =
  // 16 block + 4 k + j; result lane of (block, i, j) = 16 block + 4 ? ...  print the fitted rule
  int ok_rule = 1;
  for (int l = 0; l < 64; ++l) {
    if (pairs[l].size() != 4) { ok_rule = 0; continue; }
  }
  printf("every result lane has 4 contributions: %s\n", ok_rule ? "yes" : "NO");

  // accumulation order: random operands, compare with fma chains over the discovered contributions in the
  // order of ascending A lane (= ascending k if the rule above holds)
  std::vector<double> a(64), b(64), c(64), d(64);
  srand(7);
  int same_fma = 0, same_rev = 0, same_unfused = 0;
  double *da, *db, *dc, *dd;
  CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dc, 512)); CK(hipMalloc(&dd, 512));
  const int trials = 200;
  for (int t = 0; t < trials; ++t) {
    for (int l = 0; l < 64; ++l) {
      a[l] = (rand() / (double)RAND_MAX - 0.5) * pow(2.0, rand() % 20 - 10);
      b[l] = (rand() / (double)RAND_MAX - 0.5) * pow(2.0, rand() % 20 - 10);
      c[l] = (rand() / (double)RAND_MAX - 0.5);
    }
    CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    CK(hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost));
    for (int l = 0; l < 64; ++l) {
      double f = c[l], r = c[l], u = c[l];
      for (size_t k = 0; k < pairs[l].size(); ++k) f = fma(a[pairs[l][k].first], b[pairs[l][k].second], f);
      for (size_t k = pairs[l].size(); k-- > 0;) r = fma(a[pairs[l][k].first], b[pairs[l][k].second], r);
      for (size_t k = 0; k < pairs[l].size(); ++k) { volatile double p = a[pairs[l][k].first] * b[pairs[l][k].second]; u = u + p; }
      same_fma += (f == d[l]);
      same_rev += (r == d[l]);
      same_unfused += (u == d[l]);
    }
  }
  printf("results equal to: ascending fma chain %d / %d, descending chain %d, unfused ascending %d\n", same_fma, trials * 64,
         same_rev, same_unfused);

  // issue rate
  double *sink;
  CK(hipMalloc(&sink, 1024 * 256 * sizeof(double)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, 0));
      if (which == 0) hipLaunchKernelGGL(k_rate4, dim3(1024), dim3(256), 0, 0, sink, iters);
      else hipLaunchKernelGGL(k_rate16, dim3(1024), dim3(256), 0, 0, sink, iters);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double ninstr = 1024.0 * 4 * iters * (which == 0 ? 16 : 8);
      const double flop = ninstr * (which == 0 ? 512.0 : 2048.0);
      if (rep == 1)
        printf("%s: %.3f ms, %.1f TFLOP/s, %.1f clocks per instruction per SIMD at 2.4 GHz\n",
               which == 0 ? "4x4x4_4b " : "16x16x4  ", ms, flop / (ms * 1e-3) / 1e12,
               (ms * 1e-3 * 2.4e9) / (ninstr / 1024.0));
    }
  }
  return 0;
}
