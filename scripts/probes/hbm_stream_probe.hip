// Probe (round 5): what do hand-written streaming kernels reach on THIS pool's MI355X boxes?  The builder's
// ceiling argument for k_prep4 (0.110 ms = 4.96 TB/s of its 546 MB) and the likelihood rows (4.85 TB/s) rested on torch's copy
// kernel (4.8 TB/s, profiles/r02_hbm_copy_rate.json); the guide measures 6.29 TB/s for a float4 copy.  Variants, each over a
// 400 MB read (10^6 x 50 doubles, k_prep4's input) with 16-byte accesses, grid = CUs x waves/CU swept:
//   read      : read-only (sum into a register, one store per thread at the end)
//   copy      : read + write of the same size
//   prep_mix  : read + write of 137/400 of it (k_prep4's mix: 400 B in, 128 + 9 B out per proposal)
//   read_lds  : read-only through LDS-DMA (global_load_lds, 16-byte pieces -- the form k_prep4 uses), no register traffic
//   nt        : copy with non-temporal loads and stores
// Per variant: best-of-5 time, GB/s of (bytes read + bytes written).
//   hipcc --offload-arch=gfx950 -O3 -o bin/hbm_stream_probe hbm_stream_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

typedef float float4v __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const float4v *__restrict__ src, size_t n16, float *sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float4v acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    float4v v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  for (; i < n16; i += stride) acc += src[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = acc[0];
}

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_copy(const float4v *__restrict__ src, float4v *__restrict__ dst, size_t n16, int wnum, int wden) {
  // writes wnum of every wden 16-byte pieces (wnum = wden: a plain copy)
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    float4v v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t j = i + u * stride;
      if (wnum == wden) {
        if (NT) __builtin_nontemporal_store(v[u], dst + j); else dst[j] = v[u];
      } else if ((j / 64) % wden < (size_t)wnum) {   // whole waves write or do not (wave-uniform), contiguous output
        const size_t o = (j / 64 / wden * wnum + (j / 64) % wden) * 64 + (j & 63);
        if (NT) __builtin_nontemporal_store(v[u], dst + o); else dst[o] = v[u];
      }
    }
  }
  for (; i < n16; i += stride)
    if (wnum == wden) dst[i] = src[i];
}

// read-only through LDS-DMA: each wave keeps DEPTH KiB in flight, never touches the data
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read_lds(const float4v *__restrict__ src, size_t n16, float *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned char *mine = lds + (size_t)wv * DEPTH * 1024;
  const size_t nwaves = (size_t)gridDim.x * 4;
  const size_t wave = (size_t)blockIdx.x * 4 + wv;
  for (size_t base = wave * 64 * DEPTH; base + 64 * DEPTH <= n16; base += nwaves * 64 * DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
      __builtin_amdgcn_global_load_lds((gptr_t *)(src + base + u * 64 + lane), (lptr_t *)(mine + u * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (mine[lane] == 77 && sink[1] == 3.0f) sink[0] = 1.0f;
}



int main() {
  const size_t bytes = (size_t)1000000 * 50 * 8, n16 = bytes / 16;
  float4v *src, *dst;
  float *sink;
  CK(hipMalloc(&src, bytes));
  CK(hipMalloc(&dst, bytes));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(src, 1, bytes));
  CK(hipMemset(dst, 0, bytes));
  CK(hipMemset(sink, 0, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_read_lds<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_read_lds<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  printf("{\"device\": \"%s\", \"cus\": %d, \"bytes_read\": %zu, \"rows\": [\n", prop.name, cus, bytes);
  bool first = true;
  auto run = [&](const char *name, int wg_per_cu, double moved, auto &&launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, 0));
      launch(cus * wg_per_cu);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%s {\"variant\": \"%s\", \"workgroups_per_cu\": %d, \"ms\": %.4f, \"GBps\": %.0f}", first ? " " : ",", name, wg_per_cu, best,
           moved / (best * 1e-3) / 1e9);
    printf("\n");
    first = false;
  };
  for (int w : {2, 4, 8, 16, 32}) {
    run("read, 4 x 16 B in flight per thread", w, (double)bytes, [&](int g) { hipLaunchKernelGGL((k_read<4>), dim3(g), dim3(256), 0, 0, src, n16, sink); });
    run("read, 8 x 16 B in flight per thread", w, (double)bytes, [&](int g) { hipLaunchKernelGGL((k_read<8>), dim3(g), dim3(256), 0, 0, src, n16, sink); });
    run("copy", w, 2.0 * bytes, [&](int g) { hipLaunchKernelGGL((k_copy<4, false>), dim3(g), dim3(256), 0, 0, src, dst, n16, 1, 1); });
    run("copy, non-temporal", w, 2.0 * bytes, [&](int g) { hipLaunchKernelGGL((k_copy<4, true>), dim3(g), dim3(256), 0, 0, src, dst, n16, 1, 1); });
    run("read + 11/32 write (k_prep4's mix)", w, bytes * (1.0 + 11.0 / 32.0), [&](int g) { hipLaunchKernelGGL((k_copy<4, false>), dim3(g), dim3(256), 0, 0, src, dst, n16, 11, 32); });
    run("read + 11/32 write, non-temporal", w, bytes * (1.0 + 11.0 / 32.0), [&](int g) { hipLaunchKernelGGL((k_copy<4, true>), dim3(g), dim3(256), 0, 0, src, dst, n16, 11, 32); });
  }
  for (int w : {1, 2, 4}) {
    run("read through LDS-DMA, 8 KiB in flight per wave", w, (double)bytes, [&](int g) { hipLaunchKernelGGL((k_read_lds<8>), dim3(g), dim3(256), 4 * 8 * 1024, 0, src, n16, sink); });
    run("read through LDS-DMA, 4 KiB in flight per wave", w, (double)bytes, [&](int g) { hipLaunchKernelGGL((k_read_lds<4>), dim3(g), dim3(256), 4 * 4 * 1024, 0, src, n16, sink); });
  }
  printf("]}\n");
  return 0;
}
