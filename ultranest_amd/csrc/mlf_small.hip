// mlf_small.hip -- MLFriends.inside (reference mlfriends.pyx:1186-1211) for 1 ... 256 proposals in a single launch.
//
// The scalar step samplers of the reference call region.inside() with ONE point per call (stepsampler.py:296-330,
// 1060-1071) and the stock driver with the N live points once per iteration (integrator.py:1854-1855); the batched
// pipeline (per-proposal stage, pre-filter sweeps, re-check, exact scan: 5-7 launches, two memcpy calls) costs
// 60-140 us for such a call whatever the batch size.  Here the whole test is one kernel:
//   workgroup (p, part): proposal p, share `part` of the live points
//     1. the proposal's row -> LDS (straight from the caller's pinned staging buffer)
//     2. H3 ellipsoid test: |L^T delta|^2 against the band eps = 2^-34 |A|_F |delta|^2 (as k_ell_exact); inside the
//        band -- practically never -- the reference's own order: one accumulator, j outer, (d_j * A_jk) * d_k
//     3. T1 whitening: k-ascending binary64 FMA chain per output coordinate (as k_prep / k_recheck_whiten);
//        ScalingLayer: (w - mean) / std
//     4. neighbour scan of this workgroup's live points: lane = live point, acc += (a_k - t_k)^2 for k ascending,
//        not fused, `<=` r2 (as k_scan)
//     5. the last workgroup of a proposal writes its mask byte (straight into the caller's pinned buffer) and
//        returns the two scratch words of the proposal to zero
// Same arithmetic as the batched kernels => same masks (tests/test_small_path.py compares the two on the GPU and
// both with the CPU restatement of the reference).  -ffp-contract=off; FMAs only where written.
#include "mlf_small.hpp"

#include <math.h>

namespace mlf {

// Completion protocol.  The caller spins on a host-visible flag instead of waiting for the kernel's completion signal
// (~4 us later; the next launch is ordered behind this one by the stream), so the flag is the ONLY thing that orders the
// mask bytes before the caller's read.  Everything the workgroups tell each other travels in ATOMIC read-modify-write
// words, which meet at the device's coherence point whatever XCD (and L2) a workgroup runs on:
//   state[p]   += 1 per reporting workgroup, += 0x10000 if it found a neighbour (its return value tells the last one)
//   finished   += 1 per completed proposal (its return value tells the last proposal's workgroup: the publisher)
// A workgroup issues the second update only after the first has returned, so when the publisher's count comes back
// every state word is final; its threads fetch (and zero) them with atomic exchanges, pack the answers 16 to a store
// into the host buffer, fence at system scope and raise the flag.  No other workgroup writes to host memory, and nobody
// needs a cache write-back.  (Round 2 let each finishing workgroup write its own byte into host memory, only the last
// one fencing: bytes of other workgroups could still be in flight when the flag arrived -- a stale answer of the previous
// call, ADVICE r2.  Fencing in every workgroup closes that, at +50 % latency for 128 proposals; so does staging the
// answers in plain device memory, at the price of an L2 write-back per workgroup across the 8 XCDs.)
__device__ __forceinline__ bool finish_point(const SmallArgs &a) {
  if (a.np == 1) return true;
  // release: this workgroup's state word is visible before the count moves; acquire: the workgroup that completes the
  // count sees every other workgroup's state word (a relaxed count left the two updates unordered)
  return __hip_atomic_fetch_add(a.finished, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.np - 1u;
}

// LDS: d x (d|1) doubles per matrix when STAGE (d <= 64): rows of L^T, then the layer matrix; all loads of a call --
// the proposal's row from the host, both matrices, the first share of live points -- are in flight together, the
// arithmetic starts when the slowest (the row, ~2 us over PCIe) has landed.
template <bool STAGE>
__global__ __launch_bounds__(256) void k_inside_small(SmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) double mats[];
  __shared__ double xs[kSmallMaxDim], dl[kSmallMaxDim], dw[kSmallMaxDim], tq[kSmallMaxDim];
  __shared__ double red[2][4];
  __shared__ int gate_s, last_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int p = blockIdx.x / a.wpp, part = blockIdx.x - p * a.wpp;
  const int d = a.d, ls = d | 1;
  const bool affine = a.use_scan && a.layer_kind == 0;
  double x = 0.0;
  if (tid < d) x = a.pts[(size_t)p * d + tid];
  double *ltl = mats, *tl = mats + (size_t)d * ls;
  if (STAGE) {
    if (a.chol_ok)
      for (int e = tid; e < d * d; e += 256) {
        const int i = e / d, j = e - i * d;
        ltl[i * ls + j] = a.ell_Lt[(size_t)i * a.dp + j];
      }
    if (affine)
      for (int e = tid; e < d * d; e += 256) {
        const int k = e / d, c = e - k * d;
        tl[k * ls + c] = a.lay_T8[(size_t)k * a.ldt8 + c];
      }
  }
  // the first live point of this thread, coordinate by coordinate (registers; only when the loop bound is small)
  constexpr int kPre = STAGE ? 64 : 1;
  double pre[kPre];
  const int i0 = part * 256 + tid;
  if (STAGE && a.use_scan) {
#pragma unroll
    for (int k = 0; k < kPre; ++k) pre[k] = (k < d && i0 < a.n) ? a.refT[(size_t)k * a.npad + i0] : 0.0;
  }
  if (tid < d) {
    xs[tid] = x;
    dl[tid] = x - a.ell_ctr[tid];
    if (a.use_scan) {
      double w = x;
      if (a.wrap) {
        const double sh = a.wrap[tid];
        if (sh == sh) w = fmod(w + sh, 1.0);
      }
      dw[tid] = w - a.lay_ctr[tid];
    }
  }
  __syncthreads();

  // ---- threads 0..127: ellipsoid bound; threads 128..255: whitening
  double qt = 0.0, n2 = 0.0;
  if (tid < 128) {
    if (a.chol_ok && tid < d) {
      double y = 0.0;
      if (STAGE) {
        for (int j = tid; j < d; ++j) y = __builtin_fma(ltl[tid * ls + j], dl[j], y);
      } else {
        const double *lrow = a.ell_Lt + (size_t)tid * a.dp;
#pragma unroll 8
        for (int j = tid; j < d; ++j) y = __builtin_fma(lrow[j], dl[j], y);
      }
      qt = y * y;
      n2 = dl[tid] * dl[tid];
    }
    for (int o = 32; o > 0; o >>= 1) {
      qt += __shfl_xor(qt, o, 64);
      n2 += __shfl_xor(n2, o, 64);
    }
    if (lane == 0) {
      red[0][wv] = qt;
      red[1][wv] = n2;
    }
  } else if (a.use_scan) {
    const int c = tid - 128;
    if (c < d) {
      double acc;
      if (a.layer_kind == 0) {
        acc = 0.0;
        if (STAGE) {
          for (int k = 0; k < d; ++k) acc = __builtin_fma(dw[k], tl[k * ls + c], acc);
        } else {
#pragma unroll 8
          for (int k = 0; k < d; ++k) acc = __builtin_fma(dw[k], a.lay_T8[(size_t)k * a.ldt8 + c], acc);
        }
      } else {
        acc = dw[c] / a.lay_std[c];
      }
      tq[c] = acc;
    }
  }
  __syncthreads();
  bool decided = false, inside = false;
  if (a.chol_ok) {
    qt = red[0][0] + red[0][1];
    n2 = red[1][0] + red[1][1];
    const double eps = a.eps_scale * n2;
    if (qt + eps < a.enlarge) {
      decided = true;
      inside = true;
    } else if (qt - eps > a.enlarge) {
      decided = true;
    }
  }
  if (!decided) {   // workgroup-uniform: inside the band (or no factorisation): the reference's own order
    if (tid == 0) {
      double acc = 0.0;
      for (int j = 0; j < d; ++j) {
        const double dj = dl[j];
        const double *arow = a.ell_A + (size_t)j * a.dp;
        for (int k = 0; k < d; ++k) acc += (dj * arow[k]) * dl[k];
      }
      gate_s = acc <= a.enlarge ? 1 : 0;
    }
    __syncthreads();
    inside = gate_s != 0;
  }

  bool publish = false;   // thread 0: this workgroup completed the last proposal of the call
  if (!a.use_scan) {
    if (part == 0 && tid == 0) {
      // the returned value is consumed, so the update is a returning atomic that has completed before the count moves
      const unsigned old = __hip_atomic_fetch_add(a.state + p, (inside ? 0x10000u : 0u) + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      publish = (old & 0xffffu) == 0u && finish_point(a);
    }
  } else {

  // ---- scan of this workgroup's live points
  int found = 0;
  if (inside) {   // workgroup-uniform
    for (int i = i0; i < a.n; i += a.wpp * 256) {
      double acc = 0.0;
      if (STAGE && i == i0) {
#pragma unroll
        for (int k = 0; k < kPre; ++k)
          if (k < d) {
            const double diff = pre[k] - tq[k];
            acc += diff * diff;
          }
      } else {
        const double *col = a.refT + i;
#pragma unroll 8
        for (int k = 0; k < d; ++k) {
          const double diff = col[(size_t)k * a.npad] - tq[k];
          acc += diff * diff;
        }
      }
      if (acc <= a.r2) found = 1;
    }
  }
  found = __syncthreads_or(found);
  if (tid == 0) {
    const unsigned old = atomicAdd(a.state + p, ((found && inside) ? 0x10000u : 0u) + 1u);
    if ((old & 0xffffu) == (unsigned)a.wpp - 1u) publish = finish_point(a);   // the last workgroup of this proposal
  }
  }
  if (tid == 0) last_s = publish ? 1 : 0;
  __syncthreads();
  if (last_s) {   // workgroup-uniform: fetch (and zero) every proposal's word, 16 answers per thread and host store
    if (tid * 16 < a.np) {
      unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int t = tid * 16 + q;
        const unsigned word = t < a.np ? atomicExch(a.state + t, 0u) : 0u;
        w[q >> 2] |= ((word >> 16) != 0u ? 1u : 0u) << (8 * (q & 3));
      }
      reinterpret_cast<uint4 *>(a.mask)[tid] = make_uint4(w[0], w[1], w[2], w[3]);   // the buffer is 16-byte aligned, kSmallMaxPoints long
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
      if (a.np > 1) (void)atomicExch(a.finished, 0u);
      __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

void launch_inside_small(const SmallArgs &a, hipStream_t s) {
  if (a.np <= 0) return;
  const dim3 grid((unsigned)(a.np * a.wpp));
  if (a.d <= 64) {
    const size_t lds = (size_t)2 * a.d * (a.d | 1) * sizeof(double);
    static DeviceGrant grant;
    (void)grant.ensure([] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_inside_small<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 65 * (int)sizeof(double));
    });
    hipLaunchKernelGGL(k_inside_small<true>, grid, dim3(256), lds, s, a);
  } else {
    hipLaunchKernelGGL(k_inside_small<false>, grid, dim3(256), 0, s, a);
  }
}

}  // namespace mlf
