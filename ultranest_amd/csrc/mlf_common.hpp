// mlf_common.hpp -- shared declarations of the gfx950 MLFriends kernels (internal; the public
// boundary is include/mlfriends_hip.h).
//
// Data layouts in HBM (all float64):
//   refT  : live points, COORDINATE-major  [DP][npad]   (npad = n rounded up to 64, zero padded)
//           -> a wave reads 64 consecutive live points of one coordinate in one 512-byte load
//   refR  : live points, row-major         [npad][DP]   (zero padded), for LDS tile staging
//   query : row-major (nq, ldq) as handed over by the caller
// DP is the compile-time padded dimensionality of a kernel instance (d <= DP, DP even).  Zero
// padding is exact: (0-0)^2 adds +0.0 to a non-negative accumulator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mlf {

constexpr int kWave = 64;
constexpr int kScanQB = 64;        // queries staged in LDS per scan workgroup
constexpr int kScanThreads = 256;  // 4 waves; wave w owns live-point tiles w, w+4, ...
constexpr int kBootTI = 32;        // live points per LDS sub-tile in the bootstrap kernel
constexpr int kBootGroup = 32;     // bootstrap rounds handled per launch (one selection bit each)
constexpr int kNone = 0x7fffffff;  // "no neighbour found yet"

enum ScanMode : int {
  SCAN_FIRST = 0,  // K1: lowest live-point index within r2, else -1        (int64 out)
  SCAN_COUNT = 1,  // K2: number of live points within r2                   (int64 out)
  SCAN_FLAGS = 2,  // K3 pass 1: 64-bit hit ballots per (query, tile)       (u64 out)
  SCAN_MASK = 3    // R3: byte mask "gate && any live point within r2"     (u8 out)
};

struct ScanArgs {
  const double *refT;
  int n, npad, ntiles;
  const double *q;        // query element (j, k) at q[j*ldq + k*ldk]
  long long ldq;
  long long ldk;          // 0 or 1: row-major; > 1: coordinate-major
  long long nq;
  int d;
  double r2;
  int mode;
  const uint8_t *gate;            // optional, per query: 0 = do not scan (result: none)
  int only_gated;                 // 1: write outputs only for gated-in queries; idle workgroups exit early
  const int *slot;                // optional: query j reads row slot[j] of q (negative: no coordinates, never scanned)
  const unsigned *any_flag;       // optional (device): 0 = no query is gated in, every workgroup returns at once
  long long *out_idx;             // SCAN_FIRST / SCAN_COUNT
  unsigned long long *out_flags;  // SCAN_FLAGS  [nq][ntiles]
  uint8_t *out_mask;              // SCAN_MASK
};

struct BootArgs {
  const double *refT;   // [DP][npad]
  const double *refR;   // [npad][DP]
  const unsigned *sel;  // [npad] bit b set = live point selected in bootstrap round b of this group
  int n, npad;
  int chunk;            // live points per blockIdx.y (multiple of kBootTI)
  unsigned long long *M;  // [kBootGroup][npad] running minima as ordered bit patterns
};

struct PrepArgs {
  const double *pts;
  long long np;
  int d;
  int do_ell;
  const double *ell_ctr;  // [DP] zero padded
  const double *ell_A;    // [d][DP] rows zero padded
  double enlarge;
  uint8_t *mask;          // out (do_ell) : inside ellipsoid
  double *q_out;          // optional
  int do_tr;
  const double *lay_ctr;      // [DP]
  const double *lay_Tt;       // [d][DP]: row c holds T[:, c], zero padded
  const double *wrap_shift;   // [DP] (NaN = not wrapped) or nullptr
  double *t_out;
  long long ldt;
};

struct QuadMaxArgs {
  const double *u;          // (n, d) row-major points
  int n, d;
  const uint8_t *selected;  // (B, n)
  const double *ctr;        // [B][DP]
  const double *invcov;     // [B][d][DP]
  double *part;             // [B][ceil(n/256)] per-workgroup maxima
};

// smallest instantiated DP >= d, or -1 (d > MLF_MAX_DIM)
int pick_dp(int d);

hipError_t launch_scan(int dp, const ScanArgs &a, hipStream_t s);
hipError_t launch_boot(int dp, const BootArgs &a, int nchunks, hipStream_t s);
hipError_t launch_prep(int dp, const PrepArgs &a, hipStream_t s);
hipError_t launch_boot_quadmax(int dp, const QuadMaxArgs &a, int B, hipStream_t s);

// instantiated dimensionalities: every even value up to 32, every 4th up to 64 (+50, the
// headline configuration), every 16th up to 128
#define MLF_FOR_EACH_DP(X)                                                                   \
  X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) \
  X(36) X(40) X(44) X(48) X(50) X(52) X(56) X(60) X(64) X(80) X(96) X(112) X(128)

}  // namespace mlf
