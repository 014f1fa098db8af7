// mlf_walk.hpp -- population step-sampler state machine on the device (mlf_walk.hip).
// reference ultranest/stepfuncs.pyx (evolve :189-282, evolve_update :99-183,
// step_back :285-334, direction generators :348-533, update_vectorised_slice_sampler :537-630)
// and ultranest/popstepsampler.py (unitcube_line_intersection :26-61, diagnose_move_distances
// :64-94, PopulationSliceSampler :347-697).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mlf {

// Device-resident state of one walker population (PopulationSliceSampler attributes of the same
// names, reference popstepsampler.py:383-437).  P walkers, G = nsteps + 1 chain slots, d dims.
struct WalkState {
  int P, G, d, nparams;
  double *allu;            // [P][G][d]
  double *allL;            // [P][G]
  long long *generation;   // [P]   -1 = not started
  double *currentt;        // [P]   NaN = bracket undefined
  double *currentv;        // [P][d]
  double *left, *right;    // [P]
  uint8_t *sl, *sr;        // [P]   searching_left / searching_right
  double *currentp;        // [P][nparams]
  // per-step scratch
  double *unew;            // [P][d]
  uint8_t *movable;        // [P]   generation < nsteps at propose time
  uint8_t *acceptable;     // [P]
  uint8_t *success;        // [P]
  double *pnew;            // [P][nparams]   (full size; host results are expanded into it)
  double *Lnew;            // [P]
  double *dist2;           // [P]   t-space move distance^2 of successful walkers, NaN otherwise
};

// layer description for the move diagnostics (T1 of the region path)
struct WalkLayer {
  int kind;                // 0 affine (ctr, T row-major d x d), 1 scaling (mean, std), -1 none
  const double *ctr;
  const double *mat;
  const double *wrap;      // (1 - cut) per wrapped axis, NaN elsewhere; nullptr = no wraps
  double r2;               // maxradiussq
};

// Per-call scalars of the whole-step path when it is replayed as a hipGraph: kernel arguments are frozen
// at capture time, so the values that change from call to call are read from device memory instead.
struct StepParams {
  double Lmin, scale, dirscale, r2;
  unsigned long long seed, offset;
};

// direction generators on the device (Philox); kinds follow the reference functions
enum WalkDirection {
  DIR_CUBE_ORIENTED = 0,          // stepfuncs.pyx:348-370
  DIR_CUBE_ORIENTED_SCALED = 1,   // :373-399
  DIR_RANDOM = 2,                 // :401-422
  DIR_REGION_ORIENTED = 3,        // :425-450
  DIR_REGION_RANDOM = 4,          // :453-478
  DIR_DIFFERENTIAL = 5,           // :480-508
  DIR_MIXTURE = 6                 // :512-535
};

struct WalkDirData {
  const double *axes;      // transformLayer.axes, d x d row-major (kinds 3, 4, 6)
  const double *live;      // region.u, nlive x d (kinds 5, 6)
  int nlive;
  const double *std;       // region.u.std(axis=0) (kind 1)
};

// several rounds of the whole-step path in one launch sequence (mlf_walk.hip: k_walk_round0 / k_walk_pick / k_walk_rounds /
// k_walk_round_stats)
struct RoundsArgs {
  WalkState w;
  const double *live, *Ls;   // device copy of the live points and their likelihoods (restarts, differential directions)
  int nlive;
  int dirkind;
  WalkDirData dd;
  int tkind;
  double ta, tb;
  int lkind;
  const double *aux;
  double sigma;
  WalkLayer ly;
  uint8_t *was_starting;     // [P]
  const StepParams *sp;      // device: Lmin, scale, dirscale, r2, seed, offset of round 0
  long long *ring;           // device ring index
  int *ctl;                  // [0] rounds made R, [1] ring walker of this call, [2] harvested, [3] last round the ring walker has
                             // committed to (+ 2^30 once it is through), [5] a follower gave up waiting (8 words)
  uint8_t *rflags;           // [max_rounds][P] bit0 movable, bit1 acceptable, bit2 success, bit3 was (re)starting
  double *rdist2;            // [max_rounds][P] move distance^2 of the walkers that succeeded in the round
  int *rlast;                // [P] rounds [0, rlast) of this call wrote the walker's flags
  int force_memory_form;     // tests: every round through global memory (dw_round), as the first form of this path did
  double *rows;              // [max_rounds][5] step statistics per round
  double *rec;               // [0] harvested, [1] L, [2] left, [3] right, [4] R, [9 ..] u (d), p (nparams), next ring index
  int max_rounds;
  unsigned long long per_call;   // Philox counters one call of the call-by-call path consumes
  double *rparts;            // [max_rounds][chunks of 1024 walkers][5]: per-chunk statistics (populations above 1024 walkers)
  int phase;                 // k_walk_rounds: 0 = ring walker and followers in one launch, 1 = the ring walker only, 2 = the followers only
};
void launch_walk_rounds(const RoundsArgs &a, hipStream_t s);
// rows idx[j] of the device copy of the live points (and their likelihoods) replaced
void launch_walk_scatter_live(const double *rows, const double *Ls, const long long *idx, int n, int d, double *live, double *liveL,
                              hipStream_t s);

void launch_walk_reset(const WalkState &w, hipStream_t s);
// step_back + snapshot: flags[i] = bit0 !isfinite(currentt) | bit1 searching_left | bit2 searching_right
void launch_walk_step_back(const WalkState &w, double Lmin, long long *gmax_scratch, uint8_t *flags, hipStream_t s,
                           const StepParams *sp = nullptr);
void launch_walk_start(const WalkState &w, const long long *idx, int n, const double *rows, const double *L,
                       hipStream_t s);
void launch_walk_points(const WalkState &w, const long long *idx, int n, double *out, hipStream_t s);
void launch_walk_brackets(const WalkState &w, const long long *idx, int n, double scale, const double *v_rows,
                          hipStream_t s);
void launch_walk_brackets_philox(const WalkState &w, double scale, int kind, double dirscale, WalkDirData dd,
                                 unsigned long long seed, unsigned long long offset, hipStream_t s,
                                 const StepParams *sp = nullptr);
// unif: one U[0,1) per walker (host stream) or nullptr -> Philox(seed, offset + walker)
void launch_walk_propose(const WalkState &w, const double *unif, unsigned long long seed,
                         unsigned long long offset, hipStream_t s, const StepParams *sp = nullptr);
// p = transform(unew) for every walker: tkind 0 identity, 1 x*a + b, 2 (x*a)*b
void launch_walk_transform(const WalkState &w, int tkind, double a, double b, hipStream_t s);
// host likelihood: compacted (pnew, Lnew) of the acceptable walkers -> full-size arrays
void launch_walk_expand(const WalkState &w, const unsigned *blk, const double *pc, const double *Lc,
                        hipStream_t s);
void launch_walk_update(const WalkState &w, double Lmin, WalkLayer layer, hipStream_t s, const StepParams *sp = nullptr);
// rec: [0] harvested flag, [1] L, [2] left, [3] right, [4] nc, [5] nmovable, [6] nsuccess, [7] nfar,
//      [8] sum log(dist/ref + 1e-10), [9 ..] u (d) then p (nparams)
// ring_dev != nullptr: the ring index lives on the device (read, advanced when a walker was harvested, and
// reported in rec[9 + d + nparams])
// partials: scratch of 6 * ceil(P / 1024) doubles (two-stage reduction of the step statistics)
void launch_walk_harvest(const WalkState &w, long long ring, long long *ring_dev, double r2, double *rec, double *partials,
                         hipStream_t s, const StepParams *sp = nullptr, const uint8_t *was_starting = nullptr);
// front half of a whole step in one kernel: step_back, restart, new slice, proposal, prior transform
void launch_walk_prologue(const WalkState &w, const double *live, const double *Ls, int nlive, int dirkind, WalkDirData dd,
                          int tkind, double ta, double tb, uint8_t *was_starting, const StepParams &p, const StepParams *sp,
                          hipStream_t s);
// device-side setup_start: ring index skips restarting walkers, restarts draw live points with L > Lmin
void launch_walk_restart_philox(const WalkState &w, const double *live, const double *Ls, int nlive, double Lmin,
                                unsigned long long seed, unsigned long long offset, long long *ring, hipStream_t s,
                                const StepParams *sp = nullptr);

// ---- stateless forms on device arrays (the parity boundary of ultranest.stepfuncs) -------------
void launch_within_unit_cube(const double *u, int n, int d, uint8_t *out, hipStream_t s);
void launch_evolve_propose(const double *currentu, const double *currentv, const double *left,
                           const double *right, const uint8_t *sl, const uint8_t *sr, const double *currentt,
                           int n, int d, double *unew, hipStream_t s);
void launch_bisect_draw(const double *left, const double *right, const uint8_t *sl, const uint8_t *sr,
                        const double *unif_full, int n, double *currentt, hipStream_t s);
void launch_evolve_update(const uint8_t *acceptable, const double *Lnew_full, double Lmin, double *currentt,
                          double *left, double *right, uint8_t *sl, uint8_t *sr, uint8_t *success, int n,
                          hipStream_t s);
void launch_step_back(double Lmin, double *allL, int n, int G, long long *generation, double *currentt,
                      long long *gmax_scratch, hipStream_t s);
void launch_line_intersection(const double *origin, const double *direction, int n, int d, double *tleft,
                              double *tright, hipStream_t s);
void launch_slice_update(const double *t, double *tleft, double *tright, const double *pL, const double *pu,
                         const double *pp, long long *worker_running, long long *status, double threshold,
                         double shrink, double *allu, double *allL, double *allp, int popsize, int d, int nparams,
                         long long *discarded, hipStream_t s);
void launch_row_dist2(const double *a, const double *b, int n, int d, double *out, hipStream_t s);

}  // namespace mlf
