// hip_context.hpp -- internal: the device context behind NdlqrHipCtx* and the launch-profiling
// helper, shared by ndlqr_hip.hip and the per-size translation units (small_instance.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ndlqr.h"
#include "ndlqr_hip.h"

#include "kernels_common.hpp"

// records the message for ndlqr_hip_last_error(), prints it, returns NDLQR_ERR_NO_DEVICE (or
// NDLQR_ERR_INVALID for a rejected launch configuration / argument)
int ndlqr_hip_fail(const char* what, hipError_t e);
struct NdlqrHipCtx;
int ndlqr_hip_ensure_F(NdlqrHipCtx* c);
#define HIP_TRY(expr)                                                  \
  do {                                                                 \
    hipError_t e_ = (expr);                                            \
    if (e_ != hipSuccess) return ndlqr_hip_fail(#expr, e_);            \
  } while (0)

// ------------------------------------------------------------------------------ context

enum { SLOT_LEAF = 0, SLOT_SEP, SLOT_SCHUR, SLOT_BOUNDARY, SLOT_APPLY, SLOT_BOTTOM, SLOT_UPPER, SLOT_TOP, SLOT_COUNT };

struct PendingEvent {
  int slot;
  hipEvent_t start, stop;
};

// Second set of everything a solve writes (and of what orders it): consecutive solves of one context
// alternate between the two sets, each on its own stream, so that the thinly populated upper-level
// kernels and the HBM-bound back-substitution of one solve run beside the ALU-bound bottom kernel of
// the next (only for schedules without cross-solve state: no factor array, no kept records).
struct NdlqrAltSlot {
  bool ready = false;
  double* rec = nullptr;
  double* red = nullptr;
  size_t red_bytes = 0;
  double* ytop = nullptr;
  double* z = nullptr;
  double* rhs = nullptr;      // this set's copy of the right-hand side (ndlqr_hip_step_async replaces it per step)
  double* xfer = nullptr;     // transfer staging in HBM: flat q | r | d | x0 going up, packed [batch][nvars] coming down
  int* tree_cnt = nullptr;
  int* h_fail = nullptr;
  hipStream_t stream = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  unsigned graph_flags = 0;
  hipStream_t graph_stream = nullptr;
  bool graph_rec_complete = false;
  bool graph_rec_compact = false;
  const char* graph_schedule = "none";
  unsigned graph_apply = 0;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};

struct NdlqrHipCtx {
  ndlqr::Dims d;   // block sizes of the DEVICE layout (every kernel works on these)
  ndlqr::Dims du;  // the caller's block sizes: the same, or smaller when the problem runs zero-padded into the next
                   // size-specialised instance ("padded shapes", ndlqr_hip_create); only the boundary functions see it
  bool padded;
  double* pad_stage;      // HBM staging of caller-layout inputs / outputs of a padded shape (grown on demand)
  size_t pad_stage_cap;   // doubles
  int device;
  unsigned flags;
  hipStream_t stream;
  bool own_stream;
  double* AB;
  double* QR;
  double* rhs;
  double* F;    // complete factor array; allocated by the first solve whose schedule touches it (ndlqr_hip_ensure_F)
  double* z;
  double* rec;  // [batch][N][2 n^2 + n] separator records f_a | f_bb | z_sep (compact forms use the front of a record and its last n entries)
  double* ytop; // [batch][N/8][n] multipliers of the separators of level >= 3 (rb_backsub_top -> rb_backsub)
  double* red;  // accumulators of the separator-only schedules: [batch][N/4][slot] (size-specialised shapes, allocated with the context) or [batch][N/2][4 n^2 + 2 n] (runtime-sized schedule, on its first solve)
  size_t red_bytes;
  int rowbcast;  // bottom levels of the separator-only schedule on the row-broadcast core (rb_bottom): NDLQR_ROWBCAST=1 always, 0 never (bottom_reduced_mc), unset (-1): by block size
  int tree;  // tree schedule (bottom_reduced_mc<TREE>: one launch for the whole factorisation, wavefronts climbing on arrival counters): NDLQR_TREE=1 always, 0 never, unset (-1): when all bottom wavefronts are resident at once (small batches: fewer launches win; large ones: a launch per level is faster)
  int fuse2;  // tree level 2 inside the bottom launch (bottom8_reduced_mc) instead of as a launch of its own: NDLQR_FUSE2=1 always, 0 never, unset (-1): where it measured faster -- the (12,4) instance (launch_small.hpp)
  int* tree_cnt;  // arrival counters of the separators of level >= 2, [batch][N / 4]; zero between solves (reset by the root's wavefront)
  NdlqrAltSlot alt;       // the other buffer set / stream of the two-deep solve pipeline
  int pipeline;           // 1: stream-ordered solves; 2 (default, NDLQR_PIPELINE): consecutive solves alternate slots
  unsigned solve_count;   // solves enqueued so far (parity picks the slot)
  bool in_alt;            // the context's buffer / stream fields currently hold the alternate set
  const double* z_latest; // solution of the most recent solve (either slot)
  hipStream_t stream_latest;
  int* h_fail_other;      // the failure word of the slot that is not current
  bool state_dirty;  // a solve failed to launch or to complete: counters / failure words are zeroed before the next one
  int fail_base;     // value of the (cumulative) batch-wide failure counter at the last synchronisation
  int* info;
  const char* schedule;  // name of the launch sequence the last solve used (ndlqr_hip_schedule)
  int* h_fail;      // pinned host word: the batch-wide failure count, copied behind the last kernel of a solve
  double* kkt_out;  // [2 batch] scratch of ndlqr_hip_kkt_residual (allocated on first use)
  // several right-hand sides per problem (ndlqr_hip_solve_multi_rhs): buffers for `multi_cap` right-hand sides, grown on demand
  size_t multi_cap;
  double *multi_rhs, *multi_z, *multi_zsep, *multi_fsum, *multi_ytop, *multi_in, *multi_out;
  double* sep_scratch;  // S-bar and panel of every level-0 separator in global memory: blocks beyond the LDS of separator_generic (allocated on first use)
  double* xfer;     // transfer staging of the current buffer set (see NdlqrAltSlot::xfer; allocated on first use)
  double* h_stage[2];  // pinned bounce buffers of the downloads into pageable host memory (allocated on first use)
  bool no_top;        // NDLQR_NO_TOP=1: the last three tree levels as launches of their own (A/B timing of reduced_top_mc)
  int top_levels;     // tree levels inside reduced_top_mc (NDLQR_TOP_LEVELS, 3 .. 5): beyond three a wavefront takes several separators of the first ones in turn
  bool no_mfma;       // NDLQR_NO_MFMA=1: keep the scalar Schur kernel for large blocks (A/B timing)
  bool rec_complete;  // last factorisation left every separator record and factor (fast mode + KEEP / KEEP_RECORDS)
  bool rec_compact;   // ... in the compact form of the default schedule (level-0 records = L, the factors of the upper
                      // separators in the slack of those slots): the re-solve is rb_forward / rb_forward_top / rb_backsub
  bool graph_rec_complete;  // the same for the captured launch sequence (replays do not re-enter the launch code)
  bool graph_rec_compact;
  const char* graph_schedule;  // and its name
  int sep_threads;    // NDLQR_SEP_THREADS: workgroup size of the matrix-core separator (0 = by block size)
  hipEvent_t ev_start, ev_stop;
  hipEvent_t ev_step[2];  // end of the steps of ndlqr_hip_step_async, alternating (ndlqr_hip_synchronize_previous)
  unsigned step_count;
  hipEvent_t ev_inputs;  // orders the other buffer set's stream behind a device-side replacement of the inputs
  // One LOGICAL right-hand side, two physical copies (one per buffer set): generation counters per part -- 0: q,
  // 1: r, 2: d, 3: x0 -- of the latest write and of each set's copy ([0] primary set, [1] alternate). Whoever writes
  // (uploads, device packing, an MPC step) writes the CURRENT set and bumps its generations; a solve or step that
  // lands on a set whose copy is behind in a part it does not replace copies that part over first (rhs_make_current).
  unsigned long long rhs_latest[4];
  unsigned long long rhs_gen[2][4];
  // what an MPC step brings down (ndlqr_hip_set_step_selection): sel_nknots == 0: every solution, [batch][nvars]
  int sel_knot0, sel_nknots;
  unsigned sel_blocks;  // NDLQR_SOLN_* bits; with NDLQR_SOLN_ONLY (8) a step computes nothing but the selected knots:
  // the last launch of the back-substitution covers workgroups [apply_blk0, apply_blk0 + apply_nblk) of eight knots only
  // (apply_nblk == 0: all; set by ndlqr_hip_step_async around its launches, honoured by the schedules that end in
  // rb_backsub, part of the key of the captured launch sequence), and z_partial says that the latest solution is such a
  // slice -- [z_blk0, z_blk0 + z_nblk) -- so that nothing else is handed out until the next complete solve
  int apply_blk0, apply_nblk;
  unsigned graph_apply;  // (apply_blk0 << 16 | apply_nblk) of the captured launch sequence of the current buffer set
  bool z_partial;
  int z_blk0, z_nblk;
  int step_set[2];   // buffer set (0 primary, 1 alternate) of the steps behind ev_step[0 / 1]
  // One-shot solve of a small batch from / into pinned host staging (ndlqr_hip_solve_staged; the drop-in ndlqr_Solve):
  // AB | QR | rhs going up, the solution blocks [batch][N][2n+m] coming down, all in the caller's block size; the whole
  // sequence -- three copies up, the launch chain, the copy down -- is ONE captured graph.
  double* h_io;            // pinned: AB | QR | rhs | z
  hipGraphExec_t graph_staged;
  unsigned graph_staged_flags;
  bool timing_pending;
  double last_ms;
  int last_failures;
  // the launch sequence captured as a hipGraph (replayed when nothing that shapes it changed)
  hipGraphExec_t graph_exec;
  unsigned graph_flags;
  hipStream_t graph_stream;
  bool fact_valid;   // the device holds a complete factorisation (last solve ran with KEEP_FACT)
  // profile
  std::vector<PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;
  double slot_ms[SLOT_COUNT];
  int slot_launches[SLOT_COUNT];
};

static inline hipEvent_t take_event(NdlqrHipCtx* c) {
  if (!c->event_pool.empty()) {
    hipEvent_t ev = c->event_pool.back();
    c->event_pool.pop_back();
    return ev;
  }
  hipEvent_t ev = nullptr;
  (void)hipEventCreate(&ev);
  return ev;
}

struct ScopedSlot {  // brackets one kernel launch with events when profiling is on
  NdlqrHipCtx* c;
  PendingEvent pe;
  bool on;
  ScopedSlot(NdlqrHipCtx* ctx, int slot) : c(ctx), on((ctx->flags & NDLQR_FLAG_PROFILE) != 0) {
    if (!on) return;
    pe.slot = slot; pe.start = take_event(c); pe.stop = take_event(c);
    (void)hipEventRecord(pe.start, c->stream);
  }
  ~ScopedSlot() {
    if (!on) return;
    (void)hipEventRecord(pe.stop, c->stream);
    c->pending.push_back(pe);
  }
};
