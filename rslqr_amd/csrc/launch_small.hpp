// launch_small.hpp -- internal: launch sequences of the size-specialised kernels, instantiated
// once per (nstates, ninputs) in its own translation unit (small_instance.hip) so that the
// instances compile in parallel.
#pragma once
#include "hip_context.hpp"
#include "kernels_leaf.hpp"
#include "kernels_small.hpp"
#include "kernels_bottom_reduced.hpp"
#include "kernels_rowbcast.hpp"

// Size-specialised launch sequences.
//   separator-only ("reduced"): bottom kernel (leaf phase + levels 0, 1) -> one launch per upper level
//       -> back-substitution; fast mode without KEEP, shapes with matrix-core products
//   knot-based: bottom_small (leaf phase + levels 0, 1 on the knot states) -> level_small per upper
//       level (separator + the two boundary knots of every subtree) -> backsub_small (fast mode
//       without KEEP: solution from the records) or apply_small (strict / KEEP: every knot through all
//       upper levels in registers, one pass)
constexpr int kBottomLevels = 2;  // tree levels fused with the leaf phase

// What launch_small is going to do for this context: decided once, before the launch sequence is
// enqueued (and possibly captured), so that ndlqr_hip.hip can allocate what the schedule needs.
struct SmallPlan {
  bool lean;     // solution by back-substitution from the separator records (fast mode, no KEEP)
  int store_l;   // keep the separator factors for a record-based re-solve (KEEP_RECORDS)
  bool reduced;  // separator-only schedule (bottom_reduced_mc + reduced_level_mc)
  bool tree;     // ... with the whole factorisation in one launch (small batches)
  bool compact;  // ... with compact level-0 records and the two-launch back-substitution (kernels_rowbcast.hpp)
  bool rowbcast; // ... and the bottom levels on the row-broadcast core (four separators per wavefront)
  bool needs_F;  // the schedule reads or writes the factor array
};

// An MPC step that wants nothing but a knot range (NDLQR_SOLN_ONLY): the last launch of a back-substitution whose workgroups
// take eight knots each runs those of the range -- grid and Dims::xoff of that launch.
static inline unsigned apply_grid(const NdlqrHipCtx* c, const ndlqr::Dims& d) {
  return (unsigned)(c->apply_nblk > 0 ? c->apply_nblk : d.N / 8);
}
static inline ndlqr::Dims apply_dims(const NdlqrHipCtx* c, const ndlqr::Dims& d) {
  ndlqr::Dims da = d;
  if (c->apply_nblk > 0) da.xoff += c->apply_blk0;
  return da;
}

template <int NX, int NU, bool STRICT, bool KEEP>
static SmallPlan plan_small(const NdlqrHipCtx* c) {
  const ndlqr::Dims& d = c->d;
  SmallPlan p;
  // fast mode without KEEP: solution by back-substitution from the separator records (backsub_small
  // resolves K + 4 separators of NX rows in one 256-thread workgroup)
  p.lean = !STRICT && !KEEP && (d.K + 4) * NX <= 256;
  p.store_l = (c->flags & NDLQR_FLAG_KEEP_RECORDS) ? 1 : 0;  // factors for a record-based re-solve
  p.reduced = false;
  p.tree = false;
  p.rowbcast = false;
  p.compact = false;
  if constexpr (!STRICT && !KEEP && ndlqr::P1OnMatrixCores<NX, NU>::value) {
    if (p.lean && c->red) {
      p.reduced = true;
      // tree schedule for small batches (at most half a resident round of bottom wavefronts): three
      // launches instead of K + 1; measured cross-over at batch x N / 4 ~ 4096 wavefronts
      p.tree = c->tree_cnt && (c->tree == 1 || (c->tree < 0 && (size_t)d.batch * (d.N >> 2) <= 2048));
      // compact level-0 records (L of S-bar only) and the two-launch back-substitution: the tree schedule keeps the
      // one-kernel back-substitution; rb_backsub's thread roles need 8 (2 nx + nu) <= 256 (and rb_backsub_top's sweep
      // array, N / 8 multipliers, has to fit the LDS of its one workgroup per problem). With KEEP_RECORDS (round 4): the
      // same schedule, which then also keeps the factors of the separators of level >= 1 in the slack of the level-0
      // record slots (store_l = 2; the record-based re-solve rb_forward / rb_forward_top works on that: four sweep
      // arrays in the LDS of its one workgroup per problem)
      p.compact = !p.tree && d.N >= 16 && 8 * (2 * NX + NU) <= 256 &&
                  sizeof(double) * (size_t)(d.N >> 3) * NX * (p.store_l ? 4 : 1) <= 160 * 1024;
      if (p.compact && p.store_l) p.store_l = 2;
      // row-broadcast bottom kernel (one DPP row holds the rows of S-bar and of [A | B]'): its cost falls
      // with the block size, the matrix-core kernel's does not (16x16 tiles whatever n is). Measured bottom
      // kernel, N = 256 x 1024: (6,3) 0.105 vs 0.170 ms, (8,4) 0.144 vs 0.190, (9,3) 0.202 vs 0.244,
      // (10,4) 0.230 vs 0.271, (12,4) 0.301 vs 0.290 in round 2. Round 3 (paired Cholesky pass that carries the panel):
      // (10,4) 0.230 vs 0.207, (9,3) 0.203 vs 0.187, (8,4) 0.143 vs 0.152, (6,3) 0.104 vs 0.134 -- so it serves n <= 8
      // (NDLQR_ROWBCAST=0/1 overrides)
      p.rowbcast = p.compact && !p.store_l && NX <= 16 && NX + NU <= 16 && (c->rowbcast == 1 || (c->rowbcast < 0 && NX <= 8));
    }
  }
  // the separator-only schedule touches F only to park the factors of KEEP_RECORDS under its full-record forms
  p.needs_F = !(p.reduced && p.store_l != 1);
  return p;
}

template <int NX, int NU, bool STRICT, bool KEEP>
static int launch_small(NdlqrHipCtx* c) {
  const ndlqr::Dims& d = c->d;
  using Sh = ndlqr::SchurShape<NX, NU>;
  constexpr int JB = kBottomLevels;
  const SmallPlan plan = plan_small<NX, NU, STRICT, KEEP>(c);
  const bool lean = plan.lean;
  const int store_l = plan.store_l;
  // the record-based re-solve needs every separator's record and factor: KEEP writes them all,
  // KEEP_RECORDS adds the factors to the lean schedule
  c->rec_complete = !STRICT && (KEEP || (lean && store_l));
  if constexpr (!STRICT && !KEEP && ndlqr::P1OnMatrixCores<NX, NU>::value) {
    if (plan.reduced) {
      const bool tree = plan.tree;
      // compact level-0 records + the two-launch back-substitution (kernels_rowbcast.hpp), unless the
      // records have to serve a record-based re-solve (KEEP_RECORDS) or the tree schedule runs
      const bool compact = plan.compact;
      c->schedule = tree ? "reduced-tree" : (compact ? (store_l ? "reduced-compact-records" : "reduced") : "reduced-records");
      c->rec_compact = compact && store_l == 2;
      bool fuse2 = false;
      {
        ScopedSlot t(c, SLOT_BOTTOM);
        bool launched = false;
        if constexpr (NX <= 16 && NX + NU <= 16) {
          if (plan.rowbcast) {  // one separator per DPP row, four per wavefront
            hipLaunchKernelGGL((ndlqr::rb_bottom<NX, NU>), dim3(d.N >> 4, d.batch), dim3(64), 0, c->stream, d, c->AB,
                               c->QR, c->rhs, c->red, c->rec, c->info);
            launched = true;
          }
        }
        // Levels 0-2 in one launch (bottom8_reduced_mc: two wavefronts per eight knots, the level-2 slot in LDS): 12 % less
        // HBM traffic and one launch less per step; the level-2 work costs inside the bottom launch about what it costs
        // outside, so the step gains little -- and only at the (12,4) instance, where it is the default (same box,
        // profiles/r04_fuse2_ab.txt: (12,4,256) x 1024 0.587 -> 0.579 ms, (12,4,1024) x 512 1.20 -> 1.16, the padded (11,3)
        // 0.578 -> 0.569; (12,8) +2.5 %, (13,4) +0.9 %, (9,3) / (10,4) / (15,2) +-0). NDLQR_FUSE2=0 / 1 overrides.
        fuse2 = !launched && !tree && compact && !store_l && d.N >= 16 &&
                (c->fuse2 > 0 || (c->fuse2 < 0 && NX == 12 && NU == 4));
        if (fuse2) c->schedule = "reduced-fused2";
        if (launched) {
        } else if (fuse2) {
          hipLaunchKernelGGL((ndlqr::bottom8_reduced_mc<NX, NU>), dim3(d.N >> 3, d.batch), dim3(128), 0, c->stream, d,
                             c->AB, c->QR, c->rhs, c->red, c->rec, c->info);
        } else if (tree)
          hipLaunchKernelGGL((ndlqr::bottom_reduced_mc<NX, NU, true>), dim3(d.N >> 2, d.batch), dim3(64), 0, c->stream,
                             d, c->AB, c->QR, c->rhs, c->red, c->rec, c->F, c->info, store_l, c->tree_cnt, 0);
        else if (compact)
          hipLaunchKernelGGL((ndlqr::bottom_reduced_mc<NX, NU, false, true>), dim3(d.N >> 2, d.batch), dim3(64), 0, c->stream,
                             d, c->AB, c->QR, c->rhs, c->red, c->rec, c->F, c->info, store_l, nullptr, 1);
        else
          hipLaunchKernelGGL((ndlqr::bottom_reduced_mc<NX, NU, false>), dim3(d.N >> 2, d.batch), dim3(64), 0, c->stream,
                             d, c->AB, c->QR, c->rhs, c->red, c->rec, c->F, c->info, store_l, nullptr, 0);
      }
      // upper levels: one launch per level while a level has more than four separators per problem, then the
      // last three levels in one launch (reduced_top_mc; NDLQR_NO_TOP=1: a launch per level to the root)
      const int top_levels = d.K - c->top_levels >= 3 ? c->top_levels : 3;
      const int ltop = (d.K >= 5 && !c->no_top) ? d.K - top_levels : d.K;
      for (int l = fuse2 ? 3 : 2; l < ltop && !tree; ++l) {
        ScopedSlot t(c, SLOT_UPPER);
        hipLaunchKernelGGL((ndlqr::reduced_level_mc<NX, NU>), dim3(d.N >> (l + 1), d.batch), dim3(64), 0, c->stream,
                           d, l, c->AB, c->QR, c->rhs, c->red, c->rec, c->F, c->info, store_l);
      }
      // ... which also runs the top-down sweep over the records of level >= 3 when the back-substitution is the
      // two-launch form and its array fits the workgroup's LDS
      const bool top_sweeps = !tree && ltop < d.K && compact &&
                              sizeof(double) * (size_t)(d.N >> 3) * NX <= 4 * sizeof(ndlqr::ReducedLds<NX, NU, false>);
      if (!tree && ltop < d.K) {
        ScopedSlot t(c, SLOT_TOP);  // (a profile slot of its own: one kernel name per slot, like rocprofv3's per-kernel averages)
        const int l0 = (fuse2 && ltop < 3) ? 3 : ltop;  // (level 2 went with the bottom launch)
        hipLaunchKernelGGL((ndlqr::reduced_top_mc<NX, NU>), dim3(d.batch), dim3(256), 0, c->stream, d, l0, c->AB,
                           c->QR, c->rhs, c->red, c->rec, c->F, c->info, store_l, top_sweeps ? c->ytop : (double*)nullptr);
      }
      ScopedSlot t(c, SLOT_APPLY);
      if (compact) {
        if (!top_sweeps) {
          const size_t top_lds = sizeof(double) * (size_t)(d.N >> 3) * NX;
          if (top_lds > 64 * 1024)  // (beyond the default limit of dynamic LDS: horizons of 8192 knots at 12 states)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ndlqr::rb_backsub_top<NX>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)top_lds);
          hipLaunchKernelGGL((ndlqr::rb_backsub_top<NX>), dim3(d.batch), dim3(256), top_lds, c->stream, d, c->rec, c->ytop);
        }
        // (an MPC step that asked for nothing but a knot range -- NDLQR_SOLN_ONLY -- runs the workgroups of that range)
        hipLaunchKernelGGL((ndlqr::rb_backsub<NX, NU>), dim3(apply_grid(c, d), d.batch), dim3(256), 0, c->stream,
                           apply_dims(c, d), c->AB, c->QR, c->rhs, c->rec, c->ytop, c->z);
      } else {
        hipLaunchKernelGGL((ndlqr::backsub_small<NX, NU>), dim3(apply_grid(c, d), d.batch), dim3(256), 0, c->stream, apply_dims(c, d), c->AB,
                           c->QR, c->rhs, c->rec, c->z);
      }
      return NDLQR_OK;
    }
  }
  c->schedule = lean ? "knot-lean" : (STRICT ? "knot-strict" : "knot-keep");
  {
    ScopedSlot t(c, SLOT_BOTTOM);
    hipLaunchKernelGGL((ndlqr::bottom_small<NX, NU, STRICT, KEEP, JB>), dim3(d.N >> JB, d.batch), dim3(32 << JB), 0,
                       c->stream, d, c->AB, c->QR, c->rhs, c->F, c->z, c->info, c->rec, lean ? 1 : 0,
                       ((lean || (KEEP && !STRICT)) ? 1 : 0) | (store_l ? 2 : 0));
  }
  for (int l = JB; l < d.K; ++l) {  // separator + boundary update of a level in one launch
    ScopedSlot t(c, SLOT_UPPER);
    hipLaunchKernelGGL((ndlqr::level_small<NX, NU, STRICT, KEEP>), dim3(d.N >> (l + 1), d.batch), dim3(64), 0,
                       c->stream, d, l, c->AB, c->F, c->z, c->rec, c->info, store_l);
  }
  ScopedSlot t(c, SLOT_APPLY);
  if (lean) {
    if constexpr (!STRICT && !KEEP)
      hipLaunchKernelGGL((ndlqr::backsub_small<NX, NU>), dim3(apply_grid(c, d), d.batch), dim3(256), 0, c->stream, apply_dims(c, d), c->AB,
                         c->QR, c->rhs, c->rec, c->z);
    return NDLQR_OK;
  }
  const size_t lds = sizeof(double) * (size_t)(d.K - JB) * Sh::REC;
  hipLaunchKernelGGL((ndlqr::apply_small<NX, NU, STRICT, KEEP>), dim3(d.N / Sh::KPB, d.batch), dim3(256), lds,
                     c->stream, d, JB, c->F, c->z, c->rec);
  return NDLQR_OK;
}

// Record-based re-solve (fast mode, specialised sizes): forward pass over the separators, then
// the same back-substitution as the full solve. Returns false when the shape has no instance.
template <int NX, int NU>
static void launch_rhs_records(NdlqrHipCtx* c) {
  const ndlqr::Dims& d = c->d;
  if constexpr (ndlqr::P1OnMatrixCores<NX, NU>::value && 8 * (2 * NX + NU) <= 256) {
    if (c->rec_compact) {
      // the compact records of the default schedule (round 4): forward pass over the separators with the right-hand-side
      // column alone, then the back-substitution of a full solve. c->red (the accumulator slots, idle here) holds what
      // the eight-knot blocks push to the separators between them: [batch][N / 8][2][NX].
      {
        ScopedSlot t(c, SLOT_SEP);
        hipLaunchKernelGGL((ndlqr::rb_forward<NX, NU>), dim3(d.N / 8, d.batch), dim3(256), 0, c->stream, d, c->AB, c->QR,
                           c->rhs, c->rec, c->red);
      }
      {
        ScopedSlot t(c, SLOT_UPPER);
        const size_t lds = sizeof(double) * 4 * (size_t)(d.N >> 3) * NX;
        if (lds > 64 * 1024)
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ndlqr::rb_forward_top<NX, NU>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ndlqr::rb_forward_top<NX, NU>), dim3(d.batch), dim3(256), lds, c->stream, d, c->AB, c->QR,
                           c->rhs, c->rec, (const double*)c->red, c->ytop);
      }
      ScopedSlot t(c, SLOT_APPLY);
      hipLaunchKernelGGL((ndlqr::rb_backsub<NX, NU>), dim3(apply_grid(c, d), d.batch), dim3(256), 0, c->stream,
                         apply_dims(c, d), c->AB, c->QR, c->rhs, c->rec, c->ytop, c->z);
      return;
    }
  }
  {
    ScopedSlot t(c, SLOT_SEP);
    hipLaunchKernelGGL((ndlqr::rhs_forward_small<NX, NU>), dim3(d.N / 8, d.batch), dim3(64), 0, c->stream, d, c->AB,
                       c->QR, c->rhs, c->F, c->rec, c->z);
  }
  if (d.K > 3) {
    ScopedSlot t(c, SLOT_UPPER);
    const size_t lds = sizeof(double) * (size_t)(d.N / 8) * NX;
    hipLaunchKernelGGL((ndlqr::rhs_forward_upper<NX, NU>), dim3(d.batch), dim3(512), lds, c->stream, d, c->AB, c->QR,
                       c->rhs, c->F, c->rec, c->z);
  }
  {
    ScopedSlot t(c, SLOT_APPLY);
    hipLaunchKernelGGL((ndlqr::backsub_small<NX, NU>), dim3(apply_grid(c, d), d.batch), dim3(256), 0, c->stream, apply_dims(c, d), c->AB,
                       c->QR, c->rhs, c->rec, c->z);
  }
}

// Several right-hand sides per problem against the compact records (SURVEY.md 8(f)-2 "multiple right-hand sides";
// ndlqr_hip_solve_multi_rhs): `count` right-hand sides in all, right-hand side j belongs to problem j % batch; rhs / zsep /
// fsum / ytop / z are arrays of `count` entries, the inputs and records those of the context. Returns false when the
// shape has no such form.
template <int NX, int NU>
static bool launch_multi_rhs(NdlqrHipCtx* c, const int count, const double* rhs, double* zsep, double* fsum, double* ytop,
                             double* z) {
  if constexpr (ndlqr::P1OnMatrixCores<NX, NU>::value && 8 * (2 * NX + NU) <= 256) {
    const ndlqr::Dims& d = c->d;
    const size_t lds = sizeof(double) * 4 * (size_t)(d.N >> 3) * NX;
    if (lds > 160 * 1024) return false;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ndlqr::rb_forward_top<NX, NU, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((ndlqr::rb_forward<NX, NU, true>), dim3(d.N / 8, count), dim3(256), 0, c->stream, d, c->AB, c->QR, rhs,
                       c->rec, fsum, d.batch, zsep);
    hipLaunchKernelGGL((ndlqr::rb_forward_top<NX, NU, true>), dim3(count), dim3(256), lds, c->stream, d, c->AB, c->QR, rhs,
                       c->rec, (const double*)fsum, ytop, d.batch, zsep);
    // (a knot range alone: ndlqr_hip_solve_multi_rhs_slices)
    hipLaunchKernelGGL((ndlqr::rb_backsub<NX, NU, true>), dim3(apply_grid(c, d), count), dim3(256), 0, c->stream,
                       apply_dims(c, d), c->AB, c->QR, rhs, (const double*)c->rec, (const double*)ytop, z, d.batch,
                       (const double*)zsep);
    return true;
  } else {
    return false;
  }
}

// Time-axis sharding of the separator-only schedule: the horizon is cut into G
// chunks of N / G knots, rank g works on chunk g. The tree levels 0 .. K - log2(G) - 1 lie inside a chunk; what a
// chunk exposes to the rest of the tree is what any subtree exposes: the blocks it adds to the slots of the G - 1
// separators between the chunks (and the couplings between those). Phase 0 (here): bottom kernel and level launches
// restricted to the chunk (Dims::xoff). Between the phases the caller sums the top slots over the ranks (one
// all-reduce of (G - 1) slots per problem: every rank contributes the halves its chunk wrote, zeros elsewhere).
// Phase 1: the top log2(G) levels -- G - 1 separators, eliminated REDUNDANTLY by every rank, which saves sending
// multipliers back --, the top-down sweep, and the back-substitution of the chunk's knots.
// Inputs are resident for the whole horizon on every rank (this prototype shards the work, not the storage).
template <int NX, int NU>
static int launch_time_shard(NdlqrHipCtx* c, const int phase, const int g, const int G) {
  ndlqr::Dims d = c->d;
  int lg = 0;
  while ((1 << lg) < G) ++lg;
  if (G < 2 || (1 << lg) != G || g < 0 || g >= G) return NDLQR_ERR_INVALID;
  if constexpr (!ndlqr::P1OnMatrixCores<NX, NU>::value) {
    return NDLQR_ERR_INVALID;
  } else {
    const int ltop = d.K - lg;  // levels [0, ltop) lie inside a chunk
    if (!c->red || !c->ytop || ltop < 4 || 8 * (2 * NX + NU) > 256 || (c->flags & ~NDLQR_FLAG_PROFILE)) return NDLQR_ERR_INVALID;
    const size_t top_lds = sizeof(double) * (size_t)(d.N >> 3) * NX;
    if (top_lds > 160 * 1024) return NDLQR_ERR_INVALID;  // (rb_backsub_top's sweep array has to fit one workgroup's LDS)
    if (top_lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ndlqr::rb_backsub_top<NX>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)top_lds);
    constexpr size_t SLOT = ndlqr::RedSlot<NX>::SIZE;
    if (phase == 0) {
      // whatever an earlier exchange left in the top slots goes: this rank's chunk writes its halves afresh
      for (int j = 1; j < G; ++j) {
        const int s = j * (d.N / G) - 1;
        double* p = c->red + (size_t)(s >> 2) * SLOT;
        if (hipMemset2DAsync(p, sizeof(double) * (size_t)(d.N >> 2) * SLOT, 0, sizeof(double) * SLOT, (size_t)d.batch,
                             c->stream) != hipSuccess)
          return NDLQR_ERR_NO_DEVICE;
      }
      {
        ScopedSlot t(c, SLOT_BOTTOM);
        const int cnt = (d.N >> 2) / G;
        d.xoff = g * cnt;
        hipLaunchKernelGGL((ndlqr::bottom_reduced_mc<NX, NU, false, true>), dim3(cnt, d.batch), dim3(64), 0, c->stream, d,
                           c->AB, c->QR, c->rhs, c->red, c->rec, c->F, c->info, 0, nullptr, 1);
      }
      for (int l = 2; l < ltop; ++l) {
        ScopedSlot t(c, SLOT_UPPER);
        const int cnt = (d.N >> (l + 1)) / G;
        d.xoff = g * cnt;
        hipLaunchKernelGGL((ndlqr::reduced_level_mc<NX, NU>), dim3(cnt, d.batch), dim3(64), 0, c->stream, d, l, c->AB,
                           c->QR, c->rhs, c->red, c->rec, c->F, c->info, 0);
      }
    } else {
      d.xoff = 0;
      for (int l = ltop; l < d.K; ++l) {
        ScopedSlot t(c, SLOT_TOP);
        hipLaunchKernelGGL((ndlqr::reduced_level_mc<NX, NU>), dim3(d.N >> (l + 1), d.batch), dim3(64), 0, c->stream, d,
                           l, c->AB, c->QR, c->rhs, c->red, c->rec, c->F, c->info, 0);
      }
      ScopedSlot t(c, SLOT_APPLY);
      // (the sweep runs over every separator of level >= 3; those of other chunks have no records here and resolve to
      //  garbage nobody reads: a separator depends on its ancestors only, which are in this chunk or among the top ones)
      hipLaunchKernelGGL((ndlqr::rb_backsub_top<NX>), dim3(d.batch), dim3(256), top_lds, c->stream, d, c->rec, c->ytop);
      const int cnt = (d.N >> 3) / G;
      d.xoff = g * cnt;
      hipLaunchKernelGGL((ndlqr::rb_backsub<NX, NU>), dim3(cnt, d.batch), dim3(256), 0, c->stream, d, c->AB, c->QR,
                         c->rhs, c->rec, c->ytop, c->z);
    }
    c->schedule = "reduced-time-shard";
    return NDLQR_OK;
  }
}
