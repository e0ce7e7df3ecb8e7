// launch_small.hpp -- internal: launch sequences of the size-specialised kernels, instantiated
// once per (nstates, ninputs) in its own translation unit (small_instance.hip) so that the
// instances compile in parallel.
#pragma once
#include "hip_context.hpp"
#include "kernels_leaf.hpp"
#include "kernels_small.hpp"
#include "kernels_bottom_reduced.hpp"

// Size-specialised launch sequence. Levels below J: one separator + one Schur launch each.
// Levels J..K-1 ("boundary-first"): separator + Schur on the two boundary knots of every subtree
// (tiny grids), then ONE apply_small pass that takes every knot through all those levels in
// registers. J = K disables the second form (pure level-by-level streaming).
template <int NX, int NU, bool STRICT, bool KEEP, int JB, bool REDUCED = false>
static void launch_bottom(NdlqrHipCtx* c, bool lean) {
  const ndlqr::Dims& d = c->d;
  ScopedSlot t(c, SLOT_BOTTOM);
  const size_t pad = (size_t)c->bottom_lds_pad;  // occupancy experiments (NDLQR_BOTTOM_LDS_PAD)
  if constexpr (REDUCED) {
    hipLaunchKernelGGL((ndlqr::bottom_small<NX, NU, STRICT, KEEP, JB, true>), dim3(d.N >> JB, d.batch),
                       dim3(32 << JB), pad, c->stream, d, c->AB, c->QR, c->rhs, c->F, c->z, c->info, c->rec, 1,
                       1 | ((c->flags & NDLQR_FLAG_KEEP_RECORDS) ? 2 : 0), c->red);
    return;
  }
  hipLaunchKernelGGL((ndlqr::bottom_small<NX, NU, STRICT, KEEP, JB>), dim3(d.N >> JB, d.batch), dim3(32 << JB), pad,
                     c->stream, d, c->AB, c->QR, c->rhs, c->F, c->z, c->info, c->rec, lean ? 1 : 0,
                     ((lean || (KEEP && !STRICT)) ? 1 : 0) | ((c->flags & NDLQR_FLAG_KEEP_RECORDS) ? 2 : 0));
}

// What launch_small is going to do for this context: decided once, before the launch sequence is
// enqueued (and possibly captured), so that ndlqr_hip.hip can allocate what the schedule needs.
struct SmallPlan {
  int JB;        // tree levels fused with the leaf phase in the bottom kernel
  bool lean;     // solution by back-substitution from the separator records (fast mode, no KEEP)
  int store_l;   // keep the separator factors for a record-based re-solve (KEEP_RECORDS)
  bool reduced;  // separator-only schedule (bottom_reduced_mc + reduced_level_mc)
  bool tree;     // ... with the whole factorisation in one launch (small batches)
  bool needs_F;  // the schedule reads or writes the factor array
};

template <int NX, int NU, bool STRICT, bool KEEP>
static SmallPlan plan_small(const NdlqrHipCtx* c, int J) {
  const ndlqr::Dims& d = c->d;
  SmallPlan p;
  // leaf + levels 0..JB-1 fused on chip when the horizon is long enough, else the leaf kernel
  int JB = c->bottom_levels;
  if (JB > 3) JB = 3;
  while (JB > 0 && d.K <= JB) --JB;
  if (JB > J) JB = J;
  p.JB = JB;
  // fast mode without KEEP: solution by back-substitution from the separator records (needs the
  // boundary-first schedule right after the bottom kernel, so that no level reads interior knots)
  p.lean = !STRICT && !KEEP && JB >= J && JB >= 1 && JB < d.K && c->upper_mode != 0 &&
           (d.K + 4) * NX <= 256 && !c->no_backsub;
  p.store_l = (c->flags & NDLQR_FLAG_KEEP_RECORDS) ? 1 : 0;  // factors for a record-based re-solve
  p.reduced = false;
  p.tree = false;
  // separator-only schedule of the upper levels (see reduced_level): the bottom kernel pushes
  // 12x12 blocks instead of handing knot rows over
  if constexpr (!STRICT && !KEEP && ndlqr::P1OnMatrixCores<NX, NU>::value) {
    if (p.lean && c->reduced && JB == 2 && d.K > 2 && c->red) {
      p.reduced = true;
      // tree schedule for small batches (at most half a resident round of bottom wavefronts): three
      // launches instead of K + 1; measured cross-over at batch x N / 4 ~ 4096 wavefronts
      p.tree = c->bottom_reduced && c->mcore && c->tree_cnt &&
               (c->tree == 1 || (c->tree < 0 && (size_t)d.batch * (d.N >> 2) <= 2048));
    }
  }
  // the separator-only schedule touches F only to park the factors of KEEP_RECORDS
  p.needs_F = !(p.reduced && !p.store_l);
  return p;
}

template <int NX, int NU, bool STRICT, bool KEEP>
static int launch_small(NdlqrHipCtx* c, int J) {
  const ndlqr::Dims& d = c->d;
  using Sh = ndlqr::SchurShape<NX, NU>;
  const SmallPlan plan = plan_small<NX, NU, STRICT, KEEP>(c, J);
  const int JB = plan.JB;
  const bool lean = plan.lean;
  const int store_l = plan.store_l;
  // the record-based re-solve needs every separator's record and factor: KEEP writes them all,
  // KEEP_RECORDS adds the factors to the lean schedule
  c->rec_complete = !STRICT && (KEEP || (lean && store_l));
  if constexpr (!STRICT && !KEEP && ndlqr::P1OnMatrixCores<NX, NU>::value) {
    if (plan.reduced) {
      const bool tree = plan.tree;
      c->schedule = tree ? "reduced-tree" : "reduced";
      if (tree) {
        // arrival counters start from zero in every solve: a launch that did not run to completion
        // (error mid-graph, aborted stream) cannot leave odd counters behind for the next one
        HIP_TRY(hipMemsetAsync(c->tree_cnt, 0, sizeof(int) * (size_t)d.batch * (d.N >> 2), c->stream));
      }
      if (c->bottom_reduced) {
        ScopedSlot t(c, SLOT_BOTTOM);
        if (tree)
          hipLaunchKernelGGL((ndlqr::bottom_reduced_mc<NX, NU, true>), dim3(d.N >> 2, d.batch), dim3(64),
                             (size_t)c->bottom_lds_pad, c->stream, d, c->AB, c->QR, c->rhs, c->red, c->rec, c->F,
                             c->info, store_l, c->tree_cnt);
        else if (c->mcore)
          hipLaunchKernelGGL((ndlqr::bottom_reduced_mc<NX, NU, false>), dim3(d.N >> 2, d.batch), dim3(64),
                             (size_t)c->bottom_lds_pad, c->stream, d, c->AB, c->QR, c->rhs, c->red, c->rec, c->F,
                             c->info, store_l, nullptr);
        else
          hipLaunchKernelGGL((ndlqr::bottom_reduced<NX, NU>), dim3(d.N >> 2, d.batch), dim3(64), 0, c->stream, d,
                             c->AB, c->QR, c->rhs, c->red, c->rec, c->F, c->info, store_l);
      } else {
        launch_bottom<NX, NU, STRICT, KEEP, 2, true>(c, true);
      }
      for (int l = 2; l < d.K && !tree; ++l) {
        ScopedSlot t(c, SLOT_UPPER);
        if (c->mcore)
          hipLaunchKernelGGL((ndlqr::reduced_level_mc<NX, NU>), dim3(d.N >> (l + 1), d.batch), dim3(64), 0, c->stream,
                             d, l, c->AB, c->QR, c->rhs, c->red, c->rec, c->F, c->info, store_l);
        else
          hipLaunchKernelGGL((ndlqr::reduced_level<NX, NU>), dim3(d.N >> (l + 1), d.batch), dim3(64), 0, c->stream, d,
                             l, c->AB, c->QR, c->rhs, c->red, c->rec, c->F, c->info, store_l);
      }
      ScopedSlot t(c, SLOT_APPLY);
      hipLaunchKernelGGL((ndlqr::backsub_small<NX, NU>), dim3(d.N / 8, d.batch), dim3(256), 0, c->stream, d, c->AB,
                         c->QR, c->rhs, c->rec, c->z);
      return NDLQR_OK;
    }
  }
  c->schedule = lean ? "knot-lean" : (STRICT ? "knot-strict" : "knot-keep");
  switch (JB) {
    case 3: launch_bottom<NX, NU, STRICT, KEEP, 3>(c, lean); break;
    case 2: launch_bottom<NX, NU, STRICT, KEEP, 2>(c, lean); break;
    case 1: launch_bottom<NX, NU, STRICT, KEEP, 1>(c, lean); break;
    default: {
      ScopedSlot t(c, SLOT_LEAF);
      hipLaunchKernelGGL((ndlqr::leaf_generic<STRICT>), dim3(d.N, d.batch), dim3(128), 0, c->stream, d,
                         c->AB, c->QR, c->rhs, c->F, c->z, c->info);
    }
  }
  if (JB >= J && JB >= 1 && JB < d.K && c->upper_mode == 1) {
    // no full-level Schur pass left: separator + boundary update of a level in one launch
    for (int l = JB; l < d.K; ++l) {
      ScopedSlot t(c, SLOT_UPPER);
      hipLaunchKernelGGL((ndlqr::level_small<NX, NU, STRICT, KEEP>), dim3(d.N >> (l + 1), d.batch), dim3(64), 0,
                         c->stream, d, l, c->AB, c->F, c->z, c->rec, c->info, store_l);
    }
  } else if (JB >= J && JB >= 1 && JB < d.K && c->upper_mode == 2) {
    // no full-level Schur pass left: all remaining levels of a problem in one launch
    ScopedSlot t(c, SLOT_UPPER);
    auto kern = ndlqr::upper_small<NX, NU, STRICT, KEEP>;
    int nw = 8;
    while (nw > 1 && nw / 2 >= (d.N >> (JB + 1))) nw /= 2;  // not more wavefronts than subtrees
    const size_t lds = (size_t)nw * (sizeof(ndlqr::SepIn<NX, NU>) + sizeof(ndlqr::SepOut<NX>));
    if (lds > 64 * 1024 && c->big_lds_kernel != reinterpret_cast<const void*>(kern)) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
      c->big_lds_kernel = reinterpret_cast<const void*>(kern);
    }
    hipLaunchKernelGGL(kern, dim3(d.batch), dim3(64 * nw), lds, c->stream, d, JB, c->AB, c->F, c->z, c->rec,
                       c->info, store_l);
  } else
  for (int l = JB; l < d.K; ++l) {
    {
      ScopedSlot t(c, SLOT_SEP);
      const int nsep = d.N >> (l + 1);
      hipLaunchKernelGGL((ndlqr::separator_one<NX, NU, STRICT, KEEP>), dim3(nsep, d.batch), dim3(64), 0, c->stream,
                         d, l, c->AB, c->F, c->z, c->rec, c->info);
    }
    if (l < J) {
      ScopedSlot t(c, SLOT_SCHUR);
      hipLaunchKernelGGL((ndlqr::schur_small<NX, NU, STRICT, false>), dim3(d.N / Sh::KPB, d.batch), dim3(256),
                         0, c->stream, d, l, c->F, c->z, c->rec);
    } else if (l < d.K - 1) {
      ScopedSlot t(c, SLOT_BOUNDARY);
      const int nsub = d.N >> (l + 1);
      hipLaunchKernelGGL((ndlqr::schur_small<NX, NU, STRICT, true>),
                         dim3((nsub + Sh::WAVES - 1) / Sh::WAVES, d.batch), dim3(256), 0, c->stream, d, l,
                         c->F, c->z, c->rec);
    }
  }
  if (lean) {
    ScopedSlot t(c, SLOT_APPLY);
    if constexpr (!STRICT && !KEEP)
      hipLaunchKernelGGL((ndlqr::backsub_small<NX, NU>), dim3(d.N / 8, d.batch), dim3(256), 0, c->stream, d, c->AB,
                         c->QR, c->rhs, c->rec, c->z);
    return NDLQR_OK;
  }
  if (J < d.K) {
    ScopedSlot t(c, SLOT_APPLY);
    if constexpr (!STRICT && !KEEP) {
      // only the solution is wanted: two dot products per knot row against the top-down vectors w
      if (J >= 2 && !c->no_finish) {
        const size_t lds = sizeof(double) * (size_t)(d.K - J) * (Sh::REC + 2 * 2 * NX);
        hipLaunchKernelGGL((ndlqr::finish_small<NX, NU>), dim3(d.N / Sh::KPB, d.batch), dim3(256), lds, c->stream,
                           d, J, c->F, c->z, c->rec);
        return NDLQR_OK;
      }
    }
    const size_t lds = sizeof(double) * (size_t)(d.K - J) * Sh::REC;
    hipLaunchKernelGGL((ndlqr::apply_small<NX, NU, STRICT, KEEP>), dim3(d.N / Sh::KPB, d.batch), dim3(256), lds,
                       c->stream, d, J, c->F, c->z, c->rec);
  }
  return NDLQR_OK;
}

// Record-based re-solve (fast mode, specialised sizes): forward pass over the separators, then
// the same back-substitution as the full solve. Returns false when the shape has no instance.
template <int NX, int NU>
static void launch_rhs_records(NdlqrHipCtx* c) {
  const ndlqr::Dims& d = c->d;
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
    hipLaunchKernelGGL((ndlqr::backsub_small<NX, NU>), dim3(d.N / 8, d.batch), dim3(256), 0, c->stream, d, c->AB,
                       c->QR, c->rhs, c->rec, c->z);
  }
}
