// small_instance.hip -- one size-specialised instance of the kernels in kernels_small.hpp.
// Compiled once per line of small_instances.def with -DNDLQR_INST_NX=<nstates>
// -DNDLQR_INST_NU=<ninputs>; exports the two entry points ndlqr_hip.hip dispatches to:
//   ndlqr_small_solve_<nx>_<nu>(ctx, strict, keep)   factor + solve launch sequence
//   ndlqr_small_needs_F_<nx>_<nu>(ctx, strict, keep)  does that sequence touch the factor array?
//   ndlqr_small_rhs_<nx>_<nu>(ctx)                      record-based right-hand-side re-solve
//   ndlqr_small_kpb_<nx>_<nu>()                         knots per workgroup of its Schur kernels
//   ndlqr_small_tshard_<nx>_<nu>(ctx, phase, g, G)      time-axis sharding: chunk g of G, phase 0 / 1 (launch_time_shard)
//   ndlqr_small_slot_<nx>_<nu>()                        doubles per accumulator slot
//   ndlqr_small_multi_<nx>_<nu>(ctx, count, rhs, zsep, fsum, ytop, z)   several right-hand sides per problem (launch_multi_rhs)
#include "launch_small.hpp"

#if !defined(NDLQR_INST_NX) || !defined(NDLQR_INST_NU)
#error "compile with -DNDLQR_INST_NX=... -DNDLQR_INST_NU=..."
#endif

#define NDLQR_PASTE_(prefix, a, b) prefix##a##_##b
#define NDLQR_PASTE(prefix, a, b) NDLQR_PASTE_(prefix, a, b)
#define NDLQR_INST_NAME(prefix) NDLQR_PASTE(prefix, NDLQR_INST_NX, NDLQR_INST_NU)

namespace {
constexpr int NX = NDLQR_INST_NX, NU = NDLQR_INST_NU;
}

int NDLQR_INST_NAME(ndlqr_small_solve_)(NdlqrHipCtx* c, bool strict, bool keep) {
  if (strict) return keep ? launch_small<NX, NU, true, true>(c) : launch_small<NX, NU, true, false>(c);
  return keep ? launch_small<NX, NU, false, true>(c) : launch_small<NX, NU, false, false>(c);
}

int NDLQR_INST_NAME(ndlqr_small_needs_F_)(const NdlqrHipCtx* c, bool strict, bool keep) {
  if (strict) return keep ? plan_small<NX, NU, true, true>(c).needs_F : plan_small<NX, NU, true, false>(c).needs_F;
  return keep ? plan_small<NX, NU, false, true>(c).needs_F : plan_small<NX, NU, false, false>(c).needs_F;
}

void NDLQR_INST_NAME(ndlqr_small_rhs_)(NdlqrHipCtx* c) { launch_rhs_records<NX, NU>(c); }

int NDLQR_INST_NAME(ndlqr_small_kpb_)(void) { return ndlqr::SchurShape<NX, NU>::KPB; }

int NDLQR_INST_NAME(ndlqr_small_tshard_)(NdlqrHipCtx* c, int phase, int g, int G) {
  return launch_time_shard<NX, NU>(c, phase, g, G);
}

int NDLQR_INST_NAME(ndlqr_small_slot_)(void) { return (int)ndlqr::RedSlot<NX>::SIZE; }

int NDLQR_INST_NAME(ndlqr_small_multi_)(NdlqrHipCtx* c, int count, const double* rhs, double* zsep, double* fsum,
                                        double* ytop, double* z) {
  return launch_multi_rhs<NX, NU>(c, count, rhs, zsep, fsum, ytop, z) ? 1 : 0;
}
