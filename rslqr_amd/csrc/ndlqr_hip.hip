// ndlqr_hip.hip -- gfx950 (MI355X, CDNA4) implementation of the device boundary in
// include/ndlqr_hip.h: context/memory management, launch sequence, D2H converters and the
// dense Matrix* helpers. Kernels live in kernels_generic.hpp (any n, m) and kernels_small.hpp
// (size-specialised, one knot row per lane).
//
// Written for wave64 / gfx950 only; built with -ffp-contract=off so that every fused
// multiply-add in the kernels is an explicit fma() (fast mode) or an explicit mul + add
// (NDLQR_FLAG_STRICT_FP, which reproduces the reference's default CPU build bit for bit).
#include <mutex>
#include <utility>

#include "hip_context.hpp"
#include "kernels_generic.hpp"
#include "kernels_mfma.hpp"
#include "kernels_reduced_mfma.hpp"
#ifdef NDLQR_SINGLE_TU  // developer builds (tools/segtime.py): every instance in this translation unit
#include "launch_small.hpp"
#endif

// ------------------------------------------------------------------------------ errors

static thread_local std::string g_last_error = "";

const char* ndlqr_hip_last_error(void) { return g_last_error.c_str(); }

int ndlqr_hip_fail(const char* what, hipError_t e) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  fprintf(stderr, "ndlqr_hip: %s\n", g_last_error.c_str());
  // a launch configuration / argument the device rejects is the caller's (or this library's) error,
  // not a missing device
  if (e == hipErrorInvalidConfiguration || e == hipErrorInvalidValue) return NDLQR_ERR_INVALID;
  return NDLQR_ERR_NO_DEVICE;
}
static int fail(const char* what, hipError_t e) { return ndlqr_hip_fail(what, e); }

int ndlqr_hip_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

static const char* kSlotNames[SLOT_COUNT] = {"leaf", "separator", "schur", "schur_boundary", "apply", "bottom", "upper", "top"};

static bool has_small_instance(int nstates, int ninputs);  // defined with the instance table below
static void pick_pad_instance(int nstates, int ninputs, int* pn, int* pm);

static size_t bytes_AB(const ndlqr::Dims& d) { return sizeof(double) * (size_t)d.batch * d.N * d.n * d.w; }
static size_t bytes_QR(const ndlqr::Dims& d) { return sizeof(double) * (size_t)d.batch * d.N * d.w; }
static size_t bytes_z(const ndlqr::Dims& d) { return sizeof(double) * (size_t)d.batch * d.N * d.rows; }
static size_t bytes_rec(const ndlqr::Dims& d) { return sizeof(double) * (size_t)d.batch * d.N * (2 * d.n * d.n + d.n); }
static size_t bytes_F(const ndlqr::Dims& d) { return sizeof(double) * (size_t)d.batch * d.K * d.N * d.fb; }

NdlqrHipCtx* ndlqr_hip_create(int nstates, int ninputs, int nhorizon, int batch, int device) {
  return ndlqr_hip_create_ex(nstates, ninputs, nhorizon, batch, device, 0u);
}

NdlqrHipCtx* ndlqr_hip_create_ex(int nstates, int ninputs, int nhorizon, int batch, int device, unsigned create_flags) {
  if (nstates <= 0 || ninputs <= 0 || batch <= 0 || nhorizon < 2 || (nhorizon & (nhorizon - 1))) {
    g_last_error = "invalid dimensions";
    return nullptr;
  }
  if (batch > 65535) {  // the batch index rides on gridDim.y
    g_last_error = "batch > 65535 problems per solver: split the batch over several solvers";
    fprintf(stderr, "ndlqr_hip: %s\n", g_last_error.c_str());
    return nullptr;
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_last_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    return nullptr;
  }
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  if (device >= count) {
    g_last_error = "device index out of range";
    return nullptr;
  }
  if ((e = hipSetDevice(device)) != hipSuccess) { fail("hipSetDevice", e); return nullptr; }

  NdlqrHipCtx* c = new NdlqrHipCtx();
  ndlqr::Dims& d = c->d;
  auto set_dims = [&](ndlqr::Dims& x, int n_, int m_) {
    x.n = n_; x.m = m_; x.N = nhorizon; x.batch = batch;
    x.K = 0; while ((1 << x.K) < nhorizon) ++x.K;
    x.rows = 2 * n_ + m_; x.w = n_ + m_; x.fb = x.rows * n_; x.xoff = 0;
  };
  set_dims(c->du, nstates, ninputs);
  // Padded shapes: a block size without a size-specialised instance runs zero-padded inside the cheapest instance
  // that contains it (dummy states and inputs with unit weights and no coupling: they solve to exactly zero and the
  // real variables see the same arithmetic plus exact zeros) instead of the runtime-sized kernels, which are 3-4x
  // slower below 16 states. NDLQR_NO_PAD=1 keeps the caller's block size (A/B, tests).
  int pn = nstates, pm = ninputs;
  if (!has_small_instance(nstates, ninputs) && nhorizon >= 8 && !getenv("NDLQR_NO_PAD") &&
      !(create_flags & NDLQR_CREATE_NO_PAD))
    pick_pad_instance(nstates, ninputs, &pn, &pm);
  // ... and beyond 128 states (the knot-based kernels: launch_generic) a block that does not fill 16 x 16 tiles is padded
  // to the next one that does: separator_mfma instead of separator_generic, 3-7x faster there (round 4)
  if (nstates > 128 && (nstates % 16 != 0 || (nstates + ninputs) % 4 != 0) && !getenv("NDLQR_NO_PAD") &&
      !(create_flags & NDLQR_CREATE_NO_PAD)) {
    pn = (nstates + 15) / 16 * 16;
    pm = ninputs + (4 - (pn + ninputs) % 4) % 4;
  }
  set_dims(d, pn, pm);
  c->padded = pn != nstates || pm != ninputs;
  c->pad_stage = nullptr; c->pad_stage_cap = 0;
  c->device = device; c->flags = 0; c->stream = nullptr; c->own_stream = true;
  c->red_bytes = 0;
  c->AB = c->QR = c->rhs = c->F = c->z = c->rec = c->red = nullptr; c->info = nullptr; c->tree_cnt = nullptr;
  c->pipeline = getenv("NDLQR_PIPELINE") ? atoi(getenv("NDLQR_PIPELINE")) : 2;
  c->solve_count = 0; c->in_alt = false; c->z_latest = nullptr; c->stream_latest = nullptr; c->h_fail_other = nullptr;
  c->state_dirty = false; c->fail_base = 0; c->ytop = nullptr; c->schedule = "none"; c->kkt_out = nullptr; c->xfer = nullptr; c->h_stage[0] = c->h_stage[1] = nullptr; c->ev_inputs = nullptr; c->ev_step[0] = c->ev_step[1] = nullptr; c->step_count = 0; c->h_fail = nullptr; c->rec_complete = false; c->graph_rec_complete = false; c->graph_schedule = "none"; c->rec_compact = false; c->graph_rec_compact = false;

  c->tree = getenv("NDLQR_TREE") ? (atoi(getenv("NDLQR_TREE")) != 0 ? 1 : 0) : -1;  // -1: by batch size
  c->rowbcast = getenv("NDLQR_ROWBCAST") ? (atoi(getenv("NDLQR_ROWBCAST")) != 0 ? 1 : 0) : -1;  // -1: by block size
  c->fuse2 = getenv("NDLQR_FUSE2") ? (atoi(getenv("NDLQR_FUSE2")) != 0 ? 1 : 0) : -1;  // -1: by instance (launch_small)
  c->no_mfma = getenv("NDLQR_NO_MFMA") != nullptr;
  c->no_top = getenv("NDLQR_NO_TOP") != nullptr;
  c->top_levels = getenv("NDLQR_TOP_LEVELS") ? atoi(getenv("NDLQR_TOP_LEVELS")) : 3;
  if (c->top_levels < 3 || c->top_levels > 5) c->top_levels = 3;
  c->sep_threads = getenv("NDLQR_SEP_THREADS") ? atoi(getenv("NDLQR_SEP_THREADS")) : 0;
  c->timing_pending = false; c->last_ms = 0; c->last_failures = 0; c->fact_valid = false;
  memset(c->rhs_latest, 0, sizeof(c->rhs_latest)); memset(c->rhs_gen, 0, sizeof(c->rhs_gen));
  c->sel_knot0 = 0; c->sel_nknots = 0; c->sel_blocks = 7u; c->step_set[0] = c->step_set[1] = 0;
  c->apply_blk0 = c->apply_nblk = 0; c->graph_apply = 0; c->z_partial = false; c->z_blk0 = c->z_nblk = 0;
  c->h_io = nullptr; c->graph_staged = nullptr; c->graph_staged_flags = 0;
  c->graph_exec = nullptr; c->graph_flags = 0; c->graph_stream = nullptr;
  c->sep_scratch = nullptr;
  c->multi_cap = 0;
  c->multi_rhs = c->multi_z = c->multi_zsep = c->multi_fsum = c->multi_ytop = c->multi_in = c->multi_out = nullptr;
  memset(c->slot_ms, 0, sizeof(c->slot_ms));
  memset(c->slot_launches, 0, sizeof(c->slot_launches));
  bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreate(&c->ev_start) == hipSuccess && hipEventCreate(&c->ev_stop) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_inputs, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_step[0], hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_step[1], hipEventDisableTiming) == hipSuccess &&
            hipMalloc(&c->AB, bytes_AB(d)) == hipSuccess && hipMalloc(&c->QR, bytes_QR(d)) == hipSuccess &&
            hipMalloc(&c->rhs, bytes_z(d)) == hipSuccess && hipMalloc(&c->z, bytes_z(d)) == hipSuccess &&
            hipMalloc(&c->rec, bytes_rec(d)) == hipSuccess &&
            hipMalloc(&c->info, sizeof(int) * ((size_t)batch + 1)) == hipSuccess &&
            hipHostMalloc((void**)&c->h_fail, sizeof(int), hipHostMallocDefault) == hipSuccess;
  if (ok) *c->h_fail = 0;
  if (ok && nhorizon >= 8 && has_small_instance(d.n, d.m)) {
    // slot = DL | DR (packed lower triangles) | CA | CB | gL | gR, padded to whole 128-byte lines (RedSlot<NX>::SIZE)
    const size_t slot_doubles = ((size_t)d.n * (d.n + 1) + 2 * (size_t)d.n * d.n + 2 * d.n + 15) / 16 * 16;
    const size_t red_bytes = sizeof(double) * (size_t)batch * (nhorizon / 4) * slot_doubles;
    ok = hipMalloc(&c->red, red_bytes) == hipSuccess && hipMemsetAsync(c->red, 0, red_bytes, c->stream) == hipSuccess;
    if (ok) c->red_bytes = red_bytes;
    ok = ok && hipMalloc(&c->ytop, sizeof(double) * (size_t)batch * (nhorizon / 8) * d.n) == hipSuccess;
    const size_t cnt_bytes = sizeof(int) * (size_t)batch * (nhorizon / 4);
    ok = ok && hipMalloc(&c->tree_cnt, cnt_bytes) == hipSuccess &&
         hipMemsetAsync(c->tree_cnt, 0, cnt_bytes, c->stream) == hipSuccess;
  }
  if (ok && c->padded) {
    hipLaunchKernelGGL(ndlqr::pad_fill_generic, dim3(d.N, d.batch), dim3(128), 0, c->stream, d, c->AB, c->QR, c->rhs);
    ok = hipGetLastError() == hipSuccess;
  }
  if (ok) {
    // (the factor array F is allocated by the first solve whose schedule touches it: ndlqr_hip_ensure_F)
    ok = hipMemsetAsync(c->z, 0, bytes_z(d), c->stream) == hipSuccess &&
         hipMemsetAsync(c->info, 0, sizeof(int) * ((size_t)batch + 1), c->stream) == hipSuccess &&
         hipStreamSynchronize(c->stream) == hipSuccess;
  }
  if (!ok) {
    fail("device allocation", hipGetLastError());
    ndlqr_hip_destroy(c);
    return nullptr;
  }
  return c;
}

static void free_alt(NdlqrHipCtx* c);  // two-deep solve pipeline, below

void ndlqr_hip_destroy(NdlqrHipCtx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  free_alt(c);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
  if (c->graph_staged) (void)hipGraphExecDestroy(c->graph_staged);
  if (c->h_io) (void)hipHostFree(c->h_io);
  for (auto& p : c->pending) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
  for (auto& ev : c->event_pool) (void)hipEventDestroy(ev);
  (void)hipFree(c->AB); (void)hipFree(c->QR); (void)hipFree(c->rhs); (void)hipFree(c->F);
  (void)hipFree(c->z); (void)hipFree(c->rec); (void)hipFree(c->red); (void)hipFree(c->tree_cnt); (void)hipFree(c->info);
  (void)hipFree(c->sep_scratch);
  (void)hipFree(c->multi_rhs); (void)hipFree(c->multi_z); (void)hipFree(c->multi_zsep); (void)hipFree(c->multi_fsum);
  (void)hipFree(c->multi_ytop); (void)hipFree(c->multi_in); (void)hipFree(c->multi_out);
  (void)hipFree(c->kkt_out); (void)hipFree(c->ytop); (void)hipFree(c->xfer); (void)hipFree(c->pad_stage);
  for (double* h : c->h_stage) if (h) (void)hipHostFree(h);
  if (c->ev_inputs) (void)hipEventDestroy(c->ev_inputs);
  for (hipEvent_t ev : c->ev_step) if (ev) (void)hipEventDestroy(ev);
  if (c->h_fail) (void)hipHostFree(c->h_fail);
  if (c->ev_start) (void)hipEventDestroy(c->ev_start);
  if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// ------------------------------------------------------------------------------ two-deep solve pipeline

// the current buffer set holds the most recent solution -- all of it, or (a step with NDLQR_SOLN_ONLY) the knots of the
// workgroups its back-substitution ran
static void note_solution(NdlqrHipCtx* c) {
  c->z_latest = c->z;
  c->stream_latest = c->stream;
  c->z_partial = c->apply_nblk > 0;
  c->z_blk0 = c->apply_blk0;
  c->z_nblk = c->apply_nblk;
}
// Everything is idle (the caller has just waited for both streams) and the current buffer set has received a new right-hand
// side: let the next pipelined solve start on THIS set instead of the other one, whose copy of the right-hand side would
// have to be brought up to date first (copy_rhs_parts_generic: 62 us per 1024 x (12,4,256) in a loop that replaces the
// problem every iteration).
static void next_solve_on_current_set(NdlqrHipCtx* c) {
  if (((c->solve_count & 1u) != 0) != c->in_alt) ++c->solve_count;
}
// consumers of the whole solution vector refuse a slice
static int need_full_solution(const NdlqrHipCtx* c, const char* who) {
  if (!c->z_partial) return NDLQR_OK;
  g_last_error = std::string(who) + ": the last step computed only knots " + std::to_string(8 * c->z_blk0) + " .. " +
                 std::to_string(8 * (c->z_blk0 + c->z_nblk) - 1) + " (NDLQR_SOLN_ONLY); run a solve or a step without it first";
  return NDLQR_ERR_INVALID;
}

// exchange the context's per-solve buffers, stream, graph and events with the alternate set
static void swap_slot(NdlqrHipCtx* c) {
  NdlqrAltSlot& a = c->alt;
  std::swap(c->rec, a.rec); std::swap(c->red, a.red); std::swap(c->red_bytes, a.red_bytes); std::swap(c->ytop, a.ytop); std::swap(c->z, a.z);
  std::swap(c->rhs, a.rhs); std::swap(c->xfer, a.xfer);
  std::swap(c->tree_cnt, a.tree_cnt); std::swap(c->h_fail, a.h_fail); std::swap(c->stream, a.stream);
  std::swap(c->graph_exec, a.graph_exec); std::swap(c->graph_flags, a.graph_flags);
  std::swap(c->graph_stream, a.graph_stream); std::swap(c->graph_rec_complete, a.graph_rec_complete);
  std::swap(c->graph_rec_compact, a.graph_rec_compact);
  std::swap(c->graph_schedule, a.graph_schedule); std::swap(c->graph_apply, a.graph_apply);
  std::swap(c->ev_start, a.ev_start); std::swap(c->ev_stop, a.ev_stop);
  c->in_alt = !c->in_alt;
}

static void free_alt(NdlqrHipCtx* c) {
  if (c->in_alt) swap_slot(c);
  NdlqrAltSlot& a = c->alt;
  if (a.stream) (void)hipStreamSynchronize(a.stream);
  if (a.graph_exec) (void)hipGraphExecDestroy(a.graph_exec);
  (void)hipFree(a.rec); (void)hipFree(a.red); (void)hipFree(a.ytop); (void)hipFree(a.z); (void)hipFree(a.tree_cnt);
  (void)hipFree(a.rhs); (void)hipFree(a.xfer);
  if (a.h_fail) (void)hipHostFree(a.h_fail);
  if (a.ev_start) (void)hipEventDestroy(a.ev_start);
  if (a.ev_stop) (void)hipEventDestroy(a.ev_stop);
  if (a.stream) (void)hipStreamDestroy(a.stream);
  a = NdlqrAltSlot();
}

// allocate the alternate set on first use; false (and depth 1 from then on) when it does not fit
static bool ensure_alt(NdlqrHipCtx* c) {
  NdlqrAltSlot& a = c->alt;
  if (a.ready) return true;
  const ndlqr::Dims& d = c->d;
  const size_t slot_doubles = ((size_t)d.n * (d.n + 1) + 2 * (size_t)d.n * d.n + 2 * d.n + 15) / 16 * 16;  // RedSlot<NX>::SIZE
  const size_t red_bytes = sizeof(double) * (size_t)d.batch * (d.N / 4) * slot_doubles;
  const size_t cnt_bytes = sizeof(int) * (size_t)d.batch * (d.N / 4);
  // The second set's stream gets another priority than the first's: streams of one priority share a few hardware
  // queues round-robin with every other stream of the process, and two streams on ONE hardware queue run strictly one
  // after the other (measured: the step pipeline lost all its overlap in a process that had created other streams
  // before, tools/e2e_probe.py --other-solvers; with its own priority level it keeps it).
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  const int alt_prio = getenv("NDLQR_ALT_PRIORITY") ? atoi(getenv("NDLQR_ALT_PRIORITY")) : prio_greatest;
  bool ok = hipStreamCreateWithPriority(&a.stream, hipStreamNonBlocking, alt_prio) == hipSuccess &&
            hipEventCreate(&a.ev_start) == hipSuccess && hipEventCreate(&a.ev_stop) == hipSuccess &&
            hipMalloc(&a.rec, bytes_rec(d)) == hipSuccess && hipMalloc(&a.z, bytes_z(d)) == hipSuccess &&
            hipMalloc(&a.rhs, bytes_z(d)) == hipSuccess &&
            hipHostMalloc((void**)&a.h_fail, sizeof(int), hipHostMallocDefault) == hipSuccess;
  // this set's own copy of the right-hand side (a step of ndlqr_hip_step_async replaces the right-hand side of
  // ITS buffer set only; rhs_gen / rhs_make_current keep track of which copy is behind in what)
  ok = ok && hipStreamSynchronize(c->stream) == hipSuccess &&
       hipMemcpyAsync(a.rhs, c->rhs, bytes_z(d), hipMemcpyDeviceToDevice, a.stream) == hipSuccess;
  if (ok && c->tree_cnt)  // size-specialised shapes (the runtime-sized schedule's slots: ensure_red_generic)
    ok = hipMalloc(&a.red, red_bytes) == hipSuccess && hipMemsetAsync(a.red, 0, red_bytes, a.stream) == hipSuccess &&
         ((a.red_bytes = red_bytes), true) &&
         hipMalloc(&a.ytop, sizeof(double) * (size_t)d.batch * (d.N / 8) * d.n) == hipSuccess &&
         hipMalloc(&a.tree_cnt, cnt_bytes) == hipSuccess && hipMemsetAsync(a.tree_cnt, 0, cnt_bytes, a.stream) == hipSuccess;
  ok = ok && hipMemsetAsync(a.z, 0, bytes_z(d), a.stream) == hipSuccess && hipStreamSynchronize(a.stream) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    free_alt(c);
    c->pipeline = 1;
    return false;
  }
  *a.h_fail = *c->h_fail;
  a.ready = true;
  {  // (the new set's copy of the right-hand side was taken from the current one)
    const int cur = c->in_alt ? 1 : 0;
    for (int p = 0; p < 4; ++p) c->rhs_gen[1 - cur][p] = c->rhs_gen[cur][p];
  }
  return true;
}

// every solve in flight on either slot has finished
static hipError_t sync_all(NdlqrHipCtx* c) {
  hipError_t e = c->stream ? hipStreamSynchronize(c->stream) : hipSuccess;
  if (c->alt.stream) { const hipError_t e2 = hipStreamSynchronize(c->alt.stream); if (e == hipSuccess) e = e2; }
  return e;
}

static int rhs_make_current(NdlqrHipCtx* c, unsigned need);  // below

int ndlqr_hip_set_pipeline_depth(NdlqrHipCtx* c, int depth) {
  if (!c || depth < 1) return NDLQR_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(sync_all(c));
  if (c->in_alt) {  // keep the latest solution where the single-slot code expects it
    swap_slot(c);
    {  // (the right-hand side the latest solution belongs to: after ndlqr_hip_step_async the two sets may differ)
      const int merr = rhs_make_current(c, 0xFu);
      if (merr) return merr;
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (c->alt.z && c->z_latest == c->alt.z) {
      HIP_TRY(hipMemcpy(c->z, c->alt.z, bytes_z(c->d), hipMemcpyDeviceToDevice));
      HIP_TRY(hipDeviceSynchronize());  // (a device-to-device copy on the null stream need not be finished on return;
                                        //  the solver's streams do not wait for the null stream)
      c->z_latest = c->z;
    }
  }
  c->pipeline = depth > 2 ? 2 : depth;
  return NDLQR_OK;
}
int ndlqr_hip_pipeline_depth(const NdlqrHipCtx* c) { return c ? c->pipeline : 0; }

// The complete factor array [batch][K][N][2n+m][n] (5.6 GB at (12,4,256) x 1024, 87 GB at
// (64,16,512) x 256) exists only for the schedules that touch it: strict mode, KEEP_FACT, the
// knot-based and runtime-sized paths, and the factors kept by KEEP_RECORDS. The default
// separator-only fast path never allocates it. Must run outside stream capture (hipMalloc).
int ndlqr_hip_ensure_F(NdlqrHipCtx* c) {
  if (c->F) return NDLQR_OK;
  HIP_TRY(hipSetDevice(c->device));
  hipError_t e = hipMalloc(&c->F, bytes_F(c->d));
  if (e != hipSuccess) {
    c->F = nullptr;
    g_last_error = "factor array does not fit on the device (" + std::to_string(bytes_F(c->d) >> 20) +
                   " MiB): use the default fast mode without NDLQR_FLAG_KEEP_FACT, or a smaller batch";
    fprintf(stderr, "ndlqr_hip: %s\n", g_last_error.c_str());
    (void)hipGetLastError();
    return NDLQR_ERR_INVALID;
  }
  // Structural zeros of F are never written by the kernels; zero once so that the factor
  // download matches the reference's calloc'ed array (src/nddata.c:34).
  HIP_TRY(hipMemsetAsync(c->F, 0, bytes_F(c->d), c->stream));
  return NDLQR_OK;
}

int ndlqr_hip_set_flags(NdlqrHipCtx* c, unsigned flags) {
  if (!c) return NDLQR_ERR_INVALID;
  c->flags = flags;
  return NDLQR_OK;
}
unsigned ndlqr_hip_get_flags(const NdlqrHipCtx* c) { return c ? c->flags : 0u; }

int ndlqr_hip_set_stream(NdlqrHipCtx* c, void* hip_stream) {
  if (!c) return NDLQR_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  {
    const int perr = ndlqr_hip_set_pipeline_depth(c, c->pipeline);  // drains both slots, primary set current
    if (perr) return perr;
  }
  if (c->own_stream && c->stream) HIP_TRY(hipStreamDestroy(c->stream));
  if (hip_stream) {
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
  } else {
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  return NDLQR_OK;
}
void* ndlqr_hip_get_stream(NdlqrHipCtx* c) { return c ? (void*)c->stream : nullptr; }

// The right-hand side exists once per buffer set of the pipeline (hip_context.hpp: rhs_latest / rhs_gen).
// rhs_written_cur: the parts of `mask` of the CURRENT set's copy have just been (re)written.
static void rhs_written_cur(NdlqrHipCtx* c, unsigned mask) {
  const int cur = c->in_alt ? 1 : 0;
  for (int p = 0; p < 4; ++p)
    if (mask & (1u << p)) c->rhs_gen[cur][p] = ++c->rhs_latest[p];
}
// rhs_make_current: the current set's copy is brought up to date in the parts of `need` before a solve reads it. Only a
// change of flow gets here with something to do (full MPC steps followed by x0-only steps, a plain solve behind steps,
// the first solve on the other set after an upload): everything in flight is waited for, the stale parts are copied
// from the other set on this set's stream, and the other set's stream waits for that copy before it may rewrite its own.
static int rhs_make_current(NdlqrHipCtx* c, unsigned need) {
  const int cur = c->in_alt ? 1 : 0;
  unsigned stale = 0;
  for (int p = 0; p < 4; ++p)
    if ((need & (1u << p)) && c->rhs_gen[cur][p] < c->rhs_latest[p]) stale |= 1u << p;
  if (!stale || !c->alt.rhs) return NDLQR_OK;
  HIP_TRY(sync_all(c));
  hipLaunchKernelGGL(ndlqr::copy_rhs_parts_generic, dim3(c->d.N, c->d.batch), dim3(64), 0, c->stream, c->d, stale,
                     (const double*)c->alt.rhs, c->rhs);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev_inputs, c->stream));
  if (c->alt.stream) HIP_TRY(hipStreamWaitEvent(c->alt.stream, c->ev_inputs, 0));
  for (int p = 0; p < 4; ++p)
    if (stale & (1u << p)) c->rhs_gen[cur][p] = c->rhs_latest[p];
  return NDLQR_OK;
}

// the other buffer set's stream waits for everything enqueued on the current one so far
static hipError_t other_stream_waits(NdlqrHipCtx* c) {
  if (!c->alt.stream) return hipSuccess;
  const hipError_t e = hipEventRecord(c->ev_inputs, c->stream);
  return e != hipSuccess ? e : hipStreamWaitEvent(c->alt.stream, c->ev_inputs, 0);
}

// staging of caller-layout data of a padded shape
static int ensure_pad_stage(NdlqrHipCtx* c, size_t doubles) {
  if (doubles <= c->pad_stage_cap) return NDLQR_OK;
  if (c->pad_stage) { HIP_TRY(hipStreamSynchronize(c->stream)); (void)hipFree(c->pad_stage); c->pad_stage = nullptr; c->pad_stage_cap = 0; }
  if (c->graph_staged) { (void)hipGraphExecDestroy(c->graph_staged); c->graph_staged = nullptr; }  // (it holds the old address)
  HIP_TRY(hipMalloc(&c->pad_stage, sizeof(double) * doubles));
  c->pad_stage_cap = doubles;
  return NDLQR_OK;
}

int ndlqr_hip_upload_inputs(NdlqrHipCtx* c, int p0, int count, const double* AB, const double* QR,
                            const double* rhs) {
  if (!c || !AB || !QR || !rhs || p0 < 0 || count <= 0 || p0 + count > c->d.batch) return NDLQR_ERR_INVALID;
  const ndlqr::Dims& d = c->d;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(sync_all(c));  // solves in flight on either slot still read the inputs
  {  // (a partial upload lands on a complete, current copy: the other set then takes the whole of it on its next use)
    const int merr = rhs_make_current(c, 0xFu);
    if (merr) return merr;
  }
  if (c->padded) {  // the caller's layout goes to a staging array in HBM, a kernel files it into the padded arrays
    const ndlqr::Dims& u = c->du;
    const size_t uAB = (size_t)u.N * u.n * u.w * count, uQR = (size_t)u.N * u.w * count, uz = (size_t)u.N * u.rows * count;
    const int serr = ensure_pad_stage(c, uAB + uQR + uz);
    if (serr) return serr;
    double* s0 = c->pad_stage;
    HIP_TRY(hipMemcpyAsync(s0, AB, sizeof(double) * uAB, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(s0 + uAB, QR, sizeof(double) * uQR, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(s0 + uAB + uQR, rhs, sizeof(double) * uz, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(ndlqr::pad_inputs_generic, dim3(d.N, count), dim3(128), 0, c->stream, u, d, p0, s0, s0 + uAB,
                       s0 + uAB + uQR, c->AB, c->QR, c->rhs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    rhs_written_cur(c, 0xFu);
    next_solve_on_current_set(c);
    c->fact_valid = false;
    c->rec_complete = false;
    return NDLQR_OK;
  }
  const size_t sAB = (size_t)d.N * d.n * d.w, sQR = (size_t)d.N * d.w, sz = (size_t)d.N * d.rows;
  HIP_TRY(hipMemcpyAsync(c->AB + p0 * sAB, AB, sizeof(double) * sAB * count, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->QR + p0 * sQR, QR, sizeof(double) * sQR * count, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->rhs + p0 * sz, rhs, sizeof(double) * sz * count, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // the host staging buffers are reused by the caller
  rhs_written_cur(c, 0xFu);
  next_solve_on_current_set(c);
  c->fact_valid = false;  // new A, B, Q, R: a cached factorisation no longer matches the inputs
  c->rec_complete = false;
  return NDLQR_OK;
}

int ndlqr_hip_pack_flat_device(NdlqrHipCtx* c, const double* A, const double* B, const double* Q,
                               const double* R, const double* q, const double* r, const double* d,
                               const double* x0) {
  if (!c || !A || !B || !Q || !R || !q || !r || !d || !x0) return NDLQR_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(sync_all(c));  // solves in flight on either slot still read the inputs
  // (eight knots per workgroup while a thread's eight loads -- the same entry of eight consecutive blocks -- stay within what
  //  the caches hold together: at (64,16) the strided reads of eight 40 KB blocks at once took 9.5 instead of 5 ms)
  if (c->d.N % 8 == 0 && c->du.n <= 16)
    hipLaunchKernelGGL(ndlqr::pack_flat_generic<8>, dim3(c->d.N / 8, c->d.batch), dim3(128), 0, c->stream, c->du, c->d, A, B, Q,
                       R, q, r, d, x0, c->AB, c->QR, c->rhs);
  else if (c->du.n > 16 && sizeof(double) * (size_t)(c->du.n | 1) * c->du.w <= 64 * 1024)  // through LDS, whole lines in and out
    hipLaunchKernelGGL(ndlqr::pack_flat_tiled, dim3(c->d.N, c->d.batch), dim3(256), sizeof(double) * (size_t)(c->du.n | 1) * c->du.w,
                       c->stream, c->du, c->d, A, B, Q, R, q, r, d, x0, c->AB, c->QR, c->rhs);
  else
    hipLaunchKernelGGL(ndlqr::pack_flat_generic<1>, dim3(c->d.N, c->d.batch), dim3(128), 0, c->stream, c->du, c->d, A, B, Q, R,
                       q, r, d, x0, c->AB, c->QR, c->rhs);
  HIP_TRY(hipGetLastError());
  rhs_written_cur(c, 0xFu);        // (the whole batch: nothing of the older copies is needed any more)
  next_solve_on_current_set(c);
  HIP_TRY(other_stream_waits(c));  // the next solve may run on the other buffer set's stream
  c->fact_valid = false;  // new A, B, Q, R: neither a cached factor array nor cached records match
  c->rec_complete = false;
  return NDLQR_OK;
}

int ndlqr_hip_device_pointers(NdlqrHipCtx* c, void** out5) {
  if (!c || !out5) return NDLQR_ERR_INVALID;
  if (c->padded) {
    g_last_error = "no raw device pointers for a block size that runs zero-padded (the arrays have another layout): "
                   "use ndlqr_hip_pack_flat_device, or NDLQR_NO_PAD=1";
    fprintf(stderr, "ndlqr_hip: %s\n", g_last_error.c_str());
    return NDLQR_ERR_INVALID;
  }
  const int ferr = ndlqr_hip_ensure_F(c);  // the caller asks for the factor array: it has to exist
  if (ferr) return ferr;
  const int perr = ndlqr_hip_set_pipeline_depth(c, 1);  // the caller holds raw pointers: one buffer set from now on
  if (perr) return perr;
  out5[0] = c->AB; out5[1] = c->QR; out5[2] = c->rhs; out5[3] = c->F; out5[4] = c->z;
  return NDLQR_OK;
}

// ------------------------------------------------------------------------------ launches

// lean (fast mode without KEEP): the Schur passes only keep the boundary knots of every subtree
// up to date, the solution comes from the back-substitution over the separator records.
constexpr int kSepChunkTiles = 3;  // column tiles of the right-hand-side panel resident in LDS (separator_mfma)

// Separator-only schedule for blocks that fill 16x16 matrix-core tiles (kernels_reduced_mfma.hpp): workgroup
// size and LDS of separator_reduced_mfma, or false when the shape does not qualify (-> generic-lean).
struct ReducedGenericPlan {
  bool ok;
  int threads;
  size_t lds;
  int nb;    // 16 x 16 tiles per block row
  bool pad;  // the block does not fill them
  bool keep; // NDLQR_FLAG_KEEP_RECORDS
};
static ReducedGenericPlan plan_reduced_generic(const NdlqrHipCtx* c) {
  const ndlqr::Dims& d = c->d;
  ReducedGenericPlan p = {false, 0, 0, 0, false, false};
  if (c->flags & (NDLQR_FLAG_STRICT_FP | NDLQR_FLAG_KEEP_FACT)) return p;
  if (c->no_mfma || d.n > 128 || d.N < 2) return p;
  if (getenv("NDLQR_DEV_NO_REDUCED_GENERIC")) return p;  // (developer: A/B against the knot-based runtime-sized schedule)
  p.keep = (c->flags & NDLQR_FLAG_KEEP_RECORDS) != 0;  // W of every separator kept for rhs-only re-solves
  p.nb = (d.n + 15) / 16;
  const int npad = 16 * p.nb, wpad = (d.w + 3) / 4 * 4;
  p.pad = npad != d.n || wpad != d.w;  // blocks that do not fill their tiles: zero-padded in LDS (PAD instances)
  // one wavefront per 16x16 block where the weights fit its lanes: many small workgroups fill the chip better than
  // a four-wavefront workgroup whose phases are mostly serial at this size
  // a wavefront per 16-column tile where the weights fit the lanes ("two rounds", kernels_reduced_mfma.hpp: small
  // workgroups, three or more of them per CU), else twice that
  p.threads = wpad <= 64 * p.nb ? 64 * p.nb : (p.nb >= 3 ? 512 : 256);
  if (wpad > p.threads) return p;  // one weight / rhs entry per thread
  if (p.nb >= 5 && p.threads != 64 * p.nb) return p;  // (beyond 64 states the second panel array does not fit the LDS)
  p.lds = sizeof(double) * (size_t)ndlqr::reduced_lds_doubles(npad, wpad, p.threads == 64 * p.nb);
  if (p.lds > 160 * 1024) return p;
  p.ok = true;
  return p;
}

static size_t bytes_red_generic(const ndlqr::Dims& d) {
  return sizeof(double) * (size_t)d.batch * (d.N / 2) * (4 * (size_t)d.n * d.n + 2 * d.n);
}

// slots of the separators of level >= 1 (runtime-sized separator-only schedule); allocated by the first
// solve that takes it, for both buffer sets of the pipeline. Never zeroed: the level-0 launch stores
// every accumulator block. Must run outside stream capture.
static int ensure_red_generic(NdlqrHipCtx* c) {
  if (c->d.N < 4) return NDLQR_OK;  // a single separator: no slots
  const size_t need = bytes_red_generic(c->d);
  for (int which = 0; which < 2; ++which) {
    double** slot = which == 0 ? &c->red : &c->alt.red;
    size_t* have = which == 0 ? &c->red_bytes : &c->alt.red_bytes;
    if (which == 1 && !c->alt.ready) continue;
    if (*slot && *have >= need) continue;
    // (a context of a size-specialised shape under NDLQR_FLAG_GENERIC comes with the smaller array of ITS schedule)
    HIP_TRY(sync_all(c));
    if (*slot) { (void)hipFree(*slot); *slot = nullptr; *have = 0; }
    // a launch sequence captured on this buffer set holds the old address
    hipGraphExec_t* ge = which == 0 ? &c->graph_exec : &c->alt.graph_exec;
    if (*ge) { (void)hipGraphExecDestroy(*ge); *ge = nullptr; }
    if (hipMalloc(slot, need) != hipSuccess) {
      *slot = nullptr;
      (void)hipGetLastError();
      g_last_error = "accumulator slots of the separator-only schedule do not fit on the device";
      return NDLQR_ERR_INVALID;
    }
    *have = need;
    // (zeros for the size-specialised schedule, should the context go back to it. hipMemset runs on the null
    //  stream and may return before it is done; the solver's streams are non-blocking: wait here, or the
    //  level-0 launch races with it)
    HIP_TRY(hipMemset(*slot, 0, need));
    HIP_TRY(hipDeviceSynchronize());
  }
  return NDLQR_OK;
}

// back-substitution of the runtime-sized separator-only schedule: its records are compact (the Cholesky factor and y~ instead of
// f_a | f_bb | z_sep, the couplings come from the slots / the problem data), and one launch resolves the level-0
// multipliers and the states / inputs of every knot
static void launch_backsub_reduced_generic(NdlqrHipCtx* c) {
  const ndlqr::Dims& d = c->d;
  ScopedSlot t(c, SLOT_APPLY);
  const size_t lds_m = sizeof(double) * ((size_t)d.n * (d.n + 1) / 2 + 2 * (size_t)d.n + 16);
  // (rows of CA | CB eight per wavefront. Four wavefronts per separator leave a CU a quarter full at small blocks: one up to
  //  32 states, two up to 64, four beyond -- profiles/r04_mult_threads_ab.txt: (16,4,256) x 1024 2.30 -> 1.86 ms per solve,
  //  (20,20) 1.32 -> 1.23, (48,16,512) 6.44 -> 6.27, (96,16) best at four; NDLQR_MULT_THREADS overrides)
  static const int thr_env = getenv("NDLQR_MULT_THREADS") ? atoi(getenv("NDLQR_MULT_THREADS")) : 0;
  const int thr_m = (thr_env == 64 || thr_env == 128 || thr_env == 256) ? thr_env : (d.n <= 32 ? 64 : (d.n <= 64 ? 128 : 256));
  // A step that wants knots [k0, k1] alone (NDLQR_SOLN_ONLY: apply_blk0 / apply_nblk in units of eight knots): of every
  // level the separators whose subtree meets [k0 - 1, k1 + 2] -- a set closed under "needs the multipliers of the
  // separators bounding its subtree" (those are ancestors: their subtrees contain it) --, of level 0 the pairs of the range
  const bool part = c->apply_nblk > 0;
  const int k0 = 8 * c->apply_blk0;  // (< N: the selection lies inside the horizon)
  const int k1 = 8 * (c->apply_blk0 + c->apply_nblk) - 1 < d.N ? 8 * (c->apply_blk0 + c->apply_nblk) - 1 : d.N - 1;  // (horizons below 8 knots)
  const int ka = k0 > 0 ? k0 - 1 : 0, kb = k1 + 2 < d.N ? k1 + 2 : d.N - 1;
  for (int l = d.K - 1; l >= 1; --l) {
    ndlqr::Dims dl = d;
    int cnt = d.N >> (l + 1);
    if (part) { dl.xoff = ka >> (l + 1); cnt = (kb >> (l + 1)) - dl.xoff + 1; }
    hipLaunchKernelGGL(ndlqr::backsub_multipliers_compact, dim3(cnt, d.batch), dim3(thr_m), lds_m, c->stream, dl, l,
                       c->red, c->rec, c->z);
  }
  const int thr = d.n <= 16 ? 64 : (d.n <= 32 ? 128 : 256);  // (its y_s step wants n <= threads)
  const size_t lds = sizeof(double) * ((size_t)d.n * (d.n + 1) / 2 + 5 * (size_t)d.n + 4 * (size_t)d.w + 2 * (size_t)d.rows + thr);
  ndlqr::Dims d0 = d;
  if (part) d0.xoff = k0 >> 1;
  hipLaunchKernelGGL(ndlqr::backsub_level0_states_generic, dim3(part ? (k1 >> 1) - (k0 >> 1) + 1 : d.N >> 1, d.batch), dim3(thr),
                     lds, c->stream, d0, c->AB, c->QR, c->rhs, c->rec, c->z);
}

static int launch_reduced_generic(NdlqrHipCtx* c, const ReducedGenericPlan& p) {
  const ndlqr::Dims& d = c->d;
  c->schedule = p.keep ? "generic-reduced-records" : "generic-reduced";
  c->rec_complete = p.keep;  // records, slots and W of every separator stay: rhs-only re-solves (launch_rhs_reduced_generic)
  for (int l = 0; l < d.K; ++l) {
    ScopedSlot t(c, SLOT_SEP);
    const dim3 grid(d.N >> (l + 1), d.batch);
#define NDLQR_LAUNCH_SEP2(NB_, NT_, L0_, PAD_)                                                                     \
  hipLaunchKernelGGL((ndlqr::separator_reduced_mfma<NB_, NT_, L0_, PAD_>), grid, dim3(NT_), p.lds, c->stream, d, l, \
                     c->AB, c->QR, c->rhs, c->red, c->rec, c->info)
#define NDLQR_LAUNCH_SEP(NB_, NT_)                                          \
  do {                                                                      \
    if (l == 0 && p.pad) NDLQR_LAUNCH_SEP2(NB_, NT_, true, true);           \
    else if (l == 0) NDLQR_LAUNCH_SEP2(NB_, NT_, true, false);              \
    else if (p.pad) NDLQR_LAUNCH_SEP2(NB_, NT_, false, true);               \
    else NDLQR_LAUNCH_SEP2(NB_, NT_, false, false);                         \
  } while (0)
    switch (p.nb) {
      case 1:
        if (p.threads == 64) NDLQR_LAUNCH_SEP(1, 64);
        else NDLQR_LAUNCH_SEP(1, 256);
        break;
      case 2:
        if (p.threads == 128) NDLQR_LAUNCH_SEP(2, 128);
        else NDLQR_LAUNCH_SEP(2, 256);
        break;
      case 3:
        if (p.threads == 192) NDLQR_LAUNCH_SEP(3, 192);
        else NDLQR_LAUNCH_SEP(3, 512);
        break;
      case 4:
        if (p.threads == 256) NDLQR_LAUNCH_SEP(4, 256);
        else NDLQR_LAUNCH_SEP(4, 512);
        break;
      case 5:  // (beyond 64 states: one workgroup per CU, the form with one wavefront per tile column only)
        NDLQR_LAUNCH_SEP(5, 320);
        break;
      case 6:
        NDLQR_LAUNCH_SEP(6, 384);
        break;
      case 7:
        NDLQR_LAUNCH_SEP(7, 448);
        break;
      default:
        NDLQR_LAUNCH_SEP(8, 512);
        break;
    }
#undef NDLQR_LAUNCH_SEP2
#undef NDLQR_LAUNCH_SEP
  }
  launch_backsub_reduced_generic(c);
  return NDLQR_OK;
}

// rhs-only re-solve on the records, slots and separator factors of a generic-reduced-records sweep
static void launch_rhs_reduced_generic(NdlqrHipCtx* c) {
  const ndlqr::Dims& d = c->d;
  const int np = (d.n + 15) / 16 * 16;
  const size_t lds = sizeof(double) * (2 * (size_t)d.w + d.n + 3 * (size_t)np + 256 + (size_t)d.n * (d.n + 1) / 2);
  for (int l = 0; l < d.K; ++l) {
    ScopedSlot t(c, SLOT_SEP);
    hipLaunchKernelGGL(ndlqr::rhs_reduced_generic, dim3(d.N >> (l + 1), d.batch), dim3(256), lds, c->stream, d, l, np,
                       c->AB, c->QR, c->rhs, c->red, c->rec);
  }
  launch_backsub_reduced_generic(c);
}

// The separator kernel of the knot-based runtime-sized schedules and where its S-bar (n x (n+1)) and right-hand-side panel
// (n x (2n+1)) live: separator_mfma (fast mode, blocks that fill 16x16 tiles; the panel goes through LDS in chunks of
// kSepChunkTiles column tiles, + the inverses of the 16x16 diagonal blocks, pitch 17) while its LDS and its wavefronts
// fit a workgroup; else separator_generic with both arrays in LDS; else -- blocks beyond ~80 states -- separator_generic
// with both arrays in global memory (NdlqrHipCtx::sep_scratch): every mode, any block size the device holds.
struct GenericSepPlan {
  bool mfma;
  bool scratch;
  size_t lds;
  int threads;
};
static GenericSepPlan plan_generic_sep(const NdlqrHipCtx* c, const bool strict) {
  const ndlqr::Dims& d = c->d;
  GenericSepPlan p;
  p.mfma = !strict && d.n % 16 == 0 && d.w % 4 == 0 && !c->no_mfma;
  p.scratch = false;
  const int ctl = 2 * (d.n / 16) + 1, ctc = ctl < kSepChunkTiles ? ctl : kSepChunkTiles;
  const size_t lds_mfma = sizeof(double) * ((size_t)d.n * (d.n + 1) + (size_t)d.n * (16 * ctc + 1) + (size_t)d.n * 17);
  const size_t lds_generic = sizeof(double) * ((size_t)d.n * (d.n + 1) + (size_t)d.n * (2 * d.n + 1));
  // matrix-core separator: one wavefront per 16x16 tile of the products / updates; 512 threads let two
  // workgroups (67 KB of LDS each at n = 64) share a CU
  p.threads = c->sep_threads > 0 ? c->sep_threads : (d.n >= 32 ? 512 : 256);
  if (p.mfma) {  // separator_mfma: a wavefront per tile of a block row of W, at most three panel tiles per wavefront
    const int need = (d.n / 16) > ((d.n / 16) * ctc + 2) / 3 ? (d.n / 16) : ((d.n / 16) * ctc + 2) / 3;
    if (p.threads < 64 * need) p.threads = 64 * need;
    if (p.threads > 1024) p.mfma = false;
  }
  if (p.mfma) {
    // (beyond 112 states S-bar / L goes to global memory: the panel chunk and the inverses of the diagonal blocks stay)
    const size_t lds_no_s = lds_mfma - sizeof(double) * (size_t)d.n * (d.n + 1);
    if (lds_mfma <= 160 * 1024) { p.lds = lds_mfma; return p; }
    if (lds_no_s <= 160 * 1024) { p.lds = lds_no_s; p.scratch = true; return p; }
    p.mfma = false;
  }
  p.scratch = lds_generic > 160 * 1024;
  p.lds = p.scratch ? 0 : lds_generic;
  p.threads = p.scratch ? 1024 : 256;
  return p;
}

template <bool STRICT>
static int launch_generic(NdlqrHipCtx* c, bool lean) {
  const ndlqr::Dims& d = c->d;
  double* rec = lean ? c->rec : nullptr;
  c->schedule = lean ? "generic-lean" : (STRICT ? "generic-strict" : "generic-keep");
  {
    ScopedSlot t(c, SLOT_LEAF);
    hipLaunchKernelGGL((ndlqr::leaf_generic<STRICT>), dim3(d.N, d.batch), dim3(128), 0, c->stream, d,
                       c->AB, c->QR, c->rhs, c->F, c->z, c->info, lean ? 1 : 0);
  }
  // S (n x (n+1)) + right-hand-side panel (n x (2n+1)); on the matrix-core path the panel goes through
  // LDS in chunks of kSepChunkTiles column tiles (+ the inverses of the 16x16 diagonal blocks, pitch 17)
  const GenericSepPlan sp = plan_generic_sep(c, STRICT);
  const bool p1mfma = sp.mfma;
  const size_t lds = sp.lds;
  const int sep_threads = sp.threads;
  double* sep_scratch = nullptr;
  if (sp.scratch) {
    sep_scratch = c->sep_scratch;
    if (!sep_scratch) {
      g_last_error = "nstates too large for the separator kernel's LDS staging and no global scratch";
      return NDLQR_ERR_INVALID;
    }
  }
  for (int l = 0; l < d.K; ++l) {
    const int nsub = d.N >> (l + 1);
    {
      ScopedSlot t(c, SLOT_SEP);
      if (p1mfma)
        hipLaunchKernelGGL((ndlqr::separator_mfma<kSepChunkTiles>), dim3(nsub, d.batch), dim3(sep_threads), lds,
                           c->stream, d, l, c->AB, c->F, c->z, c->info, rec, sep_scratch,
                           (size_t)d.n * (d.n + 1) + (size_t)d.n * (2 * d.n + 1));
      else
        hipLaunchKernelGGL((ndlqr::separator_generic<STRICT>), dim3(nsub, d.batch), dim3(sep_threads), lds,
                           c->stream, d, l, c->AB, c->F, c->z, c->info, rec, sep_scratch);
    }
    if (lean && l == d.K - 1) break;  // nothing above the root separator
    {
      ScopedSlot t(c, lean ? SLOT_BOUNDARY : SLOT_SCHUR);
      const int bnd = lean ? 1 : 0;
      const unsigned gx = lean ? 2u * (unsigned)nsub : (unsigned)d.N;  // knots this pass updates
      // block sizes that fill 16x16 MFMA tiles: Schur update on the fp64 matrix cores (fast mode)
      const bool mfma = !STRICT && d.n % 16 == 0 && d.rows % 16 == 0 && d.n <= 64 && !c->no_mfma;
      const size_t flds = sizeof(double) * (size_t)d.n * (d.n + 16);
      if (mfma && d.n == 64)
        hipLaunchKernelGGL((ndlqr::schur_mfma<4>), dim3(gx, d.batch), dim3(256), flds, c->stream, d, l, c->F, c->z, bnd,
                           (const double*)rec);
      else if (mfma && d.n == 48)
        hipLaunchKernelGGL((ndlqr::schur_mfma<3>), dim3(gx, d.batch), dim3(256), flds, c->stream, d, l, c->F, c->z, bnd,
                           (const double*)rec);
      else if (mfma && d.n == 32)
        hipLaunchKernelGGL((ndlqr::schur_mfma<2>), dim3(gx, d.batch), dim3(256), flds, c->stream, d, l, c->F, c->z, bnd,
                           (const double*)rec);
      else if (mfma && d.n == 16)
        hipLaunchKernelGGL((ndlqr::schur_mfma<1>), dim3(gx, d.batch), dim3(256), flds, c->stream, d, l, c->F, c->z, bnd,
                           (const double*)rec);
      else if (!STRICT && d.n % 16 == 0 && d.n > 64 && !c->no_mfma)  // (runtime-sized matrix-core form: blocks beyond 64 states)
        hipLaunchKernelGGL(ndlqr::schur_mfma_rt, dim3(gx, d.batch), dim3(256), 0, c->stream, d, l, c->F, c->z, bnd,
                           (const double*)rec);
      else {
        const long work = (long)gx * d.rows * d.n;
        hipLaunchKernelGGL((ndlqr::schur_generic<STRICT>), dim3((unsigned)((work + 255) / 256), d.batch),
                           dim3(256), 0, c->stream, d, l, c->F, c->z, bnd, (const double*)rec);
      }
    }
  }
  if (lean) {
    ScopedSlot t(c, SLOT_APPLY);
    for (int l = d.K - 1; l >= 0; --l) {
      hipLaunchKernelGGL(ndlqr::backsub_multipliers_generic, dim3(d.N >> (l + 1), d.batch), dim3(64), 0, c->stream,
                         d, l, c->rec, c->z);
    }
    const int work = d.N * d.rows;
    hipLaunchKernelGGL(ndlqr::backsub_states_generic, dim3((work + 255) / 256, d.batch), dim3(256), 0, c->stream, d,
                       c->AB, c->QR, c->rhs, c->z);
  }
  return NDLQR_OK;
}

// ---- size-specialised instances: one translation unit each (small_instance.hip), listed in
//      small_instances.def
#define NDLQR_SMALL_INSTANCE(NX_, NU_)                                              \
  int ndlqr_small_solve_##NX_##_##NU_(NdlqrHipCtx* c, bool strict, bool keep);       \
  int ndlqr_small_needs_F_##NX_##_##NU_(const NdlqrHipCtx* c, bool strict, bool keep); \
  void ndlqr_small_rhs_##NX_##_##NU_(NdlqrHipCtx* c);                               \
  int ndlqr_small_kpb_##NX_##_##NU_(void);                                          \
  int ndlqr_small_tshard_##NX_##_##NU_(NdlqrHipCtx* c, int phase, int g, int G);      \
  int ndlqr_small_slot_##NX_##_##NU_(void);                                          \
  int ndlqr_small_multi_##NX_##_##NU_(NdlqrHipCtx* c, int count, const double* rhs, double* zsep, double* fsum, double* ytop, double* z);
#include "small_instances.def"
#undef NDLQR_SMALL_INSTANCE

struct SmallInstance {
  int nx, nu;
  int (*solve)(NdlqrHipCtx*, bool, bool);
  int (*needs_F)(const NdlqrHipCtx*, bool, bool);
  void (*rhs)(NdlqrHipCtx*);
  int (*kpb)(void);
  int (*tshard)(NdlqrHipCtx*, int, int, int);
  int (*slot)(void);
  int (*multi)(NdlqrHipCtx*, int, const double*, double*, double*, double*, double*);
};
#ifdef NDLQR_SINGLE_TU
#define NDLQR_SMALL_INSTANCE(NX_, NU_)                                                              \
  int ndlqr_small_solve_##NX_##_##NU_(NdlqrHipCtx* c, bool strict, bool keep) {                     \
    if (strict) return keep ? launch_small<NX_, NU_, true, true>(c) : launch_small<NX_, NU_, true, false>(c); \
    return keep ? launch_small<NX_, NU_, false, true>(c) : launch_small<NX_, NU_, false, false>(c);           \
  }                                                                                                 \
  int ndlqr_small_needs_F_##NX_##_##NU_(const NdlqrHipCtx* c, bool strict, bool keep) {             \
    if (strict) return keep ? plan_small<NX_, NU_, true, true>(c).needs_F : plan_small<NX_, NU_, true, false>(c).needs_F; \
    return keep ? plan_small<NX_, NU_, false, true>(c).needs_F : plan_small<NX_, NU_, false, false>(c).needs_F;           \
  }                                                                                                 \
  void ndlqr_small_rhs_##NX_##_##NU_(NdlqrHipCtx* c) { launch_rhs_records<NX_, NU_>(c); }           \
  int ndlqr_small_kpb_##NX_##_##NU_(void) { return ndlqr::SchurShape<NX_, NU_>::KPB; }              \
  int ndlqr_small_tshard_##NX_##_##NU_(NdlqrHipCtx* c, int phase, int g, int G) { return launch_time_shard<NX_, NU_>(c, phase, g, G); } \
  int ndlqr_small_slot_##NX_##_##NU_(void) { return (int)ndlqr::RedSlot<NX_>::SIZE; }              \
  int ndlqr_small_multi_##NX_##_##NU_(NdlqrHipCtx* c, int count, const double* rhs, double* zsep, double* fsum, double* ytop, double* z) { return launch_multi_rhs<NX_, NU_>(c, count, rhs, zsep, fsum, ytop, z) ? 1 : 0; }
#include "small_instances.def"
#undef NDLQR_SMALL_INSTANCE
#endif

static const SmallInstance kSmallInstances[] = {
#define NDLQR_SMALL_INSTANCE(NX_, NU_) \
  {NX_, NU_, ndlqr_small_solve_##NX_##_##NU_, ndlqr_small_needs_F_##NX_##_##NU_, ndlqr_small_rhs_##NX_##_##NU_, \
   ndlqr_small_kpb_##NX_##_##NU_, ndlqr_small_tshard_##NX_##_##NU_, ndlqr_small_slot_##NX_##_##NU_,            \
   ndlqr_small_multi_##NX_##_##NU_},
#include "small_instances.def"
#undef NDLQR_SMALL_INSTANCE
};

static bool has_small_instance(int nstates, int ninputs) {
  for (const SmallInstance& s : kSmallInstances)
    if (s.nx == nstates && s.nu == ninputs) return true;
  return false;
}

// cheapest instance (with a matrix-core path: six states or more) that contains the block size; *pn, *pm untouched
// when there is none
static void pick_pad_instance(int nstates, int ninputs, int* pn, int* pm) {
  long best = -1;
  for (const SmallInstance& si : kSmallInstances) {
    if (si.nx < nstates || si.nu < ninputs || si.nx < 6) continue;
    const long cost = (long)si.nx * si.nx * (si.nx + si.nu);
    if (best < 0 || cost < best) { best = cost; *pn = si.nx; *pm = si.nu; }
  }
}

static const SmallInstance* find_small(const ndlqr::Dims& d) {
  for (const SmallInstance& s : kSmallInstances)
    if (s.nx == d.n && s.nu == d.m) return &s;
  return nullptr;
}

// the size-specialised instance that serves this context (nullptr: runtime-sized kernels)
static const SmallInstance* pick_small(const NdlqrHipCtx* c) {
  const ndlqr::Dims& d = c->d;
  if (c->flags & NDLQR_FLAG_GENERIC) return nullptr;
  const SmallInstance* inst = find_small(d);
  // the fused kernels own eight knots per workgroup and two tree levels: shorter horizons run the
  // runtime-sized kernels
  if (!inst || d.N < inst->kpb() || d.K < 3) return nullptr;
  return inst;
}

// returns true when (n, m, N) has a size-specialised instance and it was launched
static bool try_launch_small(NdlqrHipCtx* c, bool strict, int* err) {
  const SmallInstance* inst = pick_small(c);
  if (!inst) return false;
  *err = inst->solve(c, strict, (c->flags & NDLQR_FLAG_KEEP_FACT) != 0);
  return true;
}

// does the launch sequence enqueue_solve is about to issue touch the factor array?
static bool solve_needs_F(const NdlqrHipCtx* c) {
  const SmallInstance* inst = pick_small(c);
  if (!inst) return !plan_reduced_generic(c).ok;  // the other runtime-sized kernels work on F
  return inst->needs_F(c, (c->flags & NDLQR_FLAG_STRICT_FP) != 0, (c->flags & NDLQR_FLAG_KEEP_FACT) != 0) != 0;
}

// NDLQR_FLAG_KEEP_RECORDS where no schedule keeps records -- the knot-based runtime-sized path: blocks beyond 128 states,
// inputs wider than a workgroup -- keeps the factor array instead: the right-hand-side re-solve is the factor-based sweep
static bool records_kept_as_factors(const NdlqrHipCtx* c) {
  return (c->flags & NDLQR_FLAG_KEEP_RECORDS) && !(c->flags & NDLQR_FLAG_STRICT_FP) && !pick_small(c) &&
         !plan_reduced_generic(c).ok;
}

// Enqueue leaf/bottom + per-level + apply launches on the context's stream.
static int enqueue_solve(NdlqrHipCtx* c) {
  const ndlqr::Dims& d = c->d;
  // (the failure counters are cumulative: no memset node; the host subtracts what it has seen)
  const bool strict = (c->flags & NDLQR_FLAG_STRICT_FP) != 0;
  int err = NDLQR_OK;
  bool done = false;
  c->rec_complete = false;
  c->rec_compact = false;
  done = try_launch_small(c, strict, &err);
  if (!done) {
    const ReducedGenericPlan rp = plan_reduced_generic(c);
    const bool lean = !strict && !(c->flags & NDLQR_FLAG_KEEP_FACT) && !records_kept_as_factors(c);
    if (rp.ok) err = launch_reduced_generic(c, rp);
    else err = strict ? launch_generic<true>(c, false) : launch_generic<false>(c, lean);
  }
  // the batch-wide failure count travels to pinned host memory behind the last kernel: the host
  // reads it after the stream synchronisation without another blocking copy
  if (!err) HIP_TRY(hipMemcpyAsync(c->h_fail, c->info + d.batch, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  return err;
}

// first half of a solve: allocations, recovery from a failed solve, choice of the buffer set (the context's
// buffer / stream fields hold it on return). *pipelined_out: this solve runs on the two-deep pipeline.
static int prepare_solve(NdlqrHipCtx* c, bool* pipelined_out) {
  HIP_TRY(hipSetDevice(c->device));
  if (solve_needs_F(c)) {  // before any capture starts: allocation is not a stream operation
    const int ferr = ndlqr_hip_ensure_F(c);
    if (ferr) return ferr;
  }
  const bool red_generic = !pick_small(c) && plan_reduced_generic(c).ok;
  if (!pick_small(c) && !red_generic && !c->sep_scratch) {
    // knot-based runtime-sized path with a block too large for the separator kernel's LDS (launch_generic): S-bar and
    // the panel of every level-0 separator in global memory. Before any capture starts: allocation is not a stream operation.
    const ndlqr::Dims& dd = c->d;
    const size_t per_sep = (size_t)dd.n * (dd.n + 1) + (size_t)dd.n * (2 * dd.n + 1);
    if (plan_generic_sep(c, (c->flags & NDLQR_FLAG_STRICT_FP) != 0).scratch) {
      const size_t bytes = sizeof(double) * per_sep * (size_t)dd.batch * (dd.N / 2 > 0 ? dd.N / 2 : 1);
      if (hipMalloc(&c->sep_scratch, bytes) != hipSuccess) {
        c->sep_scratch = nullptr;
        (void)hipGetLastError();
        g_last_error = "global scratch of the large-block separator kernel does not fit on the device (" +
                       std::to_string(bytes >> 20) + " MiB): use a smaller batch";
        fprintf(stderr, "ndlqr_hip: %s\n", g_last_error.c_str());
        return NDLQR_ERR_INVALID;
      }
    }
  }
  if (c->state_dirty) {
    // the previous solve did not launch or complete: its arrival counters may be odd and its failure
    // words meaningless -- start from zero (the kernels themselves leave both clean)
    (void)sync_all(c);
    if (c->tree_cnt)
      HIP_TRY(hipMemsetAsync(c->tree_cnt, 0, sizeof(int) * (size_t)c->d.batch * (c->d.N / 4), c->stream));
    if (c->alt.tree_cnt)
      HIP_TRY(hipMemsetAsync(c->alt.tree_cnt, 0, sizeof(int) * (size_t)c->d.batch * (c->d.N / 4), c->stream));
    HIP_TRY(hipMemsetAsync(c->info, 0, sizeof(int) * ((size_t)c->d.batch + 1), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *c->h_fail = 0;
    if (c->alt.h_fail) *c->alt.h_fail = 0;
    c->fail_base = 0;
    c->state_dirty = false;
  }
  // Two-deep pipeline: solves that leave nothing behind but records and the solution alternate between two
  // buffer sets, each on its own stream (the other set may still be in flight). Everything else -- factor
  // array, kept records, per-kernel events, a caller-owned stream -- stays stream-ordered on the primary set.
  const bool pipelined = c->pipeline >= 2 && c->own_stream && !solve_needs_F(c) &&
                         !(c->flags & (NDLQR_FLAG_PROFILE | NDLQR_FLAG_KEEP_RECORDS | NDLQR_FLAG_KEEP_FACT));
  const bool want_alt = pipelined && (c->solve_count & 1u) && ensure_alt(c);
  if (!pipelined && c->alt.stream) HIP_TRY(hipStreamSynchronize(c->in_alt ? c->stream : c->alt.stream));
  if (red_generic) {  // (after ensure_alt: both buffer sets get their slots)
    const int rerr = ensure_red_generic(c);
    if (rerr) return rerr;
  }
  if (want_alt != c->in_alt) swap_slot(c);
  ++c->solve_count;
  c->state_dirty = true;  // until this solve is known to have been enqueued completely
  if (pipelined_out) *pipelined_out = pipelined;
  return NDLQR_OK;
}

// second half: the launch sequence (replayed as a hipGraph) on the current buffer set's stream
static int launch_solve(NdlqrHipCtx* c) {
  int err = NDLQR_OK;
  if (c->flags & NDLQR_FLAG_PROFILE) {
    err = enqueue_solve(c);  // per-kernel events need eager launches
  } else {
    // The sequence is a fixed chain of up to 1 + 2K short launches: capture it once as a hipGraph
    // and replay it (launch-bound single solves -- batch 1 -- gain the most).
    const unsigned apply_key = ((unsigned)c->apply_blk0 << 16) | (unsigned)c->apply_nblk;  // (restricted back-substitution of a step)
    const bool stale = !c->graph_exec || c->graph_flags != c->flags || c->graph_stream != c->stream || c->graph_apply != apply_key;
    if (stale) {
      if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
      hipGraph_t graph = nullptr;
      HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      err = enqueue_solve(c);
      hipError_t e = hipStreamEndCapture(c->stream, &graph);
      if (err) { if (graph) (void)hipGraphDestroy(graph); return err; }
      if (e != hipSuccess) return fail("hipStreamEndCapture", e);
      e = hipGraphInstantiate(&c->graph_exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) { c->graph_exec = nullptr; return fail("hipGraphInstantiate", e); }
      c->graph_flags = c->flags;
      c->graph_stream = c->stream;
      c->graph_apply = apply_key;
      c->graph_rec_complete = c->rec_complete;
      c->graph_rec_compact = c->rec_compact;  // what the captured sequence leaves behind
      c->graph_schedule = c->schedule;
    }
    HIP_TRY(hipGraphLaunch(c->graph_exec, c->stream));
    c->rec_complete = c->graph_rec_complete;
    c->rec_compact = c->graph_rec_compact;
    c->schedule = c->graph_schedule;
  }
  if (err) return err;
  HIP_TRY(hipGetLastError());
  note_solution(c);
  // a complete factor array is on the device with KEEP, and on the strict runtime-sized path
  c->fact_valid = (c->flags & NDLQR_FLAG_KEEP_FACT) != 0 || records_kept_as_factors(c) ||
                  ((c->flags & NDLQR_FLAG_GENERIC) && (c->flags & NDLQR_FLAG_STRICT_FP));
  return NDLQR_OK;
}

int ndlqr_hip_solve_async(NdlqrHipCtx* c) {
  if (!c) return NDLQR_ERR_INVALID;
  int err = prepare_solve(c, nullptr);
  if (err) return err;
  err = rhs_make_current(c, 0xFu);  // (this buffer set's copy of the right-hand side may be behind: steps write one set)
  if (err) return err;
  HIP_TRY(hipEventRecord(c->ev_start, c->stream));
  err = launch_solve(c);
  if (err) return err;
  HIP_TRY(hipEventRecord(c->ev_stop, c->stream));
  c->timing_pending = true;
  c->state_dirty = false;
  return NDLQR_OK;
}

// ------------------------------------------------------------------------------ one-shot solve from pinned staging
// The drop-in ndlqr_Solve is a batch of one whose inputs come from the host and whose solution goes back to it on
// every call (src/solve.c:38-201 works on host memory). Doing that with the batch functions costs three blocking
// uploads, a launch, and a blocking download -- four host synchronisations around 40 us of kernels. Here the caller
// packs straight into pinned staging (ndlqr_hip_staged_io), and ndlqr_hip_solve_staged replays ONE captured graph:
// AB, QR, rhs up (copy nodes), the launch chain of the schedule, the solution blocks down; one launch, one
// synchronisation. Stream-ordered on the primary buffer set (pipeline depth 1 from then on).
static size_t staged_doubles(const ndlqr::Dims& u, size_t* oAB, size_t* oQR, size_t* orhs, size_t* oz) {
  const size_t nAB = (size_t)u.batch * u.N * u.n * u.w, nQR = (size_t)u.batch * u.N * u.w, nz = (size_t)u.batch * u.N * u.rows;
  *oAB = 0; *oQR = nAB; *orhs = nAB + nQR; *oz = nAB + nQR + nz;
  return nAB + nQR + 2 * nz;
}

int ndlqr_hip_staged_io(NdlqrHipCtx* c, double** AB, double** QR, double** rhs, double** z) {
  if (!c || !AB || !QR || !rhs || !z) return NDLQR_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  size_t oAB, oQR, orhs, oz;
  const size_t total = staged_doubles(c->du, &oAB, &oQR, &orhs, &oz);
  if (!c->h_io) {
    HIP_TRY(hipHostMalloc((void**)&c->h_io, sizeof(double) * total, hipHostMallocDefault));
    if (c->padded) {  // (caller-layout staging in HBM for both directions; allocated outside any capture)
      const int serr = ensure_pad_stage(c, total);
      if (serr) return serr;
    }
    const int perr = ndlqr_hip_set_pipeline_depth(c, 1);
    if (perr) return perr;
  }
  *AB = c->h_io + oAB; *QR = c->h_io + oQR; *rhs = c->h_io + orhs; *z = c->h_io + oz;
  return NDLQR_OK;
}

// the copies around the launch chain, on the context's stream (captured, or eager under NDLQR_FLAG_PROFILE)
static int enqueue_staged(NdlqrHipCtx* c) {
  const ndlqr::Dims& d = c->d;
  const ndlqr::Dims& u = c->du;
  size_t oAB, oQR, orhs, oz;
  (void)staged_doubles(u, &oAB, &oQR, &orhs, &oz);
  const size_t nAB = oQR, nQR = orhs - oQR, nz = oz - orhs;
  hipStream_t st = c->stream;
  if (c->padded) {
    double* s0 = c->pad_stage;
    HIP_TRY(hipMemcpyAsync(s0, c->h_io, sizeof(double) * (nAB + nQR + nz), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(ndlqr::pad_inputs_generic, dim3(d.N, d.batch), dim3(128), 0, st, u, d, 0, (const double*)s0,
                       (const double*)(s0 + oQR), (const double*)(s0 + orhs), c->AB, c->QR, c->rhs);
    HIP_TRY(hipGetLastError());
  } else {
    HIP_TRY(hipMemcpyAsync(c->AB, c->h_io + oAB, sizeof(double) * nAB, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->QR, c->h_io + oQR, sizeof(double) * nQR, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->rhs, c->h_io + orhs, sizeof(double) * nz, hipMemcpyHostToDevice, st));
  }
  const int err = enqueue_solve(c);
  if (err) return err;
  if (c->padded) {
    hipLaunchKernelGGL(ndlqr::unpad_blocks_generic, dim3(d.N * d.batch), dim3(64), 0, st, u, d, (const double*)c->z,
                       c->pad_stage + oz);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_io + oz, c->pad_stage + oz, sizeof(double) * nz, hipMemcpyDeviceToHost, st));
  } else {
    HIP_TRY(hipMemcpyAsync(c->h_io + oz, c->z, sizeof(double) * nz, hipMemcpyDeviceToHost, st));
  }
  return NDLQR_OK;
}

int ndlqr_hip_solve_staged(NdlqrHipCtx* c) {
  if (!c || !c->h_io) return NDLQR_ERR_INVALID;
  if (c->pipeline != 1) {
    const int perr = ndlqr_hip_set_pipeline_depth(c, 1);
    if (perr) return perr;
  }
  int err = prepare_solve(c, nullptr);  // (allocations, recovery from a failed solve; depth 1: the primary set)
  if (err) return err;
  rhs_written_cur(c, 0xFu);
  c->fact_valid = false;  // new A, B, Q, R: neither a cached factor array nor cached records match
  c->rec_complete = false;
  HIP_TRY(hipEventRecord(c->ev_start, c->stream));
  if (c->flags & NDLQR_FLAG_PROFILE) {
    err = enqueue_staged(c);  // per-kernel events need eager launches
    if (err) return err;
  } else {
    const bool stale = !c->graph_staged || c->graph_staged_flags != c->flags;
    if (stale) {
      if (c->graph_staged) { (void)hipGraphExecDestroy(c->graph_staged); c->graph_staged = nullptr; }
      hipGraph_t graph = nullptr;
      HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      err = enqueue_staged(c);
      hipError_t e = hipStreamEndCapture(c->stream, &graph);
      if (err) { if (graph) (void)hipGraphDestroy(graph); return err; }
      if (e != hipSuccess) return fail("hipStreamEndCapture", e);
      e = hipGraphInstantiate(&c->graph_staged, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) { c->graph_staged = nullptr; return fail("hipGraphInstantiate", e); }
      c->graph_staged_flags = c->flags;
      c->graph_rec_complete = c->rec_complete;
      c->graph_rec_compact = c->rec_compact;
      c->graph_schedule = c->schedule;
    }
    HIP_TRY(hipGraphLaunch(c->graph_staged, c->stream));
    c->rec_complete = c->graph_rec_complete;
    c->rec_compact = c->graph_rec_compact;
    c->schedule = c->graph_schedule;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev_stop, c->stream));
  note_solution(c);
  c->fact_valid = (c->flags & NDLQR_FLAG_KEEP_FACT) != 0 || records_kept_as_factors(c) ||
                  ((c->flags & NDLQR_FLAG_GENERIC) && (c->flags & NDLQR_FLAG_STRICT_FP));
  c->timing_pending = true;
  c->state_dirty = false;
  return ndlqr_hip_synchronize(c);
}

// ------------------------------------------------------------------------------ time-axis sharding
// One problem (or a small batch) over G ranks along the horizon: launch_time_shard (launch_small.hpp) says what the
// phases do. The G - 1 top slots of every problem travel packed as [G - 1][batch][slot doubles].
static const SmallInstance* time_shard_instance(NdlqrHipCtx* c, int G) {
  if (!c || G < 2 || c->padded) return nullptr;
  const SmallInstance* inst = pick_small(c);
  if (!inst || c->d.N % G || (c->d.N / G) < 16) return nullptr;
  return inst;
}

int ndlqr_hip_time_shard_top_doubles(NdlqrHipCtx* c, int G) {
  const SmallInstance* inst = time_shard_instance(c, G);
  return inst ? (G - 1) * c->d.batch * inst->slot() : NDLQR_ERR_INVALID;
}

static int time_shard_copy_slots(NdlqrHipCtx* c, int G, double* buf, bool to_buf) {
  const SmallInstance* inst = time_shard_instance(c, G);
  if (!inst || !buf) return NDLQR_ERR_INVALID;
  const ndlqr::Dims& d = c->d;
  const size_t slot = (size_t)inst->slot();
  const size_t pitch_red = sizeof(double) * (size_t)(d.N >> 2) * slot, width = sizeof(double) * slot;
  HIP_TRY(hipSetDevice(c->device));
  for (int j = 1; j < G; ++j) {
    const int s = j * (d.N / G) - 1;
    double* p = c->red + (size_t)(s >> 2) * slot;
    double* q = buf + (size_t)(j - 1) * d.batch * slot;
    if (to_buf) HIP_TRY(hipMemcpy2DAsync(q, width, p, pitch_red, width, (size_t)d.batch, hipMemcpyDefault, c->stream));
    else HIP_TRY(hipMemcpy2DAsync(p, pitch_red, q, width, width, (size_t)d.batch, hipMemcpyDefault, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return NDLQR_OK;
}
int ndlqr_hip_time_shard_export(NdlqrHipCtx* c, int G, double* buf) { return time_shard_copy_slots(c, G, buf, true); }
int ndlqr_hip_time_shard_import(NdlqrHipCtx* c, int G, const double* buf) {
  return time_shard_copy_slots(c, G, const_cast<double*>(buf), false);
}

static int time_shard_phase(NdlqrHipCtx* c, int phase, int g, int G) {
  const SmallInstance* inst = time_shard_instance(c, G);
  if (!inst) {
    g_last_error = "time-axis sharding: needs a size-specialised block size with a matrix-core instance, G a power of two, "
                   "N / G >= 16";
    return NDLQR_ERR_INVALID;
  }
  if (c->flags & ~NDLQR_FLAG_PROFILE) { g_last_error = "time-axis sharding runs the default fast mode only"; return NDLQR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  if (phase == 0) {
    if (c->pipeline != 1 || c->in_alt) {
      const int perr = ndlqr_hip_set_pipeline_depth(c, 1);  // stream-ordered on the primary buffer set
      if (perr) return perr;
    }
    const int merr = rhs_make_current(c, 0xFu);
    if (merr) return merr;
    HIP_TRY(hipEventRecord(c->ev_start, c->stream));
  }
  const int err = inst->tshard(c, phase, g, G);
  if (err) {
    if (g_last_error.empty() || err == NDLQR_ERR_INVALID) g_last_error = "time-axis sharding: horizon / chunk not supported by this instance";
    return err;
  }
  HIP_TRY(hipGetLastError());
  if (phase == 1) {
    HIP_TRY(hipMemcpyAsync(c->h_fail, c->info + c->d.batch, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->ev_stop, c->stream));
    c->timing_pending = true;
    note_solution(c);
    c->fact_valid = false;
    c->rec_complete = false;
  }
  return NDLQR_OK;
}
int ndlqr_hip_time_shard_factor(NdlqrHipCtx* c, int g, int G) { return time_shard_phase(c, 0, g, G); }
int ndlqr_hip_time_shard_finish(NdlqrHipCtx* c, int g, int G) { return time_shard_phase(c, 1, g, G); }

// transfer staging of the current buffer set: max(flat right-hand side, packed solutions) doubles
static int ensure_xfer(NdlqrHipCtx* c) {
  if (c->xfer) return NDLQR_OK;
  const ndlqr::Dims& d = c->du;  // (caller-layout data: flat right-hand side going up, packed solutions coming down)
  HIP_TRY(hipMalloc(&c->xfer, sizeof(double) * ((size_t)d.batch * d.N * d.rows + (size_t)d.batch * d.n)));
  return NDLQR_OK;
}

// address under which kernels of device `device` read `p` directly: pinned host memory (hipHostMalloc / hipHostRegister)
// through its device-side view, memory of that device as it is; else (pageable memory, another device's) null
static const double* pinned_device_view(const double* p, int device) {
  if (!p) return nullptr;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (a.type == hipMemoryTypeDevice) return a.device == device ? p : nullptr;
  if (a.type != hipMemoryTypeHost) return nullptr;
  void* dv = nullptr;
  if (hipHostGetDevicePointer(&dv, const_cast<double*>(p), 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return static_cast<const double*>(dv);
}

static bool try_launch_rhs_records(NdlqrHipCtx* c);  // below

int ndlqr_hip_step_async(NdlqrHipCtx* c, const double* q, const double* r, const double* dd, const double* x0,
                         double* soln) {
  if (!c || !x0 || !soln) return NDLQR_ERR_INVALID;
  const ndlqr::Dims& d = c->d;
  int err = prepare_solve(c, nullptr);
  if (err) return err;
  err = ensure_xfer(c);  // (before anything is captured: allocation is not a stream operation)
  if (err) return err;
  // the parts of the right-hand side this step does not replace: this buffer set's copy of them may be behind the other's
  const unsigned written = (q ? 1u : 0u) | (r ? 2u : 0u) | (dd ? 4u : 0u) | 8u;
  err = rhs_make_current(c, ~written & 0xFu);
  if (err) return err;
  // Everything of this step is ordered on the stream of its buffer set. Right-hand side: a streaming kernel reads
  // pinned host arrays over the host link directly; pageable ones go through the staging first (the runtime stages
  // those copies itself and blocks). Solutions: pack kernel, then ONE copy-engine transfer. With the two-deep
  // pipeline the transfers of one step run beside the kernels of the other set's step; otherwise (factor array /
  // records kept, caller-owned stream, depth 1) the steps are simply stream-ordered.
  hipStream_t st = c->stream;
  HIP_TRY(hipEventRecord(c->ev_start, st));
  const ndlqr::Dims& u = c->du;
  const size_t nq = (size_t)u.batch * u.N * u.n, nr = (size_t)u.batch * u.N * u.m, nx = (size_t)u.batch * u.n;
  const double* src[4] = {q, r, dd, x0};
  const size_t cnt[4] = {nq, nr, nq, nx};
  const double* view[4];
  double* stage = c->xfer;
  for (int k = 0; k < 4; ++k) {
    view[k] = pinned_device_view(src[k], c->device);
    if (src[k] && !view[k]) {
      HIP_TRY(hipMemcpyAsync(stage, src[k], sizeof(double) * cnt[k], hipMemcpyDefault, st));
      view[k] = stage;
    }
    stage += cnt[k];
  }
  hipLaunchKernelGGL(ndlqr::pack_rhs_stream_generic, dim3(512), dim3(256), 0, st, u, d, view[0], view[1], view[2],
                     view[3], c->rhs);
  HIP_TRY(hipGetLastError());
  rhs_written_cur(c, written);
  // NDLQR_SOLN_ONLY: nothing but the selected knots is wanted -- the last launch of the back-substitution runs the
  // workgroups (eight knots each) that hold them (launch_small.hpp; schedules without that launch compute everything)
  struct ApplyRange {
    NdlqrHipCtx* c;
    ApplyRange(NdlqrHipCtx* c_) : c(c_) {
      if (c->sel_nknots > 0 && (c->sel_blocks & 8u)) {
        c->apply_blk0 = c->sel_knot0 >> 3;
        c->apply_nblk = ((c->sel_knot0 + c->sel_nknots - 1) >> 3) - c->apply_blk0 + 1;
      }
    }
    ~ApplyRange() { c->apply_blk0 = c->apply_nblk = 0; }
  } apply_range(c);
  // A step never changes A, B, Q, R. Under NDLQR_FLAG_KEEP_RECORDS the first step (or a solve before it) leaves the
  // compact records of the default schedule, and every further step is the right-hand-side re-solve on them (rb_forward,
  // rb_forward_top, rb_backsub: 0.46 instead of 0.59 ms per (12,4,256) x 1024) -- until new inputs are uploaded, which
  // clears rec_complete. Stream-ordered on the primary buffer set like every solve with that flag.
  // (the runtime-sized separator-only schedule likewise where its re-solve is the faster one -- beyond 32 states: 8.0
  //  against 11.5 ms at (64,16,512) x 256, 4.8 against 6.4 at (48,16,512) x 256; at (32,8) the two are equal and at
  //  (16,4,256) x 1024 the re-solve takes 2.8 ms against 1.9 for factor + solve, profiles/r04_mpc_steps.txt. The
  //  full-record form of the small shapes -- tree schedule -- re-solves no faster than it factors and keeps factoring.)
  const bool generic_records = !pick_small(c) && c->d.n > 32;
  if ((c->flags & NDLQR_FLAG_KEEP_RECORDS) && !(c->flags & (NDLQR_FLAG_STRICT_FP | NDLQR_FLAG_KEEP_FACT)) && c->rec_complete &&
      (c->rec_compact || generic_records) && !c->in_alt && try_launch_rhs_records(c)) {
    HIP_TRY(hipGetLastError());
    note_solution(c);
    c->schedule = generic_records ? "generic-reduced-records (re-solve)" : "reduced-compact-records (re-solve)";
  } else {
    err = launch_solve(c);
    if (err) return err;
  }
  // (the staging has been consumed by the pack kernel: it now takes the packed solutions -- all of them, or the slice
  //  chosen with ndlqr_hip_set_step_selection)
  //  chosen with ndlqr_hip_set_step_selection); a `soln` in this device's memory is written by the pack kernel itself
  double* packed = c->xfer;
  {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, soln) == hipSuccess && a.type == hipMemoryTypeDevice && a.device == c->device) packed = soln;
    else (void)hipGetLastError();
  }
  if (c->sel_nknots > 0) {
    const size_t width = ((c->sel_blocks & 1u) ? u.n : 0) + ((c->sel_blocks & 2u) ? u.n : 0) + ((c->sel_blocks & 4u) ? u.m : 0);
    hipLaunchKernelGGL(ndlqr::pack_selection_generic, dim3(c->sel_nknots, d.batch), dim3(64), 0, st, u, d, c->sel_knot0,
                       c->sel_nknots, c->sel_blocks & 7u, (const double*)c->z, packed);
    HIP_TRY(hipGetLastError());
    if (packed != soln)
      HIP_TRY(hipMemcpyAsync(soln, packed, sizeof(double) * width * c->sel_nknots * d.batch, hipMemcpyDefault, st));
  } else {
    hipLaunchKernelGGL(ndlqr::pack_solutions_generic, dim3(ndlqr::pack_solutions_chunks(u), d.batch), dim3(256), 0, st, u, d, c->z, packed);
    HIP_TRY(hipGetLastError());
    const size_t nvars = (size_t)u.rows * u.N - u.m;
    if (packed != soln) HIP_TRY(hipMemcpyAsync(soln, packed, sizeof(double) * nvars * d.batch, hipMemcpyDefault, st));
  }
  HIP_TRY(hipEventRecord(c->ev_stop, st));
  HIP_TRY(hipEventRecord(c->ev_step[c->step_count & 1u], st));
  c->step_set[c->step_count & 1u] = c->in_alt ? 1 : 0;
  ++c->step_count;
  c->timing_pending = true;
  c->state_dirty = false;
  return NDLQR_OK;
}

// Factor + solve of the resident problems, of which knots [knot0, knot0 + nknots) -- blocks `blocks` -- are computed by the
// last launch and written to `out` ([batch][nknots][width]; host, pinned or this device's memory), asynchronously: the
// solve of a loop that replaces A, B, Q, R as well (ndlqr_hip_pack_flat_device / uploads) and consumes u of knot 0.
// Consecutive calls alternate between the buffer sets like ndlqr_hip_solve_async; complete after ndlqr_hip_synchronize.
int ndlqr_hip_solve_slices_async(NdlqrHipCtx* c, int knot0, int nknots, unsigned blocks, double* out) {
  if (!c || !out || knot0 < 0 || nknots <= 0 || knot0 + nknots > c->d.N || !(blocks & 7u) || (blocks & ~15u))
    return NDLQR_ERR_INVALID;
  const ndlqr::Dims& d = c->d;
  const ndlqr::Dims& u = c->du;
  int err = prepare_solve(c, nullptr);
  if (err) return err;
  err = ensure_xfer(c);
  if (err) return err;
  err = rhs_make_current(c, 0xFu);
  if (err) return err;
  hipStream_t st = c->stream;
  HIP_TRY(hipEventRecord(c->ev_start, st));
  c->apply_blk0 = knot0 >> 3;
  c->apply_nblk = ((knot0 + nknots - 1) >> 3) - c->apply_blk0 + 1;
  err = launch_solve(c);
  c->apply_blk0 = c->apply_nblk = 0;
  if (err) return err;
  double* packed = c->xfer;
  {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, out) == hipSuccess && a.type == hipMemoryTypeDevice && a.device == c->device) packed = out;
    else (void)hipGetLastError();
  }
  const size_t width = ((blocks & 1u) ? u.n : 0) + ((blocks & 2u) ? u.n : 0) + ((blocks & 4u) ? u.m : 0);
  hipLaunchKernelGGL(ndlqr::pack_selection_generic, dim3(nknots, d.batch), dim3(64), 0, st, u, d, knot0, nknots, blocks & 7u,
                     (const double*)c->z, packed);
  HIP_TRY(hipGetLastError());
  if (packed != out) HIP_TRY(hipMemcpyAsync(out, packed, sizeof(double) * width * nknots * d.batch, hipMemcpyDefault, st));
  HIP_TRY(hipEventRecord(c->ev_stop, st));
  c->timing_pending = true;
  c->state_dirty = false;
  return NDLQR_OK;
}

// the step before the most recent one is complete (its `soln` may be read) -- whichever stream it ran on
int ndlqr_hip_synchronize_previous(NdlqrHipCtx* c) {
  if (!c) return NDLQR_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  if (c->step_count < 2) return NDLQR_OK;
  const unsigned slot = c->step_count & 1u;  // the step before the most recent one: step_count - 2
  HIP_TRY(hipEventSynchronize(c->ev_step[slot]));
  // The failure word of that step's buffer set holds the cumulative count of non-positive pivots as of the end of its
  // solve: anything beyond what has been reported so far belongs to it (or to a step before it).
  const bool cur_is_alt = c->in_alt;
  const int* word = (c->step_set[slot] == (cur_is_alt ? 1 : 0)) ? c->h_fail : c->alt.h_fail;
  if (word && *word > c->fail_base) {
    c->last_failures = *word - c->fail_base;
    c->fail_base = *word;
    return NDLQR_ERR_NOT_SPD;
  }
  return NDLQR_OK;
}

// What a step of ndlqr_hip_step_async brings down: knots [knot0, knot0 + nknots) of every problem, of each knot the
// blocks of `blocks` (NDLQR_SOLN_LAMBDA | NDLQR_SOLN_STATE | NDLQR_SOLN_INPUT), packed [batch][nknots][width].
// nknots == 0: every solution, [batch][nvars] (the default). The reference hands back the whole vector
// (src/solve.c:192-201); an MPC loop consumes u of knot 0.
int ndlqr_hip_set_step_selection(NdlqrHipCtx* c, int knot0, int nknots, unsigned blocks) {
  if (!c) return NDLQR_ERR_INVALID;
  if (nknots == 0) { c->sel_knot0 = 0; c->sel_nknots = 0; c->sel_blocks = 7u; return NDLQR_OK; }
  if (knot0 < 0 || nknots < 0 || knot0 + nknots > c->d.N || !(blocks & 7u) || (blocks & ~15u)) return NDLQR_ERR_INVALID;
  c->sel_knot0 = knot0; c->sel_nknots = nknots; c->sel_blocks = blocks;
  return NDLQR_OK;
}

// the same slice of the most recent solve, synchronously: out = [batch][nknots][width] doubles
int ndlqr_hip_download_selection(NdlqrHipCtx* c, int knot0, int nknots, unsigned blocks, double* out) {
  if (!c || !out || knot0 < 0 || nknots <= 0 || knot0 + nknots > c->d.N || !(blocks & 7u) || (blocks & ~7u))
    return NDLQR_ERR_INVALID;
  const ndlqr::Dims& d = c->d;
  const ndlqr::Dims& u = c->du;
  if (c->z_partial && (knot0 < 8 * c->z_blk0 || knot0 + nknots > 8 * (c->z_blk0 + c->z_nblk)))
    return need_full_solution(c, "ndlqr_hip_download_selection");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(sync_all(c));
  const int xerr = ensure_xfer(c);
  if (xerr) return xerr;
  const size_t width = ((blocks & 1u) ? u.n : 0) + ((blocks & 2u) ? u.n : 0) + ((blocks & 4u) ? u.m : 0);
  hipLaunchKernelGGL(ndlqr::pack_selection_generic, dim3(nknots, d.batch), dim3(64), 0, c->stream, u, d, knot0, nknots,
                     blocks, c->z_latest ? c->z_latest : (const double*)c->z, c->xfer);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, c->xfer, sizeof(double) * width * nknots * d.batch, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return NDLQR_OK;
}

void* ndlqr_hip_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void ndlqr_hip_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}
// device memory for callers without HIP headers of their own (the arrays a device-resident MPC loop hands to
// ndlqr_hip_step_async), and a synchronous copy in any direction
void* ndlqr_hip_device_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipMalloc(&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void ndlqr_hip_device_free(void* p) {
  if (p) (void)hipFree(p);
}
int ndlqr_hip_copy(void* dst, const void* src, size_t bytes) {
  if (!dst || !src) return NDLQR_ERR_INVALID;
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
  return NDLQR_OK;
}

int ndlqr_hip_upload_rhs(NdlqrHipCtx* c, int p0, int count, const double* rhs) {
  if (!c || !rhs || p0 < 0 || count <= 0 || p0 + count > c->d.batch) return NDLQR_ERR_INVALID;
  const ndlqr::Dims& d = c->d;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(sync_all(c));
  {
    const int merr = rhs_make_current(c, 0xFu);
    if (merr) return merr;
  }
  const size_t sz = (size_t)d.N * d.rows;
  if (c->padded) {
    const size_t uz = (size_t)c->du.N * c->du.rows * count;
    const int serr = ensure_pad_stage(c, uz);
    if (serr) return serr;
    HIP_TRY(hipMemcpyAsync(c->pad_stage, rhs, sizeof(double) * uz, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(ndlqr::pad_inputs_generic, dim3(d.N, count), dim3(128), 0, c->stream, c->du, d, p0,
                       (const double*)nullptr, (const double*)nullptr, (const double*)c->pad_stage, c->AB, c->QR, c->rhs);
    HIP_TRY(hipGetLastError());
  } else {
    HIP_TRY(hipMemcpyAsync(c->rhs + p0 * sz, rhs, sizeof(double) * sz * count, hipMemcpyHostToDevice, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  rhs_written_cur(c, 0xFu);
  next_solve_on_current_set(c);
  return NDLQR_OK;
}

template <bool STRICT>
static void launch_rhs_sweep(NdlqrHipCtx* c) {
  const ndlqr::Dims& d = c->d;
  {
    ScopedSlot t(c, SLOT_LEAF);
    hipLaunchKernelGGL((ndlqr::rhs_leaf_generic<STRICT>), dim3(d.N, d.batch), dim3(64), 0, c->stream, d, c->QR,
                       c->rhs, c->z);
  }
  for (int l = 0; l < d.K; ++l) {
    {
      ScopedSlot t(c, SLOT_SEP);
      const size_t lds_staged = sizeof(double) * ((size_t)d.n * (d.n + 1) + d.n);
      const int staged = lds_staged <= 160 * 1024 ? 1 : 0;
      hipLaunchKernelGGL((ndlqr::rhs_separator_generic<STRICT>), dim3(d.N >> (l + 1), d.batch), dim3(64),
                         staged ? lds_staged : sizeof(double) * (size_t)d.n, c->stream, d, l, c->AB, c->F, c->z, staged);
    }
    {
      ScopedSlot t(c, SLOT_SCHUR);
      const int work = d.N * d.rows;
      hipLaunchKernelGGL((ndlqr::rhs_update_generic<STRICT>), dim3((work + 255) / 256, d.batch), dim3(256), 0,
                         c->stream, d, l, c->F, c->z);
    }
  }
}

static bool try_launch_rhs_records(NdlqrHipCtx* c) {
  const ndlqr::Dims& d = c->d;
  if (!c->rec_complete || (c->flags & NDLQR_FLAG_STRICT_FP)) return false;
  if (!pick_small(c)) {  // runtime-sized separator-only schedule: records + slots + W of every separator
    if (!plan_reduced_generic(c).ok) return false;
    launch_rhs_reduced_generic(c);
    return true;
  }
  if (c->flags & NDLQR_FLAG_GENERIC) return false;
  const SmallInstance* inst = find_small(d);
  if (!inst) return false;
  // (the full-record forms: sweep array of rhs_forward_upper within the default dynamic LDS, backsub_small's K + 4
  //  separators of nx rows in one workgroup; the compact form -- rb_forward / rb_forward_top -- was checked by its plan)
  if (!c->rec_compact && (d.N < 8 || (size_t)(d.N / 8) * d.n * sizeof(double) > 60 * 1024 || (d.K + 4) * inst->nx > 256))
    return false;
  inst->rhs(c);
  return true;
}

int ndlqr_hip_solve_rhs_async(NdlqrHipCtx* c) {
  if (!c) return NDLQR_ERR_INVALID;
  if (!c->fact_valid && !c->rec_complete) {
    g_last_error = "rhs-only solve needs a previous solve with NDLQR_FLAG_KEEP_FACT or "
                   "NDLQR_FLAG_KEEP_RECORDS (cached factorisation)";
    fprintf(stderr, "ndlqr_hip: %s\n", g_last_error.c_str());
    return NDLQR_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(sync_all(c));
  if (c->in_alt) swap_slot(c);  // cached records / factors live in the primary set
  {
    const int merr = rhs_make_current(c, 0xFu);
    if (merr) return merr;
  }
  HIP_TRY(hipEventRecord(c->ev_start, c->stream));
  if (!try_launch_rhs_records(c)) {
    if (!c->fact_valid) {  // records only, but this shape / horizon has no record-based re-solve
      g_last_error = "rhs-only solve: this configuration needs NDLQR_FLAG_KEEP_FACT";
      fprintf(stderr, "ndlqr_hip: %s\n", g_last_error.c_str());
      return NDLQR_ERR_INVALID;
    }
    if (c->flags & NDLQR_FLAG_STRICT_FP) launch_rhs_sweep<true>(c); else launch_rhs_sweep<false>(c);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev_stop, c->stream));
  c->timing_pending = true;
  note_solution(c);
  return NDLQR_OK;
}

// Several right-hand sides per problem against ONE kept factorisation each (SURVEY.md 8(f)-2 "multiple right-hand
// sides"; the reference's NdData holds a single one, src/nddata.h:70-75): nrhs x batch right-hand sides, flat host arrays
// q, d [nrhs][batch][N][n], r [nrhs][batch][N][m], x0 [nrhs][batch][n] in the layout of ndlqr_BatchSetRhsFlat, solutions
// [nrhs][batch][nvars] into `soln`. Needs the compact records of a solve with NDLQR_FLAG_KEEP_RECORDS on the
// level-per-launch schedule (rec_compact). The right-hand sides are solved in chunks of at most 65 535 / batch sets;
// right-hand side j of a chunk reads the inputs and records of problem j % batch (they stay in the caches when batch is
// small: one problem x 1024 right-hand sides moves the right-hand sides and solutions and little else), its z_sep go to
// an array of their own. Blocking; the device time of the kernels alone is what ndlqr_hip_last_solve_ms reports afterwards.
// nknots == 0: whole solution vectors [nrhs][batch][nvars]; else knots [knot0, knot0 + nknots), blocks of `blocks`,
// [nrhs][batch][nknots][width] -- and only the workgroups of the last launch that hold them run (the rest of a vector is
// never produced: nothing keeps these solutions on the device anyway)
static int solve_multi_rhs(NdlqrHipCtx* c, int nrhs, const double* q, const double* r, const double* dd, const double* x0,
                           int knot0, int nknots, unsigned blocks, double* soln) {
  if (!c || nrhs <= 0 || !q || !r || !dd || !x0 || !soln) return NDLQR_ERR_INVALID;
  if (nknots < 0 || knot0 < 0 || knot0 + nknots > c->d.N || (nknots > 0 && (!(blocks & 7u) || (blocks & ~15u)))) return NDLQR_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(sync_all(c));
  if (c->in_alt) swap_slot(c);
  const SmallInstance* inst = pick_small(c);
  if (!inst || !c->rec_complete || !c->rec_compact) {
    g_last_error = "multiple right-hand sides need the compact records of a solve with NDLQR_FLAG_KEEP_RECORDS on a "
                   "size-specialised shape (level-per-launch schedule: batch x N / 4 > 2048, or NDLQR_TREE=0)";
    fprintf(stderr, "ndlqr_hip: %s\n", g_last_error.c_str());
    return NDLQR_ERR_INVALID;
  }
  const ndlqr::Dims& d = c->d;
  const ndlqr::Dims& u = c->du;
  const size_t nvars = (size_t)u.rows * u.N - u.m;
  // sets of right-hand sides per chunk: the chunk's count rides on gridDim.y, and its buffers stay within ~2 GB
  size_t per_set = (size_t)d.batch;
  size_t sets = 65535 / per_set;
  const size_t bytes_per = sizeof(double) * (size_t)d.N * (2 * d.rows + d.n + 2 * u.rows);
  while (sets > 1 && sets * per_set * bytes_per > ((size_t)2 << 30)) sets >>= 1;
  if (sets > (size_t)nrhs) sets = (size_t)nrhs;
  if (sets == 0) return NDLQR_ERR_INVALID;
  const size_t cap = sets * per_set;
  if (c->multi_cap < cap) {
    (void)hipFree(c->multi_rhs); (void)hipFree(c->multi_z); (void)hipFree(c->multi_zsep); (void)hipFree(c->multi_fsum);
    (void)hipFree(c->multi_ytop); (void)hipFree(c->multi_in); (void)hipFree(c->multi_out);
    c->multi_rhs = c->multi_z = c->multi_zsep = c->multi_fsum = c->multi_ytop = c->multi_in = c->multi_out = nullptr;
    c->multi_cap = 0;
    const size_t nz = cap * d.N * d.rows;
    const size_t nin = cap * ((size_t)u.N * (2 * u.n + u.m) + u.n);
    bool ok = hipMalloc(&c->multi_rhs, sizeof(double) * nz) == hipSuccess &&
              hipMalloc(&c->multi_z, sizeof(double) * nz) == hipSuccess &&
              hipMalloc(&c->multi_zsep, sizeof(double) * cap * d.N * d.n) == hipSuccess &&
              hipMalloc(&c->multi_fsum, sizeof(double) * cap * (d.N / 8) * 2 * d.n) == hipSuccess &&
              hipMalloc(&c->multi_ytop, sizeof(double) * cap * (d.N / 8) * d.n) == hipSuccess &&
              hipMalloc(&c->multi_in, sizeof(double) * nin) == hipSuccess &&
              hipMalloc(&c->multi_out, sizeof(double) * cap * nvars) == hipSuccess;
    // (padded shapes: the pad entries of the right-hand side are zero and stay zero -- the pack kernel never touches them)
    ok = ok && hipMemsetAsync(c->multi_rhs, 0, sizeof(double) * nz, c->stream) == hipSuccess &&
         hipMemsetAsync(c->multi_z, 0, sizeof(double) * nz, c->stream) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      g_last_error = "buffers of the multiple right-hand sides do not fit on the device";
      return NDLQR_ERR_INVALID;
    }
    c->multi_cap = cap;
  }
  double total_ms = 0.0;
  for (size_t s0 = 0; s0 < (size_t)nrhs; s0 += sets) {
    const size_t ns = (size_t)nrhs - s0 < sets ? (size_t)nrhs - s0 : sets, count = ns * per_set;
    ndlqr::Dims uc = u, dc = d;
    uc.batch = dc.batch = (int)count;  // (the pack kernels index problems by their position alone)
    const size_t nq = count * u.N * u.n, nr = count * u.N * u.m, nx = count * u.n;
    double* in = c->multi_in;
    HIP_TRY(hipMemcpyAsync(in, q + s0 * per_set * u.N * u.n, sizeof(double) * nq, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(in + nq, r + s0 * per_set * u.N * u.m, sizeof(double) * nr, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(in + nq + nr, dd + s0 * per_set * u.N * u.n, sizeof(double) * nq, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(in + 2 * nq + nr, x0 + s0 * per_set * u.n, sizeof(double) * nx, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(ndlqr::pack_rhs_stream_generic, dim3(512), dim3(256), 0, c->stream, uc, dc, (const double*)in,
                       (const double*)(in + nq), (const double*)(in + nq + nr), (const double*)(in + 2 * nq + nr), c->multi_rhs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_start, c->stream));
    if (nknots > 0) { c->apply_blk0 = knot0 >> 3; c->apply_nblk = ((knot0 + nknots - 1) >> 3) - c->apply_blk0 + 1; }
    const bool launched = inst->multi(c, (int)count, c->multi_rhs, c->multi_zsep, c->multi_fsum, c->multi_ytop, c->multi_z);
    c->apply_blk0 = c->apply_nblk = 0;
    if (!launched) {
      g_last_error = "multiple right-hand sides: this shape / horizon has no such form";
      return NDLQR_ERR_INVALID;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_stop, c->stream));
    if (nknots > 0) {
      const size_t width = ((blocks & 1u) ? u.n : 0) + ((blocks & 2u) ? u.n : 0) + ((blocks & 4u) ? u.m : 0);
      hipLaunchKernelGGL(ndlqr::pack_selection_generic, dim3(nknots, (unsigned)count), dim3(64), 0, c->stream, uc, dc, knot0,
                         nknots, blocks & 7u, (const double*)c->multi_z, c->multi_out);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(soln + s0 * per_set * nknots * width, c->multi_out, sizeof(double) * count * nknots * width,
                             hipMemcpyDeviceToHost, c->stream));
    } else {
      hipLaunchKernelGGL(ndlqr::pack_solutions_generic, dim3(ndlqr::pack_solutions_chunks(uc), (unsigned)count), dim3(256), 0, c->stream, uc, dc,
                         (const double*)c->multi_z, c->multi_out);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(soln + s0 * per_set * nvars, c->multi_out, sizeof(double) * count * nvars, hipMemcpyDeviceToHost,
                             c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) == hipSuccess) total_ms += ms;
  }
  c->last_ms = total_ms;
  c->timing_pending = false;
  return NDLQR_OK;
}
int ndlqr_hip_solve_multi_rhs(NdlqrHipCtx* c, int nrhs, const double* q, const double* r, const double* dd,
                              const double* x0, double* soln) {
  return solve_multi_rhs(c, nrhs, q, r, dd, x0, 0, 0, 7u, soln);
}
int ndlqr_hip_solve_multi_rhs_slices(NdlqrHipCtx* c, int nrhs, const double* q, const double* r, const double* dd,
                                     const double* x0, int knot0, int nknots, unsigned blocks, double* out) {
  if (nknots <= 0) return NDLQR_ERR_INVALID;
  return solve_multi_rhs(c, nrhs, q, r, dd, x0, knot0, nknots, blocks, out);
}

int ndlqr_hip_synchronize(NdlqrHipCtx* c) {
  if (!c) return NDLQR_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  {
    const hipError_t se = sync_all(c);
    if (se != hipSuccess) { c->state_dirty = true; return fail("hipStreamSynchronize", se); }
  }
  if (c->timing_pending) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev_start, c->ev_stop));
    c->last_ms = ms;
    c->timing_pending = false;
    // info[batch] = cumulative batch-wide count of non-positive pivots: failures since the last synchronisation
    int seen = *c->h_fail;  // the counter is cumulative: the larger of the two slots' copies is the newer one
    if (c->alt.h_fail && *c->alt.h_fail > seen) seen = *c->alt.h_fail;
    c->last_failures = seen - c->fail_base;
    c->fail_base = seen;
  }
  for (auto& p : c->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
      c->slot_ms[p.slot] += ms;
      c->slot_launches[p.slot] += 1;
    }
    c->event_pool.push_back(p.start);
    c->event_pool.push_back(p.stop);
  }
  c->pending.clear();
  return NDLQR_OK;
}

double ndlqr_hip_last_solve_ms(NdlqrHipCtx* c) { return c ? c->last_ms : -1.0; }
int ndlqr_hip_cholesky_failures(NdlqrHipCtx* c) { return c ? c->last_failures : NDLQR_ERR_INVALID; }

int ndlqr_hip_profile_slots(NdlqrHipCtx* c) { return c ? (int)SLOT_COUNT : 0; }
int ndlqr_hip_profile_get(NdlqrHipCtx* c, int slot, char* name, int name_cap, double* total_ms, int* launches) {
  if (!c || slot < 0 || slot >= SLOT_COUNT) return NDLQR_ERR_INVALID;
  if (name && name_cap > 0) { strncpy(name, kSlotNames[slot], (size_t)name_cap - 1); name[name_cap - 1] = '\0'; }
  if (total_ms) *total_ms = c->slot_ms[slot];
  if (launches) *launches = c->slot_launches[slot];
  return NDLQR_OK;
}
int ndlqr_hip_profile_reset(NdlqrHipCtx* c) {
  if (!c) return NDLQR_ERR_INVALID;
  memset(c->slot_ms, 0, sizeof(c->slot_ms));
  memset(c->slot_launches, 0, sizeof(c->slot_launches));
  return NDLQR_OK;
}

// ------------------------------------------------------------------------------ downloads

// is `p` pinned (hipHostMalloc / hipHostRegister) host memory?
static bool host_ptr_is_pinned(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // an ordinary malloc'ed pointer is "invalid value" to older runtimes
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

// Solutions of problems [p0, p0 + count) as [count][nvars]: a pack kernel gathers them into the transfer staging
// (the device layout carries the unused trailing input slot of every problem, src/solver.c:64), then ONE contiguous
// copy brings them down -- straight into `soln` when that is pinned memory (ndlqr_hip_host_alloc), through two
// pinned 8 MB bounce buffers otherwise (the copy of chunk i overlaps the host memcpy of chunk i-1). The strided
// hipMemcpy2D into pageable memory this replaces ran at 5.4 GB/s.
int ndlqr_hip_download_solutions(NdlqrHipCtx* c, int p0, int count, double* soln) {
  if (!c || !soln || p0 < 0 || count <= 0 || p0 + count > c->d.batch) return NDLQR_ERR_INVALID;
  if (c->z_partial) return need_full_solution(c, "ndlqr_hip_download_solutions");
  const ndlqr::Dims& d = c->d;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(sync_all(c));
  const int xerr = ensure_xfer(c);
  if (xerr) return xerr;
  const double* zl = c->z_latest ? c->z_latest : c->z;
  const size_t nvars = (size_t)c->du.rows * d.N - c->du.m, pitch = (size_t)d.rows * d.N;
  hipStream_t st = c->stream;
  hipLaunchKernelGGL(ndlqr::pack_solutions_generic, dim3(ndlqr::pack_solutions_chunks(c->du), count), dim3(256), 0, st, c->du, d, zl + p0 * pitch, c->xfer);
  HIP_TRY(hipGetLastError());
  const size_t total = nvars * count;
  if (host_ptr_is_pinned(soln)) {
    HIP_TRY(hipMemcpyAsync(soln, c->xfer, sizeof(double) * total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NDLQR_OK;
  }
  const size_t chunk = (8u << 20) / sizeof(double);
  if (total <= chunk / 8) {  // small: one synchronous copy (the runtime stages it)
    HIP_TRY(hipMemcpyAsync(soln, c->xfer, sizeof(double) * total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NDLQR_OK;
  }
  for (int i = 0; i < 2; ++i)
    if (!c->h_stage[i]) HIP_TRY(hipHostMalloc((void**)&c->h_stage[i], sizeof(double) * chunk, hipHostMallocDefault));
  hipEvent_t done[2] = {take_event(c), take_event(c)};
  const size_t nchunks = (total + chunk - 1) / chunk;
  hipError_t e = hipSuccess;
  for (size_t i = 0; i <= nchunks && e == hipSuccess; ++i) {
    if (i < nchunks) {  // (buffer i & 1 held chunk i - 2, which the previous iteration copied out)
      const size_t off = i * chunk, len = total - off < chunk ? total - off : chunk;
      e = hipMemcpyAsync(c->h_stage[i & 1], c->xfer + off, sizeof(double) * len, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipEventRecord(done[i & 1], st);
    }
    if (i > 0 && e == hipSuccess) {
      const size_t off = (i - 1) * chunk, len = total - off < chunk ? total - off : chunk;
      e = hipEventSynchronize(done[(i - 1) & 1]);
      if (e == hipSuccess) memcpy(soln + off, c->h_stage[(i - 1) & 1], sizeof(double) * len);
    }
  }
  c->event_pool.push_back(done[0]);
  c->event_pool.push_back(done[1]);
  if (e != hipSuccess) return fail("ndlqr_hip_download_solutions", e);
  return NDLQR_OK;
}

const char* ndlqr_hip_schedule(const NdlqrHipCtx* c) { return c ? c->schedule : "none"; }

int ndlqr_hip_factors_valid(const NdlqrHipCtx* c) { return c && c->fact_valid ? 1 : 0; }

int ndlqr_hip_pack_solutions_device(NdlqrHipCtx* c, double* dst) {
  if (!c || !dst) return NDLQR_ERR_INVALID;
  if (c->z_partial) return need_full_solution(c, "ndlqr_hip_pack_solutions_device");
  const ndlqr::Dims& d = c->d;
  HIP_TRY(hipSetDevice(c->device));
  // on the stream of the latest solve: ordered behind it, asynchronous for the caller
  hipLaunchKernelGGL(ndlqr::pack_solutions_generic, dim3(ndlqr::pack_solutions_chunks(c->du), d.batch), dim3(256), 0,
                     c->stream_latest ? c->stream_latest : c->stream, c->du, d, c->z_latest ? c->z_latest : c->z, dst);
  HIP_TRY(hipGetLastError());
  return NDLQR_OK;
}

int ndlqr_hip_kkt_residual(NdlqrHipCtx* c, double* res, double* bnorm) {
  if (!c || !res) return NDLQR_ERR_INVALID;
  if (c->z_partial) return need_full_solution(c, "ndlqr_hip_kkt_residual");
  const ndlqr::Dims& d = c->d;
  HIP_TRY(hipSetDevice(c->device));
  if (!c->kkt_out) HIP_TRY(hipMalloc(&c->kkt_out, sizeof(double) * 2 * (size_t)d.batch));
  double* out = c->kkt_out;
  HIP_TRY(sync_all(c));
  hipLaunchKernelGGL(ndlqr::kkt_residual_generic, dim3(d.batch), dim3(256), 0, c->stream, d, c->AB, c->QR, c->rhs,
                     c->z_latest ? c->z_latest : c->z, out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(res, out, sizeof(double) * d.batch, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess && bnorm)
    e = hipMemcpyAsync(bnorm, out + d.batch, sizeof(double) * d.batch, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail("ndlqr_hip_kkt_residual", e);
  return NDLQR_OK;
}

int ndlqr_hip_download_rhs_blocks(NdlqrHipCtx* c, int p, double* z_full) {
  if (!c || !z_full || p < 0 || p >= c->d.batch) return NDLQR_ERR_INVALID;
  if (c->z_partial) return need_full_solution(c, "ndlqr_hip_download_rhs_blocks");
  const ndlqr::Dims& d = c->d;
  HIP_TRY(hipSetDevice(c->device));
  const size_t pitch = (size_t)d.rows * d.N;
  HIP_TRY(sync_all(c));
  const double* zp = (c->z_latest ? c->z_latest : c->z) + p * pitch;
  if (c->padded) {
    const size_t upitch = (size_t)c->du.rows * d.N;
    const int serr = ensure_pad_stage(c, upitch);
    if (serr) return serr;
    hipLaunchKernelGGL(ndlqr::unpad_blocks_generic, dim3(d.N), dim3(64), 0, c->stream, c->du, d, zp, c->pad_stage);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(z_full, c->pad_stage, sizeof(double) * upitch, hipMemcpyDeviceToHost, c->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(z_full, zp, sizeof(double) * pitch, hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return NDLQR_OK;
}

int ndlqr_hip_download_factors(NdlqrHipCtx* c, int p, double* fact) {
  if (!c || !fact || p < 0 || p >= c->d.batch) return NDLQR_ERR_INVALID;
  if (!c->fact_valid) {
    g_last_error = "factor download needs NDLQR_FLAG_KEEP_FACT set before the solve";
    fprintf(stderr, "ndlqr_hip: %s\n", g_last_error.c_str());
    return NDLQR_ERR_INVALID;
  }
  const ndlqr::Dims& d = c->d;
  HIP_TRY(hipSetDevice(c->device));
  const size_t count = (size_t)d.K * d.N * d.fb;
  std::vector<double> tmp(count);
  HIP_TRY(hipMemcpyAsync(tmp.data(), c->F + p * count, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  // device: [level][knot][row][col] row-major -> reference: block (k,level) at (k + N*level)*fb,
  // sub-blocks lambda (n x n), state (n x n), input (m x n), each column-major (src/nddata.c:40-53)
  // (a padded shape: the device blocks have np >= n columns and rows lambda [0, np), state [np, 2 np), input
  //  [2 np, 2 np + mp); the caller gets the real rows and columns)
  const int n = c->du.n, m = c->du.m, np = d.n;
  const size_t ufb = c->du.fb;
  for (int lvl = 0; lvl < d.K; ++lvl)
    for (int k = 0; k < d.N; ++k) {
      const double* src = tmp.data() + ((size_t)lvl * d.N + k) * d.fb;
      double* dst = fact + ((size_t)k + (size_t)d.N * lvl) * ufb;
      for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) dst[i + n * j] = src[i * np + j];
        for (int i = 0; i < n; ++i) dst[n * n + i + n * j] = src[(np + i) * np + j];
        for (int i = 0; i < m; ++i) dst[2 * n * n + i + m * j] = src[(2 * np + i) * np + j];
      }
    }
  return NDLQR_OK;
}

// ------------------------------------------------------------------------------ dense helpers

// Host matrices in, host matrices out (the reference's Matrix* layer, src/linalg.c:55-190). One call = one
// packed H2D copy of the operands, one kernel, one D2H copy of the result, all on the stream of a pooled
// scratch (a device buffer + a pinned staging buffer, grown on demand): no allocation and no blocking
// null-stream copy per call, and calls from different host threads (the reference's tests call these from
// an OpenMP team, test/parallel_test.c:30-239) run side by side on different scratches.
namespace {
struct DenseScratch {
  double* dev = nullptr;
  double* host = nullptr;
  size_t cap = 0;  // doubles
  hipStream_t stream = nullptr;
  int device = -1;  // the device its stream and buffer live on: leased only to calls whose current device is this one
};
std::mutex g_dense_mu;
std::vector<DenseScratch*> g_dense_pool;  // idle scratches; never freed (bounded by the peak concurrency)

struct DenseLease {
  DenseScratch* s = nullptr;
  DenseLease() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    std::lock_guard<std::mutex> lock(g_dense_mu);
    for (size_t i = g_dense_pool.size(); i-- > 0;)
      if (g_dense_pool[i]->device == dev) {
        s = g_dense_pool[i];
        g_dense_pool.erase(g_dense_pool.begin() + (long)i);
        return;
      }
    s = new DenseScratch();
    s->device = dev;
  }
  ~DenseLease() {
    std::lock_guard<std::mutex> lock(g_dense_mu);
    g_dense_pool.push_back(s);
  }
  int ensure(size_t doubles) {
    if (!s->stream) HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    if (doubles <= s->cap) return NDLQR_OK;
    size_t cap = s->cap ? s->cap : 4096;
    while (cap < doubles) cap *= 2;
    if (s->dev) { (void)hipFree(s->dev); s->dev = nullptr; }
    if (s->host) { (void)hipHostFree(s->host); s->host = nullptr; }
    s->cap = 0;
    HIP_TRY(hipMalloc(&s->dev, sizeof(double) * cap));
    HIP_TRY(hipHostMalloc((void**)&s->host, sizeof(double) * cap, hipHostMallocDefault));
    s->cap = cap;
    return NDLQR_OK;
  }
};
}  // namespace

static int dense_ready() {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    g_last_error = "no HIP device for the dense Matrix* helpers (no CPU fallback)";
    fprintf(stderr, "ndlqr_hip: %s\n", g_last_error.c_str());
    return NDLQR_ERR_NO_DEVICE;
  }
  return NDLQR_OK;
}

int ndlqr_hip_gemm(int tA, int tB, int m, int n, int k, double alpha, const double* A, int lda,
                   const double* B, int ldb, double beta, double* C, int ldc) {
  if (dense_ready()) return NDLQR_ERR_NO_DEVICE;
  if (m <= 0 || n <= 0 || k < 0 || !A || !B || !C) return NDLQR_ERR_INVALID;
  // stored shapes: A is (Ar x Ac) with leading dimension lda, ...; the last column ends after its rows
  const int Ar = tA ? k : m, Ac = tA ? m : k, Br = tB ? n : k, Bc = tB ? k : n;
  if (k > 0 && (lda < Ar || ldb < Br)) return NDLQR_ERR_INVALID;
  if (ldc < m) return NDLQR_ERR_INVALID;
  const size_t nA = k > 0 ? (size_t)lda * (Ac - 1) + Ar : 0, nB = k > 0 ? (size_t)ldb * (Bc - 1) + Br : 0,
               nC = (size_t)ldc * (n - 1) + m;
  DenseLease L;
  const int err = L.ensure(nA + nB + nC);
  if (err) return err;
  DenseScratch* s = L.s;
  memcpy(s->host, A, sizeof(double) * nA);
  memcpy(s->host + nA, B, sizeof(double) * nB);
  memcpy(s->host + nA + nB, C, sizeof(double) * nC);
  HIP_TRY(hipMemcpyAsync(s->dev, s->host, sizeof(double) * (nA + nB + nC), hipMemcpyHostToDevice, s->stream));
  const int total = m * n;
  hipLaunchKernelGGL(ndlqr::dense_gemm, dim3((total + 255) / 256), dim3(256), 0, s->stream, tA, tB, m, n, k, alpha,
                     s->dev, lda, s->dev + nA, ldb, beta, s->dev + nA + nB, ldc);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->host + nA + nB, s->dev + nA + nB, sizeof(double) * nC, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  memcpy(C, s->host + nA + nB, sizeof(double) * nC);
  return NDLQR_OK;
}

int ndlqr_hip_potrf_lower(int n, double* A, int lda) {
  if (dense_ready()) return NDLQR_ERR_NO_DEVICE;
  if (n <= 0 || !A) return NDLQR_ERR_INVALID;
  if (lda < n) return NDLQR_ERR_INVALID;
  const size_t nA = (size_t)lda * (n - 1) + n;
  DenseLease L;
  const int err = L.ensure(nA + 1);  // + one slot for the failure word
  if (err) return err;
  DenseScratch* s = L.s;
  memcpy(s->host, A, sizeof(double) * nA);
  s->host[nA] = 0.0;  // (all-zero bits: the int the kernel writes into starts at 0)
  HIP_TRY(hipMemcpyAsync(s->dev, s->host, sizeof(double) * (nA + 1), hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(ndlqr::dense_potrf, dim3(1), dim3(256), 0, s->stream, n, s->dev, lda,
                     reinterpret_cast<int*>(s->dev + nA));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->host, s->dev, sizeof(double) * (nA + 1), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  memcpy(A, s->host, sizeof(double) * nA);
  int info = 0;
  memcpy(&info, s->host + nA, sizeof(int));
  return info ? -1 : 0;
}

// which: 0 = L L' x = b (both substitutions), 1 = L x = b, 2 = L' x = b
static int dense_tri_solve(int which, int n, int nrhs, const double* Lm, int ldl, double* B, int ldb) {
  if (dense_ready()) return NDLQR_ERR_NO_DEVICE;
  if (n <= 0 || nrhs <= 0 || !Lm || !B) return NDLQR_ERR_INVALID;
  if (ldl < n || ldb < n) return NDLQR_ERR_INVALID;
  const size_t nL = (size_t)ldl * (n - 1) + n, nB = (size_t)ldb * (nrhs - 1) + n;
  DenseLease L;
  const int err = L.ensure(nL + nB);
  if (err) return err;
  DenseScratch* s = L.s;
  memcpy(s->host, Lm, sizeof(double) * nL);
  memcpy(s->host + nL, B, sizeof(double) * nB);
  HIP_TRY(hipMemcpyAsync(s->dev, s->host, sizeof(double) * (nL + nB), hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(ndlqr::dense_potrs, dim3((nrhs + 63) / 64), dim3(64), 0, s->stream, n, nrhs, s->dev, ldl,
                     s->dev + nL, ldb, which);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->host + nL, s->dev + nL, sizeof(double) * nB, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  memcpy(B, s->host + nL, sizeof(double) * nB);
  return NDLQR_OK;
}

int ndlqr_hip_potrs_lower(int n, int nrhs, const double* L, int ldl, double* B, int ldb) {
  return dense_tri_solve(0, n, nrhs, L, ldl, B, ldb);
}

int ndlqr_hip_trsv_lower(int n, int nrhs, const double* L, int ldl, double* B, int ldb, int transposed) {
  return dense_tri_solve(transposed ? 2 : 1, n, nrhs, L, ldl, B, ldb);
}

// developer hook (tools/debug_compare.py; not part of include/*.h): raw copy of an internal array,
// which = 0: separator records [batch][N][2 n^2 + n], 1: accumulator slots of the separator-only schedule
extern "C" long ndlqr_hip_debug_download(NdlqrHipCtx* c, int which, double* host, long count) {
  if (!c || !host) return -1;
  const ndlqr::Dims& d = c->d;
  const double* src = which == 0 ? c->rec : c->red;
  const size_t slot_doubles = ((size_t)d.n * (d.n + 1) + 2 * (size_t)d.n * d.n + 2 * d.n + 15) / 16 * 16;  // RedSlot<NX>::SIZE
  const size_t have = which == 0 ? (size_t)d.batch * d.N * (2 * d.n * d.n + d.n) : (size_t)d.batch * (d.N / 4) * slot_doubles;
  if (!src) return -1;
  const size_t n = (size_t)count < have ? (size_t)count : have;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
  if (hipMemcpy(host, src, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (long)n;
}

#ifdef NDLQR_SEGTIME
// developer instrumentation only (tools/segtime.py): read / clear the per-segment cycle sums
extern "C" int ndlqr_hip_debug_segments(unsigned long long* out, int n, int reset) {
  unsigned long long host[128];
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ndlqr::ndlqr_seg), sizeof(host)) != hipSuccess) return -1;
  for (int i = 0; i < n && i < 128; ++i) out[i] = host[i];
  if (reset) {
    for (int i = 0; i < 128; ++i) host[i] = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(ndlqr::ndlqr_seg), host, sizeof(host)) != hipSuccess) return -1;
  }
  return 128;
}
#endif
