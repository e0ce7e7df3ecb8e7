"""Algorithmic bytes and useful fp64 flops of the launch sequences this library implements, per
solve and per kernel kind (the profile slots of include/ndlqr_hip.h) -- the denominators of
bench.py's `roofline` object, derived in DESIGN.md section 4.

"Algorithmic" = what the kernel has to move / compute as designed with perfect caching: every
operand block read once, every result block written once (an atomic add into an accumulator counts
as one write), no tile padding, no redundant lanes. It is NOT the traffic of the reference's dense
level-streaming schedule (SURVEY.md 8(d) model (B), kept below as `model_b_bytes` for the
`vs_level_streaming_model` ratio): the separator-only schedules move ~20x less than that.

Symbols: n states, m inputs, w = n + m, rows = 2n + m, N horizon, K = log2 N.
"""
import math

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6300 achievable)
FP64_PEAK_TFLOPS = 78.6   # dense fp64: matrix-core rate = vector rate on gfx950 (same guide)


def record_doubles(n, level, compact_level0=False):
    """Separator record f_a | f_bb | z_sep (2 n^2 + n). With the compact level-0 records of the
    row-broadcast schedule a level-0 separator keeps only the packed inverse factor (n (n+1) / 2)."""
    if compact_level0 and level == 0:
        return n * (n + 1) // 2
    return 2 * n * n + n


def separator_flops(n, w):
    """Useful flops of eliminating one separator on the reduced system: S-bar = [A|B] D [A|B]' (+ rhs
    column), Cholesky, the solves of the 2n+1 panel columns, and the Schur blocks pushed upwards
    (a'a and bb'bb symmetric, a'bb, two vectors)."""
    leaf = 2 * n * n * w + 2 * n * w           # products of S-bar and of the rhs column
    chol = n ** 3 / 3.0
    solves = 2 * n * n * (2 * n + 1)           # forward + backward substitution, 2n+1 columns
    schur = 4 * n ** 3 + 4 * n * n             # n^3 + n^3 + 2 n^3, plus the two rhs vectors
    return leaf + chol + solves + schur


def backsub_flops(n, m, N):
    """Back-substitution: two n x n matrix-vector products per separator, A'y and B'y per knot."""
    w = n + m
    return (N - 1) * 4 * n * n + N * (2 * n * w + 3 * (2 * n + m))


def inputs_doubles(n, m, N):
    """[A|B], diag(Q), diag(R) and the raw right-hand side of every knot."""
    return N * (n * (n + m) + (n + m) + (2 * n + m))


def compulsory_bytes(n, m, N):
    """SURVEY.md 8(d) floor (A): inputs read once + solution written once."""
    return 8 * (inputs_doubles(n, m, N) + N * (2 * n + m))


def reduced_model(n, m, N, compact_level0=False, fused2=False):
    """Separator-only ("reduced") schedule of the size-specialised shapes: bottom kernel (leaf phase +
    tree levels 0, 1), one launch per upper level, back-substitution. Returns
    {slot: {"bytes": per solve, "flops": per solve, "launches": per solve}}.
    fused2 (round 4, bottom8_reduced_mc): the bottom launch also eliminates the level-2 separators, whose slots live in
    LDS -- per eight knots it stores the two groups' shares of the outer separators, adds the level-2 separator's
    (two symmetric blocks + two vectors each, one coupling block) and writes the level-2 record."""
    K = int(math.log2(N))
    if fused2 and N >= 16:
        base = reduced_model(n, m, N, compact_level0, False)
        w, rows = n + m, 2 * n + m
        tri = n * (n + 1) // 2
        rec0, rec = record_doubles(n, 0, compact_level0), record_doubles(n, 1)
        fs = separator_flops(n, w)
        push = 2 * tri + n * n + 2 * n
        slot = 2 * tri + 2 * n * n + 2 * n
        per_sep = slot + n * w + w + n + rows + 2 * n + rec + push   # an upper-level separator in a launch of its own
        nsep2 = N // 8
        base["bottom"] = {"bytes": 8 * (inputs_doubles(n, m, N) + (N // 2) * rec0 + (N // 4 + nsep2) * rec +
                                       nsep2 * (4 * (tri + n) + n * n)),
                          "flops": (3 * N // 4 + nsep2) * fs, "launches": 1}
        if "upper" in base:
            left = base["upper"]["bytes"] // 8 // per_sep - nsep2   # separators still eliminated by level launches
            if left > 0:
                base["upper"] = {"bytes": 8 * left * per_sep, "flops": left * fs, "launches": base["upper"]["launches"] - 1}
            else:
                del base["upper"]
        elif "top" in base and K == 5:   # the top launch is left with levels 3 and 4
            sweep_b = (N // 8 - 1) * rec + (N // 8) * n
            base["top"] = {"bytes": 8 * (3 * per_sep + sweep_b), "flops": 3 * fs + (N // 8 - 1) * 4 * n * n, "launches": 1}
        return base
    w, rows = n + m, 2 * n + m
    tri = n * (n + 1) // 2        # DL, DR are symmetric: packed lower triangles (round 3)
    push = 2 * tri + n * n + 2 * n      # what one four-knot group / one upper separator pushes to its neighbours
    slot = 2 * tri + 2 * n * n + 2 * n  # DL | DR | CA | CB | gL | gR of one separator of level >= 2
    rec0 = record_doubles(n, 0, compact_level0)
    rec = record_doubles(n, 1)
    fs = separator_flops(n, w)
    nsep_upper = max(N // 4 - 1, 0)
    bottom_b = inputs_doubles(n, m, N) + (N // 2) * rec0 + (N // 4) * rec + (N // 4) * push
    upper_b = nsep_upper * (slot + n * w + w + n + rows + 2 * n + rec + push)
    recs_read = (N // 2) * rec0 + (N // 4) * rec + nsep_upper * rec
    # back-substitution: every record once, the inputs again, the solution; the compact form adds the
    # multipliers of the top of the tree (written by rb_backsub_top, two read per eight knots)
    apply_b = recs_read + inputs_doubles(n, m, N) + N * rows + ((N // 8) * n + (N // 8) * 2 * n if compact_level0 else 0)
    out = {
        "bottom": {"bytes": 8 * bottom_b, "flops": (3 * N // 4) * fs, "launches": 1},
        "apply": {"bytes": 8 * apply_b, "flops": backsub_flops(n, m, N), "launches": 1},
    }
    if K > 2:
        out["upper"] = {"bytes": 8 * upper_b, "flops": nsep_upper * fs, "launches": K - 2}
    if K >= 5 and compact_level0:
        # round 3: the last three levels (7 separators) run in ONE launch (reduced_top_mc, profile slot "top") together
        # with the top-down sweep of the back-substitution over the records of level >= 3 (N / 8 - 1 of them, written
        # into ytop): their bytes and flops move from "upper" / "apply" to "top"
        per_sep = upper_b // nsep_upper
        sweep_b = (N // 8 - 1) * rec + (N // 8) * n
        sweep_f = (N // 8 - 1) * 4 * n * n
        out["top"] = {"bytes": 8 * (7 * per_sep + sweep_b), "flops": 7 * fs + sweep_f, "launches": 1}
        out["upper"] = {"bytes": 8 * (nsep_upper - 7) * per_sep, "flops": (nsep_upper - 7) * fs, "launches": K - 5}
        out["apply"] = {"bytes": out["apply"]["bytes"] - 8 * sweep_b, "flops": out["apply"]["flops"] - sweep_f,
                        "launches": 1}
        if nsep_upper == 7:
            del out["upper"]
    return out


def knot_lean_model(n, m, N):
    """Knot-based lean schedule (bottom_small + level_small + backsub_small): the bottom kernel hands
    the first / last knot of every four-knot group over (E, one live outer column, rhs)."""
    K = int(math.log2(N))
    w, rows = n + m, 2 * n + m
    fb = rows * n
    rec = 2 * n * n + n
    fs = separator_flops(n, w)
    bottom_b = inputs_doubles(n, m, N) + (3 * N // 4) * rec + (N // 2) * (2 * fb + rows)
    out = {"bottom": {"bytes": 8 * bottom_b, "flops": (3 * N // 4) * fs + N * 4 * fb * n, "launches": 1},
           "apply": {"bytes": 8 * ((N - 1) * rec + inputs_doubles(n, m, N) + N * rows),
                     "flops": backsub_flops(n, m, N), "launches": 1}}
    if K > 2:
        nsep = N // 4 - 1
        # per subtree: stage E / outer columns of knots s, s+1 (w n + w n + 2 n^2), rhs, [A|B](s); write the
        # record; read + write the two boundary knots' live columns
        per = (2 * w * n + 2 * n * n + rows + 2 * n + n * w) + rec + 2 * (3 * fb + 2 * rows)
        out["upper"] = {"bytes": 8 * nsep * per, "flops": nsep * (fs + 2 * 4 * fb * n), "launches": K - 2}
    return out


def generic_lean_model(n, m, N):
    """Runtime-sized lean schedule (leaf_generic, separator_generic + boundary Schur per level,
    back-substitution over the records): config 5's path."""
    K = int(math.log2(N))
    w, rows = n + m, 2 * n + m
    fb_xu = w * n                   # state + input rows of a factor block (lambda rows are dead data here)
    rec = 2 * n * n + n
    out = {
        # leaf phase: inputs in, two factor blocks' state/input rows and the rhs block out
        "leaf": {"bytes": 8 * (inputs_doubles(n, m, N) + N * (2 * fb_xu + rows)),
                 "flops": N * (2 * w * n + 4 * rows), "launches": 1},
    }
    sep_b = sep_f = sch_b = sch_f = 0
    for l in range(K):
        L = N >> (l + 1)
        # separator: [A|B](s), E(s) and the outer column of s (state+input rows), state rows of E(s+1) and of
        # its outer column, rhs of both knots; writes the record
        sep_b += L * (n * w + 2 * fb_xu + 2 * n * n + rows + 2 * n + rec)
        sep_f += L * (4 * n * n * w + n ** 3 / 3.0 + 2 * n * n * (2 * n + 1))
        if l < K - 1:
            # boundary Schur: first and last knot of every subtree: read E, f (record), read+write two columns + rhs
            sch_b += 2 * L * (fb_xu + 3 * fb_xu + 2 * rows) + L * 2 * n * n
            sch_f += 2 * L * (2 * 2 * fb_xu * n + 2 * w * n)
    out["separator"] = {"bytes": 8 * sep_b, "flops": sep_f, "launches": K}
    out["schur_boundary"] = {"bytes": 8 * sch_b, "flops": sch_f, "launches": K - 1}
    out["apply"] = {"bytes": 8 * ((N - 1) * rec + inputs_doubles(n, m, N) + N * rows),
                    "flops": backsub_flops(n, m, N), "launches": 1}  # K + 1 launches inside ONE event bracket
    return out


def generic_reduced_model(n, m, N):
    """Separator-only schedule of the runtime-sized blocks (separator_reduced_mfma once per level,
    back-substitution over the records): config 5's path since round 2. DL, DR (and S-bar) are symmetric: only
    the 16x16 tiles on and below the diagonal are computed, stored and read; the records are compact at every level
    (round 3: the Cholesky factor, packed lower triangle, and y~ instead of f_a | f_bb | z_sep -- the
    back-substitution re-forms their action from the couplings: the slots above level 0, the inputs at level 0)."""
    K = int(math.log2(N))
    w, rows = n + m, 2 * n + m
    rec0 = n * (n + 1) // 2 + n          # compact record: factor (packed lower triangle) | y~
    sym = min(n * n, n * (n + 16) // 2)  # doubles of the lower tiles of an n x n block
    slot = 2 * sym + 2 * n * n + 2 * n   # DL, DR (lower tiles) | CA | CB | gL | gR
    push = 2 * sym + n * n + 2 * n       # DR + gR, DL + gL, one coupling block
    sep_b = 0
    for l in range(K):
        L = N >> (l + 1)
        own = n * w + (w + n) + (rows + 2 * n)              # [A_s | B_s], weights and rhs of knots s, s+1
        if l == 0:
            sep_b += L * (own + n * n + rec0 + push)        # + A_{s+1} (r_bb); the pushes are stores
        else:
            sep_b += L * (own + slot + rec0 + push + (2 * sym + 2 * n))  # + own slot; the pushes read-modify-write
    return {
        "separator": {"bytes": 8 * sep_b, "flops": (N - 1) * separator_flops(n, w), "launches": K},
        # every record once, the couplings CA | CB of the separators above level 0, the inputs, the solution; K launches
        # inside ONE event bracket
        "apply": {"bytes": 8 * ((N - 1) * rec0 + (N // 2 - 1) * 2 * n * n + inputs_doubles(n, m, N) + N * rows),
                  "flops": backsub_flops(n, m, N) + (N - 1) * 2 * n * n, "launches": 1},
    }


def model_for(schedule, n, m, N):
    """Per-slot model of the named launch sequence (ndlqr_hip_schedule), or None when this file has no
    model for it (strict / KEEP schedules stream the whole factor array: model (B) is their roofline)."""
    if schedule == "reduced":  # compact level-0 records, two-launch back-substitution
        return reduced_model(n, m, N, compact_level0=True)
    if schedule == "reduced-fused2":  # ... with tree level 2 inside the bottom launch (NDLQR_FUSE2=1)
        return reduced_model(n, m, N, compact_level0=True, fused2=True)
    if schedule in ("reduced-tree", "reduced-records"):
        return reduced_model(n, m, N)
    if schedule == "knot-lean":
        return knot_lean_model(n, m, N)
    if schedule == "generic-lean":
        return generic_lean_model(n, m, N)
    if schedule in ("generic-reduced", "generic-reduced-records"):
        return generic_reduced_model(n, m, N)
    return None


def model_b_bytes(n, m, N):
    """SURVEY.md 8(d) 'level-streaming' algorithmic bytes per solve of the reference's dense schedule."""
    K = int(math.log2(N))
    Fb = (2 * n + m) * n
    total = N * (4 * n * n + 3 * m * n + m * m + 5 * n + 3 * m)
    for l in range(K):
        L = 1 << (K - l - 1)
        P1 = L * (K - l) * (4 * n * n + 2 * m * n) + L * (n * n + n * m)
        P2 = 2 * L * n * n
        P3 = L * n * n + 2 * L * (K - l - 1) * n * n
        P4 = (N * Fb if l < K - 1 else 0) + L * (K - l - 1) * n * n + 2 * N * (K - l - 1) * Fb
        S = L * (2 * n * n + n * m + 8 * n + 2 * m) + N * (Fb + 5 * n + 2 * m)
        total += P1 + P2 + P3 + P4 + S
    return 8 * total


def model_b_flops(n, m, N):
    """SURVEY.md 8(d) algorithmic flops per solve of the reference's dense schedule."""
    K = int(math.log2(N))
    Fb = (2 * n + m) * n
    fl = N * (n ** 3 / 3 + m ** 3 / 3 + 4 * n ** 3 + 2 * m * m * n)
    for l in range(K):
        L = 1 << (K - l - 1)
        fl += L * (K - l) * 4 * n * n * (n + m) + L * n ** 3 / 3 + L * (K - l - 1) * 2 * n ** 3
        fl += N * (K - l - 1) * 2 * Fb * n
        fl += L * (4 * n * (n + m) + 2 * n * n) + N * 2 * Fb
    return fl


def kernel_roofline(slot_model, batch, avg_launch_ms, launches=None):
    """Roofline entry of one kernel kind from its per-solve model and its measured average launch
    duration: both fractions, the larger one decides the bound label. launches: measured launches of this
    kind per solve (the last tree levels may share a launch), default: the model's."""
    launches = launches or slot_model["launches"]
    b = slot_model["bytes"] * batch / launches
    f = slot_model["flops"] * batch / launches
    gbs = b / (avg_launch_ms * 1e-3) / 1e9
    tfl = f / (avg_launch_ms * 1e-3) / 1e12
    hbm_frac, fp_frac = gbs / HBM_PEAK_GBS, tfl / FP64_PEAK_TFLOPS
    if hbm_frac >= fp_frac:
        head = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac}
    else:
        head = {"bound": "mfma", "achieved": tfl, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fp_frac}
    head.update(avg_launch_ms=avg_launch_ms, algorithmic_bytes_per_launch=b, useful_flops_per_launch=f,
                hbm_gbs=gbs, hbm_frac=hbm_frac, fp64_tflops=tfl, fp64_frac=fp_frac)
    return head
