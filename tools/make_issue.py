#!/usr/bin/env python3
"""profiles/TAG_issue.json from the SQ-counter passes of tools/sq_counters.sh: instructions per wavefront of every
kernel kind (the profile slots bench.py reports) and the time they take on the fp64 pipe of a SIMD.

On gfx950 the fp64 matrix-core instruction and the vector instructions of all wavefronts of a SIMD execute one after
the other (profiles/r03_issue_rates.txt): v_mfma_f64_16x16x4 holds the pipe for 64 cycles, a vector instruction for
4 (its issue rate with many wavefronts; 4.4-5.3 with one to three). `exec_ms` = that sum over the wavefronts of a
launch, spread over the 1 024 SIMDs at 2.4 GHz: the time below which the launch cannot go without executing fewer
instructions. The file records the workload and the content hash of rslqr_amd/csrc; bench.py prints `fp64_pipe` only
when both match its own run.

    python tools/make_issue.py TAG nx nu N batch flags > profiles/TAG_issue.json
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (csrc_sha)
from make_traffic import SLOTS  # noqa: E402

SIMDS, CLOCK_GHZ, CYC_VECTOR, CYC_MATRIX = 1024, 2.4, 4.0, 64.0


def main():
    tag = sys.argv[1]
    n, m, N, batch, flags = [int(v) for v in sys.argv[2:7]]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))  # kernel -> counter -> [launches, sum]
    for f in sorted(glob.glob("gpurun_out/%s_sq_*/**/*counter_collection.csv" % tag, recursive=True)):
        for row in csv.DictReader(open(f)):
            c = acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]]
            c[0] += 1
            c[1] += float(row["Counter_Value"])
    # solves in the profiled process = launches of a once-per-solve kernel
    once = [acc[k]["SQ_WAVES"][0] for k in acc if any(t in k for t in ("bottom_", "bottom8_", "rb_bottom", "leaf_generic",
                                                                        "backsub_states_generic", "backsub_level0_states_generic"))]
    solves = max(once) if once else 1
    kernels = {}
    for slot, pats in SLOTS:
        names = [k for k in acc if any(p in k for p in pats)]
        if not names:
            continue
        tot = collections.defaultdict(float)
        launches = 0
        for k in names:
            launches += acc[k]["SQ_WAVES"][0]
            for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM",
                      "SQ_WAVE_CYCLES"):
                # per-launch averages of a counter x its own launch count = totals over the same set of launches
                # (every pass profiles the same launches)
                tot[c] += acc[k][c][1] / max(1, acc[k][c][0]) * acc[k]["SQ_WAVES"][0]
        waves = tot["SQ_WAVES"]
        cycles = CYC_VECTOR * tot["SQ_INSTS_VALU"] + CYC_MATRIX * tot["SQ_INSTS_MFMA"]
        kernels[slot] = {
            "kernels": sorted(n_.replace("void ndlqr::", "").replace("ndlqr::", "") for n_ in names),
            "launches_profiled": launches,
            "wavefronts_per_launch": waves / launches,
            "vector_per_wavefront": tot["SQ_INSTS_VALU"] / waves,
            "matrix_per_wavefront": tot["SQ_INSTS_MFMA"] / waves,
            "scalar_per_wavefront": tot["SQ_INSTS_SALU"] / waves,
            "lds_per_wavefront": tot["SQ_INSTS_LDS"] / waves,
            "vmem_per_wavefront": tot["SQ_INSTS_VMEM"] / waves,
            "wavefront_life_cycles": 4.0 * tot["SQ_WAVE_CYCLES"] / waves,  # (the counter ticks every four cycles)
            "launches_per_solve": launches / solves,
            "exec_ms_per_solve": cycles / solves / SIMDS / (CLOCK_GHZ * 1e6),
        }
    json.dump({"tag": tag, "workload": [n, m, N, batch, flags], "csrc_sha": bench.csrc_sha(), "solves_profiled": solves,
               "source": "rocprofv3 --pmc SQ_* in separate passes (tools/sq_counters.sh)",
               "model": "exec_ms = (%g cycles x vector + %g cycles x fp64 matrix-core instructions) / %d SIMDs / %g GHz: "
                        "the fp64 pipe of a SIMD executes them one after the other (profiles/r03_issue_rates.txt)"
                        % (CYC_VECTOR, CYC_MATRIX, SIMDS, CLOCK_GHZ),
               "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
