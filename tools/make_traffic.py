#!/usr/bin/env python3
"""profiles/TAG_traffic.json from the two PMC passes of tools/profile_round.sh: HBM bytes per launch
of every kernel kind (the profile slots bench.py reports), corrected as MI355X_MICROARCH.md (HBM)
prescribes -- FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B), WRITE_SIZE as
reported; both counters are in KB. The file records the workload and the content hash of
rslqr_amd/csrc it was taken on; bench.py prints `traffic` only when both match its own run.

    python tools/make_traffic.py TAG FETCH.csv WRITE.csv nx nu N batch flags > profiles/TAG_traffic.json
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (csrc_sha)

SLOTS = [("bottom", ("bottom_reduced_mc", "bottom8_reduced_mc", "bottom_small", "rb_bottom")),
         ("upper", ("reduced_level_mc", "level_small")),
         ("top", ("reduced_top_mc",)),
         ("apply", ("backsub_small", "apply_small", "rb_backsub", "backsub_multipliers_generic", "backsub_multipliers_compact", "backsub_states_generic",
                    "backsub_level0_states_generic")),
         ("leaf", ("leaf_generic",)),
         ("separator", ("separator_generic", "separator_mfma", "separator_reduced_mfma")),
         ("schur_boundary", ("schur_mfma", "schur_generic")),
         ]


def agg(path):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        c = out.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += float(r["Counter_Value"])
    return out


def main():
    tag, fpath, wpath = sys.argv[1:4]
    n, m, N, batch, flags = [int(v) for v in sys.argv[4:9]]
    fetch, write = agg(fpath), agg(wpath)
    K = N.bit_length() - 1
    # solves in the profiled process = launches of the once-per-solve kernel
    # (the two passes are separate processes whose untimed spin-up runs by the clock: each has its own solve count)
    def count(tab):
        once = [c[0] for k, c in tab.items() if any(s in k for s in ("bottom_", "bottom8_", "rb_bottom", "leaf_generic",
                                                                     "backsub_states_generic", "backsub_level0_states_generic"))]
        return max(once) if once else 1
    solves, solves_w = count(fetch), count(write)
    kernels = {}
    for slot, pats in SLOTS:
        fkb = sum(c[1] for k, c in fetch.items() if any(p in k for p in pats))
        wkb = sum(c[1] for k, c in write.items() if any(p in k for p in pats))
        calls = sum(c[0] for k, c in fetch.items() if any(p in k for p in pats))
        if calls == 0:
            continue
        # launches of the slot as bench.py counts them (one HIP-event bracket each)
        per_solve = calls / solves
        brackets = 1 if slot == "apply" else per_solve
        kernels[slot] = {"kernels": [k.replace("void ndlqr::", "") for k in fetch if any(p in k for p in pats)],
                         "launches_per_solve": per_solve,
                         "fetch_mb_raw_per_solve": fkb / solves / 1024,
                         "write_mb_per_solve": wkb / solves_w / 1024,
                         "hbm_bytes_per_launch": int((2 * fkb / solves + wkb / solves_w) * 1024 / brackets)}
    json.dump({"tag": tag, "workload": [n, m, N, batch, flags], "csrc_sha": bench.csrc_sha(),
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/profile_round.sh)",
               "correction": "FETCH_SIZE x 2 (gfx950 counts 64 B per 128-B read request), WRITE_SIZE as reported; KB -> bytes",
               "solves_profiled": solves, "tree_levels": K, "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
