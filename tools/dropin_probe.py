"""Developer tool (GPU box): the drop-in call sequence timed per call on the two JSON fixtures (bench.py's dropin leg)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rslqr_amd  # noqa: E402

for f in ("lqr_prob.json", "lqr_prob_256.json"):
    print(json.dumps(bench.dropin_leg(rslqr_amd, os.path.join(ROOT, "tests", "golden", f)), indent=1))
