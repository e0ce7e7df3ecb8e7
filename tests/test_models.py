"""CPU checks of the host-side models bench.py reports with (no GPU): the roofline denominators of every schedule
(rslqr_amd/roofline.py) and the chunk arithmetic of the time-axis sharding helpers."""
import numpy as np
import pytest

from rslqr_amd import roofline as rf
from rslqr_amd import sharding


@pytest.mark.parametrize("schedule", ["reduced", "reduced-fused2", "reduced-tree", "reduced-records", "knot-lean",
                                      "generic-lean", "generic-reduced", "generic-reduced-records"])
@pytest.mark.parametrize("n,m,N", [(12, 4, 256), (12, 4, 1024), (6, 3, 16), (12, 4, 32), (64, 16, 512)])
def test_roofline_models_are_consistent(schedule, n, m, N):
    model = rf.model_for(schedule, n, m, N)
    assert model, schedule
    total_b = sum(v["bytes"] for v in model.values())
    total_f = sum(v["flops"] for v in model.values())
    assert all(v["bytes"] > 0 and v["flops"] > 0 and v["launches"] >= 1 for v in model.values())
    # never below the compulsory floor (inputs once + solution once), never above the reference's dense level-streaming
    # schedule (SURVEY.md 8(d) models (A) and (B))
    assert rf.compulsory_bytes(n, m, N) <= total_b <= rf.model_b_bytes(n, m, N)
    assert total_f <= rf.model_b_flops(n, m, N) * 1.01
    # a measured launch at exactly the roofs has fraction 1 on the bound side
    for slot, v in model.items():
        ms_hbm = v["bytes"] / v["launches"] / (rf.HBM_PEAK_GBS * 1e9) * 1e3
        ms_fp = v["flops"] / v["launches"] / (rf.FP64_PEAK_TFLOPS * 1e12) * 1e3
        r = rf.kernel_roofline(v, 1, max(ms_hbm, ms_fp))
        assert abs(r["frac"] - 1.0) < 1e-9 and 0 < min(r["hbm_frac"], r["fp64_frac"]) <= 1.0


def test_fused_bottom_model_moves_fewer_bytes():
    """NDLQR_FUSE2=1 (bottom8_reduced_mc): no level-2 slots, no second read of [A | B] for them, one level launch less;
    the same separators are eliminated (same useful flops)."""
    a, b = rf.model_for("reduced", 12, 4, 256), rf.model_for("reduced-fused2", 12, 4, 256)
    sa, sb = sum(v["bytes"] for v in a.values()), sum(v["bytes"] for v in b.values())
    assert 0.85 * sa < sb < 0.90 * sa
    assert abs(sum(v["flops"] for v in a.values()) - sum(v["flops"] for v in b.values())) < 1e-6
    assert b["upper"]["launches"] == a["upper"]["launches"] - 1 and b["bottom"]["bytes"] > a["bottom"]["bytes"]


def test_headline_model_numbers():
    """The figures DESIGN.md section 4 quotes for (12,4,256): 2.5 MB and 5.4 MFLOP per solve on the implemented schedule,
    68.75 MB / 83.6 MFLOP for the reference's level-streaming schedule, 0.54 MB compulsory."""
    m = rf.model_for("reduced", 12, 4, 256)
    assert abs(sum(v["bytes"] for v in m.values()) / 1e6 - 2.50) < 0.02
    assert abs(sum(v["flops"] for v in m.values()) / 1e6 - 5.43) < 0.02
    assert abs(rf.model_b_bytes(12, 4, 256) / 1e6 - 68.75) < 0.01
    assert abs(rf.model_b_flops(12, 4, 256) / 1e6 - 83.6) < 0.2
    assert abs(rf.compulsory_bytes(12, 4, 256) / 1e6 - 0.54) < 0.005


def test_time_axis_chunks_partition_the_solution_vector():
    n, m, N = 12, 4, 64
    nvars = (2 * n + m) * N - m
    for world in (2, 4, 8):
        seen = np.zeros(nvars, dtype=int)
        for r in range(world):
            seen[sharding.chunk_of_solution(None, n, m, N, r, world)] += 1
        assert (seen == 1).all()
    assert sharding.shard_range(3, 8, 512) == (1536, 2048) and sharding.shard_seed0(3, 512) == 1537
