"""the device path runs the post-filters of outer iterations 1 and 2 as a BATCH (estimate every image, then
filter every image); the reference filters image k right after its own estimate (single-thread event order,
SceneDensify.cpp:3889-3965: EVTEstimateDepthMap(k) queues EVTOptimizeDepthMap(k) first), so RemoveSmallSegments(k) fuses against the
images > k as the PREVIOUS outer iteration left them and zeroes depths in them before they are estimated again.  Both orders are
run here through the scene-level oracle harness (three outer iterations, filters after 1 and 2, six images) and compared in the
terms of the north star; the measured row is in BASELINE.md section 3.  The exact order is available as an opt-in mode
(--n-postfilter-interleave 1, densify_scene(interleave=True)) and is GPU-tested bit for bit in tests/test_gpu_schedule.py."""
import numpy as np
import pytest

import oracle_lib as O
import scene_oracle as S

# the tolerance BASELINE.md section 3 states for the batch schedule against the reference's interleaved one.  fused_points: the
# north star's 1 % holds from 8 images of 256x192 on; the six-image 160x128 scene whose LAST outer iteration is a filtered one is the
# worst case measured (tools/schedule_compare.py: +1.3 ... +1.7 %), hence its own bar
TOL = dict(valid_agree=0.97, within_1pct=0.96, fused_points=0.01, fused_points_tiny=0.02, accuracy=0.01)


def compare(a, b, views):
    out = dict(valid_agree=[], within_1pct=[], acc_a=[], acc_b=[])
    for i in sorted(a["maps"]):
        da, db = a["maps"][i][0], b["maps"][i][0]
        va, vb = da > 0, db > 0
        both = va & vb
        out["valid_agree"].append(float((va == vb).mean()))
        out["within_1pct"].append(float((np.abs(da - db)[both] / da[both] < 0.01).mean()))
        gt = views[i]["depth"]
        out["acc_a"].append(float((np.abs(da - gt)[va] / gt[va] < 0.01).mean()))
        out["acc_b"].append(float((np.abs(db - gt)[vb] / gt[vb] < 0.01).mean()))
    m = {k: float(np.mean(v)) for k, v in out.items()}
    m["worst_valid_agree"] = float(np.min(out["valid_agree"])); m["worst_within_1pct"] = float(np.min(out["within_1pct"]))
    m["points_a"], m["points_b"] = a["cloud"]["n_points"], b["cloud"]["n_points"]
    return m


CASES = [
    dict(id="6x160x128 device", mode=O.ARITH_DEVICE, scene=dict(n=6, w=160, h=128, f=150.0), points="fused_points_tiny"),
    dict(id="6x160x128 reference", mode=O.ARITH_REFERENCE, scene=dict(n=6, w=160, h=128, f=150.0), points="fused_points_tiny"),
    dict(id="8x256x192 reference", mode=O.ARITH_REFERENCE, scene=dict(n=8, w=256, h=192, f=240.0, n_src=4, n_points=200), points="fused_points"),
]


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_batch_and_interleaved_postfilter_schedules_agree_within_tolerance(case):
    views, srcs, neighbors, order, init = S.ring_scene(**case["scene"])
    kw = dict(n_external_iters=3, postfilter=True, mode=case["mode"], seed=900, adapthalfwin=6, n_estimation_iters=2, propagate_halfwin=5, propagate_step=4)
    batch = S.densify(views, srcs, neighbors, order, init, interleave=False, **kw)
    inter = S.densify(views, srcs, neighbors, order, init, interleave=True, **kw)
    m = compare(inter, batch, views)
    print("schedule (interleaved = a, batch = b):", {k: round(v, 4) if isinstance(v, float) else v for k, v in m.items()},
          "filled:", sum(inter["filled"]), sum(batch["filled"]))
    assert not all(np.array_equal(inter["maps"][i][0], batch["maps"][i][0]) for i in inter["maps"])   # they ARE different schedules
    assert m["worst_valid_agree"] >= TOL["valid_agree"] and m["worst_within_1pct"] >= TOL["within_1pct"]
    assert abs(m["points_a"] - m["points_b"]) <= TOL[case["points"]] * m["points_a"]
    assert abs(m["acc_a"] - m["acc_b"]) <= TOL["accuracy"]
    assert sum(inter["filled"]) > 0 and sum(batch["filled"]) > 0
