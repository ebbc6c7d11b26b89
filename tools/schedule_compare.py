"""Batch vs interleaved post-filter schedule through the scene-level oracle harness on a few scenes: the
numbers behind the BASELINE.md section 3 row.  CPU only; test infrastructure."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import scene_oracle as S  # noqa: E402
from test_oracle_schedule import compare  # noqa: E402

CASES = [
    dict(name="6 x 160x128, 3 src, 3 outer x 2", scene=dict(n=6, w=160, h=128, f=150.0), outer=3, iters=2, a=6),
    dict(name="8 x 256x192, 4 src, 3 outer x 2", scene=dict(n=8, w=256, h=192, f=240.0, n_src=4, n_points=200), outer=3, iters=2, a=6),
    dict(name="8 x 256x192, 6 src, 4 outer x 3 (authors' shape)", scene=dict(n=8, w=256, h=192, f=240.0, n_src=6, n_points=200), outer=4, iters=3, a=7),
]
for c in CASES:
    views, srcs, nb, order, init = S.ring_scene(**c["scene"])
    for mode, mname in ((O.ARITH_DEVICE, "device"), (O.ARITH_REFERENCE, "reference")):
        kw = dict(n_external_iters=c["outer"], postfilter=True, mode=mode, seed=900, adapthalfwin=c["a"], n_estimation_iters=c["iters"], propagate_halfwin=5,
                  propagate_step=4)
        t = time.time()
        b = S.densify(views, srcs, nb, order, init, interleave=False, **kw)
        i = S.densify(views, srcs, nb, order, init, interleave=True, **kw)
        m = compare(i, b, views)
        print("%-50s %-9s valid_agree %.4f (worst %.4f) within1%% %.4f (worst %.4f) acc %.4f / %.4f points %d / %d (%+.2f%%) filled %d / %d  [%.0f s]" % (
            c["name"], mname, m["valid_agree"], m["worst_valid_agree"], m["within_1pct"], m["worst_within_1pct"], m["acc_a"], m["acc_b"], m["points_a"], m["points_b"],
            100.0 * (m["points_b"] - m["points_a"]) / m["points_a"], sum(i["filled"]), sum(b["filled"]), time.time() - t), flush=True)
