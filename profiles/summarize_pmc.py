#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tests/prof_bench.sh into the two files kept under profiles/:
   python3 profiles/summarize_pmc.py gpurun_out/<tag> r01 [batch]
-> profiles/r01_bench_kernel_stats.csv (copy of the --kernel-trace --stats summary)
-> profiles/<tag>_pmc.json (per kernel: launches and mean per launch of every counter; FETCH_SIZE / WRITE_SIZE are
   reported by rocprofv3 in KiB -- checked against import_kernel, which writes exactly 20 B/px = 40500 KiB at 1080p)"""
import csv
import glob
import json
import re
import shutil
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32  # units per step of the profiled bench run
spl = int(sys.argv[4]) if len(sys.argv) > 4 else (8 if batch >= 16 else 1)  # sweeps per sweep-kernel launch (round 4: a batch of 16+ runs its 8 sweeps in one launch)
stats = glob.glob(src + "/trace/*/*_kernel_stats.csv")[0]
shutil.copy(stats, "profiles/%s_bench_kernel_stats.csv" % tag)
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(src + "/pmc_*/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"])
        if name.startswith("__amd") or "at::" in name or "hipcub" in name or "rocprim" in name:
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"command": "rocprofv3 --pmc <counter> --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "
                  "(one pass per counter set: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | SQ_*)",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (uncorrected; the gfx950 x2 correction of "
                "MI355X_MICROARCH.md applies to 16-B-per-lane streaming reads only, the sweep kernel issues 4-B gathers)",
       "counters": {}}
for k, cs in sorted(acc.items()):
    out["counters"][k] = {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for c, v in sorted(cs.items())}
for k, cs in out["counters"].items():
    if "sweep_kernel<8, 1, false, false, false" in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        out["sweep_kernel_batch%d_hbm_bytes_per_launch" % batch] = int((cs["FETCH_SIZE"]["mean_per_launch"] + cs["WRITE_SIZE"]["mean_per_launch"]) * 1024)
    if "sweep_kernel<8, 1, false, false, false" in k and "SQ_INSTS_VALU" in cs:
        out["sweep_kernel_batch%d_valu_insts_per_launch" % batch] = int(cs["SQ_INSTS_VALU"]["mean_per_launch"])
        out["sweep_kernel_batch%d_valu_busy_quadcycles_per_launch" % batch] = int(cs["SQ_ACTIVE_INST_VALU"]["mean_per_launch"])
out["sweeps_per_launch"] = spl
json.dump(out, open("profiles/%s_pmc.json" % tag, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("counters",)}, indent=1))
for k, cs in out["counters"].items():
    print(k, {c: round(v["mean_per_launch"]) for c, v in cs.items()})
