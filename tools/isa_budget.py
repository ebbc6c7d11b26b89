#!/usr/bin/env python3
"""Static instruction budget of the sweep worker by source phase.

Compiles pm_kernels.hip to gfx950 assembly with -DHCMVS_MARK (the phase boundaries of the stamps build + sub-phases of the scorer become
`; HCMARK <name>` comments), takes one kernel (default sweep_kernel<8,1,false,false,false>, the kernel of the benchmark) and attributes
every instruction to the marker that precedes it in the listing.  The listing is in source order for this kernel (loop bodies are
contiguous), so "between marker a and the next marker" is "phase a".  Per phase: VALU / scalar / LDS / vector-memory instruction counts
and, of the VALU ones, the classes the microbenchmark prices above 2 cycles (profiles/r02_valu_issue_bench.jsonl): lane movement
(v_readlane / v_writelane / v_readfirstlane = SGPR spills and uniform broadcasts), v_mov, conversions, DPP, the IEEE division sequence,
transcendental approximations.  Dynamic counts = static counts x trip counts; the trip counts (rounds, chunks, evaluations per pixel-sweep)
come from a run of the stamps build (tools/stamp_run.py prints them) and are passed with --trips.

usage: tools/isa_budget.py [--kernel SUBSTR] [--trips rounds=2.9,chunks=3.0,evals=8.0] [--md out.md]
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hc-mvs_amd", "csrc", "pm_kernels.hip")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"]

# which trip count multiplies a phase: per pixel (1), per hypothesis round, per chunk of <= 8 hypotheses, per evaluation
PHASES = collections.OrderedDict([
    ("13", ("prefetch: own state + neighbour slots, next pixel's patch inputs", "pixel")),
    ("12", ("fill_patch (bilateral weights, 64-lane sums)", "pixel")),
    ("1", ("wait for the row above + late neighbour loads", "pixel")),
    ("0", ("slots: interpolate (f64) + correct normal + park", "pixel")),
    ("2", ("pixel set-up: InitPlane, candidate shuffles, RNG key (scalar), the 3 x 6 refinement draws (hoisted)", "pixel")),
    ("blk_pick", ("PICK: publish hook, thresholds, Normal2Dir (atan2 + acos)", "pick")),
    ("5", ("publish hook: drain stores", "pick")),
    ("6", ("publish hook: poll the row above + rest of PICK", "pick")),
    ("blk_gen_prop", ("hypotheses: propagation candidates", "gen_prop")),
    ("blk_gen_rand", ("hypotheses: full random (rare)", "gen_rand")),
    ("blk_gen_refine", ("hypotheses: refinement trials (3 draws, 2 sincos)", "gen_refine")),
    ("3", ("share of the round", "round")),
    ("blk_chunk", ("chunk set-up: 5 shuffles, InitPlane", "chunk")),
    ("blk_smooth_chunks", ("smooth_pass entry", "chunk")),
    ("blk_smooth_chunk", ("smooth_pass: 8 slots x 8 hypotheses (2 exp, acos, div)", "smooth_chunks")),
    ("blk_smooth_end", ("smooth_pass exit", "chunk")),
    ("4", ("chunk dispatch", "chunk")),
    ("blk_score_chunk", ("score_chunk entry", "score_chunk")),
    ("sc_hom", ("score_chunk: homographies of the chunk", "score_chunk")),
    ("sc_taps", ("score_chunk: hypothesis loop control", "eval")),
    ("tap_warp", ("tap loop: warp + reciprocal tree + inside test", "eval")),
    ("tap_sample", ("tap loop: addresses, gathers, bilinear, sums", "eval")),
    ("tap_reduce", ("tap loop: group butterflies", "eval")),
    ("tap_end", ("tap loop: park sums in LDS + loop", "eval")),
    ("sc_epi", ("score_chunk: per-view ZNCC epilogue + two best views", "score_chunk")),
    ("sc_end", ("after the chunk", "score_chunk")),
    ("7", ("score exchange (NW > 1 only)", "round")),
    ("8", ("accept logic head", "round")),
    ("blk_acc_prop", ("accept: propagation", "acc_prop")),
    ("blk_acc_prop_cand", ("accept: per candidate (readlanes)", "acc_prop_cand")),
    ("blk_acc_rand", ("accept: random", "acc_rand")),
    ("blk_acc_refine", ("accept: refinement (ballot)", "acc_refine")),
    ("11", ("round loop tail", "round")),
    ("blk_pixel_tail", ("restore hint (off) + store result + ring", "pixel")),
    ("9", ("pixel loop tail", "pixel")),
    ("10", ("row end / ticket", "row")),
])


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def sub_class(line):
    op = line.split()[0]
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_mov_b32") and "dpp" not in line:
        return "mov"
    if "dpp" in line or op.startswith("v_permlane"):
        return "dpp"
    if op.startswith("v_cvt") or op.startswith("v_fract") or op.startswith("v_floor") or op.startswith("v_rndne") or op.startswith("v_trunc"):
        return "cvt/fract"
    if op.startswith(("v_div_", "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos", "v_ldexp", "v_frexp")):
        return "div/trans"
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mad_i32", "v_mad_u32", "v_mul_u32", "v_mul_i32", "v_mad_u64")):
        return "int mul"
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f64", "v_div_scale_f64", "v_rcp_f64")) or "_f64" in op:
        return "f64"
    return "plain"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="sweep_kernelILi8ELi1ELb0ELb0ELb0E")
    ap.add_argument("--trips", default="")
    ap.add_argument("--asm", default="/tmp/pm_mark.s")
    ap.add_argument("--md", default="")
    ap.add_argument("--no-build", action="store_true")
    a = ap.parse_args()
    if not a.no_build:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + FLAGS + ["-DHCMVS_MARK", "-S", "--cuda-device-only", "-o", a.asm, SRC],
                              stderr=subprocess.DEVNULL)
    lines = open(a.asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN5hcmvs") and ":" in l and a.kernel in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    phase = "prologue"
    per = collections.OrderedDict()
    marks = collections.Counter()
    inst = -1          # instance of score_taps in the listing (6 / 7 / 8 tap rows x call sites), told apart by its gather count
    inst_loads = collections.Counter()
    for l in lines[start:end]:
        s = l.strip()
        if s.startswith("; HCMARK"):
            phase = s.split()[-1]
            marks[phase] += 1
            if phase == "tap_warp":
                inst += 1
            if phase.startswith("tap_"):
                phase = "%s#%d" % (phase, inst)
            continue
        if phase.startswith("tap_sample") and s.startswith("global_load_dwordx4"):
            inst_loads[inst] += 2       # one 16-byte gather per tap row (2 x 2 footprint layout)
        elif phase.startswith("tap_sample") and s.startswith("global_load"):
            inst_loads[inst] += 1       # two 8-byte gathers per tap row
        if not l.startswith("\t") or not s or s.startswith((";", ".")):
            continue
        d = per.setdefault(phase, collections.Counter())
        k = classify(s.split()[0])
        d[k] += 1
        if k == "valu":
            d["valu:" + sub_class(s)] += 1
    trips = dict(pixel=1.0, row=0.0, round=0.0, chunk=0.0, eval=0.0)
    for kv in a.trips.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            trips[{"rounds": "round", "chunks": "chunk", "evals": "eval"}.get(k, k)] = float(v)
    trips.setdefault("round", 0.0)
    # fold the score_taps instances into one weighted phase each: rows(instance) = gathers / 2; the hint call site (second half of
    # the instances) never runs in this workload; weights from the mean taps per evaluation (36 / 49 / 64 taps = 6 / 7 / 8 rows)
    rows_w = {6: 0.0, 7: 1.0, 8: 0.0}
    if "taps" in trips:
        t = trips["taps"]
        ahw = trips.get("ahw", 6.0)
        big = {5.0: 36.0, 6.0: 49.0, 7.0: 64.0}[ahw]
        w_big = (t - 36.0) / (big - 36.0) if big > 36.0 else 1.0
        rows_w = {6: 1.0 - w_big, 7: 0.0, 8: 0.0}
        rows_w[{36.0: 6, 49.0: 7, 64.0: 8}[big]] += w_big
    n_inst = inst + 1
    folded = collections.OrderedDict()
    for p, d in per.items():
        if "#" not in p:
            folded[p] = d
            continue
        base, k = p.split("#")
        k = int(k)
        if k >= n_inst // 2 and n_inst > 3:
            continue                       # second call site (restore hint)
        wgt = rows_w.get(inst_loads[k] // 2, 0.0)
        f = folded.setdefault(base, collections.Counter())
        for kk, v in d.items():
            f[kk] += v * wgt
    per = folded
    for p in list(marks):
        if p.startswith("tap_"):
            marks[p] = 1
    out = []
    w = out.append
    w("| phase | what | per | static VALU (lane / mov / dpp / cvt+fract / div+trans / int mul / f64) | scalar | LDS | vmem | x trips | dynamic VALU per pixel-sweep |")
    w("|---|---|---|---|---|---|---|---|---|")
    tot_dyn = 0.0
    tot_static = collections.Counter()
    rows = []
    order = list(PHASES.keys()) + [p for p in per if p not in PHASES]
    for p in order:
        if p not in per:
            continue
        d = per[p]
        what, unit = PHASES.get(p, (p, "pixel"))
        n = marks.get(p, 1) or 1
        # a marker inside a function that is instantiated several times (score_taps for 6 / 7 / 8 tap rows) appears once per
        # instance: the instances are alternatives, one runs per evaluation, so the phase's static count is the mean of the instances
        div = n if p.startswith(("sc_", "blk_smooth")) else 1
        # sc_* markers sit in score_chunk, which is inlined once per call site (process_pixel main loop + restore hint): alternatives too
        valu = d["valu"] / div
        dyn = valu * trips.get(unit, 0.0)
        tot_dyn += dyn
        for k, v in d.items():
            tot_static[k] += v
        sc = " / ".join("%.0f" % (d["valu:" + k] / div) for k in ("lane", "mov", "dpp", "cvt/fract", "div/trans", "int mul", "f64"))
        rows.append((p, what, unit, valu, sc, d["salu"] / div, d["lds"] / div, d["vmem"] / div, trips.get(unit, 0.0), dyn))
    for r in rows:
        w("| %s | %s | %s | %.0f (%s) | %.0f | %.0f | %.0f | %.2f | %.0f |" % r)
    w("| **total** | | | static %d VALU, %d scalar, %d LDS, %d vmem in the listing | | | | | **%.0f** |" %
      (tot_static["valu"], tot_static["salu"], tot_static["lds"], tot_static["vmem"], tot_dyn))
    txt = "\n".join(out)
    print(txt)
    print("\nmarkers seen:", dict(marks))
    if a.md:
        open(a.md, "w").write(txt + "\n")


if __name__ == "__main__":
    sys.exit(main())
