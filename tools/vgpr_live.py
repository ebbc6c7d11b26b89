#!/usr/bin/env python3
"""Rough VGPR liveness of one kernel in an AMDGPU assembly listing made with -gline-tables-only:
   python3 tests/tools/vgpr_live.py /tmp/pm_g.s sweep_kernelILi8ELi1
Prints the maximum number of live VGPRs, where it occurs, and for every register live at that point the source line
of the instruction that defined it (a hint for what to park).  Defs are treated as kills (exec masks are ignored)."""
import re
import sys
from collections import defaultdict

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().split(":")[0].endswith(tuple("i")) or (l.startswith("_ZN") and key in l and ":" in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
ins = []      # (mnemonic, defs, uses, srcline, text)
labels = {}
cur_loc = ""
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    if t.startswith(".loc"):
        m = re.findall(r"pm_kernels\.hip:(\d+):\d+", t)
        cur_loc = "/".join(m[-3:]) if m else t.split(";")[-1].strip()[-40:]
        continue
    if t.endswith(":") or re.match(r"^\.?[A-Za-z_0-9$]+:", t):
        labels[t.split(":")[0]] = len(ins)
        continue
    if t.startswith("."):
        continue
    code = t.split(";")[0].strip()
    if not code:
        continue
    parts = code.split(None, 1)
    mn = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    def regs(o):
        r = set()
        for a, b, c in REG.findall(o):
            if a: r.add(int(a))
            else: r.update(range(int(b), int(c) + 1))
        return r
    alluse = mn.startswith(("global_store", "ds_write", "ds_store", "buffer_store", "flat_store", "scratch_store", "v_cmp", "v_readlane", "v_readfirstlane", "s_", "global_atomic", "ds_add", "v_cmpx")) and "_rtn" not in mn
    if mn.startswith("global_atomic") and "sc0" in code:  # returning atomic
        alluse = False
    d, u = set(), set()
    for k, o in enumerate(ops):
        if k == 0 and not alluse:
            d |= regs(o)
        else:
            u |= regs(o)
    if mn.startswith(("v_writelane", "v_mac", "v_fmac", "v_pk_fmac", "v_dot")) or "dpp" in code or "sdwa" in code:
        u |= d
    if mn in ("v_div_fmas_f32",):
        pass
    ins.append((mn, d, u, cur_loc, code))
n = len(ins)
succ = [[] for _ in range(n)]
for i, (mn, d, u, loc, code) in enumerate(ins):
    tgt = code.split()[-1] if mn.startswith(("s_cbranch", "s_branch")) else None
    if mn == "s_branch":
        if tgt in labels: succ[i].append(labels[tgt])
        continue
    if mn.startswith("s_cbranch") and tgt in labels:
        succ[i].append(labels[tgt])
    if mn == "s_endpgm":
        continue
    if i + 1 < n:
        succ[i].append(i + 1)
livein = [set() for _ in range(n)]
changed = True
while changed:
    changed = False
    for i in range(n - 1, -1, -1):
        out = set()
        for s_ in succ[i]:
            out |= livein[s_]
        mn, d, u, loc, code = ins[i]
        new = (out - d) | u
        if new != livein[i]:
            livein[i] = new
            changed = True
mx = max(range(n), key=lambda i: len(livein[i]))
print("instructions", n, "max live", len(livein[mx]), "at", mx, ins[mx][3], ins[mx][4])
# definition sites of the registers live at the peak: nearest preceding def in listing order
hint = defaultdict(list)
for r in sorted(livein[mx]):
    j = mx - 1
    while j >= 0 and r not in ins[j][1]:
        j -= 1
    hint[ins[j][3] if j >= 0 else "?"].append(r)
for loc, rs in sorted(hint.items(), key=lambda kv: -len(kv[1])):
    print("%3d regs  defined at line(s) %-28s %s" % (len(rs), loc, " ".join("v%d" % r for r in rs)))
