#!/usr/bin/env python3
"""Instruction breakdown of phase B's small-node path (blas_small_kernel, try_group: nodes of <= 32 triangles, several per
wave) from the ISA: blas.hip compiled with -DVD_ISA_MARKS carries comment markers (between scheduling barriers) at the
section boundaries; this script counts the instructions between them by class and prices a batch.
    python tools/blas_small_isa.py > profiles/r04_blas_small_isa.txt        (no GPU needed: hipcc cross-compiles)"""
import os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = "/tmp/blas_marks.s"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-DVD_ISA_MARKS", "-S", "--cuda-device-only", "-o", ASM, os.path.join(ROOT, "voidin_amd", "csrc", "blas.hip")]
subprocess.run(cmd, check=True, capture_output=True)
lines = open(ASM).read().split("\n")
i0 = next(i for i, l in enumerate(lines) if "blas_small_kernel" in l and l.rstrip().endswith(":") or ("blas_small_kernel" in l and ": ;" in l))
i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[i0:i1]


def cls(op):
    if op.startswith("ds_bpermute") or op.startswith("ds_permute"): return "crossbar (ds_permute / ds_bpermute)"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait / nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "SALU"
    if op.startswith("v_") and ("_dpp" in op): return "VALU (DPP)"
    if op.startswith("v_"): return "VALU"
    return "other"


def count(a, b):
    c = collections.Counter()
    for l in body[a:b]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
        op = t.split()[0]
        dpp = " row_" in t or " quad_perm" in t or "row_shr" in t or "row_bcast" in t
        k = cls(op)
        if k == "VALU" and dpp: k = "VALU (DPP)"
        c[k] += 1
    return c


marks = [(i, l.split("VDMARK")[1].strip()) for i, l in enumerate(body) if "VDMARK" in l]
# first copy of the lambda only (the compiler makes one per call site)
def first_pair(name, nth=0):
    b = [i for i, n in marks if n == name + "_begin"]; e = [i for i, n in marks if n == name + "_end"]
    return b[nth], next(x for x in e if x > b[nth])

print("# blas_small_kernel, nodes of <= 32 triangles (try_group): instructions per section, from the ISA of blas.hip built with -DVD_ISA_MARKS")
print("# (markers sit between scheduling barriers; the compiler still hoists loop invariants out of a section, so the counts are the work that")
print("#  stays inside).  One instruction of a 64-lane wave occupies its SIMD for 4 cycles (DPP 4-8); crossbar moves run on the LDS pipe.")
sec = {}
sec["group set-up (claim nodes, load centroids, centroid bounds: 6 group reductions, 21 predicate bits)"] = count(*first_pair("setup"))
# the 21-trial loop is emitted rotated: "trial_end" (top of the loop, after set-up) precedes "trial_begin" in the text; one trial =
# [setup_end .. trial_end) + [trial_begin .. eval_begin)
mk = {n: [i for i, m in marks if m == n] for n in set(m for _, m in marks)}
t_end, t_beg = mk["trial_end"][0], mk["trial_begin"][0]
if t_end < t_beg:
    trial = count(mk["setup_end"][0], t_end) + count(t_beg, mk["eval_begin"][0])
else:
    trial = count(t_beg, t_end)
sec["ONE trial = one partition_shuffle in closed form (blas.rs:168-182)"] = trial
eb, ee = first_pair("eval_elem", 0)
sec["cost evaluation, ONE element of ONE (node, candidate) pair walk (7 crossbar fetches + box min/max)"] = count(eb, ee)
cb, ce = first_pair("cost", 0)
sec["cost evaluation, per pass of 64 pairs: two areas, cost key, atomic min"] = count(cb, ce)
fb, fe = first_pair("finish")
sec["finish (children boxes: 12 group reductions; nodes; queue the children)"] = count(fb, fe)
evb, eve = first_pair("eval")
whole_eval = count(evb, eve)
for k, c in sec.items():
    tot = sum(c.values())
    print(f"\n{k}: {tot} instructions")
    for kk, v in sorted(c.items(), key=lambda x: -x[1]): print(f"    {kk:40s} {v}")
print(f"\nwhole evaluation section as emitted (three unrolled passes + reductions): {sum(whole_eval.values())} instructions")
T = sum(sec["ONE trial = one partition_shuffle in closed form (blas.rs:168-182)"].values())
E = sum(sec["cost evaluation, ONE element of ONE (node, candidate) pair walk (7 crossbar fetches + box min/max)"].values())
C = sum(sec["cost evaluation, per pass of 64 pairs: two areas, cost key, atomic min"].values())
S = sum(list(sec.values())[0].values()); F = sum(sec["finish (children boxes: 12 group reductions; nodes; queue the children)"].values())
print("\n# a batch = one wave splitting 8 nodes of <= 8 / 4 of 9..16 / 2 of 17..32 triangles: 22 trials + the evaluation of 21 x nodes pairs")
for name, gw, nodes in (("8 nodes <= 8", 8, 8), ("4 nodes 9..16", 16, 4), ("2 nodes 17..32", 32, 2)):
    passes = (21 * nodes + 63) // 64
    ev = passes * (gw * E + C)
    tot = S + 22 * T + ev + F
    print(f"{name:16s}: set-up {S} + 22 trials x {T} = {22 * T} + evaluation {passes} passes x ({gw} x {E} + {C}) = {ev} + finish {F} = {tot} instructions per batch,"
          f" {tot / nodes:.0f} per node; trials {100 * 22 * T / tot:.0f} %, evaluation {100 * ev / tot:.0f} %")
