#!/usr/bin/env python
"""Compare two per-op tables of the step program (`bench.py --ops-json`): per kernel and per op, which launches got slower / faster.
A throughput number hides a slow kernel behind a fast one (round 3: +1 % end to end with layernorm_kernel 2.6x slower); this does not.
Box-to-box spread is +-3-5 % on MFMA-bound kernels: pass --scale to normalise table B by the ratio of the two step totals.

Usage: python tools/ops_diff.py profiles/r03e_ops_b128.json profiles/r03d_ops_b128.json [--thresh 1.15] [--scale]"""
import argparse
import collections
import json


def load(path):
    ops = json.load(open(path))
    return ops, sum(r["ms"] for r in ops)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a"); ap.add_argument("b")
    ap.add_argument("--thresh", type=float, default=1.15, help="report ops / kernels whose time ratio B / A is outside [1 / thresh, thresh]")
    ap.add_argument("--scale", action="store_true", help="divide B's times by total(B) / total(A) first (different boxes)")
    a = ap.parse_args()
    A, ta = load(a.a)
    B, tb = load(a.b)
    f = tb / ta if a.scale else 1.0
    print(f"A = {a.a}: {ta:.2f} ms per step, {len(A)} ops;  B = {a.b}: {tb:.2f} ms, {len(B)} ops" + (f";  B scaled by 1 / {f:.3f}" if a.scale else ""))
    ka, kb = collections.defaultdict(lambda: [0.0, 0]), collections.defaultdict(lambda: [0.0, 0])
    for tab, ops in ((ka, A), (kb, B)):
        for r in ops:
            tab[r["kernel"]][0] += r["ms"]; tab[r["kernel"]][1] += 1
    print("\nper kernel (ms per step, launches):")
    for k in sorted(set(ka) | set(kb), key=lambda k: -(ka[k][0] + kb[k][0])):
        x, y = ka[k][0], kb[k][0] / f
        flag = ""
        if x > 0 and y > 0 and not (1 / a.thresh <= y / x <= a.thresh) and max(x, y) > 0.05:
            flag = f"   <-- x{y / x:.2f}"
        elif (x == 0) != (y == 0):
            flag = "   <-- only in one table"
        print(f"  {k:52s} {x:9.3f} ({ka[k][1]:3d})   {y:9.3f} ({kb[k][1]:3d}){flag}")
    bn = {r["name"]: r for r in B}
    rows = []
    for r in A:
        q = bn.get(r["name"])
        if q is None or r["ms"] <= 0:
            continue
        ratio = q["ms"] / f / r["ms"]
        if not (1 / a.thresh <= ratio <= a.thresh) and max(r["ms"], q["ms"]) > 0.02:
            rows.append((ratio, r["name"], r["kernel"], q["kernel"], r["ms"], q["ms"] / f))
    print(f"\nops outside x{a.thresh} ({len(rows)}):")
    for ratio, name, k1, k2, x, y in sorted(rows, reverse=True)[:60]:
        print(f"  x{ratio:5.2f}  {name:40s} {x:8.4f} -> {y:8.4f} ms   {k1}" + ("" if k1 == k2 else f" -> {k2}"))
    only = [r["name"] for r in A if r["name"] not in bn] + [n for n in bn if n not in {r["name"] for r in A}]
    if only:
        print(f"\nops present in only one table ({len(only)}): " + ", ".join(only[:12]) + (" ..." if len(only) > 12 else ""))


if __name__ == "__main__":
    main()
