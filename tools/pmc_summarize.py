#!/usr/bin/env python
"""Average the rocprofv3 CSVs written by tools/pmc_collect.sh per (case of tools/kall.py, counter) and write the summary the bench
line's roofline.traffic reads: profiles/r02_pmc_summary.json.  FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md §HBM:
the counter tallies 128-byte requests at 64 bytes for wide coalesced streams); sizes are reported by rocprofv3 in KB.
Usage: python tools/pmc_summarize.py <dir> [<out.json>]"""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_pmc_summary.json")
cases = json.load(open(os.path.join(d, "cases.json")))
reps = cases["reps"]
by_kernel = collections.defaultdict(list)        # kernel name prefix -> [case]
for name, c in cases["cases"].items():
    by_kernel[c["kernel"]].append(name)


def match(kname, case_kernel):
    """rocprofv3 prints the C++ template instance; kall.py records mdx_last_kernel()'s short tag."""
    base = case_kernel.split("<")[0]
    if "+" in base:                                 # an op made of two launches (GroupNorm: statistics + apply): either kernel matches, values add up
        return any(part in kname for part in base.split("+"))
    if base not in kname:
        return False
    tag = case_kernel[len(base):]
    if base == "gemm_xl_kernel":
        bn = tag.split("x")[1].split(",")[0]; conv = "conv" in tag
        return f"<{bn}, {'true' if conv else 'false'}" in kname.replace("(int)", "").replace("Li", "") or f"{bn}, {'true' if conv else 'false'}" in kname
    if base == "gemm_xlp_kernel":                   # gemm_xlp_kernel<GEGLU, HAS_R>
        args = kname.split("gemm_xlp_kernel<")[1].split(">")[0].replace(" ", "").split(",")
        return (args[0] == "true") == ("geglu" in tag) and (args[1] == "true") == ("+res" in tag)
    if base == "attn_kernel":
        return ("true" in kname.split("attn_kernel")[1]) == ("xview" in tag)
    if base == "attn2_kernel":                      # attn2_kernel<D8, TWO, QT, FOLD>: second argument = the two-neighbour (cross-view) form
        args = kname.split("attn2_kernel<")[1].split(">")[0].replace(" ", "").split(",")
        res = len(args) > 4 and args[4] == "true"     # (round 4) fifth argument: K / V^T resident in LDS
        return (args[1] == "true") == ("xview" in tag) and (args[3] == "true") == ("fold" in tag) and res == ("resident" in tag) and (args[0] == tag.split(",")[0].lstrip("<").replace("40", "5").replace("80", "10"))
    if base == "gemm_ws_kernel":
        return ("<true" in kname.replace(" ", "")) == ("geglu" in tag)
    return True


counters = collections.defaultdict(lambda: collections.defaultdict(list))       # case -> counter -> [values in launch order]
for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    per_kernel_rows = collections.defaultdict(list)
    for r in rows:
        per_kernel_rows[r["Kernel_Name"]].append(r)
    for kname, rs in per_kernel_rows.items():
        for case_kernel, names in by_kernel.items():
            if not match(kname, case_kernel):
                continue
            by_counter = collections.defaultdict(list)
            for r in rs:
                by_counter[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for cn, vals in by_counter.items():
                # launches appear in program order: reps per case, cases of one kernel in kall.py's order
                for i, name in enumerate(names):
                    seg = vals[i * reps:(i + 1) * reps]
                    if seg:
                        prev = counters[name].get(cn)
                        counters[name][cn] = [a + b for a, b in zip(prev, seg)] if (prev and len(prev) == len(seg) and "+" in case_kernel) else seg
durations = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "**", "stats_kernel_trace.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    per_kernel_rows = collections.defaultdict(list)
    for r in rows:
        per_kernel_rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for kname, rs in per_kernel_rows.items():
        rs.sort()
        for case_kernel, names in by_kernel.items():
            if match(kname, case_kernel):
                for i, name in enumerate(names):
                    seg = [(e - s) / 1e3 for s, e in rs[i * reps:(i + 1) * reps]]
                    prev = durations.get(name)
                    durations[name] = [a + b for a, b in zip(prev, seg)] if (prev and len(prev) == len(seg) and "+" in case_kernel) else seg
summary = {"note": "rocprofv3 --kernel-trace --pmc <group> (separate passes; tools/pmc_collect.sh over tools/kall.py), %d views, averages over %d launches; "
                   "fetch = FETCH_SIZE x 2 (gfx950 correction), sizes in bytes; *_frac counters are ratios of the SQ sums" % (cases["views"], reps),
           "build_id": cases.get("build_id"),      # mdx_build_id() of the library the passes ran on: bench.py reports these counters only beside the same build
           "cases": {}, "kernels": {}}
for name, c in cases["cases"].items():
    cs = {k: sum(v) / len(v) for k, v in counters[name].items()}
    row = dict(kernel=c["kernel"], gflop=round(c["flop"] / 1e9, 1), alg_read_bytes=c["alg_read_bytes"], alg_write_bytes=c["alg_write_bytes"])
    if durations.get(name):
        us = sum(durations[name]) / len(durations[name])
        row["avg_us_profiled"] = round(us, 1)
        if c["flop"]:
            row["tflops_profiled"] = round(c["flop"] / us / 1e6, 1)
    if "FETCH_SIZE" in cs:
        row["fetch_bytes"] = int(cs["FETCH_SIZE"] * 1024 * 2)
    if "WRITE_SIZE" in cs:
        row["write_bytes"] = int(cs["WRITE_SIZE"] * 1024)
    if "fetch_bytes" in row and "write_bytes" in row:
        row["traffic_bytes"] = row["fetch_bytes"] + row["write_bytes"]
        row["traffic_over_algorithmic"] = round(row["traffic_bytes"] / (c["alg_read_bytes"] + c["alg_write_bytes"]), 3)
    if "TCC_HIT_sum" in cs:
        row["l2_hit_rate"] = round(cs["TCC_HIT_sum"] / (cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"]), 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and cs.get("SQ_BUSY_CYCLES"):
        row["mfma_busy_over_sq_busy"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / cs["SQ_BUSY_CYCLES"], 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and cs.get("GRBM_GUI_ACTIVE"):
        # matrix-pipe utilisation from the counters: busy cycles summed over the 1024 SIMDs (256 CUs x 4) / (kernel cycles x 1024).
        # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (8 x the kernel's cycles: the ratio to the wall time is 15-19 "GHz"),
        # so the kernel's cycles are GRBM_GUI_ACTIVE / 8 — the effective clock under load then reads 1.9-2.4 GHz as the guide says.
        cycles = cs["GRBM_GUI_ACTIVE"] / 8.0
        row["mfma_util"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0), 4)
        if durations.get(name):
            row["effective_clock_ghz"] = round(cycles / (sum(durations[name]) / len(durations[name])) / 1e3, 3)
    if "fetch_bytes" in row and "write_bytes" in row and durations.get(name):
        row["hbm_gbps"] = round((row["fetch_bytes"] + row["write_bytes"]) / (sum(durations[name]) / len(durations[name])) / 1e3, 1)
    if "SQ_WAVE_CYCLES" in cs and cs["SQ_WAVE_CYCLES"]:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in cs:
                row[k.lower() + "_frac"] = round(cs[k] / cs["SQ_WAVE_CYCLES"], 4)
    row["raw"] = {k: round(v, 1) for k, v in sorted(cs.items())}
    summary["cases"][name] = row
    if "traffic_bytes" in row:          # what bench.py's roofline.traffic / per_kernel report for this kernel: every profiled case is kept
        # (bench.py picks the case whose duration is closest to the kernel's average launch in the timed run); top level = largest case
        k = summary["kernels"].get(c["kernel"])
        entry = dict(case=name, views=cases["views"], gflop=row["gflop"], bytes=row["traffic_bytes"], fetch_bytes=row["fetch_bytes"],
                     write_bytes=row["write_bytes"], algorithmic_bytes=c["alg_read_bytes"] + c["alg_write_bytes"],
                     traffic_over_algorithmic=row.get("traffic_over_algorithmic"), l2_hit_rate=row.get("l2_hit_rate"), mfma_util=row.get("mfma_util"),
                     hbm_gbps=row.get("hbm_gbps"), effective_clock_ghz=row.get("effective_clock_ghz"), avg_us_profiled=row.get("avg_us_profiled"))
        prev_cases = (k or {}).get("cases", [])
        if k is None or row["gflop"] > k["gflop"]:
            summary["kernels"][c["kernel"]] = dict(case=name, views=cases["views"], gflop=row["gflop"], bytes=row["traffic_bytes"], fetch_bytes=row["fetch_bytes"],
                                                   write_bytes=row["write_bytes"], algorithmic_bytes=c["alg_read_bytes"] + c["alg_write_bytes"],
                                                   l2_hit_rate=row.get("l2_hit_rate"), mfma_util=row.get("mfma_util"), hbm_gbps=row.get("hbm_gbps"),
                                                   effective_clock_ghz=row.get("effective_clock_ghz"), avg_us_profiled=row.get("avg_us_profiled"))
        summary["kernels"][c["kernel"]]["cases"] = prev_cases + [entry]
with open(out_path, "w") as f:
    json.dump(summary, f, indent=1)
for name, row in summary["cases"].items():
    print(name, {k: v for k, v in row.items() if k != "raw"})
print("wrote", out_path)
