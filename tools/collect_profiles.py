#!/usr/bin/env python3
"""Copies the summaries of a tools/profile.sh run (gpurun_out/prof_<tag>) and of the bench runs next to it into
profiles/r01/ and refreshes profiles/pmc_traffic.json.   python tools/collect_profiles.py <tag>"""
import collections, csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles", "r01")


def agg(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return d


f = agg(os.path.join(base, "pmc_fetch", tag + "_counter_collection.csv"), "FETCH_SIZE")
w = agg(os.path.join(base, "pmc_write", tag + "_counter_collection.csv"), "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    fv, wv = f.get(k, [0]), w.get(k, [0])
    fm = sum(fv[1:]) / max(len(fv) - 1, 1) if len(fv) > 2 else sum(fv) / len(fv)  # skip the warm-up launch
    wm = sum(wv[1:]) / max(len(wv) - 1, 1) if len(wv) > 2 else sum(wv) / len(wv)
    out[k] = dict(launches=len(fv), fetch_bytes_raw=int(fm * 1024), write_bytes=int(wm * 1024))
json.dump(out, open(os.path.join(dst, "pmc_fetch_write_bench_ci_500k_final.json"), "w"), indent=1)
shutil.copy(os.path.join(base, "trace", tag + "_kernel_stats.csv"), os.path.join(dst, "kernel_stats_bench_ci_500k_final.csv"))
shutil.copy(os.path.join(base, "bench_trace.json"), os.path.join(dst, "bench_ci_under_rocprof_final.json"))
for src, name in (("bench_ci.json", "bench_ci_final.json"), ("bench_default.json", "bench_default_opts_final.json"),
                  ("bench_chrM.json", "bench_chrM_config2.json"), ("bench_150.json", "bench_150bp_band64.json")):
    p = os.path.join(ROOT, "gpurun_out", src)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, name))
e = [v for k, v in out.items() if "extend_kernel" in k][0]
tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
t = json.load(open(tp))
t["extend_kernel_hbm_bytes_per_launch"] = e["fetch_bytes_raw"] + e["write_bytes"]
t["fetch_bytes_raw"], t["write_bytes"] = e["fetch_bytes_raw"], e["write_bytes"]
json.dump(t, open(tp, "w"), indent=1)
for r in csv.DictReader(open(os.path.join(dst, "kernel_stats_bench_ci_500k_final.csv"))):
    if float(r["Percentage"]) > 0.3:
        print("%-45s calls=%s avg=%.4f ms pct=%s" % (r["Name"].split("(")[0][-45:], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
for n in ("bench_ci_final", "bench_default_opts_final", "bench_chrM_config2", "bench_150bp_band64"):
    d = json.load(open(os.path.join(dst, n + ".json")))
    print(n, d["value"], d["roofline"]["stage_ms"], d["roofline"]["frac"], d["roofline"]["traffic"])
