#!/usr/bin/env python3
"""Summaries of a tools/profile_r03.sh run (gpurun_out/prof_<tag>) -> profiles/r03/ and the two files bench.py reads
(profiles/pmc_traffic.json, profiles/sq_counters.json).   python tools/collect_r03.py <tag> [suffix]"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
suffix = sys.argv[2] if len(sys.argv) > 2 else ""
base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles", "r03")
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("thm::dev::", "")


def counters(sub):
    """kernel -> counter -> list of per-launch values (summed over the dimensions rocprofv3 splits a counter into)"""
    fs = glob.glob(os.path.join(base, sub, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        return {}
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        per[(short(r["Kernel_Name"]), r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for (k, c, _), v in sorted(per.items(), key=lambda kv: int(kv[0][2])):
        out[k][c].append(v)
    return out


def mean_after_warmup(v, skip):
    v = v[skip:] if len(v) > skip else v
    return sum(v) / max(len(v), 1)


bench = None
bt = os.path.join(base, "bench_trace.json")
if os.path.exists(bt) and os.path.getsize(bt):
    bench = json.loads(open(bt).read().strip().splitlines()[-1])
    shutil.copy(bt, os.path.join(dst, "bench_under_rocprof%s.json" % suffix))
ks = glob.glob(os.path.join(base, "trace", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(dst, "kernel_stats_bench%s.csv" % suffix))
    for r in csv.DictReader(open(ks[0])):
        if float(r["Percentage"]) > 0.3:
            print("%-55s calls=%s avg=%.4f ms pct=%s" % (short(r["Name"])[-55:], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))

n_reads = 500000
csrc_hash = open(os.path.join(base, "csrc_hash.txt")).read().strip() if os.path.exists(os.path.join(base, "csrc_hash.txt")) else None
# the workload the profile was taken on (bench.py matches on these keys)
wl = {"reads_per_gpu": n_reads, "ref_len": 46709983, "opts": "ci", "read_len": 91, "percent": None, "wide": False}
if bench:
    cfg = bench.get("config", {})
    wl.update(reads_per_gpu=cfg.get("reads_per_gpu_per_step", n_reads), ref_len=cfg.get("ref_len", 46709983), opts=cfg.get("opts", "ci"),
              read_len=cfg.get("read_len", 91), percent=cfg.get("percent"), wide=cfg.get("coord_bytes", 4) == 8)
    n_reads = wl["reads_per_gpu"]


def put_entry(path, entry):
    """profiles/<file>.json holds one entry per workload"""
    entries = []
    if os.path.exists(path):
        try:
            old = json.load(open(path))
            entries = old if isinstance(old, list) else [old]
        except Exception:
            entries = []
    keys = ("reads_per_gpu", "ref_len", "opts", "read_len", "percent", "wide")
    entries = [e for e in entries if any(e.get(k) != entry.get(k) for k in keys)]
    entries.append(entry)
    json.dump(entries, open(path, "w"), indent=1)

# the bench makes 4 pool-sizing launches (one per resident batch) + warm-up before the timed ones: skip them
SKIP = 5
f, w = counters("pmc_fetch"), counters("pmc_write")
if f or w:
    out = {}
    for k in sorted(set(f) | set(w)):
        fm = mean_after_warmup(f.get(k, {}).get("FETCH_SIZE", [0]), SKIP)
        wm = mean_after_warmup(w.get(k, {}).get("WRITE_SIZE", [0]), SKIP)
        out[k] = dict(launches=len(f.get(k, {}).get("FETCH_SIZE", [])), fetch_bytes_raw=int(fm * 1024), write_bytes=int(wm * 1024))
    json.dump(out, open(os.path.join(dst, "pmc_fetch_write_bench%s.json" % suffix), "w"), indent=1)
    e = [v for k, v in out.items() if k.startswith("extend_kernel")]
    e = max(e, key=lambda v: v["fetch_bytes_raw"] + v["write_bytes"])
    cal = None
    cp = os.path.join(dst, "fetch_calibration.json")
    if not os.path.exists(cp):
        cp = os.path.join(ROOT, "profiles", "r02", "fetch_calibration.json")  # (the load shapes did not change)
    if os.path.exists(cp):
        cal = json.load(open(cp))
    t = dict(wl, round=3, csrc_hash=csrc_hash)
    t.update({"fetch_bytes_raw": e["fetch_bytes_raw"], "write_bytes": e["write_bytes"],
         "source": "profiles/r03/pmc_fetch_write_bench%s.json" % suffix})
    factor = cal["pattern2_window_runs"]["fetch_size_per_requested_byte"] if cal else None
    if factor:
        t["fetch_bytes_calibrated"] = int(e["fetch_bytes_raw"] / factor)
        t["extend_kernel_hbm_bytes_per_launch"] = t["fetch_bytes_calibrated"] + e["write_bytes"]
        t["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_r03.sh), KB*1024, mean over the timed "
                     "launches; FETCH_SIZE divided by the factor measured on a known-size gather in this kernel's dominant load pattern "
                     "(256-byte runs of 16-byte lane loads: profiles/r03/fetch_calibration.json)")
    else:
        t["extend_kernel_hbm_bytes_per_launch"] = e["fetch_bytes_raw"] + e["write_bytes"]
        t["note"] = "raw FETCH_SIZE (uncalibrated for gathers) + WRITE_SIZE"
    put_entry(os.path.join(ROOT, "profiles", "pmc_traffic.json"), t)
    print("traffic", t)

s1, s2 = counters("sq1"), counters("sq2")
if s1:
    summ = {}
    for k in sorted(s1):
        c = {n: mean_after_warmup(v, SKIP) for n, v in s1[k].items()}
        c.update({n: mean_after_warmup(v, SKIP) for n, v in s2.get(k, {}).items()})
        d = {n: round(v, 1) for n, v in c.items()}
        wc = c.get("SQ_WAVE_CYCLES", 0)
        if wc:
            # per-wave view (all four counters in the same unit; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES,
            # MI355X_MICROARCH.md): share of a resident wave's time in which it issues a VALU instruction / any
            # instruction, waits to issue, or is parked at s_waitcnt
            d["valu_issue_share_of_wave_cycles"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4)
            d["any_issue_share_of_wave_cycles"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4)
            d["wait_any_share_of_wave_cycles"] = round(c.get("SQ_WAIT_ANY", 0) / wc, 4)
            d["wait_inst_share_of_wave_cycles"] = round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        summ[k] = d
    json.dump(summ, open(os.path.join(dst, "sq_counters_bench%s.json" % suffix), "w"), indent=1)
    ek, e = max(((k, v) for k, v in summ.items() if k.startswith("extend_kernel")), key=lambda kv: kv[1].get("SQ_INSTS_VALU", 0))
    # SIMD view: a SIMD issues one VALU instruction at a time; with W waves resident per SIMD its vector ALU is busy
    # W x (per-wave VALU share) of the time.  W = the kernel's launch bound (third template argument: waves per SIMD
    # its register budget is compiled for).
    W = int(ek.split("<")[1].split(">")[0].split(",")[2])
    sq = dict(wl, round=3, csrc_hash=csrc_hash)
    sq.update({"extend_valu_busy_frac": round(min(1.0, W * e.get("valu_issue_share_of_wave_cycles", 0)), 4),
          "extend_waves_per_simd": W,
          "extend_valu_share_per_wave": e.get("valu_issue_share_of_wave_cycles"),
          "extend_wait_inst_share": e.get("wait_inst_share_of_wave_cycles"),
          "extend_valu_insts_per_read": round(e.get("SQ_INSTS_VALU", 0) / n_reads, 1),
          "extend_wait_any_share": e.get("wait_any_share_of_wave_cycles"),
          "source": "profiles/r03/sq_counters_bench%s.json (rocprofv3 --pmc SQ_*; valu_busy = waves per SIMD x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)" % suffix})
    put_entry(os.path.join(ROOT, "profiles", "sq_counters.json"), sq)
    for k, d in summ.items():
        if "rocclr" not in k:
            print(k, {n: d[n] for n in d if "share" in n}, "VALU/launch", d.get("SQ_INSTS_VALU"))

cal = counters("calib")
cj = os.path.join(base, "calib.json")
if cal and os.path.exists(cj):
    req = json.loads(open(cj).read().strip().splitlines()[-1])
    v = cal.get("calib_gather_kernel", {}).get("FETCH_SIZE", [])
    names = ["pattern0_8B_random", "pattern1_4B_random", "pattern2_window_runs"]
    out = {}
    for i, nme in enumerate(names):
        launches = v[3 * i: 3 * i + 3]
        if not launches:
            continue
        fetched = sum(launches[1:]) / max(len(launches) - 1, 1) * 1024
        r = req["pattern%d" % i]
        sectors = r["threads"] if i < 2 else r["threads"] // 16 * 4
        out[nme] = {"threads": r["threads"], "bytes_requested": r["bytes_requested"], "fetch_size_bytes": int(fetched),
                    "fetch_size_per_requested_byte": round(fetched / r["bytes_requested"], 4),
                    "fetch_size_per_64B_sector_touched": round(fetched / (sectors * 64), 4)}
    json.dump(out, open(os.path.join(dst, "fetch_calibration.json"), "w"), indent=1)
    print("calibration", out)
