#!/usr/bin/env python3
"""rocprofv3 passes (tools/pmc_passes.sh) -> profiles/<name>.json: per WORKLOAD, every kernel of a timed step with its
counters per launch, the steady-state launch time from the kernel trace (warm-up launches dropped), and the per-step sums
`bench.py`'s roofline is built from.  The entry carries a hash of the kernel sources it was collected on: bench.py
refuses to price counters of another build (roofline.stale).

    python tools/pmc_to_json.py gpurun_out/<dir> <workload> profiles/r03_pmc.json [--warmup W] [--steps K]

W, K: warm-up and timed steps of the profiled `bench.py --timed-only` commands (defaults 20 / 30, as pmc_passes.sh runs).
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    d = os.path.join(ROOT, "uob_raytracer_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    args = sys.argv[1:]
    W = int(args[args.index("--warmup") + 1]) if "--warmup" in args else 20
    K = int(args[args.index("--steps") + 1]) if "--steps" in args else 30
    src, workload, out = args[0], args[1], args[2]
    ours = ("uobrt::",)          # the product's kernels (torch's own fill / copy kernels are listed but not summed)

    # ---- kernel trace: per kernel, launches per step and the steady-state duration (first W steps dropped) ----------
    per = collections.defaultdict(list)
    for f in glob.glob(src + "/trace/*/*kernel_trace.csv"):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            per[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    trace = {}
    for k, v in per.items():
        lps = len(v) / float(W + K)               # launches per step
        if lps < 0.9 or abs(lps - round(lps)) > 0.05:       # (a kernel that skips a context's first frame still counts as once per step)
            trace[k] = {"launches": len(v), "launches_per_step": None, "avg_ns": sum(v) / len(v)}
            continue
        lps = int(round(lps))
        steady = v[W * lps:]
        trace[k] = {"launches": len(v), "launches_per_step": lps, "steady_launches": len(steady),
                    "avg_ns": sum(steady) / len(steady), "min_ns": min(steady), "max_ns": max(steady),
                    "avg_ns_all_launches": sum(v) / len(v)}

    # ---- counters: mean per dispatch over the steady-state dispatches of each kernel -------------------------------
    kernels = {}
    for f in glob.glob(src + "/pmc_*/*/*counter_collection.csv"):
        byk = collections.defaultdict(lambda: collections.defaultdict(dict))      # kernel -> counter -> dispatch -> value
        for r in csv.DictReader(open(f)):
            d = byk[short(r["Kernel_Name"])][r["Counter_Name"]]
            d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        for k, cs in byk.items():
            for cname, dv in cs.items():
                ids = sorted(dv)
                lps = trace.get(k, {}).get("launches_per_step")
                if lps:
                    ids = ids[W * lps:] or ids
                kernels.setdefault(k, {})[cname] = sum(dv[i] for i in ids) / len(ids)
    if not kernels:
        sys.exit("no counter CSV under %s" % src)

    def derived(c):
        e = {"valu_instructions": c.get("SQ_INSTS_VALU"), "salu_instructions": c.get("SQ_INSTS_SALU"),
             # executed FP32 operations, counted per lane as if all 64 lanes were active (an upper bound): FMA = 2 flop
             "fp32_flop_upper_bound": 64.0 * (c.get("SQ_INSTS_VALU_ADD_F32", 0) + c.get("SQ_INSTS_VALU_MUL_F32", 0) +
                                               2 * c.get("SQ_INSTS_VALU_FMA_F32", 0) + c.get("SQ_INSTS_VALU_TRANS_F32", 0)),
             # HBM bytes with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE (KB) x 2, WRITE_SIZE (KB) as read
             "hbm_bytes": (2.0 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024.0}
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_fraction"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
        if c.get("GRBM_GUI_ACTIVE"):
            e["mean_waves_per_simd"] = 4.0 * c.get("SQ_WAVE_CYCLES", 0) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if c.get("SQ_WAVE_CYCLES"):
            e["wave_cycle_shares"] = {k: c[k] / c["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in c}
        return e

    entry = {"source_sha256_16": source_hash(), "n_gpus": 1, "profiled_command": "bench.py --workload %s --timed-only --warmup %d --steps %d" % (workload, W, K),
             "kernels": {}, "other_kernels": {}}
    step = collections.defaultdict(float)
    dominant, dom_ns = None, -1.0
    for k, c in sorted(kernels.items()):
        t = trace.get(k, {})
        rec = dict(derived(c), counters_per_launch={n: c[n] for n in sorted(c)}, trace=t)
        if any(o in k for o in ours):
            entry["kernels"][k] = rec
            lps = t.get("launches_per_step") or 1
            for key in ("valu_instructions", "salu_instructions", "fp32_flop_upper_bound", "hbm_bytes"):
                if rec.get(key) is not None:
                    step[key] += lps * rec[key]
            step["kernel_ns"] += lps * t.get("avg_ns", 0.0)
            if lps * t.get("avg_ns", 0.0) > dom_ns:
                dominant, dom_ns = k, lps * t.get("avg_ns", 0.0)
        else:
            entry["other_kernels"][k] = {"trace": t, "hbm_bytes": rec["hbm_bytes"]}
    entry["dominant_kernel"] = dominant
    entry["per_step"] = dict(step)
    entry["how"] = ("tools/pmc_passes.sh: one rocprofv3 --pmc run per counter set and one --kernel-trace run over the command above; "
                    "counters and durations are means over the steady-state launches only (the first %d steps are dropped); "
                    "per_step = sum over the product's kernels of launches_per_step x per-launch value" % W)
    data = json.load(open(out)) if os.path.exists(out) else {}
    data[workload] = entry
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps({"workload": workload, "dominant": dominant, "per_step": entry["per_step"],
                      "kernels": {k: v["trace"] for k, v in entry["kernels"].items()}}, indent=1))

    # steady-state kernel statistics as a CSV beside the JSON (what profiles/*_kernel_stats_<workload>.csv holds)
    stats = os.path.splitext(out)[0].replace("_pmc", "") + "_kernel_stats_%s.csv" % workload
    with open(stats, "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "LaunchesPerStep", "SteadyLaunches", "AverageNs", "MinNs", "MaxNs", "AverageNsAllLaunches", "WarmupStepsDropped"])
        for k, t in sorted(trace.items(), key=lambda kv: -(kv[1].get("avg_ns", 0) * (kv[1].get("launches_per_step") or 0))):
            w.writerow([k, t.get("launches_per_step"), t.get("steady_launches", t["launches"]), "%.1f" % t["avg_ns"],
                        t.get("min_ns", ""), t.get("max_ns", ""), "%.1f" % t.get("avg_ns_all_launches", t["avg_ns"]), W])
    print("wrote", out, "and", stats)


if __name__ == "__main__":
    main()
