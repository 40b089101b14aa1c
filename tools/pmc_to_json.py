#!/usr/bin/env python3
"""rocprofv3 --pmc passes (tools/pmc_passes.sh) -> profiles/<name>.json: mean per dispatch of every counter for the
timed kernel of a workload, plus the quantities bench.py's roofline is built from.

    python tools/pmc_to_json.py gpurun_out/<dir> <workload> <kernel-substring> profiles/r02_pmc.json
"""
import collections
import csv
import glob
import json
import os
import sys

src, workload, kernel, out = sys.argv[1:5]
acc = collections.defaultdict(float)
disp = collections.defaultdict(set)
name = None
for f in glob.glob(src + "/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if kernel not in k:
            continue
        name = k.split("(")[0]
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        disp[r["Counter_Name"]].add(r["Dispatch_Id"])
if not acc:
    sys.exit("no dispatch of a kernel matching %r under %s" % (kernel, src))
c = {k: acc[k] / len(disp[k]) for k in sorted(acc)}
trace = {}
for f in glob.glob(src + "/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if kernel in r["Name"]:
            trace = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
entry = {
    "kernel": name, "n_gpus": 1, "dispatches_averaged": max(len(v) for v in disp.values()),
    "counters_per_launch": c,
    "valu_instructions_per_launch": c.get("SQ_INSTS_VALU"),
    "salu_instructions_per_launch": c.get("SQ_INSTS_SALU"),
    # executed FP32 operations, counted per lane as if all 64 lanes were active (an upper bound): FMA = 2 flop
    "fp32_flop_per_launch_upper_bound": 64.0 * (c.get("SQ_INSTS_VALU_ADD_F32", 0) + c.get("SQ_INSTS_VALU_MUL_F32", 0) +
                                                 2 * c.get("SQ_INSTS_VALU_FMA_F32", 0) + c.get("SQ_INSTS_VALU_TRANS_F32", 0)),
    # HBM bytes per launch with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE (KB) x 2, WRITE_SIZE (KB) as read
    "hbm_bytes_per_launch": (2.0 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024.0,
    "lds_bank_conflict_fraction": c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
    "mean_waves_per_simd": 4.0 * c.get("SQ_WAVE_CYCLES", 0) / max(c.get("GRBM_GUI_ACTIVE", 0) / 8.0 * 1024.0, 1) if "GRBM_GUI_ACTIVE" in c else None,
    "wave_cycle_shares": {k: c[k] / c["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in c and "SQ_WAVE_CYCLES" in c},
    "kernel_trace": trace,
    "how": "tools/pmc_passes.sh: one rocprofv3 --pmc run per counter set over `bench.py --steps 10 --warmup 3 --timed-only"
           "`, mean per dispatch of the timed kernel; kernel_trace = rocprofv3 --kernel-trace --stats of the same command",
}
data = json.load(open(out)) if os.path.exists(out) else {}
data[workload] = entry
json.dump(data, open(out, "w"), indent=1)
print(json.dumps({k: entry[k] for k in entry if k != "counters_per_launch"}, indent=1))
