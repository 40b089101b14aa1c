#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel (mean per dispatch).  usage: pmc_summary.py <dir> [<dir>...]"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in rows:
            k = r["Kernel_Name"].split("(")[0][-60:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        for k in acc:
            if "rt_draw" not in k:
                continue
            print("%s  [%s, %d dispatches]" % (k, d, len(disp[k])))
            for c, v in sorted(acc[k].items()):
                print("    %-24s %.5g per dispatch" % (c, v / len(disp[k])))
