#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection CSVs to per-kernel means per dispatch.

usage: pmc_summary.py OUT.json DIR [DIR ...]
Each DIR is the -d directory of one rocprofv3 --pmc pass (separate passes, as MI355X_MICROARCH.md prescribes:
FETCH_SIZE and WRITE_SIZE do not fit one pass).  The HBM figures apply the guide's gfx950 corrections: both
counters are reported in kilobytes; FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B, so it
is doubled; WRITE_SIZE is taken as is.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                    name = name.split("(")[0].replace("sfmloc::", "")
                    if not name.startswith("k_hamming"):   # the Hamming kernels keep their template arguments: the full-bank
                        name = name.split("<")[0]          # scan <8, 10, 1> and the short-list form <8, 10, 4> are different kernels
                    a = acc[name][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    out = {}
    for k, cs in sorted(acc.items()):
        o = {c: {"mean_per_dispatch": v[0] / v[1], "dispatches": v[1]} for c, v in sorted(cs.items())}
        if "FETCH_SIZE" in o or "WRITE_SIZE" in o:
            rd = 2.0 * 1024.0 * o.get("FETCH_SIZE", {}).get("mean_per_dispatch", 0.0)
            wr = 1024.0 * o.get("WRITE_SIZE", {}).get("mean_per_dispatch", 0.0)
            o["hbm_bytes_per_dispatch"] = {"read(FETCH_SIZE*1024*2)": rd, "write(WRITE_SIZE*1024)": wr, "total": rd + wr}
        out[k] = o
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    for k, o in out.items():
        if "hbm_bytes_per_dispatch" in o:
            print(k, o["hbm_bytes_per_dispatch"])


if __name__ == "__main__":
    main()
