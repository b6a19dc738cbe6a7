"""Summarise a tools/prof_bench.sh output directory into a small text/JSON report (for profiles/)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read_csvs(pattern):
    rows = []
    for p in glob.glob(pattern, recursive=True):
        with open(p) as f:
            rows += list(csv.DictReader(f))
    return rows


def main(d):
    out = {"dir": d}
    stats = read_csvs(os.path.join(d, "trace", "**", "*kernel_stats.csv"))
    out["kernel_stats"] = [
        {k: r.get(k) for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")}
        for r in stats]
    for tag in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
        rows = read_csvs(os.path.join(d, tag, "**", "*counter_collection.csv"))
        agg = defaultdict(lambda: defaultdict(list))
        for r in rows:
            agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out[tag] = {k: {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()}
                    for k, cs in agg.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
