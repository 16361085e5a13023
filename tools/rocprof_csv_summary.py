#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output.  kernel_trace.csv -> per-kernel calls / total / avg / min / max;
counter_collection.csv -> per-kernel, per-counter average per launch.
Usage: rocprof_csv_summary.py DIR [substring filter for the counter tables]"""
import collections
import csv
import glob
import re
import sys


def main(d: str, flt: str = "") -> None:
    for f in sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)):
        k = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k[re.sub(r"\s+", " ", row["Kernel_Name"])[:110]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        tot = sum(sum(v) for v in k.values()) or 1.0
        print(f"# kernel trace {f}\n# columns: kernel | calls | total_us | avg_us | min_us | max_us | pct")
        for name, v in sorted(k.items(), key=lambda kv: -sum(kv[1])):
            print(f"{name:110s} {len(v):6d} {sum(v):12.1f} {sum(v)/len(v):10.2f} {min(v):10.2f} {max(v):10.2f} {100.0*sum(v)/tot:6.2f}")
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(float)
        disp = collections.defaultdict(set)
        for row in csv.DictReader(open(f)):
            key = (re.sub(r"\s+", " ", row["Kernel_Name"])[:70], row["Counter_Name"])
            acc[key] += float(row["Counter_Value"])
            disp[key].add(row["Dispatch_Id"])
        print(f"# counters {f}: average per launch")
        for (name, c), v in sorted(acc.items()):
            if flt in name:
                print(f"{name:72s} {c:28s} {v / len(disp[(name, c)]):18.1f}  launches {len(disp[(name, c)])}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
