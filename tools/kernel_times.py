#!/usr/bin/env python3
"""Per-kernel launch count, total, median and longest duration from a rocprofv3 --kernel-trace CSV
(searched below the given directory).  Development aid."""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit(f"no *kernel_trace.csv below {root}")
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        d[r["Kernel_Name"][:44]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
        v2 = sorted(v)
        print(f"{k:46s} n={len(v):5d} total={sum(v) / 1e3:8.2f} ms  med={v2[len(v2) // 2]:8.1f} us  max={v2[-1]:8.1f} us")


if __name__ == "__main__":
    main()
