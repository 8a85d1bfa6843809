#!/usr/bin/env python3
"""Prints the kernel timeline of the LAST full step found in a rocprofv3 kernel trace
(`rocprofv3 --kernel-trace --output-format csv`): start offset, duration, queue, kernel name, and
the idle gap on the critical queue.  Development aid: shows what sits between the LSTM sweeps.

usage: trace_timeline.py <kernel_trace.csv> [first_critic_iteration_only]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    return name[:60]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
                         int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
    rows.sort()
    # a full step ends with the generator's output-layer update (the largest s2g_k_adam grid, or the fused
    # weight-gradient + Adam kernel); take the last two
    adam = [i for i, r in enumerate(rows) if (r[3].startswith("s2g_k_adam") and r[4] > 1000)
            or r[3].startswith("s2g_k_linear_wgrad_adam")]
    lo, hi = adam[-2] + 1, adam[-1] + 1
    step = rows[lo:hi]
    t0 = step[0][0]
    print(f"step: {len(step)} kernels, {(step[-1][1] - t0) / 1e3:.1f} us")
    last_end = {}
    busy_until = t0
    for s, e, q, name, blocks in step:
        gap_all = s - busy_until            # time with NO kernel running anywhere
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}us q{q} blocks={blocks:<7d} idle_before={max(gap_all, 0) / 1e3:6.1f} {name}")
        busy_until = max(busy_until, e)
        last_end[q] = e


if __name__ == "__main__":
    main()
