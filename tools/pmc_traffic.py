#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes over bench.py (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE,
each with --kernel-trace --output-format csv) into profiles/pmc_traffic.json: HBM bytes per launch of
every libs2g kernel.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <tag> [<kernel_stats.csv of a --kernel-trace --stats run>]

With the stats file, every kernel also gets its average duration from that (un-instrumented) run and the
HBM rate bytes / duration (counter passes serialise the kernels, so their own timings are not used).

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B read request,
i.e. half the bytes of coalesced wide reads, so it is doubled; WRITE_SIZE is taken as is.  Check of the
doubling on this code's own access pattern: s2g_k_lstm2_bwd2 reads 4 x 800 + 400 B per (row, layer) over
2 x 32768 rows = 236 MB by construction, and 2 x FETCH_SIZE = 238 MB."""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    f = (glob.glob(os.path.join(d, "*counter_collection.csv")) + glob.glob(os.path.join(d, "*", "*counter_collection.csv")))[0]
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}, dict(cnt)


def main():
    fetch_dir, write_dir, tag = sys.argv[1:4]
    durations = {}
    if len(sys.argv) > 4:
        for r in csv.DictReader(open(sys.argv[4])):
            durations[r["Name"].split("(")[0].replace("void ", "").strip()] = (float(r["AverageNs"]), int(r["Calls"]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (fe, nfe), (wr, nwr) = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    # per-kernel summaries of the two passes (the raw per-dispatch CSVs are ~0.5 MB each)
    for name, counter, vals, cnts in (("fetch", "FETCH_SIZE", fe, nfe), ("write", "WRITE_SIZE", wr, nwr)):
        with open(os.path.join(root, "profiles", f"{tag}_pmc_{name}.csv"), "w") as f:
            f.write(f"Kernel_Name,Launches,Mean_{counter}_KiB\n")
            for k in sorted(vals, key=lambda k: -vals[k] * cnts[k]):
                f.write(f"\"{k}\",{cnts[k]},{vals[k]:.1f}\n")
    out = {}
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith("s2g_k_"):
            continue
        rd, w = 2.0 * fe.get(k, 0.0) * 1024.0, wr.get(k, 0.0) * 1024.0
        extra = {}
        if k in durations:
            ns, calls = durations[k]
            extra = {"avg_ns": ns, "calls_timed": calls, "gb_per_s": (rd + w) / ns, "frac_of_8_tb_s": (rd + w) / ns / 8000.0}
        out[k] = {**extra, "bytes_per_launch": rd + w, "read_bytes": rd, "write_bytes": w,
                  "source": f"profiles/{tag}_pmc_fetch.csv + profiles/{tag}_pmc_write.csv (rocprofv3 --pmc FETCH_SIZE / "
                            "WRITE_SIZE, separate passes; FETCH_SIZE doubled per the gfx950 correction), mean over launches"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in out.items():
        print(f"{k:36s} {v['bytes_per_launch'] / 1e6:10.1f} MB/launch (read {v['read_bytes'] / 1e6:.1f}, write {v['write_bytes'] / 1e6:.1f})")


if __name__ == "__main__":
    main()
