#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.

Collection (MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots): the two counters need separate passes
(FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2), each run with --pmc only:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 1 --warmup 1 --cpu-baseline-seconds 0
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 1 --warmup 1 --cpu-baseline-seconds 0
Correction for gfx950: counter unit is KiB; FETCH_SIZE reports exactly half the bytes of a wide coalesced
streaming read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-byte-per-lane stores.
Usage: tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = load(fetch), load(write)
    rows = {}
    for k in set(f) | set(w):
        fc, fv = f.get(k, [0, 0.0]); wc, wv = w.get(k, [0, 0.0])
        calls = max(fc, wc, 1)
        rd = 2.0 * fv * 1024.0 / max(fc, 1)
        wr = wv * 1024.0 / max(wc, 1)
        rows[k] = {"launches": calls, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                   "fetch_size_kib_raw": fv / max(fc, 1), "write_size_kib_raw": wv / max(wc, 1)}
    ordered = dict(sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))
    json.dump({"note": "read = 2*FETCH_SIZE KiB (gfx950 half-count correction), write = WRITE_SIZE KiB; separate --pmc passes",
               "kernels": ordered}, open(out, "w"), indent=1)
    for k, v in list(ordered.items())[:25]:
        print(f"{k[:70]:70s} n={v['launches']:5d} read {v['read_bytes_per_launch']/1e6:10.1f} MB write {v['write_bytes_per_launch']/1e6:10.1f} MB")


if __name__ == "__main__":
    main()
