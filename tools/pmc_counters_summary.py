#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv files -> mean counter value per launch per kernel (only kernels with >= 1e6 ns-scale work:
names matching the U-Net kernels).  Usage: tools/pmc_counters_summary.py <out.json> <csv> [<csv> ...]"""
import collections, csv, json, sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("k_conv3x3", "k_tdf_linear", "k_resample2x")):
            continue
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
res = {k: {c: v[1] / max(1, v[0]) for c, v in cs.items()} for k, cs in agg.items()}
json.dump(res, open(out, "w"), indent=1)
for k, cs in res.items():
    print(k[:90])
    wc = cs.get("SQ_WAVE_CYCLES")
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {v:16.0f}" + (f"  {100.0 * v / wc:6.1f} % of SQ_WAVE_CYCLES" if wc and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" else ""))
