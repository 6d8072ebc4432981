#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv -> counter values per DISPATCH of the kernels whose name contains <substring>, in dispatch order
(for probe scripts that launch a known sequence of variants of one kernel).  With a SEQ line (the JSON a probe printed: a list of
{"launches": n, ...} entries) the dispatches are grouped by variant and averaged.
usage: tools/pmc_by_dispatch.py <substring> <csv> [<csv> ...] [--seq <probe log>]"""
import collections, csv, json, sys

args = sys.argv[1:]
seq = None
if "--seq" in args:
    i = args.index("--seq")
    for line in open(args[i + 1]):
        if line.startswith("SEQ "):
            seq = json.loads(line[4:])
    args = args[:i]
sub, paths = args[0], args[1:]
per = collections.OrderedDict()
for path in paths:
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            rows[int(r["Dispatch_Id"])]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k, (d, cs) in enumerate(sorted(rows.items())):
        per.setdefault(k, {}).update(cs)
disp = [per[k] for k in sorted(per)]
if seq is None:
    for k, cs in enumerate(disp):
        print(k, json.dumps(cs))
    sys.exit(0)
pos = 0
for v in seq:
    n = v["launches"]
    grp = disp[pos + 1:pos + n] or disp[pos:pos + n]       # the first launch of a variant is its warm-up
    pos += n
    if not grp:
        break
    mean = {c: sum(d.get(c, 0.0) for d in grp) / len(grp) for c in grp[0]}
    tag = " ".join(f"{k}={v[k]}" for k in v if k not in ("launches", "alg_read", "alg_write"))
    extra = ""
    if "FETCH_SIZE" in mean and "alg_read" in v:
        extra += f"  read(2xFETCH) {2 * mean['FETCH_SIZE'] * 1024 / 1e9:7.3f} GB = {2 * mean['FETCH_SIZE'] * 1024 / v['alg_read']:5.2f}x alg"
    if "WRITE_SIZE" in mean and "alg_write" in v:
        extra += f"  write {mean['WRITE_SIZE'] * 1024 / 1e9:7.3f} GB = {mean['WRITE_SIZE'] * 1024 / v['alg_write']:5.2f}x alg"
    print(f"{tag:24s}{extra}  " + " ".join(f"{c}={x:.0f}" for c, x in sorted(mean.items()) if c not in ("FETCH_SIZE", "WRITE_SIZE")))
