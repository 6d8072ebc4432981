#!/usr/bin/env python3
"""What runs after a track's U-Net: from a rocprofv3 --kernel-trace rocpd database of `bench.py --pipeline-depth 1`, every kernel and
copy between the LAST U-Net kernel of the last forward of a track (k_conv1x1_small) and the first kernel of the next track's U-Net
(or the end of the trace), with the idle gap in front of each one.  This is the exposed tail `single_stream_latency_ms` sees.
Usage: tools/track_tail_timeline.py <results.db> [track index from the end = 1]"""
import sqlite3, sys

db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select start, end, {name} from kernels order by start").fetchall()
try:
    rows += [(s, e, "memcpy:" + str(n)) for s, e, n in c.execute("select start, end, name from memory_copies")]
except sqlite3.Error:
    pass
rows.sort()
lasts = [i for i, r in enumerate(rows) if "k_conv1x1_small" in r[2]]
firsts = [i for i, r in enumerate(rows) if "k_mdx_stft" in r[2]]
# forwards come in pairs per track (two sub-batches of 32 items): the track's last forward is every second k_conv1x1_small
ends = lasts[1::2] if len(lasts) % 2 == 0 else lasts
i0 = ends[-back]
nxt = [f for f in firsts if f > i0]
i1 = nxt[0] if nxt else len(rows)
t0 = rows[i0][1]
print(f"tail of track {len(ends) - back + 1} of {len(ends)}: {i1 - i0 - 1} kernels/copies, {(rows[i1 - 1][1] - t0) / 1e6:.2f} ms from the end of the U-Net to the end of the last one")
prev_end = t0
busy = 0
agg = {}
for s, e, n in rows[i0 + 1:i1]:
    gap = s - prev_end
    print(f"+{(s - t0) / 1e6:8.3f} ms  gap {max(gap, 0) / 1e3:8.1f} us  run {(e - s) / 1e3:8.1f} us  {n[:90]}")
    busy += e - s
    a = agg.setdefault(n[:60], [0, 0]); a[0] += 1; a[1] += e - s
    prev_end = max(prev_end, e)
print(f"busy {busy / 1e6:.2f} ms of {(prev_end - t0) / 1e6:.2f} ms")
for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"  {t / 1e6:7.3f} ms  x{k:3d}  {n}")
