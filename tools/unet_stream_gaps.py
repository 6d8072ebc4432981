#!/usr/bin/env python3
"""Bubbles on the U-Net stream of a pipelined bench run: gaps between consecutive U-Net kernels (conv / TDF / resampling / 1x1 / MDX STFT-iSTFT)
from a rocprofv3 --kernel-trace rocpd database, with what ran inside each gap.  Usage: tools/unet_stream_gaps.py <results.db> [min_gap_us=200] [top=12]"""
import sqlite3, sys
db = sys.argv[1]
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 200e3
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select start, end, {name} from kernels order by start").fetchall()
unet = lambda n: any(k in n for k in ("k_conv3x3", "k_tdf_", "k_resample2x", "k_conv1x1", "k_mdx_"))
u = [r for r in rows if unet(r[2])]
t0 = u[0][0]
gaps = []
end = u[0][1]
prev = u[0]
for r in u[1:]:
    if r[0] - end >= min_gap:
        gaps.append((r[0] - end, end, r[0], prev[2], r[2]))
    if r[1] > end:
        end, prev = r[1], r
span = (u[-1][1] - t0) / 1e6
print(f"U-Net kernels {len(u)}, span {span:.1f} ms, gaps >= {min_gap / 1e3:.0f} us: {len(gaps)}, total {sum(g[0] for g in gaps) / 1e6:.1f} ms")
for g in sorted(gaps, key=lambda g: -g[0])[:top]:
    inside = [r for r in rows if not unet(r[2]) and r[1] > g[1] and r[0] < g[2]]
    busy = sum(min(r[1], g[2]) - max(r[0], g[1]) for r in inside) / 1e3
    names = {}
    for r in inside:
        names[r[2][:40]] = names.get(r[2][:40], 0) + 1
    print(f"  {g[0] / 1e3:9.1f} us at +{(g[1] - t0) / 1e6:9.2f} ms  after {g[3][:34]:34s} before {g[4][:34]:34s} | {len(inside)} other kernels, {busy:.0f} us busy: "
          + ", ".join(f"{k} x{v}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:4]))
