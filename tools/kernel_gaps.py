#!/usr/bin/env python3
"""GPU idle time between kernels from a rocprofv3 --kernel-trace rocpd database: the union of all kernels' [start, end) intervals
over the trace, the idle gaps between them, and which kernels sit on either side of the largest ones.  Answers "how much of a
pipelined step is the GPU not running anything" (the host-bound hand-over between two tracks' U-Nets).
Usage: tools/kernel_gaps.py <results.db> [min_gap_us=20] [top=25]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 20e3
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select start, end, {name} from kernels order by start").fetchall()
    if not rows:
        print("no kernels"); return
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy = 0
    gaps = []
    cur_s, cur_e, cur_name = rows[0]
    for s, e, n in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0, cur_name, n))
            cur_s, cur_e, cur_name = s, e, n
        elif e > cur_e:
            cur_e, cur_name = e, n
    busy += cur_e - cur_s
    span = t1 - t0
    print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms ({100.0 * busy / span:.2f} %)  idle {(span - busy) / 1e6:.2f} ms")
    big = [g for g in gaps if g[0] >= min_gap]
    print(f"gaps >= {min_gap / 1e3:.0f} us: {len(big)}, {sum(g[0] for g in big) / 1e6:.2f} ms;  gaps below: {len(gaps) - len(big)}, {sum(g[0] for g in gaps if g[0] < min_gap) / 1e6:.2f} ms")
    for g in sorted(big, key=lambda g: -g[0])[:top]:
        print(f"  {g[0] / 1e3:9.1f} us at +{g[1] / 1e6:10.2f} ms   after {g[2][:60]:60s} before {g[3][:60]}")
    # context of the largest gaps that lie between two U-Net forwards (a bubble in the pipelined bench): the kernels on either side
    if len(sys.argv) > 4:
        ctx = int(sys.argv[4])
        extra = [c_ for c_ in ("queue_id", "stream_id") if c_ in cols]
        rows2 = c.execute(f"select start, end, {name}" + "".join(", " + e for e in extra) + " from kernels order by start").fetchall()
        starts = [r[0] for r in rows2]
        import bisect
        unet_k = ("k_conv3x3", "k_tdf", "k_resample2x", "k_conv1x1")
        inner = [g for g in sorted(big, key=lambda g: -g[0]) if 0.2 * span < g[1] < 0.98 * span][:3]
        for g in inner:
            t_gap_end = t0 + g[1] + g[0]
            i = bisect.bisect_left(starts, t_gap_end)
            print(f"--- gap of {g[0] / 1e3:.1f} us ending at +{(g[1] + g[0]) / 1e6:.2f} ms; columns {extra}")
            for r in rows2[max(0, i - ctx):i + ctx]:
                print(f"   +{(r[0] - t0) / 1e6:10.3f} ms  run {(r[1] - r[0]) / 1e3:8.1f} us  {r[3:]}  {r[2][:70]}")
    # idle inside U-Net forwards vs between them: a gap whose both neighbours are U-Net kernels counts as "inside"
    unet = ("k_conv3x3", "k_tdf", "k_resample2x", "k_conv1x1")
    inside = sum(g[0] for g in gaps if any(u in g[2] for u in unet) and any(u in g[3] for u in unet))
    print(f"idle between two U-Net kernels: {inside / 1e6:.2f} ms; elsewhere: {(span - busy - inside) / 1e6:.2f} ms")


if __name__ == "__main__":
    main()
