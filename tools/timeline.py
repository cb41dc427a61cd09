#!/usr/bin/env python3
"""Developer tool: digest a rocprofv3 --kernel-trace CSV of the pipelined bench: GPU busy fraction,
concurrency histogram and a per-queue listing of one steady-state window.
    python tools/timeline.py <kernel_trace.csv> [--window-ms 8] [--list]"""
import argparse, csv, re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--window-ms", type=float, default=8.0)
ap.add_argument("--list", action="store_true")
ap.add_argument("--end-ms", type=float, default=None, help="window end, ms after the first kernel (default: 2 ms before the last)")
a = ap.parse_args()
rows = []
for r in csv.DictReader(open(a.csv)):
    name = re.sub(r"^void ", "", r["Kernel_Name"])
    name = re.sub(r"pwclo::", "", name).split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), name))
rows.sort()
t_end = rows[-1][1]
w1 = t_end - int(2e6) if a.end_ms is None else rows[0][0] + int(a.end_ms * 1e6)
w0 = w1 - int(a.window_ms * 1e6)
win = [r for r in rows if r[1] > w0 and r[0] < w1]
ev = []
for s, e, q, n in win:
    ev.append((max(s, w0), 1)); ev.append((min(e, w1), -1))
ev.sort()
hist, cur, last = {}, 0, w0
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
hist[cur] = hist.get(cur, 0) + (w1 - last)
tot = float(w1 - w0)
print("window %.2f ms, %d kernels" % (tot / 1e6, len(win)))
for k in sorted(hist):
    print("  %d kernels running: %5.1f %%" % (k, 100 * hist[k] / tot))
fam = {}
for s, e, q, n in win:
    key = n.split("<")[0]
    fam[key] = fam.get(key, 0) + (min(e, w1) - max(s, w0))
for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:25]:
    print("  %-34s %7.3f ms kernel-time in window" % (k, v / 1e6))
if a.list:
    for s, e, q, n in win:
        print("q%-2d %9.1f -> %9.1f  (%7.1f us)  %s" % (q, (s - w0) / 1e3, (e - w0) / 1e3, (e - s) / 1e3, n[:70]))
