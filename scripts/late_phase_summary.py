"""Per-kernel time of the frames >= F0 of a rocprofv3 kernel trace (frames delimited by k_cell_key_hist launches)."""
import csv, glob, sys, collections
d, f0 = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
frame = 0
agg = collections.defaultdict(lambda: [0, 0])
t_first = t_last = None
for s, e, k in rows:
    if "k_cell_key_hist" in k:
        frame += 1
    if frame >= f0:
        a = agg[k]; a[0] += 1; a[1] += e - s
        t_first = s if t_first is None else t_first
        t_last = e
nf = frame - f0 + 1
print(f"# frames {f0}-{frame}: {nf} frames, {(t_last - t_first) / nf / 1e6:.2f} ms/frame by GPU timestamps; per-kernel totals per FRAME")
print("kernel,launches_per_frame,ms_per_frame,avg_us")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'"{k[:110]}",{c / nf:.1f},{t / nf / 1e6:.3f},{t / c / 1e3:.1f}')
