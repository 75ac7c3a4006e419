"""Per-frame device timeline of the tracker bench from a rocprofv3 --kernel-trace --memory-copy-trace run (scripts/gpu_trk_trace.sh):
the kernels and copies of the LAST frames in start order, start offset from the frame's first operation, duration, gap to the previous end."""
import csv, glob, sys, statistics
d = sys.argv[1]
ops = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("mcp::", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", ""))))
ops.sort()
# frames: a frame starts with its pyramid kernel
frames, cur = [], []
for o in ops:
    if cur and o[2].startswith("k_pyr_fast"):
        frames.append(cur); cur = []
    cur.append(o)
if cur: frames.append(cur)
tail = [f for f in frames[-12:-1] if len(f) == len(frames[-2])]
print("frames in trace %d, operations per frame %d, frames averaged %d" % (len(frames), len(frames[-2]), len(tail)))
n = len(tail[0])
prev_end = None
tot_busy = 0.0
for i in range(n):
    st = statistics.median(f[i][0] - f[0][0] for f in tail)/1e3
    du = statistics.median(f[i][1] - f[i][0] for f in tail)/1e3
    gap = statistics.median(f[i][0] - max(e for _, e, _ in f[:i]) for f in tail)/1e3 if i else 0.0
    tot_busy += du
    print("%8.1f us  +%6.1f  gap %6.1f  %s" % (st, du, gap, tail[0][i][2][:60]))
span = statistics.median(max(e for _, e, _ in f) - f[0][0] for f in tail)/1e3
period = statistics.median(b[0][0] - a[0][0] for a, b in zip(tail[:-1], tail[1:]))/1e3
print("device span of a frame %.1f us (operations %.1f), frame period %.1f us" % (span, tot_busy, period))
