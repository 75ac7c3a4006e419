"""Per-frame device timeline of the tracker bench (or, with a second argument naming the first kernel of a period, of anything periodic: e.g. k_head_small for the window's LM iterations) from a rocprofv3 --kernel-trace --memory-copy-trace run (scripts/gpu_trk_trace.sh):
the kernels and copies of the LAST frames in start order, start offset from the frame's first operation, duration, gap to the previous end."""
import csv, glob, sys, statistics
d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "k_pyr_fast"      # the kernel a frame (or iteration) starts with
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
    if cur and o[2].startswith(first):
        frames.append(cur); cur = []
    cur.append(o)
if cur: frames.append(cur)
import collections
want = int(sys.argv[3]) if len(sys.argv) > 3 else collections.Counter(len(f) for f in frames[1:-1]).most_common(1)[0][0]      # periods of the most common length (or of the length given)
must = sys.argv[4] if len(sys.argv) > 4 else None            # only periods that contain this kernel (e.g. k_export_state: the end of a call)
if must:
    cand = [f for f in frames[1:-1] if any(must in o[2] for o in f)]
    want = collections.Counter(len(f) for f in cand).most_common(1)[0][0]
    tail = [f for f in cand if len(f) == want][-12:]
else:
    tail = [f for f in frames[1:-1] if len(f) == want][-12:]
print("periods in trace %d, operations per period %d, periods averaged %d; lengths seen: %s" % (len(frames), want, len(tail), dict(collections.Counter(len(f) for f in frames))))
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
