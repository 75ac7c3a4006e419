"""Critical-path view of a rocprofv3 kernel trace of bench.py: per stream busy time, idle gaps of the main stream by neighbour
kernels, and the per-iteration timeline of the last iteration."""
import collections, csv, sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("mcp::", "").replace("void ", ""),
                 r.get("Stream_Id", r.get("Queue_Id", "0"))) for r in rows)
    # a Compute call ends with k_export_state: bench.py's calls are prewarm (40 iterations), warm-up, the timed one, ...
    ends = [i for i, e in enumerate(ev) if e[2] == "k_export_state"]
    call = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    if len(ends) <= call:
        print("too few Compute calls in trace"); return
    seg = ev[ends[call - 1] + 1:ends[call] + 1]
    seg = [e for e in seg if not e[2].startswith("__amd_rocclr_fill") and e[2] not in ("k_upload_set", "k_fan_state")]
    lin = [i for i, e in enumerate(seg) if e[2].startswith("k_linearize")]
    seg = seg[max(0, lin[0] - 8):]
    print("Compute call %d of the trace: %d linearisations" % (call, len(lin)))
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    print("device span of the call: %.3f ms" % ((t1 - t0)/1e6))
    streams = collections.defaultdict(list)
    for s, e, n, q in seg:
        streams[q].append((s, e, n))
    main_q = max(streams, key=lambda q: sum(1 for x in streams[q] if x[2].startswith("k_linearize")))
    for q, L in streams.items():
        busy = sum(e - s for s, e, _ in L)
        print("stream %s%s: %d kernels, busy %.3f ms" % (q, " (main)" if q == main_q else "", len(L), busy/1e6))
    L = streams[main_q]
    per = collections.defaultdict(lambda: [0, 0])
    gaps = collections.defaultdict(lambda: [0, 0])
    prev = None
    for s, e, n in L:
        per[n][0] += 1; per[n][1] += e - s
        if prev is not None and s > prev[1]:
            g = gaps[(prev[2], n)]; g[0] += 1; g[1] += s - prev[1]
        prev = (s, e, n)
    print("-- main stream kernels (20 iterations)")
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  %-26s n=%4d total %8.1f us  avg %7.2f us" % (n, c, t/1e3, t/c/1e3))
    print("-- main stream gaps")
    tot = 0
    for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:22]:
        print("  %-22s -> %-22s n=%4d total %8.1f us avg %6.2f us" % (k[0], k[1], c, t/1e3, t/c/1e3))
    print("  all gaps: %.1f us" % (sum(t for _, t in gaps.values())/1e3))
    # last iteration timeline
    j0 = max(i for i, x in enumerate(L) if x[2].startswith("k_linearize"))
    base = L[j0][0]
    print("-- last iteration, main stream (start us, duration us)")
    last = None; run = 0
    for s, e, n in L[j0:]:
        if n == last:
            run += 1; continue
        if last is not None and run:
            print("      ... x%d more" % run)
        print("  %8.1f %7.2f  %s" % ((s - base)/1e3, (e - s)/1e3, n)); last = n; run = 0


main(sys.argv[1])
