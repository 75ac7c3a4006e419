"""Which entries of the reduced-system buffer does the per-step path read before they are written?  Binary search with MCP_BA_DEBUG_POISON_RED."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
case = sys.argv[1]; n = int(sys.argv[2]); stride = n*n + 2*n
def run(env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.join(here, "dbg_case.py"), case], env=e, capture_output=True, text=True, timeout=300)
    return r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "NO OUTPUT " + r.stderr[-300:]
ref = run({"MCP_DEV_CACHE_POISON": "1", "MCP_DEV_CACHE_POISON_CLASS": "9999"})
print("reference", ref)
for q in range(4):
    bad = run({"MCP_BA_DEBUG_POISON_RED": "%d,%d,%d" % (q, 0, stride)}) != ref
    print("system", q, "whole buffer poisoned:", "CHANGES" if bad else "same")
    if not bad: continue
    lo, hi = 0, stride
    while hi - lo > 1:                     # smallest prefix [0, hi) that changes the result -> its last element is read
        mid = (lo + hi)//2
        if run({"MCP_BA_DEBUG_POISON_RED": "%d,%d,%d" % (q, 0, mid)}) != ref: hi = mid
        else: lo = mid
    e = hi - 1
    print("   first sensitive element: %d = row %d col %d (n = %d; row n = rhs, row n+1 = J^T r)" % (e, e//n, e % n, n))
    lo2, hi2 = 0, stride
    while hi2 - lo2 > 1:                   # largest suffix start
        mid = (lo2 + hi2)//2
        if run({"MCP_BA_DEBUG_POISON_RED": "%d,%d,%d" % (q, mid, stride)}) != ref: lo2 = mid
        else: hi2 = mid
    print("   last sensitive element: %d = row %d col %d" % (lo2, lo2//n, lo2 % n))
    break
