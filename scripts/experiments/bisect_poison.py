"""Which device buffer is read before it is written?  Runs a case once to list the size classes the cache hands out, then once per
class with only that class poisoned (0xFF; the others zero-filled) and reports the classes whose poisoning changes the result."""
import os, subprocess, sys, re, json
here = os.path.dirname(os.path.abspath(__file__))
case = sys.argv[1]
def run(env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.join(here, "dbg_case.py"), case], env=e, capture_output=True, text=True, timeout=300)
    return r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "NO OUTPUT " + r.stderr[-300:], r.stderr
ref, err = run({"MCP_DEV_CACHE_POISON": "1", "MCP_DEV_CACHE_POISON_CLASS": "9999", "MCP_DEV_CACHE_LOG": "1"})
classes = {}
for m in re.finditer(r"take class (\d+) \((\d+) bytes\)", err): classes[int(m.group(1))] = int(m.group(2))
print("reference (all zero-filled):", ref)
print("classes:", sorted(classes.items()))
takes = [(int(m.group(1)), int(m.group(2)), int(m.group(3))) for m in re.finditer(r"take #(\d+) class (\d+) \((\d+) bytes\)", err)]
for c in sorted(classes):
    out, _ = run({"MCP_DEV_CACHE_POISON": "1", "MCP_DEV_CACHE_POISON_CLASS": str(c)})
    if out == ref: continue
    print("class", c, classes[c], "bytes ->", out)
    for nth, cls, nb in takes:
        if cls != c: continue
        out2, _ = run({"MCP_DEV_CACHE_POISON": "1", "MCP_DEV_CACHE_POISON_NTH": str(nth)})
        print("   take #%d of %d takes:" % (nth, len(takes)), "CHANGES" if out2 != ref else "same", out2 if out2 != ref else "")
print("takes:", " ".join("%d:%d" % (n, nb) for n, _, nb in takes[:140]))
print("done")
