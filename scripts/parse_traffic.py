"""Per-kernel average FETCH_SIZE / WRITE_SIZE per launch from rocprofv3 --pmc counter_collection CSVs."""
import csv, glob, json, os, sys
root = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, "pmc_" + c, "*", "*counter_collection.csv"))
    if not files:
        continue
    acc = {}
    for r in csv.DictReader(open(files[0])):
        if r.get("Counter_Name") != c:
            continue
        k = r["Kernel_Name"].split("(")[0]
        a = acc.setdefault(k, [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in acc.items():
        out.setdefault(k, {})[c] = s / n
        out[k]["launches_" + c] = n
print(json.dumps(out, indent=1))
