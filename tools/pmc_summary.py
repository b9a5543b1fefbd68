"""Summarises the counter CSVs written by tools/pmc.sh: per-launch mean of every
counter for the k_primary / wavefront kernels."""
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            if "rocclr" in k or "prebake" in k:
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = []
for k, cs in acc.items():
    lines.append(k)
    for c in sorted(cs):
        v = cs[c]
        lines.append(f"  {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "summary.txt"), "w").write(txt + "\n")
