"""Summarises rocprofv3 --pmc CSVs (one directory per pass) into per-kernel means for profiles/."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void "):
            k = k[5:]            # templated kernels are printed with their return type
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc summary (mean per dispatch) of", root)
for k in sorted(acc):
    if not k.startswith("nep::"):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-24s mean %16.1f  n %4d" % (c, sum(v) / len(v), len(v)))
