"""Mean per-dispatch value of every counter rocprofv3 --pmc collected, per kernel whose name contains argv[1], grouped by grid size
(the sweep launches the same kernel at several batch sizes): python3 tools/pmc_reduce.py <substr> <dir> [<dir> ...]"""
import csv, glob, json, sys
from collections import defaultdict
sub, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                key = r["Kernel_Name"].split("(")[0] + " grid=" + r.get("Grid_Size", "?")
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: (sum(v) / len(v)) for c, v in sorted(cs.items())} | {"dispatches": max(len(v) for v in cs.values())} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
