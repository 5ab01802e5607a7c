"""Per-kernel totals of arbitrary rocprofv3 --pmc counters: python tools/pmc_table.py <dir with *counter_collection.csv> [kernel-name filter ...]
One line per kernel (all its dispatches of the run summed; dispatch count beside it), one column per counter found."""
import csv, glob, os, sys
from collections import defaultdict, OrderedDict
root = sys.argv[1]
filt = sys.argv[2:]
agg = defaultdict(lambda: defaultdict(float)); disp = defaultdict(set); names = OrderedDict()
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if filt and not any(x in k for x in filt):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r.get("Dispatch_Id")); names[r["Counter_Name"]] = 1
cols = list(names)
print(f"{'kernel':52s} {'disp':>5s} " + " ".join(f"{c[:18]:>18s}" for c in cols))
for k in sorted(agg, key=lambda k: -max(agg[k].values())):
    print(f"{k[:52]:52s} {len(disp[k]):5d} " + " ".join(f"{agg[k].get(c, 0):18.4g}" for c in cols))
