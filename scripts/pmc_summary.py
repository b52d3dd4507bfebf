"""Aggregate a rocprofv3 counter_collection CSV per kernel: mean counter value per dispatch."""
import csv, sys, collections
path = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if filt and filt not in name:
            continue
        key = (name[:70], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key].add(r["Dispatch_Id"])
for key, ctrs in acc.items():
    n = len(cnt[key])
    print(f"{key[0]} grid {key[1]} wg {key[2]} dispatches {n}")
    for c, v in sorted(ctrs.items()):
        print(f"    {c:28s} {v / n:16.0f}")
