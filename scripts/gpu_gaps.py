"""Idle-gap analysis of a rocprofv3 --kernel-trace CSV: busy vs wall over the last K steps."""
import csv, sys
path, nlast = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# step boundaries: the fused optimizer kernel ends a step
ends = [e for s, e, n in rows if "sf_adamw_kernel" in n]
if len(ends) < nlast + 1:
    sys.exit("not enough steps")
t0, t1 = ends[-nlast - 1], ends[-1]
win = [(s, e, n) for s, e, n in rows if s >= t0 and e <= t1]
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, n in win:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        gaps.append((s - cur_e, prev_name, n))
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    prev_name = n
busy += cur_e - cur_s
wall = t1 - t0
print(f"steps {nlast}: wall {wall/1e6/nlast:.2f} ms/step, GPU busy {busy/1e6/nlast:.2f} ms/step ({100*busy/wall:.1f} %), "
      f"kernels/step {len(win)/nlast:.0f}")
gaps.sort(reverse=True)
tot = sum(g for g, _, _ in gaps)
print(f"idle total {tot/1e6/nlast:.2f} ms/step in {len(gaps)/nlast:.0f} gaps/step; gaps > 20 us: "
      f"{sum(g for g,_,_ in gaps if g > 20000)/1e6/nlast:.2f} ms/step")
for g, a, b in gaps[:25]:
    print(f"  {g/1e3:8.1f} us  after {a[:60]:60s} before {b[:60]}")
