"""Critical-path attribution from a rocprofv3 --kernel-trace CSV.

For the last K steps: for every kernel name, the wall time during which it was the ONLY
kernel on the GPU ("alone") and the time it shared the GPU with another stream's kernel
("shared").  A latency-bound launch that occupies a handful of CUs shows up as a large
"alone" figure although its device-time share is small.
usage: gpu_timeline.py <kernel_trace.csv> [steps]
"""
import csv, sys, collections

path, nlast = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [e for s, e, n in rows if "sf_adamw_kernel" in n]
if len(ends) < nlast + 1:
    sys.exit("not enough steps")
t0, t1 = ends[-nlast - 1], ends[-1]
win = [(s, e, n) for s, e, n in rows if s >= t0 and e <= t1]
ev = []
for i, (s, e, n) in enumerate(win):
    ev.append((s, 1, i))
    ev.append((e, 0, i))
ev.sort()
alone = collections.Counter()
shared = collections.Counter()
calls = collections.Counter(n for _, _, n in win)
active = set()
prev = None
for t, kind, i in ev:
    if prev is not None and active and t > prev:
        if len(active) == 1:
            alone[win[next(iter(active))][2]] += t - prev
        else:
            for j in active:
                shared[win[j][2]] += (t - prev) / len(active)
    prev = t
    if kind:
        active.add(i)
    else:
        active.discard(i)
wall = (t1 - t0) / nlast / 1e6
print(f"wall {wall:.2f} ms/step over {nlast} steps")
print(f"{'kernel':80s} {'calls':>6s} {'alone ms':>9s} {'shared ms':>9s}")
names = sorted(set(alone) | set(shared), key=lambda n: -(alone[n] + shared[n]))
for n in names[:45]:
    print(f"{n[:80]:80s} {calls[n]/nlast:6.0f} {alone[n]/nlast/1e6:9.3f} {shared[n]/nlast/1e6:9.3f}")
print(f"{'total':80s} {len(win)/nlast:6.0f} {sum(alone.values())/nlast/1e6:9.3f} {sum(shared.values())/nlast/1e6:9.3f}")
