"""Combine the counter_collection CSVs of a `--pmc FETCH_SIZE` pass and a `--pmc WRITE_SIZE` pass into the per-kernel HBM
traffic table kept under profiles/ (bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: MI355X_MICROARCH.md, HBM section).
usage: pmc_hbm_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<how>" """
import collections, csv, json, sys


def load(path, counter):
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            key = f'{r["Kernel_Name"].split("(")[0].replace("void ", "")} grid={r.get("Grid_Size", "")}'
            acc[key] += float(r["Counter_Value"])
            n[key].add(r["Dispatch_Id"])
    return {k: (acc[k] / len(n[k]), len(n[k])) for k in acc}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"how": sys.argv[4] if len(sys.argv) > 4 else "", "kernels": {}}
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, (0, 0))[0] + write.get(k, (0, 0))[0]) * max(fetch.get(k, (0, 1))[1], 1)):
    f, nf = fetch.get(k, (0.0, 0))
    w, _ = write.get(k, (0.0, 0))
    out["kernels"][k] = {"FETCH_SIZE_KiB_avg": round(f), "WRITE_SIZE_KiB_avg": round(w), "dispatches": nf,
                         "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=0)
print(f"{len(out['kernels'])} kernels -> {sys.argv[3]}")
