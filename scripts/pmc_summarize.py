"""Per-kernel FETCH_SIZE / WRITE_SIZE (KB per dispatch) from two rocprofv3 --pmc passes -> JSON.

FETCH_SIZE on gfx950 counts 64 B per 128-B read request of a wide coalesced stream, so the read side is
reported both raw and doubled; WRITE_SIZE is taken as is."""
import csv, glob, hashlib, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha256():
    """Hash of the kernel sources the counters belong to (bench.py only reports the traffic while it still matches)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "slam_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def load(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"].split("(")[0]
            a = acc[k]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    fa, wa = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fa) | set(wa)):
        if "slamk" not in k:
            continue
        f, fn = fa.get(k, [0.0, 0])
        w, wn = wa.get(k, [0.0, 0])
        res[k] = {"launches": max(fn, wn), "fetch_kb_per_launch_raw": f / max(fn, 1), "write_kb_per_launch": w / max(wn, 1),
                  "hbm_bytes_per_launch_raw": (f / max(fn, 1) + w / max(wn, 1)) * 1024,
                  "hbm_bytes_per_launch_fetch_doubled": (2 * f / max(fn, 1) + w / max(wn, 1)) * 1024}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 3 --warmup 1",
               "kernel_source_sha256": kernel_source_sha256(), "kernel_names": sorted(res), "kernels": res},
              open(out, "w"), indent=1)
    for k, v in res.items():
        print("%-60s n=%4d  fetch %10.0f KB  write %10.0f KB" % (k[:60], v["launches"], v["fetch_kb_per_launch_raw"], v["write_kb_per_launch"]))


main()
