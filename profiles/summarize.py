"""Reduce rocprofv3 output directories (see collect.sh) to the small summaries kept under profiles/.

    python profiles/summarize.py gpurun_out/<tag> <tag>

writes gpurun_out/<tag>/<tag>_kernel_stats.csv (library kernels only), <tag>_hbm_pmc.json (per-kernel FETCH_SIZE /
WRITE_SIZE totals of one step, raw counter units = KB) and <tag>_traffic.json (the dominant kernel, per launch).
"""
import csv, glob, hashlib, json, os, re, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def knn_sha16():
    """what bench.py compares a committed summary with: the kernel source it was measured with"""
    return hashlib.sha256(open(os.path.join(REPO, "treee_amd", "csrc", "knn.hip"), "rb").read()).hexdigest()[:16]


OURS = re.compile(r"\bk_[a-z]|\brs_[a-z]|\bos_(hist|pass)|treee::")


def short(name):
    m = re.search(r"(k_\w+|rs_\w+|os_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main(root, tag):
    stats = glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        keep = [r for r in rows if OURS.search(r["Name"])]
        with open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct_of_all_gpu_time"])
            for r in keep:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"],
                            r["MaxNs"], r["Percentage"]])
    pmc = {}
    for which in ("fetch", "write"):
        for path in glob.glob(os.path.join(root, which, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if not OURS.search(r["Kernel_Name"]):
                    continue
                k = short(r["Kernel_Name"])
                e = pmc.setdefault(k, {"calls": {}})
                key = r["Counter_Name"] + "_KB_total"
                e[key] = e.get(key, 0.0) + float(r["Counter_Value"])
                e["calls"][which] = e["calls"].get(which, set()) | {r["Dispatch_Id"]}
    for e in pmc.values():
        e["calls"] = max(len(v) for v in e["calls"].values())
    json.dump(pmc, open(os.path.join(root, f"{tag}_hbm_pmc.json"), "w"), indent=1, sort_keys=True)
    name, knn = max(((k, v) for k, v in pmc.items() if k.startswith("k_knn_")), key=lambda kv: kv[1].get("FETCH_SIZE_KB_total", 0.0), default=(None, None))
    if knn:
        n = knn["calls"]
        traffic = {
            "kernel": name,
            "knn_hip_sha16": knn_sha16(),
            "points_per_launch": 15000000,
            "launches_in_pass": n,
            "fetch_size_bytes_raw": knn.get("FETCH_SIZE_KB_total", 0.0) * 1024 / n,
            "fetch_size_bytes_corrected": 2 * knn.get("FETCH_SIZE_KB_total", 0.0) * 1024 / n,
            "write_size_bytes": knn.get("WRITE_SIZE_KB_total", 0.0) * 1024 / n,
            "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py --steps 1 "
                    "--warmup 0 (15 M points per launch); FETCH_SIZE on gfx950 counts 128-B requests as 64 B for "
                    "wide coalesced reads (MI355X_MICROARCH.md): corrected = 2 x raw. Calibration in this library's own "
                    "access pattern: k_morton_keys reads 15 M x 12 B = 180 MB and reports 90 MB raw",
        }
        json.dump(traffic, open(os.path.join(root, f"{tag}_traffic.json"), "w"), indent=1)
        print(json.dumps(traffic))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
