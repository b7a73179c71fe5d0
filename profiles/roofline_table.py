"""Per-kernel roofline table of one 15 M-point step from a kernel-stats summary (collect.sh):

    python profiles/roofline_table.py profiles/r02_v2_kernel_stats.csv

Algorithmic bytes per point are SURVEY.md 8 D / BASELINE.md 3 (a kernel's share of its stage); `frac` is against the
8 TB/s HBM3E spec.  Kernels whose bound is not HBM say what it is (measured: profiles/README.md)."""
import csv
import sys

N = 15_000_000
W, H = 1920, 1080
# kernel -> (launches per step that touch all N points, algorithmic bytes per point per launch, what bounds it)
MODEL = {
    "k_bbox_partial": (1, 12, "hbm (streaming read)"),
    "k_morton_keys_cube": (1, 20, "hbm (12 B read, 4 B key written; the index is implicit: 16 B moved)"),
    "rs_hist": (4, 4, "hbm read + LDS atomics"),
    "rs_scatter": (4, 16, "hbm (8 B read + 8 B written per pair) + LDS ranking"),
    "k_gather_sorted": (1, 28, "hbm random 12-B gather: one 64-B sector per point"),
    "k_leaf_aabb": (1, 16, "hbm (streaming read of the 16-B records)"),
    "k_knn_sorted32": (1, 340, "occupancy (3 waves/SIMD): LDS instruction rate in lane-parallel scans, VALU issue elsewhere"),
    "k_map_to_u32": (1, 8, "hbm"),
    "k_splat_points": (1, 16, "L2 / fabric request rate (one random 8-B pixel word per covered pixel)"),
    "k_oct_finish": (1, 4, "hbm (leaf id per point)"),
}


def main(path):
    rows = list(csv.DictReader(open(path)))
    print("| kernel | calls/step | avg us | bytes/pt | achieved GB/s | frac of 8 TB/s | bound |")
    print("|---|---|---|---|---|---|---|")
    steps = None
    for r in rows:
        name = r["kernel"].split("<")[0]
        if name == "k_knn_sorted32":
            steps = int(r["calls"])
    for r in rows:
        name = r["kernel"].split("<")[0]
        if name not in MODEL:
            continue
        per_step, bpp, bound = MODEL[name]
        calls = int(r["calls"])
        # the N-point launches are the long ones: average over the longest per_step * steps calls is not in the summary, so
        # use max_ns for multi-size kernels (rs_* also sort the 234 k-entry dispatch table) and avg for the others
        us = float(r["max_ns"] if name.startswith("rs_") else r["avg_ns"]) / 1e3
        gbs = bpp * N / (us * 1e-6) / 1e9
        print(f"| `{name}` | {calls // steps} | {us:.1f} | {bpp} | {gbs:.0f} | {gbs / 8000:.3f} | {bound} |")


if __name__ == "__main__":
    main(sys.argv[1])
