"""Evidence that the out-of-core tile stream overlaps PCIe copies with kernels: run under
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python profiles/stream_overlap.py run
then
    python profiles/stream_overlap.py report <dir>
prints how much of the host-to-device copy time lies under running kernels."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch

    import treee_amd as T
    from treee_amd import stream, synth

    dev = torch.device("cuda", 0)
    n, W, H = 15_000_000, 1920, 1080
    base = []
    for i in range(2):
        p, _ = synth.forest_cloud_torch(n, dev, seed=synth.SEED_FOREST + 200 + i)
        base.append(stream.las_tile_from_cloud(p))
        del p
    proj = T.Camera3D(W / H).projection()
    view = T.view_from_transform(T.rotation_x(-0.6) @ T.translation(0.0, 0.0, 700.0))
    env = T.PointCloudEnvironment(0, 0xFFFFFFFF, 0.1)
    ts = stream.TileStream(dev, n, width=W, height=H, knn_k=31)
    tiles = [stream.Tile(base[i % 2].records, n, 20, (0.001,) * 3, ((i % 4) * 200.0, (i // 4) * 200.0, 0.0), (300.0, 100.0, 0.0)) for i in range(8)]
    ts.run(tiles, view, proj, env)
    ts.close()


def report(root):
    kern = []
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    copies = []
    for path in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "HOST_TO_DEVICE" in r.get("Direction", "") or "H2D" in r.get("Direction", ""):
                copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    kern.sort()
    merged = []
    for s, e in kern:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    big = [c for c in copies if c[1] - c[0] > 1_000_000]  # the 300 MB tile copies (> 1 ms)
    total = sum(e - s for s, e in big)
    under = 0
    for s, e in big:
        for ks, ke in merged:
            lo, hi = max(s, ks), min(e, ke)
            if hi > lo:
                under += hi - lo
    out = {"tile_copies": len(big), "copy_ms_total": round(total / 1e6, 3), "copy_ms_under_kernels": round(under / 1e6, 3),
           "fraction_overlapped": round(under / total, 3) if total else None,
           "mean_copy_GBps": round(sum(1 for _ in big) * 15_000_000 * 20 / (total / 1e9) / 1e9, 1) if total else None}
    print(json.dumps(out))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
