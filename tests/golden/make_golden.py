"""Generates the committed golden fixtures under tests/golden/.

The reference (antonWetzel/treee @ 2024_10_08) has no tests, fixtures or golden vectors and
cannot be built here (Rust; no toolchain), so these vectors are outputs of the CPU oracle
(oracle/*.cpp, a line-by-line restatement) on small seeded inputs.  They pin the oracle
against silent drift and give the HIP path a fixed target; the hand-traced KATs that pin
the oracle itself live in tests/test_oracle.py.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as O  # noqa: E402
from treee_amd import synth  # noqa: E402

F32_MAX = float(np.finfo(np.float32).max)


def main():
    O.build()
    pts, seg = synth.forest_cloud(3000, plot=6.0)
    tree = O.KDTree(pts)
    ent, cnt = tree.k_nearest_batch(pts, 31, F32_MAX)
    ent_r, cnt_r = tree.k_nearest_batch(pts, 31, 0.0225)
    txyz, tidx = tree.dump()
    curve = O.curvature(pts, 31, F32_MAX)
    mn = pts.min(0).astype(np.float32)
    size = float((pts.max(0) - mn).max())
    keys = O.morton_keys(pts, mn, size)
    sk, perm = O.sort_keys(keys)
    nodes, leaf, lod = O.octree_build(sk, 64, 32)
    np.savez_compressed(
        os.path.join(HERE, "forest3k.npz"),
        points=pts, segment=seg, tree_order=tidx.astype(np.uint32),
        knn_dist=ent["distance"], knn_index=ent["index"].astype(np.uint32), knn_count=cnt,
        knn_r_dist=ent_r["distance"], knn_r_index=ent_r["index"].astype(np.int64).astype(np.int32), knn_r_count=cnt_r,
        curve=curve, corner=mn, size=np.float32(size), keys=keys, perm=perm,
        nodes=nodes.view(np.uint32).reshape(-1, 8), leaf=leaf, lod=lod,
        slice_keys=O.slice_keys(pts),
    )
    w, h = 96, 54
    view = np.eye(4, dtype=np.float32)
    view[2, 3] = -12.0  # inverse of translate(0,0,12), row-major here
    view = np.ascontiguousarray(view.T).reshape(16)  # column-major
    proj = O.perspective(w / h, 45.0 * 2 * np.pi / 360.0, 0.1, 10_000.0)
    depth, ids, index = O.splat(pts, seg, view, proj, 0.1, 0, 0xFFFFFFFF, w, h)
    np.savez_compressed(os.path.join(HERE, "splat3k.npz"), view=view, proj=proj, depth=depth, ids=ids, index=index,
                        eye_dome=O.eye_dome(depth, 0.3 ** 6))
    ints = np.array([[0, 0, 0], [1000, 2000, 3000], [-1, -2, -3], [2147483647, -2147483648, 123456789]], np.int32)
    img = synth.las_bytes(ints, scale=(0.001, 0.001, 0.001), offset=(100.0, 200.0, 50.0))
    hdr = O.las_header_parse(img)
    dec = O.decode_las_records(img[hdr.offset_to_point_data:], len(ints), 20, list(hdr.scale), list(hdr.offset), list(hdr.center))
    np.savez_compressed(os.path.join(HERE, "las4.npz"), image=np.frombuffer(img, np.uint8), ints=ints, decoded=dec,
                        center=np.array(list(hdr.center)))
    print("golden fixtures written")


if __name__ == "__main__":
    main()
