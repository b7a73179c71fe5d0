"""Soak of round 4's structure and splat paths (not collected by pytest: run on a GPU box):
    python tests/soak_structure.py [first_seed] [count]
further seeds of test_block_sorted_hilbert_order_on_ragged_clouds (the block-sorted Hilbert order against the point sort, neighbour
list for neighbour list) and test_splat_from_the_octrees_sorted_records_is_the_same_frame (frames from the caller's cloud, the
octree's records and the kNN structure's records, against each other and the software rasteriser)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import treee_amd as T  # noqa: E402
from oracle import oracle as O  # noqa: E402

import test_gpu_parity as P  # noqa: E402

O.build()
O.lib()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
for seed in range(first, first + count):
    P.test_block_sorted_hilbert_order_on_ragged_clouds(T, seed)
    P.test_splat_from_the_octrees_sorted_records_is_the_same_frame(T, O, seed)
    if (seed - first) % 10 == 9:
        print(f"seed {seed}: ok ({time.time() - t0:.0f} s)", flush=True)
print(f"{count} seeds of both generators: identical")
