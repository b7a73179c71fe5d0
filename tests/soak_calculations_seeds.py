"""Soak of the Calculations phase against the CPU restatement (not collected by pytest: run on a GPU box):
    python tests/soak_calculations_seeds.py [first_seed] [count]
runs the seeded generators of test_long_slices_fuzz_vs_oracle, test_segments_calculate_fuzz_vs_oracle and
test_long_slices_with_points_at_infinity for `count` further seeds.  (Seed 531 of the first found the hull-with-a-NaN-normal case.)"""
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
    P.test_long_slices_fuzz_vs_oracle(T, O, seed)
    P.test_segments_calculate_fuzz_vs_oracle(T, O, seed)
    P.test_long_slices_with_points_at_infinity(T, O, seed)
    if (seed - first) % 10 == 9:
        print(f"seed {seed}: ok ({time.time() - t0:.0f} s)", flush=True)
print(f"{count} seeds of the three generators: all bit-identical to the CPU restatement")
