"""Second, independent restatement of the k-nearest crate in pure Python (small cases only).

CPU ORACLE / TEST INFRASTRUCTURE.  Written separately from oracle/knn_oracle.cpp, directly
from k-nearest/src/kd_tree.rs and best_set.rs, so that a transcription slip in either copy
shows up as a disagreement (the reference itself cannot run here: Rust, no toolchain).
Uses numpy.float32 scalars so every operation rounds like Rust's f32.
"""
import numpy as np

USIZE_MAX = 0xFFFFFFFFFFFFFFFF
f32 = np.float32


def distance(left, right):  # metric.rs:21-26
    s = f32(0.0)
    for d in range(3):
        v = f32(left[d]) - f32(right[d])
        s = f32(s + f32(v * v))
    return s


def distance_plane(position, plane, dim):  # metric.rs:28-31
    diff = f32(position[dim]) - f32(plane)
    return f32(diff * diff)


def partition(target, dim, pos, lo_off, length):  # kd_tree.rs:58-80 on pos[lo_off:lo_off+length]
    lower, upper = 0, length - 1
    while True:
        pivot_index = (upper - lower) // 2 + lower
        pivot = pos[lo_off + pivot_index][0][dim]
        pos[lo_off + pivot_index], pos[lo_off + upper] = pos[lo_off + upper], pos[lo_off + pivot_index]
        store = lower
        for i in range(lower, upper):
            if pos[lo_off + i][0][dim] < pivot:
                pos[lo_off + store], pos[lo_off + i] = pos[lo_off + i], pos[lo_off + store]
                store += 1
        pos[lo_off + store], pos[lo_off + upper] = pos[lo_off + upper], pos[lo_off + store]
        if target == store:
            break
        elif target < store:
            upper = store - 1
        else:
            lower = store + 1


def create_tree(dim, tree, off, length):  # kd_tree.rs:46-56
    middle = length // 2
    partition(middle, dim, tree, off, length)
    next_dim = (dim + 1) % 3
    if 0 < middle:
        create_tree(next_dim, tree, off, middle)
    if middle < length - 1:
        create_tree(next_dim, tree, off + middle + 1, length - middle - 1)


def kdtree_new(points):  # kd_tree.rs:35-42
    tree = [([f32(c) for c in p], i) for i, p in enumerate(points)]
    create_tree(0, tree, 0, len(tree))
    return tree


class FixedSet:  # best_set.rs:9-82
    def __init__(self, max_distance, k):
        self.values = [[f32(max_distance), USIZE_MAX] for _ in range(k)]

    def fix_down(self, index, max_size):
        v = self.values
        while True:
            swap_index, swap_value = index, v[index][0]
            cl = index * 2 + 1
            if cl < max_size and v[cl][0] > swap_value:
                swap_index, swap_value = cl, v[cl][0]
            cr = index * 2 + 2
            if cr < max_size and v[cr][0] > swap_value:
                swap_index = cr
            if swap_index == index:
                break
            v[index], v[swap_index] = v[swap_index], v[index]
            index = swap_index

    def result(self):
        size = len(self.values)
        for end in reversed(range(len(self.values))):
            if self.values[0][1] == USIZE_MAX:
                size = end
            self.values[0], self.values[end] = self.values[end], self.values[0]
            self.fix_down(0, end)
        return size

    def distance(self):
        return self.values[0][0]

    def insert(self, d, idx):
        self.values[0] = [d, idx]
        self.fix_down(0, len(self.values))


def search_nearest(tree, off, length, position, dim, best):  # kd_tree.rs:104-146
    if length < 32:
        for i in range(off, off + length):
            point, idx = tree[i]
            diff = distance(point, position)
            if diff < best.distance():
                best.insert(diff, idx)
        return
    middle = length // 2
    point, idx = tree[off + middle]
    diff = distance(point, position)
    if diff < best.distance():
        best.insert(diff, idx)
    next_dim = (dim + 1) % 3
    is_left = f32(position[dim]) < point[dim]
    if is_left:
        search_nearest(tree, off, middle, position, next_dim, best)
    else:
        search_nearest(tree, off + middle + 1, length - middle - 1, position, next_dim, best)
    if distance_plane(position, point[dim], dim) < best.distance():
        if is_left:
            search_nearest(tree, off + middle + 1, length - middle - 1, position, next_dim, best)
        else:
            search_nearest(tree, off, middle, position, next_dim, best)


def k_nearest(tree, position, k, max_distance):  # kd_tree.rs:92-96
    best = FixedSet(max_distance, k)
    search_nearest(tree, 0, len(tree), [f32(c) for c in position], 0, best)
    count = best.result()
    return best.values, count
