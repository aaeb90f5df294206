"""Independent pure-Python restatement of the reference's host-side builders, used
to cross-check the C++ host library on small inputs.

TEST INFRASTRUCTURE ONLY (imported by tests/ alone).  Follows src/mesh.rs:308-355
(Mesh::from_model), src/bvh.rs:52-178 (compute_node_bounds / subdivide /
generate_skips), src/blas.rs:18-41 (triangle bounds, the four-term centroid, the
face swap) and src/tlas.rs:27-99.  All arithmetic is done on numpy float32
scalars so that every operation rounds to f32 like glam's.  PARITY UNPINNED: the
reference has no fixtures for these builders.
"""
import sys

import numpy as np

f32 = np.float32


def from_model(positions, indices, normals=None):
    """Mesh::from_model: centre, per-axis normalise by the max of the centred coords."""
    pos = [np.array([p[0], p[1], p[2], 0.0], dtype=f32) for p in np.asarray(positions, dtype=f32).reshape(-1, 3)]
    n = len(pos)
    s = np.zeros(4, f32)
    for p in pos:
        s = s + p
    center = s / f32(n)
    pos = [p - center for p in pos]
    extent = pos[0].copy()
    for p in pos[1:]:
        extent = np.maximum(p, extent)
    out = []
    for p in pos:
        q = p[:3] / extent[:3]
        out.append(np.array([q[0], q[1], q[2], 1.0], dtype=f32))
    faces = [list(map(int, t)) + [0] for t in np.asarray(indices).reshape(-1, 3)]
    return out, faces


class Node:
    __slots__ = ("lb", "ub", "left", "right", "skip", "is_leaf", "start", "end")

    def __init__(self, is_leaf=True, start=0, end=0):
        self.lb = np.zeros(3, f32)
        self.ub = np.zeros(3, f32)
        self.left = self.right = self.skip = 0
        self.is_leaf = is_leaf
        self.start, self.end = start, end


class Builder:
    """The BVH trait over abstract elements."""

    def __init__(self, n_elems, elem_bounds, elem_centroid, elem_swap):
        self.nodes = [Node(True, 0, n_elems)]
        self.elem_bounds, self.elem_centroid, self.elem_swap = elem_bounds, elem_centroid, elem_swap

    def compute_node_bounds(self, idx):
        node = self.nodes[idx]
        if not node.is_leaf:
            l, r = self.nodes[node.left], self.nodes[node.right]
            node.lb, node.ub = np.minimum(l.lb, r.lb), np.maximum(l.ub, r.ub)
        else:
            lb, ub = self.elem_bounds(node.start)
            for i in range(node.start + 1, node.end):
                l2, u2 = self.elem_bounds(i)
                lb, ub = np.minimum(lb, l2), np.maximum(ub, u2)
            node.lb, node.ub = lb, ub

    def subdivide(self, idx, threshold):
        node = self.nodes[idx]
        if not node.is_leaf:
            self.subdivide(node.left, threshold)
            self.subdivide(node.right, threshold)
            return
        if node.end - node.start <= threshold:
            return
        extent = node.ub - node.lb
        axis = 0
        if extent[1] > extent[0]:
            axis = 1
        if extent[2] > extent[axis]:
            axis = 2
        i, j = node.start, node.end - 1
        cmin = self.elem_centroid(node.start)
        cmax = cmin
        for k in range(node.start + 1, node.end):
            c = self.elem_centroid(k)
            cmin, cmax = np.minimum(cmin, c), np.maximum(cmax, c)
        split = (cmin[axis] + cmax[axis]) * f32(0.5)
        while i < j:
            if self.elem_centroid(i)[axis] < split:
                i += 1
            else:
                self.elem_swap(i, j)
                j -= 1
        if i == node.end or i == node.start:
            return
        l = len(self.nodes)
        self.nodes.append(Node(True, node.start, i))
        r = len(self.nodes)
        self.nodes.append(Node(True, i, node.end))
        self.compute_node_bounds(l)
        self.compute_node_bounds(r)
        self.subdivide(l, threshold)
        self.subdivide(r, threshold)
        node.is_leaf = False
        node.left, node.right = l, r
        self.compute_node_bounds(idx)

    def generate_skips(self, idx, nxt):
        node = self.nodes[idx]
        node.skip = nxt
        if not node.is_leaf:
            self.generate_skips(node.left, node.right)
            self.generate_skips(node.right, nxt)

    def initialize(self, threshold):
        sys.setrecursionlimit(max(sys.getrecursionlimit(), 100000))
        self.compute_node_bounds(0)
        self.subdivide(0, threshold)
        self.generate_skips(0, 0)


def build_blas(positions, faces):
    """BLAS::new: returns (nodes, reordered faces)."""
    faces = [list(f) for f in faces]

    def bounds(f):
        p = [positions[i] for i in faces[f][:3]]
        lb = np.minimum(np.minimum(p[0], p[1]), p[2])[:3]
        ub = np.maximum(np.maximum(p[0], p[1]), p[2])[:3]
        return lb, ub

    def centroid(f):
        p = [positions[i] for i in faces[f]]  # four entries, the fourth is positions[0]
        acc = p[0]
        for q in p[1:]:
            acc = acc + q
        return (acc / f32(3.0))[:3]

    def swap(a, b):
        faces[a], faces[b] = faces[b], faces[a]

    b = Builder(len(faces), bounds, centroid, swap)
    b.initialize(4)
    return b.nodes, faces


def nodes_as_records(nodes):
    """(lb3, ub3, left, skip, is_leaf, start, end) tuples, the fields of BVHNodeGPU."""
    return [(tuple(float(x) for x in n.lb), tuple(float(x) for x in n.ub), int(n.left), int(n.skip),
             int(n.is_leaf), int(n.start), int(n.end)) for n in nodes]
