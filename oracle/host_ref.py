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


# ---------------------------------------------------------------------- TLAS
# src/tlas.rs:61-158 with glam 0.30's Mat4 semantics written out: column-major
# 4x4 of f32; from_rotation_x/y/z put (cos, sin) / (-sin, cos) in the two
# rotated columns; a * b multiplies a by every column of b; a * v is
# x*col0 + y*col1 + z*col2 + w*col3, each product rounded, summed left to right.
def _mat_identity():
    return [np.array([1, 0, 0, 0], f32), np.array([0, 1, 0, 0], f32), np.array([0, 0, 1, 0], f32),
            np.array([0, 0, 0, 1], f32)]


def _mul_vec4(m, v):
    acc = m[0] * v[0]
    acc = acc + m[1] * v[1]
    acc = acc + m[2] * v[2]
    acc = acc + m[3] * v[3]
    return acc


def _mul_mat4(a, b):
    return [_mul_vec4(a, col) for col in b]


def instance_matrix(scale, rotation, translation, sin, cos):
    """translate * (rot_x * (rot_y * rot_z)) * scale (src/tlas.rs:80-92); sin / cos are f32 functions."""
    t = _mat_identity()
    t[3] = np.array([translation[0], translation[1], translation[2], 1], f32)
    sx, cx, sy, cy, sz, cz = sin(rotation[0]), cos(rotation[0]), sin(rotation[1]), cos(rotation[1]), sin(rotation[2]), cos(rotation[2])
    rx, ry, rz = _mat_identity(), _mat_identity(), _mat_identity()
    rx[1] = np.array([0, cx, sx, 0], f32)
    rx[2] = np.array([0, -sx, cx, 0], f32)
    ry[0] = np.array([cy, 0, -sy, 0], f32)
    ry[2] = np.array([sy, 0, cy, 0], f32)
    rz[0] = np.array([cz, sz, 0, 0], f32)
    rz[1] = np.array([-sz, cz, 0, 0], f32)
    rotate = _mul_mat4(rx, _mul_mat4(ry, rz))
    s = _mat_identity()
    s[0][0], s[1][1], s[2][2] = f32(scale[0]), f32(scale[1]), f32(scale[2])
    return _mul_mat4(t, _mul_mat4(rotate, s))


def build_tlas(geom_aabbs, transforms, instances, sin, cos):
    """TLAS::new: geom_aabbs[g] = (lb3, ub3); transforms[t] = (scale3, rotation3, translation3);
    instances[i] = (transform, geometry).  Returns (nodes, instance_ids)."""
    boxes = []
    for t_idx, g_idx in instances:
        lb, ub = geom_aabbs[g_idx]
        m = instance_matrix(*transforms[t_idx], sin, cos)
        acc = None
        for x in (0, 1):           # multi_cartesian_product: the last factor varies fastest
            for y in (0, 1):
                for z in (0, 1):
                    corner = np.array([lb[0] if x == 0 else ub[0], lb[1] if y == 0 else ub[1],
                                       lb[2] if z == 0 else ub[2], 1.0], f32)
                    c = _mul_vec4(m, corner)[:3]
                    acc = (c, c) if acc is None else (np.minimum(acc[0], c), np.maximum(acc[1], c))
        boxes.append(acc)
    ids = list(range(len(instances)))

    def swap(a, b):
        boxes[a], boxes[b] = boxes[b], boxes[a]
        ids[a], ids[b] = ids[b], ids[a]

    b = Builder(len(instances), lambda e: boxes[e], lambda e: (boxes[e][0] + boxes[e][1]) / f32(2.0), swap)
    b.initialize(1)
    return b.nodes, ids
