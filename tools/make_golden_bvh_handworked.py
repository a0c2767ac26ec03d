"""Writes tests/golden/bvh_handworked.npz: the GPU buffers of a six-triangle mesh, worked out BY HAND from the text of the
reference's builder (src/bvh.cu:100-207, 234-294, src/bvh.cuh:39-49, src/aabb.cpp, include/linear_math.cuh:85-121).  No
builder, ours or the oracle's, is called here: every number below is a literal from the derivation.  The tests assert that
both builders reproduce these bytes (tests/test_cpu.py::test_bvh_builders_reproduce_the_handworked_tree).

THE MESH.  Triangle T_k = (v0, v1, v2) = ((a+1, 0, 1), (a, 1, 1), (a, 0, 1)) with a = 9, 0, 2, 12, 1, 10 for k = 0..5: unit
right triangles in the plane z = 1, legs along +x and +y, spread along x, listed out of order on purpose.

BOXES (bvh.cuh:39-49).  min = (a, 0, 1), max = (a+1, 1, 1); the z extent is zero, so max.z += EPSILON: fl(1 + 0.001f) =
0x3F8020C5 = 1.00100004673004150390625, called e below.  Centroid x = a + 0.5 (exact).

ROOT, build(0, 6) (bvh.cu:100-207).  Centroid bounds x: [0.5, 12.5], y and z extents 0 -> axis 0 (aabb.cpp:32-42).  Bucket
(bvh.cu:87-90) = 12 * (c - 0.5) / 12 = a, and 12 -> 11 (bvh.cu:143): T1 -> 0, T4 -> 1, T2 -> 2, T0 -> 9, T5 -> 10, T3 -> 11.
With SA(dx) = 2 (dx * 1 + dx * (e - 1) + 1 * (e - 1)) ~ 2.002 dx + 0.002 the costs (n0 SA0 + n1 SA1) / SA + 1 (bvh.cu:162) are,
up to the common divisor:   i = 0: 1 SA(1) + 5 SA(12) ~ 122     i = 1: 2 SA(2) + 4 SA(11) ~ 96
    i = 2..8: 3 SA(3) + 3 SA(4) ~ 42 (the same boxes and counts, hence the same float, for all seven)
    i = 9: 4 SA(10) + 2 SA(3) ~ 92     i = 10: 5 SA(11) + 1 SA(1) ~ 112
The first minimum wins (strict <, bvh.cu:166-173): min_idx = 2.  std::partition (libstdc++, bidirectional form: advance
`first` over elements that satisfy the predicate, retreat `last` over those that do not, swap, repeat) with "bucket <= 2" on
[T0 n, T1 y, T2 y, T3 n, T4 y, T5 n]: first stops at T0, last retreats over T5 to T4, swap -> [T4, T1, T2, T3, T0, T5]; first
advances over T1, T2 and stops at T3, last retreats to T3: they meet, mid = 3.

LEFT, build(0, 3) = host node 1 over [T4, T1, T2].  Centroid bounds x: [0.5, 2.5], extent 2.  Buckets 12 * (c - 0.5) / 2:
T4 -> 6, T1 -> 0, T2 -> 12 -> 11.  Costs: i = 0..5: 1 SA(1) + 2 SA(2); i = 6..10: 2 SA(2) + 1 SA(1) -- the same two products
added in the other order, the same float: a tie over all eleven, so min_idx = 0.  Partition "bucket <= 0" on [T4 n, T1 y, T2 n]:
first stops at T4, last retreats over T2 to T1, swap -> [T1, T4, T2], first = 1 = last: mid = 1.
  host node 2 = build(0, 1): one triangle -> leaf {T1}, index 0.
  host node 3 = build(1, 3): two triangles [T4, T2]: already in order along x (1.5 < 2.5, bvh.cu:127) -> leaf {T4, T2}, index 1.
  host node 1: right child = host node 3, box = [0, 3] x [0, 1] x [1, e].

RIGHT, build(3, 6) = host node 4 over [T3, T0, T5].  Centroid bounds x: [9.5, 12.5], extent 3.  Buckets 12 * (c - 9.5) / 3:
T3 -> 12 -> 11, T0 -> 0, T5 -> 12 * 1 / 3 = 4.  Costs: i = 0..3: 1 SA(1) + 2 SA(3) ~ 14; i = 4..10: 2 SA(2) + 1 SA(1) ~ 10: first
minimum min_idx = 4.  Partition "bucket <= 4" on [T3 n, T0 y, T5 y]: first stops at T3, last = T5 satisfies, swap -> [T5, T0, T3];
first advances over T0 and meets last at position 2: mid = 3 + 2 = 5.
  host node 5 = build(3, 5): two triangles [T5, T0]: 10.5 < 9.5 is false -> swapped (bvh.cu:127-129) -> leaf {T0, T5}, index 3.
  host node 6 = build(5, 6): leaf {T3}, index 5.
  host node 4: right child = host node 6, box = [9, 13] x [0, 1] x [1, e].
Triangle array after the build: [T1, T4, T2, T0, T5, T3].

FLATTEN (bvh.cu:234-294).  Triangle data starts with one zero float4 (index 0).  Stack = [root].
  pop host 0: both children inner.  GPU node 0: left box = host 1's, right box = host 4's.  Push left (host 1), then right
      (host 4): the right one is popped first.
  pop host 4 (right child of GPU node 0) -> GPU node 1, so node 0's child.y = 1.  Left = leaf {T0, T5}: child.x = ~1 = -2, T0 at
      float4 1..3, T5 at 4..6, terminator (four words 0x80000000) at 7.  Right = leaf {T3}: child.y = ~8 = -9, T3 at 8..10,
      terminator at 11.  Boxes: left [9, 11], right [12, 13] in x.
  pop host 1 (left child of GPU node 0) -> GPU node 2, node 0's child.x = 2.  Left = leaf {T1}: child.x = ~12 = -13, T1 at 12..14,
      terminator at 15.  Right = leaf {T4, T2}: child.y = ~16 = -17, T4 at 16..18, T2 at 19..21, terminator at 22.  Boxes: left
      [0, 1], right [1, 3] in x.
23 float4 in all.  Node = (left min.x, max.x, min.y, max.y), (right ...), (left min.z, max.z, right min.z, max.z), (left, right, 0, 0).

WOOP ROWS (bvh.cu:214-227).  Columns of the matrix: v0 - v2 = (1, 0, 0), v1 - v2 = (0, 1, 0), their cross product (0, 0, 1), v2 =
(a, 0, 1); last row (0, 0, 0, 1).  Its inverse has rows (1, 0, 0, -a), (0, 1, 0, 0), (0, 0, 1, -1).  In the cofactor sums of
linear_math.cuh:93-108 every product is exact and the determinant is 1; an entry that is zero is a sum of six signed zeros, and in
each of those sums at least one term enters as +0, so (round to nearest: -0 + +0 = +0) every zero is +0 -- also -a for a = 0 (T1):
its sum runs ... - 0 * 1 * 1 + 0 * 0 * 0 from a partial sum that is already +0.  Emitted: (m20, m21, m22, -m23) = (0, 0, 1, 1), then
row 0 = (1, 0, 0, -a), row 1 = (0, 1, 0, 0).

PRIMITIVE IDS (this build's addition, not the reference's): the face index of a triangle at the float4 index of its first row.
"""
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

A = [9.0, 0.0, 2.0, 12.0, 1.0, 10.0]
E = struct.unpack("<f", struct.pack("<I", 0x3F8020C5))[0]


def main():
    verts = np.array([[[a + 1, 0, 1], [a, 1, 1], [a, 0, 1]] for a in A], dtype=np.float32)
    nodes = np.array([
        # left x, y          right x, y          left z, right z   children
        [0, 3, 0, 1,         9, 13, 0, 1,        1, E, 1, E,       0, 0, 0, 0],
        [9, 11, 0, 1,        12, 13, 0, 1,       1, E, 1, E,       0, 0, 0, 0],
        [0, 1, 0, 1,         1, 3, 0, 1,         1, E, 1, E,       0, 0, 0, 0],
    ], dtype=np.float32)
    # the child words are integers: write their bits, not their float values
    ch = nodes.view(np.int32)
    ch[0, 12:14] = [2, 1]
    ch[1, 12:14] = [-2, -9]
    ch[2, 12:14] = [-13, -17]

    def woop(a):
        return [[0, 0, 1, 1], [1, 0, 0, -a if a else 0.0], [0, 1, 0, 0]]

    end = [[0, 0, 0, 0]]                          # the terminator's bits are written below
    order = [[0, 5], [3], [1], [4, 2]]            # leaf runs in emission order: {T0, T5}, {T3}, {T1}, {T4, T2}
    rows = [[0, 0, 0, 0]]
    prim = {}
    for run in order:
        for k in run:
            prim[len(rows)] = k
            rows += woop(A[k])
        rows += end
    tris = np.array(rows, dtype=np.float32)
    tv = tris.view(np.uint32)
    for i in (7, 11, 15, 22):
        tv[i] = 0x80000000
    assert tris.shape == (23, 4) and sorted(prim) == [1, 4, 8, 12, 16, 19]
    assert np.float32(1) + np.float32(0.001) == np.float32(E)
    path = os.path.join(ROOT, "tests", "golden", "bvh_handworked.npz")
    np.savez(path, verts=verts, nodes=nodes, tris=tris, prim_index=np.array(sorted(prim), dtype=np.uint32),
             prim_id=np.array([prim[i] for i in sorted(prim)], dtype=np.uint32))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
