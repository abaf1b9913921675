"""TEST INFRASTRUCTURE (oracle): the reference's solid cuts (dataset.py:716-763) restated on the CPU.

The reference asks a third-party library for the work: `open3d==0.15.2` (README.md:25 of the reference; not in this
image, no network).  `sphere_split` / `cylinder_split` / `cone_split` build a triangle mesh with
`o3d.geometry.TriangleMesh.create_sphere(0.5, 50)` / `create_cylinder(0.6, 1, 50)` / `create_cone(1, 2, 50)`, move it
(translate / rotate by numpy draws), and keep the points whose `RaycastingScene.compute_signed_distance` is negative,
i.e. the points strictly INSIDE the closed mesh.

Restated here from Open3D's published mesh factory (cpp/open3d/geometry/TriangleMeshFactory.cpp, v0.15):

  create_sphere(r, res):   vertex 0 = (0,0,r), vertex 1 = (0,0,-r); step = pi / res; for i = 1 .. res-1 (alpha = step i),
                           j = 0 .. 2 res - 1 (theta = step j): r (sin a cos t, sin a sin t, cos a).  Triangles: fans to the
                           two poles, two triangles per (ring i, ring i+1, sector j) quad.
  create_cylinder(r, h, res, split = 4): vertex 0 = (0,0,h/2), 1 = (0,0,-h/2); step = 2 pi / res; for i = 0 .. split, j = 0 ..
                           res-1: (r cos(step j), r sin(step j), h/2 - (h / split) i).  Caps as fans, side quads as two triangles.
  create_cone(r, h, res, split = 1): vertex 0 = (0,0,0), 1 = (0,0,h); step = 2 pi / res; for i = 0 .. split-1 (radius
                           r (split - i) / split, height (h / split) i), j = 0 .. res-1.  Base fan, side triangles to the apex.

All three meshes are CONVEX polyhedra (vertices of the sphere mesh lie on the sphere and every ring quad is a planar
trapezoid; the cylinder is a prism and the cone a pyramid over a regular res-gon), so "strictly inside the closed mesh" is
"strictly on the inner side of every face plane": `inside_convex_mesh` below, brute force over all faces, in float64.
PARITY: pinned to this restatement only - open3d itself cannot be run here, so the product (datapipe.solid_cut_mask, a
closed-form evaluation of the same polyhedra) is checked against THIS file, not against the library."""
import numpy as np


def create_sphere(radius=1.0, resolution=20):
    res = int(resolution)
    V = np.zeros((2 * res * (res - 1) + 2, 3))
    V[0], V[1] = (0.0, 0.0, radius), (0.0, 0.0, -radius)
    step = np.pi / res
    for i in range(1, res):
        alpha = step * i
        base = 2 + 2 * res * (i - 1)
        for j in range(2 * res):
            theta = step * j
            V[base + j] = radius * np.array([np.sin(alpha) * np.cos(theta), np.sin(alpha) * np.sin(theta), np.cos(alpha)])
    T = []
    for j in range(2 * res):
        j1 = (j + 1) % (2 * res)
        base = 2
        T.append((0, base + j, base + j1))
        base = 2 + 2 * res * (res - 2)
        T.append((1, base + j1, base + j))
    for i in range(1, res - 1):
        b1, b2 = 2 + 2 * res * (i - 1), 2 + 2 * res * i
        for j in range(2 * res):
            j1 = (j + 1) % (2 * res)
            T.append((b2 + j, b1 + j1, b1 + j))
            T.append((b2 + j, b2 + j1, b1 + j1))
    return V, np.array(T, dtype=np.int64)


def create_cylinder(radius=1.0, height=2.0, resolution=20, split=4):
    res = int(resolution)
    V = np.zeros((res * (split + 1) + 2, 3))
    V[0], V[1] = (0.0, 0.0, height * 0.5), (0.0, 0.0, -height * 0.5)
    step, h_step = 2 * np.pi / res, height / split
    for i in range(split + 1):
        for j in range(res):
            theta = step * j
            V[2 + res * i + j] = (np.cos(theta) * radius, np.sin(theta) * radius, height * 0.5 - h_step * i)
    T = []
    for j in range(res):
        j1 = (j + 1) % res
        T.append((0, 2 + j, 2 + j1))
        base = 2 + res * split
        T.append((1, base + j1, base + j))
    for i in range(split):
        b1, b2 = 2 + res * i, 2 + res * (i + 1)
        for j in range(res):
            j1 = (j + 1) % res
            T.append((b2 + j, b1 + j1, b1 + j))
            T.append((b2 + j, b2 + j1, b1 + j1))
    return V, np.array(T, dtype=np.int64)


def create_cone(radius=1.0, height=2.0, resolution=20, split=1):
    res = int(resolution)
    V = np.zeros((res * split + 2, 3))
    V[0], V[1] = (0.0, 0.0, 0.0), (0.0, 0.0, height)
    step, h_step, r_step = 2 * np.pi / res, height / split, radius / split
    for i in range(split):
        base, r = 2 + res * i, r_step * (split - i)
        for j in range(res):
            theta = step * j
            V[base + j] = (np.cos(theta) * r, np.sin(theta) * r, h_step * i)
    T = []
    for j in range(res):
        j1 = (j + 1) % res
        T.append((0, 2 + j1, 2 + j))
        base = 2 + res * (split - 1)
        T.append((1, base + j, base + j1))
    for i in range(split - 1):
        b1, b2 = 2 + res * i, 2 + res * (i + 1)
        for j in range(res):
            j1 = (j + 1) % res
            T.append((b2 + j1, b1 + j, b1 + j1))
            T.append((b2 + j1, b2 + j, b1 + j))
    return V, np.array(T, dtype=np.int64)


def rotation_from_axis_angle(w):
    """open3d.geometry.get_rotation_matrix_from_axis_angle: Rodrigues rotation by |w| about w / |w|."""
    w = np.asarray(w, dtype=np.float64).reshape(3)
    th = np.linalg.norm(w)
    if th == 0:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def inside_convex_mesh(points, V, T):
    """Strictly inside the closed CONVEX mesh (V, T): on the inner side of the plane of every triangle, whatever its
    winding (the normal is oriented away from the mesh's centroid).  points [M,3] -> bool [M]; float64, brute force."""
    P = np.asarray(points, dtype=np.float64)
    a, b, c = V[T[:, 0]], V[T[:, 1]], V[T[:, 2]]
    n = np.cross(b - a, c - a)
    keep = np.linalg.norm(n, axis=1) > 0
    a, n = a[keep], n[keep]
    centre = V.mean(axis=0)
    n = n * np.sign(np.einsum("fi,fi->f", n, a - centre))[:, None]          # outward
    n = n / np.linalg.norm(n, axis=1, keepdims=True)
    d = np.einsum("fi,fi->f", n, a)
    inside = np.ones(len(P), dtype=bool)
    for s in range(0, len(n), 512):                                          # (chunks: [M, faces] stays small)
        inside &= (P @ n[s:s + 512].T < d[s:s + 512]).all(axis=1)
    return inside


def solid_mesh(kind, rot=None, shift=None):
    """The moved mesh of dataset.py:716-763 for the given draws -> (V, T)."""
    if kind == "sphere":                                           # :717-719
        V, T = create_sphere(0.5, 50)
        return V + np.asarray(shift, dtype=np.float64).reshape(1, 3), T
    if kind == "cylinder":                                         # :733-735 (rotate about the origin, then translate)
        V, T = create_cylinder(0.6, 1.0, 50)
        R = rotation_from_axis_angle(rot)
        return V @ R.T + np.asarray(shift, dtype=np.float64).reshape(1, 3), T
    if kind == "cone":                                             # :750-752 (translate (0,0,-1), then rotate about the origin)
        V, T = create_cone(1.0, 2.0, 50)
        R = rotation_from_axis_angle(rot)
        return (V + np.array([0.0, 0.0, -1.0])) @ R.T, T
    raise ValueError(kind)


def solid_cut_mask(points, kind, rot=None, shift=None):
    """up-mask of sphere_split / cylinder_split / cone_split for the given draws: signed distance < 0 = strictly inside."""
    V, T = solid_mesh(kind, rot, shift)
    return inside_convex_mesh(points, V, T)
