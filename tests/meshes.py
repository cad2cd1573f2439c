"""Synthetic meshes and prescribed fields for tests and the bench."""
import numpy as np


def brick(nx, ny, nz, lx=1.0, ly=1.0, lz=1.0):
    """Structured hex8 brick, x-fastest node numbering, standard hex8 node order."""
    xs, ys, zs = np.linspace(0, lx, nx + 1), np.linspace(0, ly, ny + 1), np.linspace(0, lz, nz + 1)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    nid = lambda i, j, k: (k * (ny + 1) + j) * (nx + 1) + i
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()
    conn = np.stack([nid(i, j, k), nid(i + 1, j, k), nid(i + 1, j + 1, k), nid(i, j + 1, k),
                     nid(i, j, k + 1), nid(i + 1, j, k + 1), nid(i + 1, j + 1, k + 1), nid(i, j + 1, k + 1)],
                    axis=1).astype(np.int32)
    tol = 1e-12
    sets = {
        "xmin": np.where(coords[:, 0] < tol)[0], "xmax": np.where(coords[:, 0] > lx - tol)[0],
        "ymin": np.where(coords[:, 1] < tol)[0], "ymax": np.where(coords[:, 1] > ly - tol)[0],
        "zmin": np.where(coords[:, 2] < tol)[0], "zmax": np.where(coords[:, 2] > lz - tol)[0],
    }
    return coords, conn, sets


def notched_bar(nx, ny, nz, lx=4.0, ly=1.0, lz=1.0, depth=0.3, halfwidth=0.25):
    """Double-edge-notched tensile specimen in hex8 (BASELINE config 3): the brick of `brick` with two V-notches cut at
    mid-length from the faces y = 0 and y = ly (elements whose centre lies inside a notch are removed, the nodes they
    leave unused are dropped and the rest renumbered in the old order).  Node degrees then vary along the notch flanks:
    this is the unstructured case of the staged assembly and of the node graphs.  Returns coords, conn, node sets."""
    coords, conn, _ = brick(nx, ny, nz, lx, ly, lz)
    ctr = coords[conn].mean(axis=1)
    reach = depth * ly * np.clip(1.0 - np.abs(ctr[:, 0] - 0.5 * lx) / (halfwidth * lx), 0.0, None)  # notch depth at x
    keep = (ctr[:, 1] >= reach) & (ctr[:, 1] <= ly - reach)
    conn = conn[keep]
    used = np.zeros(len(coords), dtype=bool)
    used[conn.ravel()] = True
    new_id = np.cumsum(used) - 1
    coords, conn = coords[used], new_id[conn].astype(np.int32)
    tol = 1e-12
    sets = {"xmin": np.where(coords[:, 0] < tol)[0], "xmax": np.where(coords[:, 0] > lx - tol)[0],
            "zmin": np.where(coords[:, 2] < tol)[0], "zmax": np.where(coords[:, 2] > lz - tol)[0]}
    return np.ascontiguousarray(coords), np.ascontiguousarray(conn), sets


def pinched_bricks():
    """Two 2 x 2 x 2 hex8 bricks that share exactly one node: the centre node of the first is also the centre node of the
    second (which is rotated and shrunk so that no other nodes coincide).  That node has sixteen elements and 53 graph
    neighbours: the case of a node with more than eight elements (unstructured hex8 meshes have them at irregular points)."""
    c1, conn1, _ = brick(2, 2, 2, 1.0, 1.0, 1.0)
    c2, conn2, _ = brick(2, 2, 2, 0.8, 0.9, 0.7)
    ctr1, ctr2 = 13, 13  # centre node of a 3 x 3 x 3 node grid
    th = 0.6
    R = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])
    c2 = (c2 - c2[ctr2]) @ R.T + c1[ctr1]
    keep = np.arange(len(c2)) != ctr2
    new_id = np.full(len(c2), -1, dtype=np.int64)
    new_id[keep] = len(c1) + np.arange(keep.sum())
    new_id[ctr2] = ctr1
    coords = np.concatenate([c1, c2[keep]])
    conn = np.concatenate([conn1, new_id[conn2]]).astype(np.int32)
    return np.ascontiguousarray(coords), np.ascontiguousarray(conn)


def tri_mesh(nx, ny, lx=1.0, ly=1.0):
    """Structured tri3 mesh of a rectangle (each cell split along alternating diagonals), counter-clockwise triangles;
    coords [n][3] with z = 0 (the layout the 2-D path takes)."""
    xs, ys = np.linspace(0, lx, nx + 1), np.linspace(0, ly, ny + 1)
    Y, X = np.meshgrid(ys, xs, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1)
    nid = lambda i, j: j * (nx + 1) + i
    tris = []
    for j in range(ny):
        for i in range(nx):
            a, b, c, d = nid(i, j), nid(i + 1, j), nid(i + 1, j + 1), nid(i, j + 1)
            tris += [[a, b, c], [a, c, d]] if (i + j) % 2 == 0 else [[a, b, d], [b, c, d]]
    tol = 1e-12
    sets = {"xmin": np.where(coords[:, 0] < tol)[0], "xmax": np.where(coords[:, 0] > lx - tol)[0],
            "ymin": np.where(coords[:, 1] < tol)[0], "ymax": np.where(coords[:, 1] > ly - tol)[0]}
    return coords, np.array(tris, dtype=np.int32), sets


def jiggle_2d(coords, sets, amp, seed=7):
    """jiggle() in the plane (z stays 0)"""
    c = jiggle(coords, sets, amp, seed)
    c[:, 2] = 0.0
    return c


def fields_for(ndims, u, p):
    """a prescribed 3-D field pair restricted to the in-plane components for a 2-D mesh"""
    if ndims == 3:
        return u, p
    return np.ascontiguousarray(u.reshape(-1, 3)[:, :2].ravel()), p


def jiggle(coords, sets, amp, seed=7):
    """Perturb interior nodes so elements are not parallelepipeds (exercises the full Jacobian)."""
    rng = np.random.default_rng(seed)
    c = coords.copy()
    bnd = np.zeros(len(c), dtype=bool)
    for v in sets.values():
        bnd[v] = True
    c[~bnd] += amp * (rng.random((int((~bnd).sum()), 3)) - 0.5)
    return c


def prescribed_fields(coords, eps_bar, E=1000.0, nu=0.25, seed=1234, ramp=False, perturb=1e-4):
    """Prescribed displacement/pressure state of SURVEY.md section 8d: uniaxial stretch along y with a
    lateral contraction, plus a small random perturbation (numpy default_rng(seed)).  With
    ramp=True the stretch grows linearly with y so that part of the bar is plastic."""
    rng = np.random.default_rng(seed)
    x, y, z = coords[:, 0], coords[:, 1], coords[:, 2]
    ly = max(y.max(), 1e-300)
    e = eps_bar * (y / ly if ramp else 1.0)
    u = np.zeros_like(coords)
    if ramp:
        u[:, 1] = 0.5 * eps_bar * y * y / ly
    else:
        u[:, 1] = eps_bar * y
    u[:, 0] = -nu * e * x
    u[:, 2] = -nu * e * z
    u += perturb * eps_bar * (rng.random(coords.shape) - 0.5)
    kappa = E / (3 * (1 - 2 * nu))
    p = -kappa * (1 - 2 * nu) * e + perturb * eps_bar * kappa * (rng.random(len(coords)) - 0.5)
    return np.ascontiguousarray(u.ravel()), np.ascontiguousarray(p)
