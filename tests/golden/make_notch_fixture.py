"""Generate tests/golden/notch_tet4.json from the reference's shipped notch mesh (PUMI .smb, one part).

Inputs (data files held by the reference's own tests, read here once):
  /root/reference/source/calibr8/test/mesh/notch/notch0.smb   SCOREC/PUMI binary mesh, 546 vertices / 1550 tets
  /root/reference/source/calibr8/test/mesh/notch/notch.txt    set associations (model-entity ids)
Output: coordinates, tet connectivity (0-based), node sets and side sets keyed by the names the test decks use.

The .smb layout read here (big-endian; SCOREC core `mds_smb.c`, version 5, written down from the file itself and
checked below against the gmsh file of the SAME cube mesh the reference ships in both formats):
  u32 magic = 0, version, dim, nparts
  u32 n[8]                      entity counts: vertex, edge, triangle, quad, hex, prism, pyramid, tet
  u32 conn                      for edge, triangle, ..., tet in that order: the entities ONE dimension down
                                (edge -> 2 vertices, triangle -> 3 edges, tet -> 4 triangles), 0-based
  f64 xyz[n_vertex][3], f64 param[n_vertex][2]
  u32 number of remote copies   (0 in a one-part mesh)
  u32 (model id, model dim)     classification of every entity, same type order
  ... tags, matches (not needed)
A tet's vertices are the union of the vertices of its triangles' edges; the local order is fixed here by positive
volume (the assembled system of linear tets does not depend on it).  A node belongs to the node set of a model face
when it is a vertex of a mesh triangle classified on that face, as in make_cube_fixture.py.

Run:  python tests/golden/make_notch_fixture.py
"""
import json
import os
import struct

import numpy as np

REF = "/root/reference/source/calibr8/test/mesh"
HERE = os.path.dirname(os.path.abspath(__file__))


def read_smb(path):
    b = open(path, "rb").read()
    magic, version, dim, nparts = struct.unpack(">4I", b[:16])
    assert magic == 0 and dim == 3 and nparts == 1, (magic, version, dim, nparts)
    n = struct.unpack(">8I", b[16:48])
    nv, ne, nt, nq, nh, npr, npy, ntet = n
    assert nq == nh == npr == npy == 0, "simplex meshes only"
    off = 48

    def u32(count, width):
        nonlocal off
        a = np.frombuffer(b, ">u4", count * width, off).reshape(count, width).astype(np.int64)
        off += 4 * count * width
        return a

    edges, tris, tets = u32(ne, 2), u32(nt, 3), u32(ntet, 4)
    xyz = np.frombuffer(b, ">f8", nv * 3, off).reshape(nv, 3).astype(np.float64)
    off += 8 * nv * 3 + 8 * nv * 2  # coordinates, then parametric coordinates
    (nremotes,) = struct.unpack(">I", b[off:off + 4])
    assert nremotes == 0
    off += 4
    cls = {}
    for name, cnt in (("vertex", nv), ("edge", ne), ("tri", nt), ("tet", ntet)):
        cls[name] = u32(cnt, 2)  # (model id, model dim)
    tri_verts = np.array([sorted(set(edges[t].ravel())) for t in tris])
    assert tri_verts.shape == (nt, 3)
    tet_verts = np.array([sorted(set(tri_verts[t].ravel())) for t in tets])
    assert tet_verts.shape == (ntet, 4)
    # positive orientation
    x = xyz[tet_verts]
    vol = np.einsum("ij,ij->i", np.cross(x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]), x[:, 3] - x[:, 0])
    flip = vol < 0
    tet_verts[flip] = tet_verts[flip][:, [0, 2, 1, 3]]
    return xyz, tet_verts, tri_verts, cls


def read_assoc(path):
    sets = {}
    lines = [l for l in open(path).read().split("\n") if l.strip()]
    i = 0
    while i < len(lines):
        kind, _, name, cnt = lines[i].split()
        sets[(kind, name)] = [tuple(int(v) for v in lines[i + 1 + k].split()) for k in range(int(cnt))]
        i += 1 + int(cnt)
    return sets


def fixture(smb, assoc_path, source):
    xyz, tets, tri_verts, cls = read_smb(smb)
    assoc = read_assoc(assoc_path)
    node_sets, side_sets = {}, {}
    for (kind, name), ents in assoc.items():
        faces = {tag for d, tag in ents if d == 2}
        on = [k for k in range(len(tri_verts)) if cls["tri"][k][1] == 2 and cls["tri"][k][0] in faces]
        if kind == "node":
            node_sets[name] = sorted({int(v) for k in on for v in tri_verts[k]})
        elif kind == "side":
            side_sets[name] = [[int(v) for v in tri_verts[k]] for k in on]
    return {"source": source, "elem_type": "tet4", "coords": xyz.tolist(), "conn": tets.tolist(),
            "node_sets": node_sets, "side_sets": side_sets}


def check_reader_on_cube():
    """The reference ships the cube in both formats: the .smb reader must reproduce the gmsh-derived fixture."""
    d = fixture(os.path.join(REF, "cube", "cube0.smb"), os.path.join(REF, "cube", "cube.txt"), "check")
    g = json.load(open(os.path.join(HERE, "cube_tet4.json")))
    key = lambda c: tuple(np.round(c, 12))
    gid = {key(c): i for i, c in enumerate(g["coords"])}
    perm = [gid[key(c)] for c in d["coords"]]  # smb vertex -> gmsh node
    assert sorted(perm) == list(range(len(g["coords"])))
    assert sorted(tuple(sorted(perm[v] for v in t)) for t in d["conn"]) == sorted(tuple(sorted(t)) for t in g["conn"])
    for name, nodes in g["node_sets"].items():
        assert sorted(perm[v] for v in d["node_sets"][name]) == sorted(nodes), name
    for name, tris in g["side_sets"].items():
        assert sorted(tuple(sorted(perm[v] for v in t)) for t in d["side_sets"][name]) == sorted(tuple(sorted(t)) for t in tris)
    print("cube0.smb == cube.msh: vertices, tets, node sets and side sets agree")


def main():
    check_reader_on_cube()
    d = fixture(os.path.join(REF, "notch", "notch0.smb"), os.path.join(REF, "notch", "notch.txt"),
                "sandialabs/calibr8 test/mesh/notch/notch0.smb + notch.txt")
    out = os.path.join(HERE, "notch_tet4.json")
    json.dump(d, open(out, "w"))
    c = np.array(d["coords"])
    print("wrote", out, len(d["coords"]), "nodes", len(d["conn"]), "tets; bounding box", c.min(0), c.max(0),
          {k: len(v) for k, v in d["node_sets"].items()})


if __name__ == "__main__":
    main()
