"""Generate tests/golden/notch2D_tri3.json from the reference's shipped 2-D notch mesh (PUMI .smb, one part).

Inputs (data files held by the reference's own tests, read here once):
  /root/reference/source/calibr8/test/mesh/notch2D/notch2D0.smb   SCOREC/PUMI binary mesh, 252 vertices / 447 triangles
  /root/reference/source/calibr8/test/mesh/notch2D/notch2D.txt    set associations (model-entity dim and id)
Output: coordinates (z = 0), triangle connectivity (0-based, counter-clockwise), node sets keyed by the names the test
decks use (`test/primal/notch2D_small_J2.yaml.in`: xmin, ymin, ymax).

The .smb layout is the one make_notch_fixture.py documents and validates (3-D: against the cube the reference ships in
two formats); a 2-D file has the same header with dim = 2 and no tets: edges -> 2 vertices, triangles -> 3 edges,
classification (model id, model dim) for vertices, edges, triangles.  A node belongs to the node set of a model edge
when it is a vertex of a mesh edge classified on that model edge (the closure of the model edge, as
apf::collectEntityModels gives it in Disc::compute_node_sets, disc.cpp:519-540).

Run:  python tests/golden/make_notch2d_fixture.py
"""
import json
import os
import struct

import numpy as np

from make_notch_fixture import read_assoc

REF = "/root/reference/source/calibr8/test/mesh"
HERE = os.path.dirname(os.path.abspath(__file__))


def read_smb_2d(path):
    b = open(path, "rb").read()
    magic, version, dim, nparts = struct.unpack(">4I", b[:16])
    assert magic == 0 and dim == 2 and nparts == 1, (magic, version, dim, nparts)
    nv, ne, nt, nq, nh, npr, npy, ntet = struct.unpack(">8I", b[16:48])
    assert nq == nh == npr == npy == ntet == 0, "triangle meshes only"
    off = 48

    def u32(count, width):
        nonlocal off
        a = np.frombuffer(b, ">u4", count * width, off).reshape(count, width).astype(np.int64)
        off += 4 * count * width
        return a

    edges, tris = u32(ne, 2), u32(nt, 3)
    xyz = np.frombuffer(b, ">f8", nv * 3, off).reshape(nv, 3).astype(np.float64)
    off += 8 * nv * 3 + 8 * nv * 2  # coordinates, then parametric coordinates
    (nremotes,) = struct.unpack(">I", b[off:off + 4])
    assert nremotes == 0
    off += 4
    cls = {}
    for name, cnt in (("vertex", nv), ("edge", ne), ("tri", nt)):
        cls[name] = u32(cnt, 2)  # (model id, model dim)
    tri_verts = np.array([sorted(set(edges[t].ravel())) for t in tris])
    assert tri_verts.shape == (nt, 3)
    x = xyz[tri_verts]
    area2 = (x[:, 1, 0] - x[:, 0, 0]) * (x[:, 2, 1] - x[:, 0, 1]) - (x[:, 1, 1] - x[:, 0, 1]) * (x[:, 2, 0] - x[:, 0, 0])
    flip = area2 < 0
    tri_verts[flip] = tri_verts[flip][:, [0, 2, 1]]
    assert np.abs(xyz[:, 2]).max() == 0.0
    return xyz, tri_verts, edges, cls


def main():
    xyz, tris, edges, cls = read_smb_2d(os.path.join(REF, "notch2D", "notch2D0.smb"))
    assoc = read_assoc(os.path.join(REF, "notch2D", "notch2D.txt"))
    node_sets = {}
    for (kind, name), ents in assoc.items():
        if kind != "node":
            continue
        tags = {tag for d, tag in ents if d == 1}
        on = [k for k in range(len(edges)) if cls["edge"][k][1] == 1 and cls["edge"][k][0] in tags]
        node_sets[name] = sorted({int(v) for k in on for v in edges[k]})
    body = {tag for d, tag in assoc[("elem", "body")] if d == 2}
    assert all(int(c[1]) == 2 and int(c[0]) in body for c in cls["tri"]), "one element set: body"
    d = {"source": "sandialabs/calibr8 test/mesh/notch2D/notch2D0.smb + notch2D.txt", "elem_type": "tri3",
         "coords": xyz.tolist(), "conn": tris.tolist(), "node_sets": node_sets}
    out = os.path.join(HERE, "notch2D_tri3.json")
    json.dump(d, open(out, "w"))
    c = np.array(d["coords"])
    x = c[tris]
    area = 0.5 * ((x[:, 1, 0] - x[:, 0, 0]) * (x[:, 2, 1] - x[:, 0, 1]) - (x[:, 1, 1] - x[:, 0, 1]) * (x[:, 2, 0] - x[:, 0, 0]))
    print("wrote", out, len(c), "nodes", len(tris), "triangles; bounding box", c.min(0), c.max(0), "area", area.sum(),
          {k: len(v) for k, v in node_sets.items()})
    for name, nodes in node_sets.items():
        print(name, "x range", c[nodes, 0].min(), c[nodes, 0].max(), "y range", c[nodes, 1].min(), c[nodes, 1].max())


if __name__ == "__main__":
    main()
