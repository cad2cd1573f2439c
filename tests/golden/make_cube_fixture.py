"""Generate tests/golden/cube_tet4.json from the reference's shipped cube mesh.

Inputs (data files held by the reference's own tests, read here once):
  /root/reference/source/calibr8/test/mesh/cube/cube.msh   gmsh 2.2 ASCII, 14 nodes / 24 tets
  /root/reference/source/calibr8/test/mesh/cube/cube.txt   set associations (model-entity ids)
Output: coordinates, tet connectivity (0-based), node sets and side sets
(boundary triangles) keyed by the names the test decks use (xmin, ymin, zmin, ymax).

Run:  python tests/golden/make_cube_fixture.py
"""
import json
import os

REF = "/root/reference/source/calibr8/test/mesh/cube"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cube_tet4.json")


def read_msh(path):
    lines = open(path).read().split("\n")
    i = lines.index("$Nodes")
    n = int(lines[i + 1])
    coords = []
    for k in range(n):
        t = lines[i + 2 + k].split()
        coords.append([float(t[1]), float(t[2]), float(t[3])])
    i = lines.index("$Elements")
    m = int(lines[i + 1])
    tets, tris = [], []
    for k in range(m):
        t = [int(v) for v in lines[i + 2 + k].split()]
        etype, ntags = t[1], t[2]
        tags = t[3:3 + ntags]
        nodes = [v - 1 for v in t[3 + ntags:]]
        if etype == 4:
            tets.append(nodes)
        elif etype == 2:
            tris.append((tags[1], nodes))  # elementary (model face) tag
    return coords, tets, tris


def read_assoc(path):
    sets = {}
    lines = [l for l in open(path).read().split("\n") if l.strip()]
    i = 0
    while i < len(lines):
        kind, _, name, cnt = lines[i].split()
        ents = []
        for k in range(int(cnt)):
            d, tag = lines[i + 1 + k].split()
            ents.append((int(d), int(tag)))
        sets[(kind, name)] = ents
        i += 1 + int(cnt)
    return sets


def main():
    coords, tets, tris = read_msh(os.path.join(REF, "cube.msh"))
    assoc = read_assoc(os.path.join(REF, "cube.txt"))
    node_sets, side_sets = {}, {}
    for (kind, name), ents in assoc.items():
        faces = {tag for d, tag in ents if d == 2}
        if kind == "node":
            nodes = sorted({n for tag, tri in tris if tag in faces for n in tri})
            node_sets[name] = nodes
        elif kind == "side":
            side_sets[name] = [tri for tag, tri in tris if tag in faces]
    out = {
        "source": "sandialabs/calibr8 test/mesh/cube/cube.msh + cube.txt",
        "elem_type": "tet4",
        "coords": coords,
        "conn": tets,
        "node_sets": node_sets,
        "side_sets": side_sets,
    }
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, len(coords), "nodes", len(tets), "tets")


if __name__ == "__main__":
    main()
