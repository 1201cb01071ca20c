"""Mesh input for the depth rasteriser: ``load_ply(path, vertex_scale)`` with the return convention of the
reference's BOP-toolkit fork (lib/pysixd/inout.py ``load_ply``: dict with 'pts' [V,3], 'faces' [F,3] and, when
present, 'normals', 'colors', 'texture_uv'), as used at gdrn_evaluator.py:58-63 and
lib/render_vispy/model3d.py.  Written from the PLY format specification (ASCII and binary little/big endian,
list properties for faces); polygons with more than 3 vertices are fan-triangulated."""
from __future__ import annotations

import struct

import numpy as np

_PLY_TYPES = {
    "char": ("b", 1), "int8": ("b", 1), "uchar": ("B", 1), "uint8": ("B", 1), "short": ("h", 2), "int16": ("h", 2),
    "ushort": ("H", 2), "uint16": ("H", 2), "int": ("i", 4), "int32": ("i", 4), "uint": ("I", 4), "uint32": ("I", 4),
    "float": ("f", 4), "float32": ("f", 4), "double": ("d", 8), "float64": ("d", 8),
}


def load_ply(path, vertex_scale: float = 1.0) -> dict:
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").strip().split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append({"name": tok[1], "count": int(tok[2]), "props": []})
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1]["props"].append(("list", tok[2], tok[3], tok[4]))
                else:
                    elements[-1]["props"].append(("scalar", tok[1], tok[2]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        endian = ">" if fmt == "binary_big_endian" else "<"
        data = {}
        for el in elements:
            rows = []
            all_scalar = all(p[0] == "scalar" for p in el["props"])
            if fmt != "ascii" and all_scalar:
                dt = np.dtype([(p[2], endian + _PLY_TYPES[p[1]][0]) for p in el["props"]])
                rows = np.frombuffer(f.read(dt.itemsize * el["count"]), dtype=dt, count=el["count"])
                data[el["name"]] = {p[2]: rows[p[2]] for p in el["props"]}
                continue
            cols = {p[-1]: [] for p in el["props"]}
            for _ in range(el["count"]):
                if fmt == "ascii":
                    vals = f.readline().split()
                    k = 0
                    for p in el["props"]:
                        if p[0] == "scalar":
                            cols[p[2]].append(float(vals[k])); k += 1
                        else:
                            n = int(vals[k]); k += 1
                            cols[p[3]].append([int(float(v)) for v in vals[k:k + n]]); k += n
                else:
                    for p in el["props"]:
                        if p[0] == "scalar":
                            c, sz = _PLY_TYPES[p[1]]
                            cols[p[2]].append(struct.unpack(endian + c, f.read(sz))[0])
                        else:
                            c, sz = _PLY_TYPES[p[1]]
                            n = struct.unpack(endian + c, f.read(sz))[0]
                            c2, sz2 = _PLY_TYPES[p[2]]
                            cols[p[3]].append(list(struct.unpack(endian + c2 * n, f.read(sz2 * n))))
            data[el["name"]] = cols
    v = data.get("vertex")
    if v is None:
        raise ValueError(f"{path}: no vertex element")
    model = {"pts": np.stack([np.asarray(v[a], np.float64) for a in ("x", "y", "z")], 1) * vertex_scale}
    if all(a in v for a in ("nx", "ny", "nz")):
        model["normals"] = np.stack([np.asarray(v[a], np.float64) for a in ("nx", "ny", "nz")], 1)
    if all(a in v for a in ("red", "green", "blue")):
        model["colors"] = np.stack([np.asarray(v[a], np.float64) for a in ("red", "green", "blue")], 1)
    for ua, va in (("texture_u", "texture_v"), ("u", "v"), ("s", "t")):
        if ua in v and va in v:
            model["texture_uv"] = np.stack([np.asarray(v[ua], np.float64), np.asarray(v[va], np.float64)], 1)
            break
    faces = []
    fel = data.get("face", {})
    key = next((k for k in ("vertex_indices", "vertex_index") if k in fel), None)
    if key is not None:
        for poly in fel[key]:
            for j in range(1, len(poly) - 1):
                faces.append((poly[0], poly[j], poly[j + 1]))
    model["faces"] = np.asarray(faces, np.int64).reshape(-1, 3)
    return model


def models_to_meshset(models, device="cuda"):
    """[{'pts','faces'}, …] -> hip_lib.MeshSet (all objects flat in HBM)."""
    from ... import hip_lib

    return hip_lib.MeshSet([np.asarray(m["pts"], np.float32) for m in models],
                           [np.asarray(m["faces"], np.int32) for m in models], device=device)
