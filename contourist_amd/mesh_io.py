"""Binary mesh writers (the step right after get_points_and_triangles for every caller of the reference,
html_demo.py:118-161, next to the three.js emitters of contourist_amd/html_demo.py).

Two kinds: `write_ply` / `write_gltf_bin` take host arrays; `write_ply_device` / `write_gltf_device` take an isosurface object of
`contourist_amd.tetrahedral` and let the library write the file straight from the Level-1 DEVICE buffers (`cx_level1_write`:
records laid out on the GPU, streamed through pinned staging buffers) -- no (points, triangles) arrays in Python."""
import struct

import numpy as np


def write_ply(path, points, triangles, comment="contourist_amd isosurface"):
    """binary little-endian PLY: float64 x y z per vertex, int32 index triples (consistently wound)."""
    P = np.ascontiguousarray(np.asarray(points, dtype="<f8").reshape(-1, 3))
    T = np.ascontiguousarray(np.asarray(triangles, dtype="<i4").reshape(-1, 3))
    header = ("ply\nformat binary_little_endian 1.0\ncomment %s\nelement vertex %d\n"
              "property double x\nproperty double y\nproperty double z\n"
              "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (comment, len(P), len(T)))
    faces = np.empty(len(T), dtype=[("n", "u1"), ("v", "<i4", (3,))])
    faces["n"] = 3
    faces["v"] = T
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(P.tobytes())
        f.write(faces.tobytes())
    return path


def read_ply(path):
    "reader for the files write_ply produces (tests, quick inspection) -> (points (V,3) float64, triangles (T,3) int32)"
    with open(path, "rb") as f:
        nv = nt = None
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                nv = int(line.split()[-1])
            elif line.startswith("element face"):
                nt = int(line.split()[-1])
            elif line == "end_header":
                break
        P = np.frombuffer(f.read(nv * 24), dtype="<f8").reshape(nv, 3).copy()
        faces = np.frombuffer(f.read(nt * 13), dtype=[("n", "u1"), ("v", "<i4", (3,))])
        assert np.all(faces["n"] == 3)
        return P, faces["v"].astype(np.int32)


def write_gltf_bin(path_gltf, points, triangles):
    """minimal glTF 2.0 (.gltf + .bin next to it): float32 positions, uint32 indices."""
    import json
    import os
    P = np.ascontiguousarray(np.asarray(points, dtype="<f4").reshape(-1, 3))
    T = np.ascontiguousarray(np.asarray(triangles, dtype="<u4").reshape(-1))
    bin_name = os.path.splitext(os.path.basename(path_gltf))[0] + ".bin"
    blob = P.tobytes() + T.tobytes()
    doc = {
        "asset": {"version": "2.0", "generator": "contourist_amd"},
        "buffers": [{"uri": bin_name, "byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": P.nbytes, "target": 34962},
                        {"buffer": 0, "byteOffset": P.nbytes, "byteLength": T.nbytes, "target": 34963}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": int(len(P)), "type": "VEC3",
                       "min": [float(x) for x in (P.min(axis=0) if len(P) else np.zeros(3))],
                       "max": [float(x) for x in (P.max(axis=0) if len(P) else np.zeros(3))]},
                      {"bufferView": 1, "componentType": 5125, "count": int(len(T)), "type": "SCALAR"}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "mode": 4}]}],
        "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}], "scene": 0,
    }
    with open(os.path.join(os.path.dirname(path_gltf) or ".", bin_name), "wb") as f:
        f.write(blob)
    with open(path_gltf, "w") as f:
        json.dump(doc, f)
    return path_gltf


def write_ply_device(surface, path):
    """`surface`: TriangulatedIsosurfaces / Delta3DContour (world coordinates) or GridContour3d (grid coordinates).
    Same bytes as write_ply(path, points, triangles) of the downloaded mesh with the faces in device order."""
    return surface.write_mesh(path, "ply")


def write_gltf_device(surface, path_gltf):
    "minimal glTF 2.0: the .bin payload comes straight from the device buffers, the JSON from the bounds the writer returns"
    import json
    import os
    bin_name = os.path.splitext(os.path.basename(path_gltf))[0] + ".bin"
    bin_path = os.path.join(os.path.dirname(path_gltf) or ".", bin_name)
    if hasattr(surface, "contour_maker"):
        info = surface.contour_maker.write_mesh(bin_path, "gltf_bin", surface.grid.mins, surface.grid.delta)
    else:
        info = surface.write_mesh(bin_path, "gltf_bin")
    nv, nt = info["n_vertices"], info["n_triangles"]
    doc = {
        "asset": {"version": "2.0", "generator": "contourist_amd"},
        "buffers": [{"uri": bin_name, "byteLength": info["bytes"]}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": nv * 12, "target": 34962},
                        {"buffer": 0, "byteOffset": nv * 12, "byteLength": nt * 12, "target": 34963}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": nv, "type": "VEC3",
                       "min": [float(x) for x in info["min"]], "max": [float(x) for x in info["max"]]},
                      {"bufferView": 1, "componentType": 5125, "count": nt * 3, "type": "SCALAR"}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "mode": 4}]}],
        "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}], "scene": 0,
    }
    with open(path_gltf, "w") as f:
        json.dump(doc, f)
    return path_gltf
