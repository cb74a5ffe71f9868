"""Vertices of triangle-mesh files (OBJ, STL), for mesh collision shapes.

The reference loads collision meshes with trimesh / pywavefront and hands them to MuJoCo, which collides a mesh geom as its
convex hull (reference mjcf.py:270-413).  Only the vertex cloud matters for that, so this reader returns vertices and
nothing else (no faces, normals, materials); the hull is taken by ``ModelBuilder.add_mesh_geom``."""
import os
import struct

import numpy as np


def _read_obj(path):
    verts = []
    with open(path, 'r', errors='replace') as f:
        for line in f:
            if line.startswith('v '):
                p = line.split()
                verts.append([float(p[1]), float(p[2]), float(p[3])])
    return np.array(verts, float).reshape(-1, 3)


def _read_stl(path):
    with open(path, 'rb') as f:
        raw = f.read()
    if len(raw) >= 84:
        n = struct.unpack('<I', raw[80:84])[0]
        if len(raw) == 84 + 50*n:                       # binary: 80-byte header, count, 50-byte facets
            tri = np.frombuffer(raw, dtype=np.dtype([('n', '<f4', 3), ('v', '<f4', (3, 3)), ('a', '<u2')]), count=n, offset=84)
            return np.unique(tri['v'].reshape(-1, 3).astype(float), axis=0)
    verts = []
    for line in raw.decode('ascii', errors='replace').splitlines():
        p = line.split()
        if len(p) == 4 and p[0] == 'vertex':
            verts.append([float(p[1]), float(p[2]), float(p[3])])
    return np.unique(np.array(verts, float).reshape(-1, 3), axis=0)


def read_vertices(path, scale=(1.0, 1.0, 1.0)):
    """[n, 3] vertices of an ``.obj`` or ``.stl`` file, times ``scale`` (the SDF mesh element's <scale>)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == '.obj':
        v = _read_obj(path)
    elif ext == '.stl':
        v = _read_stl(path)
    else:
        raise NotImplementedError(f'mesh format {ext!r} of {path}: .obj and .stl are read here (the reference converts other '
                                  'formats to .stl with trimesh first, mjcf.py:276-300)')
    assert len(v) > 0, f'no vertices in {path}'
    return v*np.asarray(scale, float).reshape(1, 3)
