"""Minimal SDF reader — stand-in for farms_core.io.sdf (ModelSDF, Link, Joint, Collision, shapes), which the
reference imports (reference mjcf.py:30-33) but does not vendor.  Only what the model compiler consumes is parsed:
link poses, inertials, collision geometry (sphere / capsule / cylinder / box / plane / heightmap) and joints."""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


def _floats(text, n=None, default=None):
    if text is None:
        return np.array(default, float) if default is not None else None
    v = np.array([float(x) for x in text.split()], float)
    assert n is None or len(v) == n, (text, n)
    return v


@dataclass
class Geometry:
    kind: str                       # 'sphere' | 'capsule' | 'cylinder' | 'box' | 'plane' | 'heightmap' | 'mesh'
    size: np.ndarray                # sphere: [r]; capsule/cylinder: [r, length]; box: [x, y, z]; plane: normal; heightmap: [x, y, z] extents
    uri: str = ''                   # heightmap: image file; mesh: .obj / .stl file (size = its <scale>); relative to the SDF's directory

    def bounding_radius(self) -> float:
        """MuJoCo geom_rbound of the shape (used for SwimmingHandler heights, reference drag.pyx:364-372)."""
        if self.kind == 'sphere':
            return float(self.size[0])
        if self.kind == 'capsule':
            return float(self.size[0] + 0.5*self.size[1])
        if self.kind == 'cylinder':
            return float(np.hypot(self.size[0], 0.5*self.size[1]))
        if self.kind == 'box':
            return float(0.5*np.linalg.norm(self.size))
        return 0.0


@dataclass
class Collision:
    name: str
    pose: np.ndarray
    geometry: Geometry


@dataclass
class Inertial:
    pose: np.ndarray
    mass: float
    inertias: np.ndarray            # ixx ixy ixz iyy iyz izz (order used at reference mjcf.py:540-551)


@dataclass
class Link:
    name: str
    pose: np.ndarray
    inertial: Optional[Inertial] = None
    collisions: List[Collision] = field(default_factory=list)


@dataclass
class Axis:
    xyz: np.ndarray
    limits: Optional[np.ndarray] = None      # lower, upper[, effort, velocity]


@dataclass
class Joint:
    name: str
    type: str
    parent: str
    child: str
    pose: np.ndarray
    axis: Axis


class ModelSDF:
    """One <model> of an SDF file."""

    def __init__(self, name, pose, links, joints, directory=''):
        self.name = name
        self.pose = np.asarray(pose, float)
        self.links: List[Link] = links
        self.joints: List[Joint] = joints
        self.directory = directory

    @classmethod
    def read(cls, filename: str) -> List['ModelSDF']:
        root = ET.parse(filename).getroot()
        models = root.findall('model') if root.tag == 'sdf' else [root]
        return [cls._from_xml(m, os.path.dirname(os.path.abspath(filename))) for m in models]

    @classmethod
    def _from_xml(cls, m, directory):
        def pose_of(e):
            p = e.find('pose')
            return _floats(p.text if p is not None else None, 6, default=[0]*6)
        links = []
        for le in m.findall('link'):
            inertial = None
            ie = le.find('inertial')
            if ie is not None:
                it = ie.find('inertia')
                ins = [float(it.find(k).text) if it is not None and it.find(k) is not None else 0.0
                       for k in ('ixx', 'ixy', 'ixz', 'iyy', 'iyz', 'izz')]
                inertial = Inertial(pose_of(ie), float(ie.find('mass').text), np.array(ins))
            cols = []
            for ce in le.findall('collision'):
                ge = ce.find('geometry')
                geo = None
                for kind in ('sphere', 'capsule', 'cylinder', 'box', 'plane', 'heightmap', 'mesh'):
                    k = ge.find(kind)
                    if k is None:
                        continue
                    if kind == 'sphere':
                        geo = Geometry(kind, np.array([float(k.find('radius').text)]))
                    elif kind in ('capsule', 'cylinder'):
                        geo = Geometry(kind, np.array([float(k.find('radius').text), float(k.find('length').text)]))
                    elif kind == 'box':
                        geo = Geometry(kind, _floats(k.find('size').text, 3))
                    elif kind == 'plane':
                        n = k.find('normal')
                        geo = Geometry(kind, _floats(n.text if n is not None else '0 0 1', 3))
                    elif kind == 'heightmap':
                        geo = Geometry(kind, _floats(k.find('size').text, 3), uri=k.find('uri').text.strip())
                    else:                               # mesh: <uri>, optional <scale>
                        sc = k.find('scale')
                        geo = Geometry(kind, _floats(sc.text if sc is not None else '1 1 1', 3),
                                       uri=k.find('uri').text.strip() if k.find('uri') is not None else '')
                cols.append(Collision(ce.get('name', f'{le.get("name")}_collision'), pose_of(ce), geo))
            links.append(Link(le.get('name'), pose_of(le), inertial, cols))
        joints = []
        for je in m.findall('joint'):
            ae = je.find('axis')
            xyz = _floats(ae.find('xyz').text, 3) if ae is not None and ae.find('xyz') is not None else np.array([0., 0., 1.])
            limits = None
            if ae is not None and ae.find('limit') is not None:
                le_ = ae.find('limit')
                lo, hi = le_.find('lower'), le_.find('upper')
                if lo is not None and hi is not None:
                    limits = np.array([float(lo.text), float(hi.text)])
            joints.append(Joint(je.get('name'), je.get('type'), je.find('parent').text.strip(), je.find('child').text.strip(),
                                pose_of(je), Axis(xyz, limits)))
        return cls(m.get('name', 'model'), pose_of(m), links, joints, directory)

    # ---- tree queries used by add_link_recursive (reference mjcf.py:603-644) ----------------------------
    def get_base_links(self) -> List[Link]:
        children = {j.child for j in self.joints}
        return [l for l in self.links if l.name not in children]

    def get_children(self, link: Link) -> List[Link]:
        names = [j.child for j in self.joints if j.parent == link.name]
        return [l for l in self.links if l.name in names]

    def get_parent_joint(self, link: Link) -> Optional[Joint]:
        for j in self.joints:
            if j.child == link.name:
                return j
        return None
