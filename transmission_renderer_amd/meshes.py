"""Procedural geometry in the reference's model-buffer layout (src/main.rs `ModelStagingBuffers`, filled by
src/model_loading.rs:96-161): shared position / normal / uv / index arrays, one PrimitiveInfo per drawable
primitive, and Instance records carrying a PackedSimilarity.  Used by the synthetic glTF writer, the culling and
rasteriser tests, and `cli.py --scene`.

Winding: counter-clockwise seen from outside (glTF 2.0), like the assets the reference loads.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

from . import wire

f32 = np.float32


@dataclass
class Mesh:
    position: np.ndarray   # (N, 3) float32
    normal: np.ndarray     # (N, 3) float32
    uv: np.ndarray         # (N, 2) float32
    index: np.ndarray      # (M,)  uint32, triangle list


def uv_sphere(radius=1.0, segments=24, rings=12) -> Mesh:
    pos, nrm, uv = [], [], []
    for r in range(rings + 1):
        phi = np.pi * r / rings
        for s in range(segments + 1):
            th = 2 * np.pi * s / segments
            n = np.array([np.sin(phi) * np.cos(th), np.cos(phi), np.sin(phi) * np.sin(th)])
            pos.append(n * radius)
            nrm.append(n)
            uv.append([s / segments, r / rings])
    idx = []
    w = segments + 1
    for r in range(rings):
        for s in range(segments):
            a, b, c, d = r * w + s, r * w + s + 1, (r + 1) * w + s, (r + 1) * w + s + 1
            if r != 0:
                idx += [a, b, c]
            if r != rings - 1:
                idx += [b, d, c]
    return Mesh(np.array(pos, f32), np.array(nrm, f32), np.array(uv, f32), np.array(idx, np.uint32))


def box(hx=0.5, hy=0.5, hz=0.5) -> Mesh:
    pos, nrm, uv, idx = [], [], [], []
    for axis in range(3):
        for sign in (-1.0, 1.0):
            n = np.zeros(3)
            n[axis] = sign
            u = np.zeros(3)
            v = np.zeros(3)
            u[(axis + 1) % 3] = 1.0
            v[(axis + 2) % 3] = 1.0
            if sign < 0:
                u, v = v, u          # keep u x v = n
            base = len(pos)
            for (a, b) in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                p = (n + a * u + b * v) * np.array([hx, hy, hz])
                pos.append(p)
                nrm.append(n)
                uv.append([(a + 1) / 2, (b + 1) / 2])
            idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    return Mesh(np.array(pos, f32), np.array(nrm, f32), np.array(uv, f32), np.array(idx, np.uint32))


def plane(size_x=1.0, size_z=1.0, cells=1, uv_repeat=1.0) -> Mesh:
    """Horizontal quad grid facing +y."""
    pos, nrm, uv, idx = [], [], [], []
    n = cells + 1
    for j in range(n):
        for i in range(n):
            pos.append([(i / cells - 0.5) * size_x, 0.0, (j / cells - 0.5) * size_z])
            nrm.append([0.0, 1.0, 0.0])
            uv.append([i / cells * uv_repeat, j / cells * uv_repeat])
    for j in range(cells):
        for i in range(cells):
            a, b, c, d = j * n + i, j * n + i + 1, (j + 1) * n + i, (j + 1) * n + i + 1
            idx += [a, c, b, b, c, d]
    return Mesh(np.array(pos, f32), np.array(nrm, f32), np.array(uv, f32), np.array(idx, np.uint32))


# ---- Similarity (shared-structs/src/lib.rs:196-236), fp32 like glam

def quat_from_axis_angle(axis, angle) -> np.ndarray:
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    s = np.sin(angle / 2)
    return np.array([axis[0] * s, axis[1] * s, axis[2] * s, np.cos(angle / 2)], dtype=f32)


def quat_mul(a, b) -> np.ndarray:
    """glam Quat * Quat (x, y, z, w)."""
    ax, ay, az, aw = [f32(x) for x in a]
    bx, by, bz, bw = [f32(x) for x in b]
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz], dtype=f32)


def quat_rotate(q, v) -> np.ndarray:
    """glam 0.19 scalar Quat * Vec3."""
    q = np.asarray(q, dtype=f32)
    v = np.asarray(v, dtype=f32)
    b = q[:3]
    w = q[3]
    b2 = f32(np.dot(b, b))
    return (v * f32(w * w - b2) + b * f32(f32(np.dot(v, b)) * f32(2.0)) + np.cross(b, v).astype(f32) * f32(w * f32(2.0))).astype(f32)


@dataclass
class Similarity:
    translation: np.ndarray = field(default_factory=lambda: np.zeros(3, f32))
    scale: float = 1.0
    rotation: np.ndarray = field(default_factory=lambda: np.array([0, 0, 0, 1], f32))

    def apply(self, v) -> np.ndarray:
        """Mul<Vec3>: translation + scale * (rotation * v)."""
        return (np.asarray(self.translation, f32) + f32(self.scale) * quat_rotate(self.rotation, v)).astype(f32)

    def __mul__(self, child: "Similarity") -> "Similarity":
        """Mul<Similarity> (:221-231)."""
        return Similarity(self.apply(child.translation), float(f32(self.scale) * f32(child.scale)),
                          quat_mul(self.rotation, child.rotation))


class ModelBuffers:
    """The accumulating model buffers of src/model_loading.rs (`ModelStagingBuffers`)."""

    def __init__(self):
        self.position: List[np.ndarray] = []
        self.normal: List[np.ndarray] = []
        self.uv: List[np.ndarray] = []
        self.index: List[np.ndarray] = []
        self.primitives: List[tuple] = []
        self.instances: List[tuple] = []
        self._num_vertices = 0
        self._num_indices = 0

    def add_primitive(self, mesh: Mesh, draw_buffer_index: int, instances: Sequence[tuple], bbox=None) -> int:
        """One PrimitiveInfo + its instances ([(Similarity, material_id), ...], contiguous from first_instance).
        Bounding sphere from the bounding box (`bbox` = (min, max), default: of the positions), like
        src/model_loading.rs:146-153."""
        first_index = self._num_indices
        self.index.append(mesh.index.astype(np.uint32) + np.uint32(self._num_vertices))
        self.position.append(mesh.position.astype(f32))
        self.normal.append(mesh.normal.astype(f32))
        self.uv.append(mesh.uv.astype(f32))
        self._num_vertices += len(mesh.position)
        self._num_indices += len(mesh.index)
        if bbox is None:
            bbox = (mesh.position.min(axis=0), mesh.position.max(axis=0))
        mn, mx = np.asarray(bbox[0], f32), np.asarray(bbox[1], f32)
        center = ((mn + mx) / f32(2.0)).astype(f32)
        d = (mn - mx).astype(f32)
        radius = f32(f32(np.sqrt(f32(f32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))) / f32(2.0))
        prim_id = len(self.primitives)
        self.primitives.append(((center[0], center[1], center[2], radius), draw_buffer_index, len(mesh.index), first_index,
                                len(self.instances)))
        for sim, material_id in instances:
            t = np.asarray(sim.translation, f32)
            self.instances.append(((t[0], t[1], t[2], f32(sim.scale)), tuple(np.asarray(sim.rotation, f32)), prim_id,
                                   material_id, (0, 0)))
        return prim_id

    def finish(self) -> dict:
        return {
            "position": np.concatenate(self.position).astype(f32) if self.position else np.zeros((0, 3), f32),
            "normal": np.concatenate(self.normal).astype(f32) if self.normal else np.zeros((0, 3), f32),
            "uv": np.concatenate(self.uv).astype(f32) if self.uv else np.zeros((0, 2), f32),
            "index": np.concatenate(self.index).astype(np.uint32) if self.index else np.zeros((0,), np.uint32),
            "primitives": np.array(self.primitives, dtype=wire.PRIMITIVE_DTYPE),
            "instances": np.array(self.instances, dtype=wire.INSTANCE_DTYPE),
        }


def mesh_scene_parts(extra_instances: bool = True, room=False) -> list:
    """The primitives of make_mesh_scene as (mesh, draw_buffer_index, instances) in model-buffer order."""
    S = Similarity
    q = quat_from_axis_angle
    parts = []
    if room:
        # A closed room around the scene, loaded FIRST like the reference's Sponza backdrop (src/main.rs:342-351), and three
        # partitions between its back wall and the objects, far to near: a pixel of the objects' region is covered by the
        # back wall, up to three partitions and the object in front — depth complexity 3 to 5, later draws nearer (the order
        # in which every fragment survives the depth comparison: the rasteriser's worst case, readme.md:74).
        wall = plane(1.0, 1.0, cells=2, uv_repeat=4.0)
        up, half = np.array([1.0, 0.0, 0.0], f32), np.pi / 2
        parts.append((wall, 0, [(S(np.array([0.0, 3.0, -9.0], f32), 24.0, q(up, half)), 9),                        # back wall, facing +z
                                (S(np.array([0.0, 9.0, -3.0], f32), 24.0, q(up, np.pi)), 11),                      # ceiling, facing down
                                (S(np.array([-9.0, 3.0, -3.0], f32), 24.0, q([0.0, 0.0, 1.0], -half)), 12),        # left wall, facing +x
                                (S(np.array([9.0, 3.0, -3.0], f32), 24.0, q([0.0, 0.0, 1.0], half)), 13),          # right wall, facing -x
                                (S(np.array([0.0, 3.0, 7.0], f32), 24.0, q(up, -half)), 9)]))                      # behind the camera (culled)
        parts.append((wall, 0, [(S(np.array([0.0, 2.6, -7.5], f32), 9.0, q(up, half)), 5),
                                (S(np.array([0.4, 2.2, -6.5], f32), 7.0, q(up, half)), 15),
                                (S(np.array([-0.3, 1.9, -5.6], f32), 5.0, q(up, half)), 0)]))
    parts.append((plane(8.0, 8.0, cells=4, uv_repeat=4.0), 0, [(S(np.array([0, 0.6, -3.0], f32)), 3)]))
    sphere = uv_sphere(1.0, 20, 10)
    inst = [(S(np.array([-0.9, 1.6, -2.6], f32), 0.55), 1), (S(np.array([1.1, 1.4, -3.4], f32), 0.7, q([0, 1, 0], 0.7)), 6)]
    if extra_instances:   # culled ones: far left of the frustum, behind the camera
        inst += [(S(np.array([-30.0, 1.0, -3.0], f32), 0.5), 1), (S(np.array([0.0, 3.0, 6.0], f32), 0.8), 6)]
    parts.append((sphere, 0, inst))
    parts.append((box(0.5, 0.5, 0.5), 0, [(S(np.array([0.2, 1.1, -4.2], f32), 0.9, q([0.3, 1, 0.1], 0.9)), 8)]))
    # transmissive: a sphere in front of the box and the far sphere, a slab intersecting the floor
    tinst = [(S(np.array([0.15, 1.7, -1.9], f32), 0.6), 4), (S(np.array([-1.6, 1.2, -3.9], f32), 0.45, q([1, 0, 0], 0.4)), 10)]
    if extra_instances:
        tinst += [(S(np.array([0.0, 40.0, -3.0], f32), 0.5), 4)]    # above the frustum
    parts.append((uv_sphere(1.0, 24, 12), 2, tinst))
    parts.append((box(0.7, 0.4, 0.08), 2, [(S(np.array([1.3, 1.0, -2.2], f32), 1.0, q([0, 1, 0], -0.5)), 14)]))
    # alpha clipped (needs a textured material; ids chosen by the caller's material table)
    parts.append((plane(1.6, 1.6, cells=1, uv_repeat=1.0), 1, [(S(np.array([-0.2, 1.5, -3.0], f32), 1.0, q([1, 0, 0], 1.2)), 2)]))
    parts.append((plane(1.2, 1.2, cells=1, uv_repeat=2.0), 3, [(S(np.array([0.9, 2.2, -2.8], f32), 1.0, q([1, 0, 0.2], 1.35)), 7)]))
    if extra_instances:   # a primitive whose only instance is culled: no draw at all
        parts.append((box(0.3, 0.3, 0.3), 0, [(S(np.array([50.0, 0.0, -3.0], f32), 1.0), 5)]))
    if room == 2:   # the same geometry drawn NEAR TO FAR (objects, partitions, walls): the order a depth-sorting host submits
        parts = [(m, d, list(reversed(i))) for m, d, i in reversed(parts)]
    return parts


def make_mesh_scene(extra_instances: bool = True, room=False) -> dict:
    """A small scene in front of the default camera (eye (0,3,1) looking down -z, pitched -15 deg): a floor, opaque
    and transmissive spheres and boxes (one transmissive object in front of opaque ones, one behind), an
    alpha-clipped quad, and objects outside the frustum / behind the camera for the culling pass.
    Material ids refer to synthetic.make_materials() (16 entries); draw buffers: 0 opaque, 1 alpha clip,
    2 transmission, 3 transmission + alpha clip.  room=True: the same objects inside a closed room with partitions behind
    them (mesh_scene_parts): every pixel covered, depth complexity 3 to 5 over most of the frame."""
    mb = ModelBuffers()
    for mesh, draw_buffer, instances in mesh_scene_parts(extra_instances, room):
        mb.add_primitive(mesh, draw_buffer, instances)
    return mb.finish()
