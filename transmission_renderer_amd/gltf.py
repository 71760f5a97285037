"""glTF 2.0 import into the reference's model buffers — the host-side mirror of src/model_loading.rs `load_gltf`.

What the reference does per file (src/model_loading.rs:12-333), and this module does the same way:
  * every node with a mesh contributes, per primitive, one PrimitiveInfo and one Instance whose transform is
    base_transform * (accumulated node TRS), a Similarity (uniform scale asserted, :478-487);
  * draw buffer from (alphaMode, has KHR_materials_transmission): 0 opaque, 1 mask, 2 transmission, 3 both (:64-76);
  * uv are pre-multiplied by the KHR_texture_transform *scale* of the base colour texture only (:85-92);
  * indices widened to u32 and rebased onto the shared vertex arrays; missing TEXCOORD_0 -> zeros (:124-137);
  * bounding sphere from the POSITION accessor's min/max box (:146-153);
  * materials -> MaterialInfo with the KHR_materials_{ior,transmission,volume,specular} extensions, defaults of
    :293-332 (attenuation distance pre-multiplied by base_transform.scale, roughness_override);
  * images are uploaded once per (image, sRGB?) pair: base colour / emissive / specular colour as sRGB, the rest
    linear, the specular (alpha) texture re-using an sRGB copy when one exists (:166-222).
The reference leans on the un-vendored `gltf` crate (fork @0324938) for parsing; here the JSON / GLB container, the
accessors and 8-bit RGB(A) PNG decoding are read directly; JPEG and the other PNG variants go through Pillow.

`write_gltf` is the inverse for test assets (there is no network for the Khronos sample models).
"""
from __future__ import annotations

import base64
import json
import os
import struct
import zlib
from typing import List, Optional, Tuple

import numpy as np

from . import wire
from .meshes import ModelBuffers, Similarity, quat_mul  # noqa: F401  (re-exported for callers)
from .png import read_png_rgba8_bytes

f32 = np.float32

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NUM = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class GltfError(ValueError):
    pass


class Document:
    """The parsed container: JSON + resolved buffers."""

    def __init__(self, path: str):
        self.dir = os.path.dirname(os.path.abspath(path))
        raw = open(path, "rb").read()
        glb_bin = None
        if raw[:4] == b"glTF":
            version, length = struct.unpack_from("<II", raw, 4)
            if version != 2:
                raise GltfError(f"GLB version {version}")
            off = 12
            chunks = []
            while off < length:
                clen, ctype = struct.unpack_from("<II", raw, off)
                chunks.append((ctype, raw[off + 8:off + 8 + clen]))
                off += 8 + clen
            self.json = json.loads(chunks[0][1].decode("utf-8"))
            glb_bin = next((c for t, c in chunks if t == 0x004E4942), None)
        else:
            self.json = json.loads(raw.decode("utf-8"))
        self.buffers = []
        for i, b in enumerate(self.json.get("buffers", [])):
            uri = b.get("uri")
            if uri is None:
                if glb_bin is None:
                    raise GltfError("buffer without uri outside a GLB")
                self.buffers.append(glb_bin)
            else:
                self.buffers.append(self._read_uri(uri))

    def _read_uri(self, uri: str) -> bytes:
        if uri.startswith("data:"):
            return base64.b64decode(uri.split(",", 1)[1])
        return open(os.path.join(self.dir, uri.replace("%20", " ")), "rb").read()

    def view_bytes(self, index: int) -> Tuple[bytes, int]:
        v = self.json["bufferViews"][index]
        start = v.get("byteOffset", 0)
        return self.buffers[v["buffer"]][start:start + v["byteLength"]], v.get("byteStride", 0)

    def accessor(self, index: int) -> np.ndarray:
        """(count, components) array in the accessor's component type (no normalisation applied)."""
        a = self.json["accessors"][index]
        dt = np.dtype(_COMPONENT[a["componentType"]]).newbyteorder("<")
        n = _NUM[a["type"]]
        count = a["count"]
        if "bufferView" not in a:
            dense = np.zeros((count, n), dtype=dt)
        else:
            dense = self._strided(a["bufferView"], a.get("byteOffset", 0), dt, n, count)
        if "sparse" not in a:
            return dense
        # glTF 2.0 5.1.3: `sparse.count` elements of the (dense or all-zero) array are replaced; the gltf crate's
        # accessor iterators (which the reference reads every attribute through, src/model_loading.rs:29, 96-137) apply it
        sp = a["sparse"]
        k = int(sp["count"])
        idt = np.dtype(_COMPONENT[sp["indices"]["componentType"]]).newbyteorder("<")
        where = self._strided(sp["indices"]["bufferView"], sp["indices"].get("byteOffset", 0), idt, 1, k).reshape(-1).astype(np.int64)
        values = self._strided(sp["values"]["bufferView"], sp["values"].get("byteOffset", 0), dt, n, k)
        if k and (where.min() < 0 or where.max() >= count or np.any(np.diff(where) <= 0)):
            raise GltfError(f"accessor {index}: sparse indices must be strictly increasing and inside the accessor")
        out = np.array(dense, copy=True)
        out[where] = values
        return out

    def _strided(self, view: int, off: int, dt: np.dtype, n: int, count: int) -> np.ndarray:
        data, stride = self.view_bytes(view)
        elem = dt.itemsize * n
        if stride in (0, elem):
            return np.frombuffer(data, dtype=dt, count=count * n, offset=off).reshape(count, n)
        buf = np.frombuffer(data, dtype=np.uint8)
        rows = np.lib.stride_tricks.as_strided(buf[off:], shape=(count, elem), strides=(stride, 1))
        return np.ascontiguousarray(rows).view(dt).reshape(count, n)

    def accessor_f32(self, index: int) -> np.ndarray:
        """`into_f32()` of the gltf crate: normalised integers are divided by their maximum."""
        a = self.json["accessors"][index]
        v = self.accessor(index)
        if a["componentType"] == 5126:
            return v.astype(f32)
        if not a.get("normalized", False) and a["componentType"] not in (5121, 5123):
            return v.astype(f32)
        return (v.astype(f32) / f32(np.iinfo(v.dtype).max)).astype(f32)

    def image_rgba8(self, index: int) -> np.ndarray:
        img = self.json["images"][index]
        if "uri" in img:
            data = self._read_uri(img["uri"])
        else:
            data, _ = self.view_bytes(img["bufferView"])
        return decode_image_rgba8(data, f"image {index}")


def decode_image_rgba8(data: bytes, what: str = "image") -> np.ndarray:
    """Encoded image bytes -> (H, W, 4) uint8, the way `gltf::import` + src/model_loading.rs:36-52, 343-351 treat them:
    RGB is widened to RGBA (alpha 255), RGBA is kept; anything else (grey, grey+alpha, 16-bit) makes the reference
    panic ("unsupported format") and is refused here.  8-bit non-interlaced RGB(A) PNGs are decoded in-module;
    every other container / variant (JPEG, palette or interlaced PNG, ...) goes through Pillow when it is installed."""
    if data[:8] == b"\x89PNG\r\n\x1a\n":
        try:
            from .png import decode_png
            img = decode_png(data, what)
        except ValueError:
            img = None                     # a PNG variant the small decoder does not read
        if img is not None:
            if img.shape[2] in (1, 2):
                raise GltfError(f"{what}: grey / grey+alpha images are an unsupported format in the reference "
                                "(src/model_loading.rs:348-351 panics)")
            return read_png_rgba8_bytes(data)
    try:
        from PIL import Image
    except ImportError as e:               # pragma: no cover - Pillow is present in the build image
        raise GltfError(f"{what}: only 8-bit RGB(A) PNGs can be decoded without Pillow") from e
    import io
    with Image.open(io.BytesIO(data)) as im:
        im.load()
        if im.mode == "P":
            im = im.convert("RGBA" if "transparency" in im.info else "RGB")
        elif im.mode in ("CMYK", "YCbCr"):
            im = im.convert("RGB")
        if im.mode == "RGB":
            rgb = np.asarray(im, dtype=np.uint8)
            out = np.full(rgb.shape[:2] + (4,), 255, dtype=np.uint8)
            out[..., :3] = rgb
            return out
        if im.mode == "RGBA":
            return np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
        raise GltfError(f"{what}: image mode {im.mode} is an unsupported format in the reference "
                        "(src/model_loading.rs:348-351 panics on anything but 8-bit RGB / RGBA)")


def _node_similarity(node: dict) -> Similarity:
    """node.transform().decomposed() (:475-493); uniform scale asserted like the reference."""
    if "matrix" in node:
        m = np.array(node["matrix"], dtype=np.float64).reshape(4, 4).T       # column-major -> [row][col]
        t = m[:3, 3]
        basis = m[:3, :3]
        s = np.linalg.norm(basis, axis=0)
        if np.linalg.det(basis) < 0:
            s[0] = -s[0]
        rot = basis / s
        q = _quat_from_matrix(rot)
        scale = s
    else:
        t = np.array(node.get("translation", [0, 0, 0]), dtype=np.float64)
        q = np.array(node.get("rotation", [0, 0, 0, 1]), dtype=np.float64)
        scale = np.array(node.get("scale", [1, 1, 1]), dtype=np.float64)
    eps = float(np.finfo(np.float32).eps) * 10.0
    if abs(f32(scale[0]) - f32(scale[1])) > eps or abs(f32(scale[0]) - f32(scale[2])) > eps:
        raise GltfError(f"non-uniform node scale {scale.tolist()} (the reference asserts, src/model_loading.rs:478-487)")
    return Similarity(t.astype(f32), float(f32(scale[0])), q.astype(f32))


def _quat_from_matrix(r: np.ndarray) -> np.ndarray:
    tr = r[0, 0] + r[1, 1] + r[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        return np.array([(r[2, 1] - r[1, 2]) / s, (r[0, 2] - r[2, 0]) / s, (r[1, 0] - r[0, 1]) / s, 0.25 * s])
    i = int(np.argmax([r[0, 0], r[1, 1], r[2, 2]]))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(1.0 + r[i, i] - r[j, j] - r[k, k]) * 2
    q = np.zeros(4)
    q[i] = 0.25 * s
    q[j] = (r[j, i] + r[i, j]) / s
    q[k] = (r[k, i] + r[i, k]) / s
    q[3] = (r[k, j] - r[j, k]) / s
    return q


class NodeTree:
    """src/model_loading.rs:466-512."""

    def __init__(self, nodes: list):
        self.inner = [[Similarity(), None] for _ in nodes]
        for i, node in enumerate(nodes):
            self.inner[i][0] = _node_similarity(node)
            for c in node.get("children", []):
                self.inner[c][1] = i

    def transform_of(self, index: Optional[int]) -> Similarity:
        total = Similarity()
        while index is not None:
            t, parent = self.inner[index]
            total = t * total
            index = parent
        return total


class Scene:
    """Accumulates what `load_gltf` appends to across calls: model buffers, materials, images."""

    def __init__(self):
        self.buffers = ModelBuffers()
        self.materials: List[wire.MaterialInfo] = []
        self.textures: List[Tuple[np.ndarray, bool]] = []      # (rgba8 image, srgb), the bindless array
        self.max_draw_counts = [0, 0, 0, 0]

    def geometry(self) -> dict:
        return self.buffers.finish()


def load_gltf(path: str, scene: Optional[Scene] = None, base_transform: Optional[Similarity] = None,
              roughness_override: Optional[float] = None) -> Scene:
    scene = scene or Scene()
    base = base_transform or Similarity()
    doc = Document(path)
    j = doc.json
    nodes = j.get("nodes", [])
    tree = NodeTree(nodes)
    materials = j.get("materials", [])
    mb = scene.buffers

    for node_index, node in enumerate(nodes):
        if "mesh" not in node:
            continue
        transform = base * tree.transform_of(node_index)
        for prim in j["meshes"][node["mesh"]]["primitives"]:
            if prim.get("mode", 4) != 4:
                raise GltfError("only TRIANGLES primitives are drawn (src/pipelines.rs:312)")
            mat_index = prim.get("material")
            mat = materials[mat_index] if mat_index is not None else {}
            ext = mat.get("extensions", {})
            alpha_mode = mat.get("alphaMode", "OPAQUE")
            transmissive = "KHR_materials_transmission" in ext
            draw_buffer_index = {("OPAQUE", False): 0, ("MASK", False): 1, ("OPAQUE", True): 2, ("MASK", True): 3}.get(
                (alpha_mode, transmissive), 0)                      # BLEND falls back to 0 like the `dbg!` arm
            scene.max_draw_counts[draw_buffer_index] += 1
            uv_scaling = np.ones(2, f32)
            bct = mat.get("pbrMetallicRoughness", {}).get("baseColorTexture")
            if bct and "KHR_texture_transform" in bct.get("extensions", {}):
                uv_scaling = np.array(bct["extensions"]["KHR_texture_transform"].get("scale", [1, 1]), dtype=f32)
            material_id = (mat_index if mat_index is not None else 0) + len(scene.materials)
            attrs = prim["attributes"]
            if "indices" not in prim or "POSITION" not in attrs or "NORMAL" not in attrs:
                raise GltfError("primitive without indices / POSITION / NORMAL (the reference unwraps these)")
            index = doc.accessor(prim["indices"]).reshape(-1).astype(np.uint32)
            position = doc.accessor_f32(attrs["POSITION"])
            normal = doc.accessor_f32(attrs["NORMAL"])
            if "TEXCOORD_0" in attrs:
                uv = (doc.accessor_f32(attrs["TEXCOORD_0"]) * uv_scaling).astype(f32)
            else:
                uv = np.zeros((len(position), 2), f32)
            pa = j["accessors"][attrs["POSITION"]]
            if "min" in pa and "max" in pa:
                mn, mx = np.array(pa["min"], f32), np.array(pa["max"], f32)
            else:
                mn, mx = position.min(axis=0), position.max(axis=0)
            from .meshes import Mesh
            mb.add_primitive(Mesh(position, normal, uv, index), draw_buffer_index, [(transform, material_id)],
                             bbox=(mn, mx))

    image_index_to_id = {}

    def load_optional_texture(info: Optional[dict], requirement: str) -> int:
        if info is None:
            return -1
        image_index = j["textures"][info["index"]]["source"]
        if requirement == "dont_care":
            if (image_index, True) in image_index_to_id:
                return image_index_to_id[(image_index, True)]
            srgb = False
        else:
            srgb = requirement == "srgb"
        key = (image_index, srgb)
        if key not in image_index_to_id:
            image_index_to_id[key] = len(scene.textures)
            scene.textures.append((doc.image_rgba8(image_index), srgb))
        return image_index_to_id[key]

    for mat in materials:
        pbr = mat.get("pbrMetallicRoughness", {})
        ext = mat.get("extensions", {})
        transmission = ext.get("KHR_materials_transmission")
        volume = ext.get("KHR_materials_volume")
        specular = ext.get("KHR_materials_specular")
        m = wire.MaterialInfo.default()
        t = m.textures
        t.diffuse = load_optional_texture(pbr.get("baseColorTexture"), "srgb")
        t.metallic_roughness = load_optional_texture(pbr.get("metallicRoughnessTexture"), "linear")
        t.normal_map = load_optional_texture(mat.get("normalTexture"), "linear")
        t.emissive = load_optional_texture(mat.get("emissiveTexture"), "srgb")
        t.occlusion = load_optional_texture(mat.get("occlusionTexture"), "linear")
        t.transmission = load_optional_texture((transmission or {}).get("transmissionTexture"), "linear")
        t.thickness = load_optional_texture((volume or {}).get("thicknessTexture"), "linear")
        t.specular_colour = load_optional_texture((specular or {}).get("specularColorTexture"), "srgb")
        t.specular = load_optional_texture((specular or {}).get("specularTexture"), "dont_care")
        m.metallic_factor = pbr.get("metallicFactor", 1.0)
        m.roughness_factor = roughness_override if roughness_override is not None else pbr.get("roughnessFactor", 1.0)
        m.alpha_clipping_cutoff = mat.get("alphaCutoff", 0.5)
        m.diffuse_factor = (wire.C.c_float * 4)(*pbr.get("baseColorFactor", [1, 1, 1, 1]))
        m.emissive_factor = (wire.C.c_float * 3)(*mat.get("emissiveFactor", [0, 0, 0]))
        m.normal_map_scale = mat["normalTexture"].get("scale", 1.0) if "normalTexture" in mat else 0.0
        m.occlusion_strength = mat["occlusionTexture"].get("strength", 1.0) if "occlusionTexture" in mat else 1.0
        m.index_of_refraction = ext.get("KHR_materials_ior", {}).get("ior", 1.5)
        m.transmission_factor = transmission.get("transmissionFactor", 0.0) if transmission is not None else 0.0
        m.thickness_factor = volume.get("thicknessFactor", 0.0) if volume is not None else 0.0
        if volume is not None:
            m.attenuation_distance = float(f32(volume.get("attenuationDistance", float("inf"))) * f32(base.scale))
            m.attenuation_colour = (wire.C.c_float * 3)(*volume.get("attenuationColor", [1, 1, 1]))
        else:
            m.attenuation_distance = float("inf")
            m.attenuation_colour = (wire.C.c_float * 3)(1, 1, 1)
        m.specular_factor = specular.get("specularFactor", 1.0) if specular is not None else 1.0
        m.specular_colour_factor = (wire.C.c_float * 3)(*(specular.get("specularColorFactor", [1, 1, 1]) if specular is not None
                                                          else [1, 1, 1]))
        scene.materials.append(m)
    return scene


# --------------------------------------------------------------------------- test-asset writer

def _png_bytes(img: np.ndarray) -> bytes:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, {3: 2, 4: 6}[c], 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_gltf(path: str, nodes: list, meshes_: list, materials: list, images: list, textures: Optional[list] = None,
               binary: bool = True, index_type=np.uint16) -> None:
    """Writes a .glb (or .gltf with embedded data URIs).
    nodes: glTF node dicts (TRS / children / mesh);  meshes_: [[(Mesh, material index or None), ...], ...];
    materials: glTF material dicts;  images: [(H, W, 3|4) uint8 arrays, written as PNG, or (encoded bytes, mime type)];
    textures: [image index] (default: 1:1)."""
    blob = bytearray()
    views, accessors = [], []

    def add(arr: np.ndarray, target: Optional[int], typ: str, ctype: int, minmax: bool = False) -> int:
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": arr.nbytes, **({"target": target} if target else {})})
        blob.extend(arr.tobytes())
        acc = {"bufferView": len(views) - 1, "componentType": ctype, "count": len(arr), "type": typ}
        if minmax:
            acc["min"] = [float(x) for x in arr.min(axis=0)]
            acc["max"] = [float(x) for x in arr.max(axis=0)]
        accessors.append(acc)
        return len(accessors) - 1

    jmeshes = []
    for prims in meshes_:
        jp = []
        for mesh, mat in prims:
            idx = mesh.index.astype(index_type)
            p = {"attributes": {"POSITION": add(mesh.position.astype(f32), 34962, "VEC3", 5126, True),
                                "NORMAL": add(mesh.normal.astype(f32), 34962, "VEC3", 5126)},
                 "indices": add(idx, 34963, "SCALAR", {np.uint8: 5121, np.uint16: 5123, np.uint32: 5125}[index_type])}
            if mesh.uv is not None:
                p["attributes"]["TEXCOORD_0"] = add(mesh.uv.astype(f32), 34962, "VEC2", 5126)
            if mat is not None:
                p["material"] = mat
            jp.append(p)
        jmeshes.append({"primitives": jp})
    jimages = []
    for img in images:
        mime = "image/png"
        if isinstance(img, tuple):          # (encoded bytes, mime type): stored as they are
            data, mime = img
        else:
            data = _png_bytes(img)
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)})
        blob.extend(data)
        jimages.append({"bufferView": len(views) - 1, "mimeType": mime})
    used = sorted({e for m in materials for e in m.get("extensions", {})} |
                  {"KHR_texture_transform" for m in materials
                   if "KHR_texture_transform" in m.get("pbrMetallicRoughness", {}).get("baseColorTexture", {}).get("extensions", {})})
    doc = {"asset": {"version": "2.0", "generator": "transmission_renderer_amd.gltf.write_gltf"},
           "scene": 0, "scenes": [{"nodes": [i for i in range(len(nodes)) if not any(i in n.get("children", []) for n in nodes)]}],
           "nodes": nodes, "meshes": jmeshes, "materials": materials, "accessors": accessors, "bufferViews": views,
           "buffers": [{"byteLength": len(blob)}]}
    if images:
        doc["images"] = jimages
        doc["textures"] = [{"source": s} for s in (textures if textures is not None else range(len(images)))]
        doc["samplers"] = []
    if used:
        doc["extensionsUsed"] = used
    if binary:
        js = json.dumps(doc, separators=(",", ":")).encode()
        js += b" " * (-len(js) % 4)
        while len(blob) % 4:
            blob.append(0)
        total = 12 + 8 + len(js) + 8 + len(blob)
        with open(path, "wb") as f:
            f.write(b"glTF" + struct.pack("<II", 2, total))
            f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
            f.write(struct.pack("<II", len(blob), 0x004E4942) + bytes(blob))
    else:
        doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode()
        with open(path, "w") as f:
            json.dump(doc, f)
