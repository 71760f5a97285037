"""glTF import (host side): the loader mirrors src/model_loading.rs; the reference holds no test
assets or fixtures for it and there is no network for the Khronos sample models, so the asset is written by
gltf.write_gltf and the expectations are the reference's rules restated (file:line in each check)."""
import numpy as np
import pytest

from transmission_renderer_amd import gltf, meshes, wire

f32 = np.float32


def _asset(tmp_path, binary=True, index_type=np.uint16):
    rng = np.random.default_rng(5)
    images = [rng.integers(0, 256, (16, 8, 4), dtype=np.uint8), rng.integers(0, 256, (4, 4, 3), dtype=np.uint8),
              rng.integers(0, 256, (8, 8, 4), dtype=np.uint8)]
    materials = [
        {"name": "plain", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.1, 0.2, 1.0], "metallicFactor": 0.0,
                                                   "roughnessFactor": 0.4}},
        {"name": "glass", "pbrMetallicRoughness": {"baseColorTexture": {"index": 0, "extensions": {"KHR_texture_transform": {"scale": [2.0, 3.0]}}},
                                                   "metallicRoughnessTexture": {"index": 1}},
         "normalTexture": {"index": 1, "scale": 0.7}, "emissiveTexture": {"index": 0}, "emissiveFactor": [1.0, 0.5, 0.25],
         "extensions": {"KHR_materials_transmission": {"transmissionFactor": 0.9, "transmissionTexture": {"index": 2}},
                        "KHR_materials_volume": {"thicknessFactor": 0.3, "attenuationDistance": 2.0, "attenuationColor": [0.9, 0.6, 0.3],
                                                 "thicknessTexture": {"index": 2}},
                        "KHR_materials_ior": {"ior": 1.33},
                        "KHR_materials_specular": {"specularFactor": 0.5, "specularColorFactor": [1.0, 0.9, 0.8],
                                                   "specularTexture": {"index": 0}, "specularColorTexture": {"index": 2}}}},
        {"name": "mask", "alphaMode": "MASK", "alphaCutoff": 0.3, "pbrMetallicRoughness": {"baseColorTexture": {"index": 2}}},
        {"name": "mask-glass", "alphaMode": "MASK", "extensions": {"KHR_materials_transmission": {}}},
    ]
    sphere, cube, quad = meshes.uv_sphere(1.0, 8, 4), meshes.box(0.5, 0.25, 1.0), meshes.plane(2.0, 2.0)
    no_uv = meshes.Mesh(cube.position, cube.normal, None, cube.index)
    mesh_list = [[(sphere, 1), (cube, 0)], [(quad, 2)], [(no_uv, None), (quad, 3)]]
    q = [float(x) for x in meshes.quat_from_axis_angle([0, 1, 0], 0.5)]
    nodes = [
        {"name": "root", "translation": [1.0, 2.0, 3.0], "rotation": q, "scale": [2.0, 2.0, 2.0], "children": [1, 2]},
        {"name": "child-a", "mesh": 0, "translation": [0.0, 1.0, 0.0]},
        {"name": "child-b", "mesh": 1, "matrix": [0.5, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0.5, 0, 4.0, 5.0, 6.0, 1.0], "children": [3]},
        {"name": "grandchild", "mesh": 2, "rotation": [float(x) for x in meshes.quat_from_axis_angle([1, 0, 0], -0.25)]},
        {"name": "empty"},
    ]
    path = str(tmp_path / ("scene.glb" if binary else "scene.gltf"))
    gltf.write_gltf(path, nodes, mesh_list, materials, images, binary=binary, index_type=index_type)
    return path, images, (sphere, cube, quad), q


@pytest.mark.parametrize("binary,index_type", [(True, np.uint16), (False, np.uint32), (True, np.uint8)])
def test_load_gltf_mirrors_model_loading(tmp_path, binary, index_type):
    path, images, (sphere, cube, quad), q = _asset(tmp_path, binary, index_type)
    base = meshes.Similarity(np.array([0.0, 2.0, 0.0], f32), 1.5)      # src/main.rs:364-368 with --scale 1.5
    scene = gltf.load_gltf(path, base_transform=base, roughness_override=None)
    geo = scene.geometry()
    prims, insts = geo["primitives"], geo["instances"]
    # one PrimitiveInfo + one Instance per (node, primitive), in node order (:57-161)
    assert len(prims) == len(insts) == 5
    np.testing.assert_array_equal(insts["primitive_id"], np.arange(5))
    np.testing.assert_array_equal(prims["first_instance"], np.arange(5))
    # draw buffers (:64-76): glass -> 2, plain -> 0, mask -> 1, no material -> 0, mask + transmission -> 3
    np.testing.assert_array_equal(prims["draw_buffer_index"], [2, 0, 1, 0, 3])
    assert scene.max_draw_counts == [2, 1, 1, 1]
    # material ids: material.index().unwrap_or(0) + materials.len() (:94)
    np.testing.assert_array_equal(insts["material_id"], [1, 0, 2, 0, 3])
    # shared buffers: indices rebased on the running vertex count (:100-107)
    counts = [len(sphere.position), len(cube.position), len(quad.position), len(cube.position), len(quad.position)]
    starts = np.cumsum([0] + counts[:-1])
    for k, (mesh, start) in enumerate(zip((sphere, cube, quad, cube, quad), starts)):
        fi, ic = int(prims["first_index"][k]), int(prims["index_count"][k])
        np.testing.assert_array_equal(geo["index"][fi:fi + ic], mesh.index + start)
        np.testing.assert_array_equal(geo["position"][start:start + counts[k]], mesh.position)
        np.testing.assert_array_equal(geo["normal"][start:start + counts[k]], mesh.normal)
    # uv: base colour KHR_texture_transform scale applied on the glass primitive only (:85-92, 127); zeros when absent (:131-135)
    np.testing.assert_array_equal(geo["uv"][:counts[0]], (sphere.uv * np.array([2.0, 3.0], f32)).astype(f32))
    np.testing.assert_array_equal(geo["uv"][starts[1]:starts[1] + counts[1]], cube.uv)
    assert (geo["uv"][starts[3]:starts[3] + counts[3]] == 0).all()
    # bounding spheres from the accessor boxes (:146-153)
    for k, mesh in enumerate((sphere, cube, quad, cube, quad)):
        mn, mx = mesh.position.min(axis=0), mesh.position.max(axis=0)
        np.testing.assert_allclose(prims["packed_bounding_sphere"][k, :3], (mn + mx) / 2, atol=1e-7)
        np.testing.assert_allclose(prims["packed_bounding_sphere"][k, 3], np.linalg.norm(mx - mn) / 2, rtol=1e-6)
    # transforms: base * root * child... as Similarity products (:57, 495-512; shared-structs:221-236)
    S = meshes.Similarity
    root = S(np.array([1, 2, 3], f32), 2.0, np.array(q, f32))
    child_a = S(np.array([0, 1, 0], f32))
    child_b = S(np.array([4, 5, 6], f32), 0.5)
    grand = S(np.zeros(3, f32), 1.0, meshes.quat_from_axis_angle([1, 0, 0], -0.25))
    expect = [base * (root * child_a)] * 2 + [base * (root * child_b)] + [base * (root * (child_b * grand))] * 2
    for k, e in enumerate(expect):
        np.testing.assert_allclose(insts["translation_and_scale"][k, :3], e.translation, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(insts["translation_and_scale"][k, 3], e.scale, rtol=1e-6)
        got_q, want_q = insts["rotation"][k], np.asarray(e.rotation)
        assert min(np.abs(got_q - want_q).max(), np.abs(got_q + want_q).max()) < 1e-6
    # materials (:224-330)
    m = scene.materials
    assert len(m) == 4
    assert list(m[0].diffuse_factor) == pytest.approx([0.8, 0.1, 0.2, 1.0]) and m[0].roughness_factor == pytest.approx(0.4)
    assert m[0].textures.diffuse == -1 and m[0].index_of_refraction == 1.5 and m[0].attenuation_distance == float("inf")
    assert m[0].transmission_factor == 0.0 and m[0].specular_factor == 1.0 and m[0].alpha_clipping_cutoff == 0.5
    g = m[1]
    assert g.index_of_refraction == pytest.approx(1.33) and g.transmission_factor == pytest.approx(0.9)
    assert g.thickness_factor == pytest.approx(0.3) and g.attenuation_distance == pytest.approx(2.0 * 1.5)   # x base scale (:317)
    assert list(g.attenuation_colour) == pytest.approx([0.9, 0.6, 0.3]) and g.specular_factor == pytest.approx(0.5)
    assert list(g.emissive_factor) == pytest.approx([1.0, 0.5, 0.25]) and g.normal_map_scale == pytest.approx(0.7)
    assert m[2].alpha_clipping_cutoff == pytest.approx(0.3)
    # images: one upload per (image, sRGB?) (:166-222): texture ids in first-use order
    t = g.textures
    tex = scene.textures
    assert (t.diffuse, t.metallic_roughness, t.normal_map, t.emissive, t.occlusion) == (0, 1, 1, 0, -1)
    assert tex[t.diffuse][1] is True and tex[t.metallic_roughness][1] is False
    assert tex[t.transmission][1] is False and t.thickness == t.transmission                 # image 2, linear, shared
    assert tex[t.specular_colour][1] is True and t.specular_colour != t.transmission         # image 2 again, as sRGB
    assert t.specular == t.diffuse                                                           # DontCare re-uses the sRGB copy
    assert m[2].textures.diffuse == t.specular_colour                                        # image 2 sRGB, cached
    np.testing.assert_array_equal(tex[t.diffuse][0], images[0])
    rgb = tex[t.metallic_roughness][0]
    np.testing.assert_array_equal(rgb[..., :3], images[1])
    assert (rgb[..., 3] == 255).all()                                                         # RGB widened (:36-52)
    # roughness_override (:294) and accumulation into an existing scene (second load_gltf call, src/main.rs:342-370)
    again = gltf.load_gltf(path, scene=scene, roughness_override=0.25)
    assert len(again.materials) == 8 and all(x.roughness_factor == 0.25 for x in again.materials[4:])
    geo2 = again.geometry()
    assert len(geo2["primitives"]) == 10 and geo2["instances"]["material_id"][5] == 1 + 4
    assert geo2["index"][int(geo2["primitives"]["first_index"][5])] >= sum(counts)


def test_loader_refuses_what_the_reference_cannot_draw(tmp_path):
    path, *_ = _asset(tmp_path)
    import json, struct
    raw = open(path, "rb").read()
    jlen = struct.unpack_from("<I", raw, 12)[0]
    doc = json.loads(raw[20:20 + jlen])
    doc["nodes"][0]["scale"] = [1.0, 2.0, 1.0]
    bad = str(tmp_path / "bad.gltf")
    rest = raw[20 + jlen + 8:]
    import base64
    doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(rest).decode()
    json.dump(doc, open(bad, "w"))
    with pytest.raises(gltf.GltfError):          # the reference asserts on non-uniform scale (:478-487)
        gltf.load_gltf(bad)


def test_jpeg_palette_and_grey_images(tmp_path):
    """Texture containers beyond 8-bit RGB(A) PNG: JPEG and palette PNG decode (RGB widened to RGBA like
    src/model_loading.rs:36-52); grey images are refused because the reference panics on them (:348-351)."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    import io
    rng = np.random.default_rng(2)
    base = (np.linspace(0, 255, 32 * 24 * 3).reshape(24, 32, 3) + rng.integers(0, 20, (24, 32, 3))).clip(0, 255).astype(np.uint8)
    def encoded(img, fmt, **kw):
        buf = io.BytesIO()
        img.save(buf, format=fmt, **kw)
        return buf.getvalue()
    jpeg = encoded(Image.fromarray(base), "JPEG", quality=90)
    pal = encoded(Image.fromarray(base).quantize(16), "PNG")
    grey = encoded(Image.fromarray(base[..., 0]), "PNG")
    got = gltf.decode_image_rgba8(jpeg)
    want = np.asarray(Image.open(io.BytesIO(jpeg)).convert("RGB"))
    assert got.shape == (24, 32, 4) and (got[..., 3] == 255).all()
    np.testing.assert_array_equal(got[..., :3], want)
    assert np.abs(got[..., :3].astype(int) - base.astype(int)).mean() < 6          # it is the picture, lossy
    got = gltf.decode_image_rgba8(pal)
    np.testing.assert_array_equal(got[..., :3], np.asarray(Image.open(io.BytesIO(pal)).convert("RGB")))
    with pytest.raises(gltf.GltfError):
        gltf.decode_image_rgba8(grey)
    # through a whole file: a JPEG base colour texture
    quad = meshes.plane(1.0, 1.0)
    path = str(tmp_path / "jpeg.glb")
    gltf.write_gltf(path, [{"mesh": 0}], [[(quad, 0)]],
                    [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}], [(jpeg, "image/jpeg")])
    scene = gltf.load_gltf(path)
    assert scene.materials[0].textures.diffuse == 0 and scene.textures[0][1] is True
    np.testing.assert_array_equal(scene.textures[0][0][..., :3], want)


def test_hand_written_adversarial_gltf(tmp_path):
    """A .gltf written by hand, none of it through this repo's write_gltf: interleaved vertex data (byteStride), UNSIGNED_BYTE
    indices, a `matrix` node under a TRS parent, a KHR_texture_transform whose offset and rotation must be IGNORED (the
    reference multiplies uv by the scale only, src/model_loading.rs:85-92), a normalised UNSIGNED_SHORT uv set, a sparse
    POSITION accessor (read like the gltf crate's iterators do), and a second primitive without TEXCOORD_0."""
    import base64
    import json
    f32 = np.float32
    # four vertices, position + normal interleaved (stride 24) — a unit quad in the xy plane, normals +z
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], f32)
    nrm = np.tile(np.array([[0, 0, 1]], f32), (4, 1))
    inter = np.concatenate([pos, nrm], axis=1).astype("<f4").tobytes()                    # 96 bytes
    uv16 = np.array([[0, 0], [65535, 0], [65535, 65535], [0, 65535]], "<u2").tobytes()     # 16 bytes, normalised
    idx8 = np.array([0, 1, 2, 0, 2, 3], np.uint8).tobytes() + b"\0\0"                      # 6 bytes (+ 2 padding)
    sparse_idx = np.array([1, 2], "<u2").tobytes()                                         # 4 bytes: vertices 1 and 2 move
    sparse_val = np.array([[2, 0, 0.5], [2, 1, 0.5]], "<f4").tobytes()                     # 24 bytes
    blob = inter + uv16 + idx8 + sparse_idx + sparse_val
    views = [
        {"buffer": 0, "byteOffset": 0, "byteLength": 96, "byteStride": 24},       # 0 interleaved position | normal
        {"buffer": 0, "byteOffset": 96, "byteLength": 16},                         # 1 uv (u16 normalised)
        {"buffer": 0, "byteOffset": 112, "byteLength": 6},                         # 2 indices (u8)
        {"buffer": 0, "byteOffset": 120, "byteLength": 4},                         # 3 sparse indices
        {"buffer": 0, "byteOffset": 124, "byteLength": 24},                        # 4 sparse values
    ]
    accessors = [
        {"bufferView": 0, "byteOffset": 0, "componentType": 5126, "count": 4, "type": "VEC3", "min": [0, 0, 0], "max": [2, 1, 0.5],
         "sparse": {"count": 2, "indices": {"bufferView": 3, "componentType": 5123}, "values": {"bufferView": 4}}},
        {"bufferView": 0, "byteOffset": 12, "componentType": 5126, "count": 4, "type": "VEC3"},
        {"bufferView": 1, "componentType": 5123, "normalized": True, "count": 4, "type": "VEC2"},
        {"bufferView": 2, "componentType": 5121, "count": 6, "type": "SCALAR"},
        {"bufferView": 0, "byteOffset": 0, "componentType": 5126, "count": 4, "type": "VEC3"},      # the dense positions
    ]
    half = [0.5, 0, 0, 0,  0, 0.5, 0, 0,  0, 0, 0.5, 0,  3, 4, 5, 1]                               # column-major: scale 0.5, then translate
    doc = {
        "asset": {"version": "2.0"},
        "extensionsUsed": ["KHR_texture_transform", "KHR_materials_transmission", "KHR_materials_volume", "KHR_materials_ior"],
        "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
        "bufferViews": views, "accessors": accessors,
        "images": [{"uri": "data:image/png;base64," + base64.b64encode(_png_2x2()).decode()}],
        "textures": [{"source": 0}],
        "materials": [
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 0, "extensions": {"KHR_texture_transform": {
                "offset": [0.25, 0.75], "rotation": 1.0, "scale": [2.0, 3.0]}}}, "metallicFactor": 0.0}},
            {"alphaMode": "MASK", "alphaCutoff": 0.3,
             "extensions": {"KHR_materials_transmission": {"transmissionFactor": 0.75}, "KHR_materials_ior": {"ior": 1.33},
                            "KHR_materials_volume": {"thicknessFactor": 0.2, "attenuationDistance": 4.0, "attenuationColor": [0.9, 0.8, 0.7]}}},
        ],
        "meshes": [{"primitives": [
            {"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0},
            {"attributes": {"POSITION": 4, "NORMAL": 1}, "indices": 3, "material": 1}]}],
        "nodes": [{"translation": [10, 0, 0], "scale": [2, 2, 2], "children": [1]}, {"matrix": half, "mesh": 0}],
        "scenes": [{"nodes": [0]}], "scene": 0,
    }
    path = str(tmp_path / "adversarial.gltf")
    json.dump(doc, open(path, "w"))
    sc = gltf.load_gltf(path, base_transform=gltf.Similarity(np.array([0, 2, 0], f32), 1.5))
    geo = sc.geometry()
    # vertices: primitive 0 with the sparse substitution applied, primitive 1 the dense ones (no uv: zeros)
    want0 = pos.copy()
    want0[1], want0[2] = [2, 0, 0.5], [2, 1, 0.5]
    np.testing.assert_array_equal(geo["position"][:4], want0)
    np.testing.assert_array_equal(geo["position"][4:8], pos)
    np.testing.assert_array_equal(geo["normal"][:8], np.tile(nrm, (2, 1)))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], f32) * np.array([2.0, 3.0], f32)           # the scale only: no offset, no rotation
    np.testing.assert_array_equal(geo["uv"][:4], uv)
    np.testing.assert_array_equal(geo["uv"][4:8], np.zeros((4, 2), f32))
    assert geo["index"].dtype == np.uint32
    np.testing.assert_array_equal(geo["index"][:6], [0, 1, 2, 0, 2, 3])
    # draw buffers: opaque textured -> 0, MASK + transmission -> 3
    assert list(geo["primitives"]["draw_buffer_index"]) == [0, 3] and sc.max_draw_counts == [1, 0, 0, 1]
    # transform: base (t = (0,2,0), s = 1.5) * parent (t = (10,0,0), s = 2) * matrix node (s = 0.5, t = (3,4,5))
    inst = geo["instances"][0]
    assert abs(float(inst["translation_and_scale"][3]) - 1.5) < 1e-6
    np.testing.assert_allclose(inst["translation_and_scale"][:3], [0 + 1.5 * (10 + 2 * 3), 2 + 1.5 * (2 * 4), 1.5 * (2 * 5)], rtol=1e-6)
    np.testing.assert_allclose(inst["rotation"], [0, 0, 0, 1], atol=1e-7)
    # materials: the extensions' factors, attenuation distance pre-multiplied by the base scale (src/model_loading.rs:317)
    m = sc.materials[1]
    assert abs(m.index_of_refraction - 1.33) < 1e-6 and abs(m.transmission_factor - 0.75) < 1e-6 and abs(m.alpha_clipping_cutoff - 0.3) < 1e-6
    assert abs(m.thickness_factor - 0.2) < 1e-6 and abs(m.attenuation_distance - 4.0 * 1.5) < 1e-5
    assert sc.materials[0].textures.diffuse == 0 and sc.textures[0][1] is True and sc.textures[0][0].shape == (2, 2, 4)
    # a sparse accessor whose indices are not increasing is refused
    doc["accessors"][0]["sparse"]["indices"]["bufferView"] = 3
    bad = np.array([2, 1], "<u2").tobytes()
    blob2 = blob[:120] + bad + blob[124:]
    doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(blob2).decode()
    json.dump(doc, open(path, "w"))
    with pytest.raises(gltf.GltfError):
        gltf.load_gltf(path)


def _png_2x2() -> bytes:
    import struct
    import zlib
    rows = b"".join(b"\0" + bytes([r, g, 0, 255, g, r, 0, 255]) for r, g in ((255, 0), (0, 255)))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 2, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(rows)) + chunk(b"IEND", b"")
