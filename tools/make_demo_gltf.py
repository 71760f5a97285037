#!/usr/bin/env python3
"""Writes a small glTF 2.0 test asset (a glass sphere and a tinted slab over a textured floor, an alpha-masked panel)
with transmission_renderer_amd.gltf.write_gltf: there is no network for the Khronos sample models.
    python tools/make_demo_gltf.py out.glb"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmission_renderer_amd import gltf, meshes, synthetic


def main(path):
    tex = synthetic.make_textures()
    images = [tex[0][0], tex[1][0], tex[2][0]]
    materials = [
        {"name": "floor", "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicFactor": 0.0, "roughnessFactor": 0.7}},
        {"name": "glass", "pbrMetallicRoughness": {"baseColorFactor": [0.95, 0.98, 1.0, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.08},
         "extensions": {"KHR_materials_transmission": {"transmissionFactor": 1.0}, "KHR_materials_ior": {"ior": 1.5},
                        "KHR_materials_volume": {"thicknessFactor": 1.0}}},
        {"name": "amber", "pbrMetallicRoughness": {"baseColorFactor": [1.0, 1.0, 1.0, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.3},
         "extensions": {"KHR_materials_transmission": {"transmissionFactor": 0.9}, "KHR_materials_ior": {"ior": 1.4},
                        "KHR_materials_volume": {"thicknessFactor": 0.4, "attenuationDistance": 0.3, "attenuationColor": [0.9, 0.45, 0.1]}}},
        {"name": "metal", "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.7, 0.3, 1.0], "metallicFactor": 1.0, "roughnessFactor": 0.35,
                                                   "metallicRoughnessTexture": {"index": 1}}, "normalTexture": {"index": 2}},
        {"name": "mask", "alphaMode": "MASK", "alphaCutoff": 0.75, "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}},
    ]
    q = meshes.quat_from_axis_angle
    nodes = [
        {"name": "floor", "mesh": 0, "translation": [0.0, -1.4, -3.0]},
        {"name": "glass sphere", "mesh": 1, "translation": [0.1, -0.3, -2.0], "scale": [0.6, 0.6, 0.6]},
        {"name": "slab", "mesh": 2, "translation": [1.3, -1.0, -2.3], "rotation": [float(x) for x in q([0, 1, 0], -0.5)]},
        {"name": "metal sphere", "mesh": 3, "translation": [-1.0, -0.5, -2.8], "scale": [0.55, 0.55, 0.55]},
        {"name": "box", "mesh": 4, "translation": [0.4, -0.9, -4.3], "rotation": [float(x) for x in q([0.3, 1, 0.1], 0.9)]},
        {"name": "panel", "mesh": 5, "translation": [-0.3, -0.5, -3.2], "rotation": [float(x) for x in q([1, 0, 0], 1.2)]},
    ]
    mesh_list = [[(meshes.plane(9.0, 9.0, cells=4, uv_repeat=6.0), 0)], [(meshes.uv_sphere(1.0, 64, 32), 1)],
                 [(meshes.box(0.7, 0.45, 0.08), 2)], [(meshes.uv_sphere(1.0, 48, 24), 3)], [(meshes.box(0.45, 0.45, 0.45), 0)],
                 [(meshes.plane(1.6, 1.6, cells=1), 4)]]
    gltf.write_gltf(path, nodes, mesh_list, materials, images, binary=path.endswith(".glb"))
    print("wrote", path)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "demo.glb")
