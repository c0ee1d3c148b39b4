"""Generates tests/golden/scenes/*.pts from the reference's scene files (run in the build container only;
/root/reference does not exist on the GPU box).  A .pts holds DATA only: camera, materials, lights and
post-transform mesh arrays as parsed by OUR host pipeline (pathtracer_cuda_interactive_amd/csrc/host)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from pathtracer_cuda_interactive_amd import HostScene  # noqa: E402

REF = "/root/reference/scenes"
SCENES = {
    "scene1": "spheres/scene1.xml",
    "scene1_phong": "spheres/scene1_spherical_light_phong.xml",
    "scene4": "spheres/scene4.xml",
    "scene0": "spheres/scene0.xml",
    "scene0_spherical_light": "spheres/scene0_spherical_light.xml",
    "scene1_spherical_light": "spheres/scene1_spherical_light.xml",
    "scene2": "spheres/scene2.xml",
    "scene3": "spheres/scene3.xml",
    "aabb_test": "aabb_test/aabb_test.xml",
    "single_triangle": "triangles/single_triangle.xml",
    "cbox": "cbox/cbox.xml",
    "bunny": "bunny/bunny.xml",
    "teapot": "teapot/teapot_constant.xml",
    "tetrahedron": "triangles/tetrahedron.xml",
}

if __name__ == "__main__":
    out = os.path.join(HERE, "scenes")
    os.makedirs(out, exist_ok=True)
    for name, rel in SCENES.items():
        hs = HostScene.load(os.path.join(REF, rel))
        path = os.path.join(out, name + ".pts")
        hs.save_pts(path)
        d = hs.finalize()
        print(f"{name}: {d.num_shapes} shapes, {d.num_nodes} nodes, depth {hs.bvh_depth}, {os.path.getsize(path)} bytes")
