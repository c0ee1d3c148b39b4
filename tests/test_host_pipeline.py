"""Host producer side: XML/OBJ/PLY parsing, flattening (scene.cpp:11-153), BVH build (bvh.cu:16-54),
camera (camera.cuh:28-43), .pts container, error behaviour (status codes instead of exit())."""
import ctypes as C
import os

import numpy as np
import pytest
from conftest import DATA, assert_bit_equal, load_scene, random_scene

from pathtracer_cuda_interactive_amd import (PT_BVH_SORT_REFERENCE, PT_BVH_SORT_TOTAL, PT_ERR_BAD_SCENE, PT_ERR_IO,
                                             PT_ERR_PARSE, PT_ERR_UNSUPPORTED, PT_LIGHT_DIFFUSE_AREA, PT_LIGHT_POINT,
                                             PT_MAT_DIFFUSE, PT_MAT_MIRROR, PT_MAT_PHONG, PT_MAT_PLASTIC,
                                             PT_SHAPE_SPHERE, PT_SHAPE_TRIANGLE, HostScene, PtError, camera_ray_data)

REF_SCENES = "/root/reference/scenes"


def test_mixed_xml_scene_is_parsed_like_the_reference_parser_would():
    hs = HostScene.load(os.path.join(DATA, "mixed.xml"))
    d = hs.finalize()
    cam = hs.camera
    assert (cam.width, cam.height, cam.spp) == (40, 30, 7)                       # <default> + $name substitution
    # fovAxis=x, 40 deg horizontal -> vertical (parse_scene.cpp:364-368)
    want = np.degrees(2 * np.arctan(np.tan(np.radians(40.0) / 2) * 30 / 40.0))
    assert abs(cam.vfov - want) < 1e-4
    assert tuple(cam.lookfrom) == (0.0, 2.0, 6.0) and tuple(cam.lookat) == (0.0, 0.5, 0.0)
    # rectangle 2 + pyramid (quad base 2 + 4 sides) 6 + wedge 2 + ply 2 + sphere 1
    assert d.num_shapes == 13 and d.num_meshes == 4 and d.num_nodes == 25 and d.root == 24
    mats = [d.materials[i] for i in range(d.num_materials)]
    assert [m.type for m in mats] == [PT_MAT_DIFFUSE, PT_MAT_MIRROR, PT_MAT_PLASTIC, PT_MAT_PHONG, PT_MAT_DIFFUSE]
    assert tuple(mats[0].reflectance) == (np.float32(0.6),) * 3                  # single value broadcast
    assert abs(mats[2].eta - 1.7) < 1e-6 and abs(mats[3].exponent - 30) < 1e-6
    srgb = np.array([0x80, 0xff, 0x40]) / 255.0
    lin = np.where(srgb <= 0.04045, srgb / 12.92, ((srgb + 0.055) / 1.055) ** 2.4)
    np.testing.assert_allclose(mats[3].reflectance, lin, rtol=1e-6)              # sRGB -> linear
    # flattened lights: one entry per emissive primitive, then the point lights (scene.cpp:77-85,115-120)
    assert d.num_lights == 2
    assert d.lights[0].type == PT_LIGHT_DIFFUSE_AREA and d.lights[0].shape_id == 12
    assert tuple(d.lights[0].radiance) == (8.0, 7.0, 6.0)
    assert d.lights[1].type == PT_LIGHT_POINT and tuple(d.lights[1].position) == (1.0, 5.0, 2.0)
    sph = d.shapes[12]
    assert sph.type == PT_SHAPE_SPHERE and sph.area_light_id == 0 and sph.material_id == 4
    assert tuple(d.background) == (np.float32(0.1), np.float32(0.2), np.float32(0.3))
    # transforms: rectangle rotated to the ground plane and scaled by 5
    m0 = d.meshes[0]
    P = np.ctypeslib.as_array(m0.positions, shape=(m0.num_vertices, 3))
    N = np.ctypeslib.as_array(m0.normals, shape=(m0.num_vertices, 3))
    assert np.abs(P[:, 1]).max() < 1e-5 and np.abs(np.abs(P[:, [0, 2]]) - 5).max() < 1e-5
    np.testing.assert_allclose(N, np.tile([0, 1, 0], (4, 1)), atol=1e-6)
    # OBJ with scale+translate, quad fan (0,1,2),(0,2,3), computed normals
    m1 = d.meshes[1]
    assert (m1.num_vertices, m1.num_faces) == (5, 6)
    P1 = np.ctypeslib.as_array(m1.positions, shape=(5, 3))
    np.testing.assert_allclose(P1[4], [-1.5, 0.75, 0.0], atol=1e-6)
    I1 = np.ctypeslib.as_array(m1.indices, shape=(6, 3))
    assert I1[0].tolist() == [0, 1, 2] and I1[1].tolist() == [0, 2, 3]
    N1 = np.ctypeslib.as_array(m1.normals, shape=(5, 3))
    np.testing.assert_allclose(np.linalg.norm(N1, axis=1), 1.0, atol=1e-5)
    # OBJ with vn and negative indices; de-duplication is on the RAW (v,vt,vn) triple (parse_obj.cpp:46-64),
    # so -4//1 and 1//1 are different keys although they name the same vertex
    m2 = d.meshes[2]
    assert (m2.num_vertices, m2.num_faces) == (6, 2)
    P2 = np.ctypeslib.as_array(m2.positions, shape=(6, 3))
    assert (P2[3] == P2[0]).all() and P2[5].tolist() == [0.0, 1.0, 0.0]
    # PLY (ascii) with a <matrix> transform
    m3 = d.meshes[3]
    P3 = np.ctypeslib.as_array(m3.positions, shape=(4, 3))
    np.testing.assert_allclose(P3[2], [2.5, 1.0, 0.5], atol=1e-6)


def test_point_light_before_area_light_reproduces_reference_lookup_quirk(oracle):
    """radiance.cuh:36 indexes the FLATTENED light list with the PARSED light id."""
    hs = HostScene.load(os.path.join(DATA, "quirk_point_light_first.xml"))
    d = hs.finalize()
    assert d.shapes[0].area_light_id == 1 and d.lights[1].type == PT_LIGHT_POINT
    img, cnt = oracle.render(d, hs.render_params())
    assert cnt.emit == 0 and float(img.max()) == 0.0


@pytest.mark.parametrize("name", ["scene1", "cbox", "teapot", "tetrahedron"])
def test_pts_roundtrip_is_lossless(tmp_path, name):
    hs, d = load_scene(name)
    path = tmp_path / "copy.pts"
    hs.save_pts(path)
    hs2 = HostScene.load(path)
    d2 = hs2.finalize(PT_BVH_SORT_REFERENCE)
    assert d2.num_shapes == d.num_shapes and d2.num_nodes == d.num_nodes
    a, b = hs.nodes_array(), hs2.nodes_array()
    assert a.tobytes() == b.tobytes()
    assert bytes(C.string_at(d.materials, d.num_materials * 24)) == bytes(C.string_at(d2.materials, d2.num_materials * 24))


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="reference scene files are only present in the build container")
@pytest.mark.parametrize("name,rel", [("scene1", "spheres/scene1.xml"), ("cbox", "cbox/cbox.xml"),
                                      ("teapot", "teapot/teapot_constant.xml"), ("tetrahedron", "triangles/tetrahedron.xml")])
def test_fixture_equals_fresh_parse_of_reference_scene(name, rel):
    hs = HostScene.load(os.path.join(REF_SCENES, rel))
    hs.finalize(PT_BVH_SORT_REFERENCE)
    hs2, _ = load_scene(name)
    assert hs.nodes_array().tobytes() == hs2.nodes_array().tobytes()
    assert bytes(hs.camera) == bytes(hs2.camera)


def check_bvh(hs):
    d = hs.desc
    nodes = hs.nodes_array()
    n = d.num_shapes
    assert len(nodes) == 2 * n - 1 and d.root == len(nodes) - 1
    seen = np.zeros(n, dtype=int)
    for k, nd in enumerate(nodes):
        if nd["prim"] != -1:
            assert nd["left"] == -1 and nd["right"] == -1
            seen[nd["prim"]] += 1
        else:
            l, r = nodes[nd["left"]], nodes[nd["right"]]
            assert nd["left"] < k and nd["right"] < k                      # post-order pool (bvh.cu:49-53)
            assert (nd["bmin"] == np.minimum(l["bmin"], r["bmin"])).all()  # merge (bbox.cuh:104-114)
            assert (nd["bmax"] == np.maximum(l["bmax"], r["bmax"])).all()
    assert (seen == 1).all()
    # median split: subtree sizes differ by at most one
    size = np.zeros(len(nodes), dtype=int)
    for k, nd in enumerate(nodes):
        size[k] = 1 if nd["prim"] != -1 else size[nd["left"]] + size[nd["right"]]
        if nd["prim"] == -1:
            assert size[nd["right"]] - size[nd["left"]] in (0, 1)
    return nodes


@pytest.mark.parametrize("mode", [PT_BVH_SORT_TOTAL, PT_BVH_SORT_REFERENCE])
def test_bvh_invariants(mode):
    for seed in range(3):
        hs = random_scene(seed, n_tris=30 + seed, n_spheres=3)
        hs.finalize(mode)
        check_bvh(hs)
    hs, _ = load_scene("teapot", mode)
    check_bvh(hs)


def test_leaf_boxes_are_the_primitive_bounds():
    hs = random_scene(11, n_tris=12, n_spheres=2)
    d = hs.finalize()
    nodes = hs.nodes_array()
    for nd in nodes:
        if nd["prim"] == -1:
            continue
        s = d.shapes[nd["prim"]]
        if s.type == PT_SHAPE_SPHERE:
            c = np.array(s.center, dtype=np.float32)
            assert (nd["bmin"] == c - np.float32(s.radius)).all() and (nd["bmax"] == c + np.float32(s.radius)).all()
        else:
            assert s.type == PT_SHAPE_TRIANGLE
            m = d.meshes[s.mesh_index]
            P = np.ctypeslib.as_array(m.positions, shape=(m.num_vertices, 3))
            I = np.ctypeslib.as_array(m.indices, shape=(m.num_faces, 3))
            tri = P[I[s.face_index]]
            assert (nd["bmin"] == tri.min(0)).all() and (nd["bmax"] == tri.max(0)).all()


def test_sort_modes_give_the_same_image_when_tie_order_differs(oracle):
    """Tie-breaking changes the tree (cbox quads share centroids) but never the closest hit."""
    hs_r, d_r = load_scene("cbox", PT_BVH_SORT_REFERENCE)
    hs_t, d_t = load_scene("cbox", PT_BVH_SORT_TOTAL)
    assert hs_r.nodes_array().tobytes() != hs_t.nodes_array().tobytes()
    p = hs_r.render_params(40, 30, 3)
    a, _ = oracle.render(d_r, p)
    b, _ = oracle.render(d_t, p)
    assert_bit_equal(a, b, "cbox sort modes")


def test_camera_ray_data_formula():
    hs, _ = load_scene("cbox")
    cam = hs.camera
    got = camera_ray_data(cam, 640, 480)
    lookfrom, lookat, up = (np.array(v, dtype=np.float64) for v in (cam.lookfrom, cam.lookat, cam.up))
    h = 2.0 * np.tan(np.radians(cam.vfov / 2))
    w = 640 / 480 * h
    fwd = (lookat - lookfrom) / np.linalg.norm(lookat - lookfrom)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    nup = np.cross(right, fwd)
    np.testing.assert_allclose(got[0], lookfrom, rtol=1e-6)
    np.testing.assert_allclose(got[2], w * right, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got[3], h * nup, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got[1], lookfrom - w * right / 2 + h * nup / 2 + fwd, rtol=1e-5, atol=1e-4)
    p = hs.render_params(640, 480, 64)
    assert (p.width, p.height, p.spp, p.seed, p.max_depth, p.rr_depth) == (640, 480, 64, 1984, 50, 5)
    assert_bit_equal(np.array(p.cam_top_left), got[1])


def test_computed_normals_are_unit_and_face_outward_for_a_closed_mesh():
    hs = HostScene()
    P = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float32)
    I = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], dtype=np.int32)
    hs.add_mesh(P, I, hs.add_material(PT_MAT_DIFFUSE, (0.5, 0.5, 0.5)))
    d = hs.finalize()
    N = np.ctypeslib.as_array(d.meshes[0].normals, shape=(4, 3))
    np.testing.assert_allclose(np.linalg.norm(N, axis=1), 1.0, atol=1e-6)
    centre = P.mean(0)
    assert ((P - centre) * N).sum(1).min() > 0


# ---- error behaviour: status codes + message, never exit() (bbox.cuh:9-17 calls exit(99)) ----
def test_missing_file_is_io_error(tmp_path):
    with pytest.raises(PtError) as e:
        HostScene.load(tmp_path / "nope.xml")
    assert e.value.status == PT_ERR_IO
    with pytest.raises(PtError) as e:
        HostScene.load(tmp_path / "nope.pts")
    assert e.value.status == PT_ERR_IO


@pytest.mark.parametrize("text,status", [
    ("<scene><sensor type='perspective'></scene>", PT_ERR_PARSE),                      # mismatched tag
    ("<scene><shape type='cube'/></scene>", PT_ERR_PARSE),                              # unknown shape
    ("<scene><bsdf type='velvet' id='x'/></scene>", PT_ERR_PARSE),                      # unknown BSDF
    ("<scene><bsdf type='blinn' id='x'/></scene>", PT_ERR_UNSUPPORTED),                 # dropped by the reference (H5d)
    ("<scene><shape type='sphere'><ref id='ghost'/></shape></scene>", PT_ERR_PARSE),    # unknown material ref
    ("<scene><sensor type='orthographic'/></scene>", PT_ERR_UNSUPPORTED),
    ("<scene><shape type='sphere'><float name='radius' value='$r'/></shape></scene>", PT_ERR_PARSE),   # undefined default
    ("<notscene/>", PT_ERR_PARSE),
    ("<scene><shape type='obj'><string name='filename' value='missing.obj'/></shape></scene>", PT_ERR_IO),
])
def test_malformed_scenes_report_status(tmp_path, text, status):
    f = tmp_path / "bad.xml"
    f.write_text(text)
    with pytest.raises(PtError) as e:
        HostScene.load(f)
    assert e.value.status == status, str(e.value)
    assert len(str(e.value)) > 10


def test_scene_without_shapes_or_material_is_rejected():
    hs = HostScene()
    with pytest.raises(PtError) as e:
        hs.finalize()
    assert e.value.status == PT_ERR_BAD_SCENE
    hs.add_sphere((0, 0, 0), 1.0, material_id=3)            # no such material
    with pytest.raises(PtError) as e:
        hs.finalize()
    assert e.value.status == PT_ERR_BAD_SCENE
    with pytest.raises(PtError):
        hs.add_mesh(np.zeros((3, 3), np.float32), np.array([[0, 1, 5]], np.int32), 0)   # index out of range


def test_corrupt_pts_is_rejected(tmp_path):
    f = tmp_path / "x.pts"
    f.write_bytes(b"PTSCENE1" + b"\x01\x00\x00\x00" + b"\x00" * 10)
    with pytest.raises(PtError) as e:
        HostScene.load(f)
    assert e.value.status == PT_ERR_IO          # truncated
    f.write_bytes(b"NOTASCENE" * 4)
    with pytest.raises(PtError) as e:
        HostScene.load(f)
    assert e.value.status == PT_ERR_PARSE


def test_image_writers_roundtrip(tmp_path):
    from pathtracer_cuda_interactive_amd import read_pfm, write_image
    rng = np.random.default_rng(0)
    img = (rng.random((5, 7, 3)) * 2).astype(np.float32)
    img[0, 0] = (0.25, 4.0, -1.0)
    write_image(tmp_path / "a.pfm", img)
    assert_bit_equal(read_pfm(tmp_path / "a.pfm"), img, "pfm")
    write_image(tmp_path / "a.ppm", img)
    raw = (tmp_path / "a.ppm").read_bytes()
    assert raw.startswith(b"P6\n7 5\n255\n")
    px = np.frombuffer(raw[len(b"P6\n7 5\n255\n"):], np.uint8).reshape(5, 7, 3)
    assert px[0, 0].tolist() == [127, 255, 0]           # sqrt(0.25)=0.5 -> int(255.99*0.5)=127; clamp; negative -> NaN -> 0
    with pytest.raises(PtError):
        write_image(tmp_path / "no_such_dir" / "a.pfm", img)
