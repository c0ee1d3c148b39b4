"""Stand-in scenes for BASELINE.json configs 4 and 5.

The reference snapshot lacks `scenes/buddha/buddha.ply` and `scenes/dragon/dragon.ply`
(`.MISSING_LARGE_BLOBS`), so those configs cannot run on their original geometry.  The stand-ins keep
everything else of the two scene files (camera, sphere/ground/Cornell-shell layout, materials) and replace the
missing mesh by instances of the bunny mesh that IS present (72,378 vertices / 144,046 triangles):

  buddha_standin : 8 bunny instances (1,152,368 triangles ~ the buddha's 1,087,474) + the two mirror spheres and
                   the ground rectangle of buddha.xml:48-67, camera of buddha.xml:6-8.
  dragon_standin : the Cornell shell of dragon_1000.xml (mirror back wall, area light 4.157/1.7272/0.69076) with
                   two bunnies in place of the dragon, one Phong and one plastic (BASELINE config 5 asks for
                   "Phong/plastic Fresnel + Russian roulette"; no shipped scene uses plastic).

Every report that uses them must say "stand-in geometry" (bench.py does).
"""
import ctypes as C
import os

import numpy as np

from .ctypes_defs import PT_MAT_DIFFUSE, PT_MAT_MIRROR, PT_MAT_PHONG, PT_MAT_PLASTIC, PT_SHAPE_TRIANGLE
from .host import HostScene


def mesh_arrays(desc, k):
    """(positions [nv,3], indices [nf,3], normals [nv,3]) copies of mesh k of a finalized scene."""
    m = desc.meshes[k]
    P = np.ctypeslib.as_array(m.positions, shape=(m.num_vertices, 3)).copy()
    I = np.ctypeslib.as_array(m.indices, shape=(m.num_faces, 3)).copy()
    N = np.ctypeslib.as_array(m.normals, shape=(m.num_vertices, 3)).copy()
    return P, I, N


def _rot_y(deg):
    a = np.radians(deg)
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def _unit_bunny(scene_dir):
    """Bunny mesh recentred on its bounding-box centre, resting on y=0, scaled to unit height."""
    hs = HostScene.load(os.path.join(scene_dir, "bunny.pts"))
    d = hs.finalize()
    P, I, N = mesh_arrays(d, 0)
    lo, hi = P.min(0).astype(np.float64), P.max(0).astype(np.float64)
    centre = (lo + hi) / 2
    P64 = (P.astype(np.float64) - np.array([centre[0], lo[1], centre[2]])) / (hi[1] - lo[1])
    return P64, I, N.astype(np.float64)


def _place(P64, N64, scale, rot_deg, translate):
    R = _rot_y(rot_deg)
    P = (P64 * scale) @ R.T + np.asarray(translate, dtype=np.float64)
    N = N64 @ R.T
    N /= np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-30)
    return P.astype(np.float32), N.astype(np.float32)


def buddha_standin(scene_dir, instances=8):
    P64, I, N64 = _unit_bunny(scene_dir)
    hs = HostScene()
    hs.set_camera((0, 1.2, -1.5), (0, 0, 0), (0, 1, 0), 45.0, 1280, 960, 256)      # buddha.xml:6-8 (size/spp: BASELINE config 4)
    gold = hs.add_material(PT_MAT_DIFFUSE, (0.75, 0.75, 0.5))
    gray = hs.add_material(PT_MAT_DIFFUSE, (0.5, 0.5, 0.5))
    mirror = hs.add_material(PT_MAT_MIRROR, (1.0, 0.9, 0.9))
    hs.add_point_light((50, 50, 2), (10000, 7000, 5000))
    hs.add_point_light((-30, 20, 5), (100, 70, 50))
    for k in range(instances):                                   # a 4 x 2 cluster around the origin, where the buddha stood
        col, row = k % 4, k // 4
        pos = (-0.45 + 0.3 * col, -0.5 + 0.0, -0.15 + 0.45 * row)
        P, N = _place(P64, N64, 0.32 + 0.02 * (k % 3), 25.0 * k + 10.0, pos)
        hs.add_mesh(P, I, gold, normals=N)
    hs.add_sphere((0.7, -0.2, 0.0), 0.3, mirror)                 # buddha.xml:48-59
    hs.add_sphere((-0.7, -0.2, 0.0), 0.3, mirror)
    g = 2000.0                                                   # ground rectangle of buddha.xml:60-67 at y = -0.5
    hs.add_mesh(np.array([[-g, -0.5, g], [g, -0.5, g], [g, -0.5, -g], [-g, -0.5, -g]], np.float32),
                np.array([[0, 1, 2], [0, 2, 3]], np.int32), gray, normals=np.array([[0, 1, 0]] * 4, np.float32))
    return hs


def dragon_standin(scene_dir):
    P64, I, N64 = _unit_bunny(scene_dir)
    cb = HostScene.load(os.path.join(scene_dir, "cbox.pts"))
    cd = cb.finalize()
    hs = HostScene()
    hs.set_camera((278, 273, -800), (278, 273, -799), (0, 1, 0), 39.3077, 1920, 1080, 1024)   # dragon_1000.xml:6-8
    hs.set_background((0.5, 0.5, 0.5))
    white = hs.add_material(PT_MAT_DIFFUSE, (0.884774, 0.699933, 0.666224))
    red = hs.add_material(PT_MAT_DIFFUSE, (0.56581, 0.0447145, 0.0441583))
    green = hs.add_material(PT_MAT_DIFFUSE, (0.105092, 0.378697, 0.0762035))
    phong = hs.add_material(PT_MAT_PHONG, (0.75, 0.4, 0.4), exponent=40.0)        # "gold" of dragon_1000.xml:33-35 as Phong
    plastic = hs.add_material(PT_MAT_PLASTIC, (0.3, 0.45, 0.75), eta=1.5)
    mirror = hs.add_material(PT_MAT_MIRROR, (1.0, 1.0, 1.0))
    light = hs.add_material(PT_MAT_DIFFUSE, (0.78, 0.78, 0.78))
    # cbox.pts meshes: 0 luminaire, 1 floor, 2 ceiling, 3 back, 4 green wall, 5 red wall (6/7 = boxes, not used here)
    shell = {0: (light, (4.157, 1.7272, 0.69076)), 1: (white, None), 2: (white, None), 3: (mirror, None),
             4: (green, None), 5: (red, None)}
    for k, (mat, rad) in shell.items():
        P, Ik, N = mesh_arrays(cd, k)
        hs.add_mesh(P, Ik, mat, normals=N, radiance=rad)
    P, N = _place(P64, N64, 300.0, 40.0, (360.0, 0.0, 300.0))       # where the dragon stood (dragon_1000.xml:52-60)
    hs.add_mesh(P, I, phong, normals=N)
    P, N = _place(P64, N64, 200.0, -60.0, (160.0, 0.0, 180.0))
    hs.add_mesh(P, I, plastic, normals=N)
    return hs


BUILDERS = {"buddha_standin": buddha_standin, "dragon_standin": dragon_standin}
