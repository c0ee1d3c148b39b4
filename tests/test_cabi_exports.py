"""The C-ABI libraries load without a GPU and export every symbol their headers declare."""
import ctypes
import os
import re

import pytest
from conftest import REPO, _gpu_available, load_scene

from pathtracer_cuda_interactive_amd import PT_ERR_NO_DEVICE, PtError, _build
from pathtracer_cuda_interactive_amd import device as dev

DECL = re.compile(r"^\s*(?:const\s+)?(?:int|void|char\s*\*|const char\s*\*)\s*\*?\s*(pt_[a-z_0-9]+)\s*\(", re.M)


def declared(header):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(DECL.findall(text)))


def test_hip_library_exports_every_declared_symbol():
    names = declared("pt_api.h")
    assert len(names) >= 12 and "pt_render" in names and "pt_scene_create" in names
    lib = ctypes.CDLL(_build.HIP_LIB)
    for n in names:
        assert hasattr(lib, n), f"libpt_hip.so does not export {n}"
    assert sorted(dev.EXPORTS) == names
    assert dev.lib().pt_api_version() == 1


def test_host_library_exports_every_declared_symbol(host_lib):
    names = declared("pt_host.h")
    assert len(names) >= 15
    for n in names:
        assert hasattr(host_lib, n), f"libpt_host.so does not export {n}"


def test_structs_have_the_documented_sizes():
    from pathtracer_cuda_interactive_amd import ctypes_defs as cd
    assert ctypes.sizeof(cd.PtShape) == 36 and ctypes.sizeof(cd.PtBvhNode) == 36
    assert ctypes.sizeof(cd.PtMaterial) == 24 and ctypes.sizeof(cd.PtLight) == 32
    assert ctypes.sizeof(cd.PtMesh) == 40 and ctypes.sizeof(cd.PtCamera) == 52
    assert ctypes.sizeof(cd.PtRenderParams) == 104 and ctypes.sizeof(cd.PtCounters) == 48


@pytest.mark.skipif(_gpu_available(), reason="only meaningful on a box without a GPU")
def test_no_gpu_means_a_loud_error_not_a_fallback():
    _, d = load_scene("cbox")
    with pytest.raises(PtError) as e:
        dev.DeviceScene(d)
    assert e.value.status == PT_ERR_NO_DEVICE
