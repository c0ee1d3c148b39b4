"""MI355X-native offline path-tracing core (gfx950) — Python host bindings.

Product code only: host scene pipeline (host.py -> libpt_host.so), device library
(device.py -> libpt_hip.so), multi-GPU row sharding (distributed.py).  The CPU
oracle under oracle/ is test infrastructure and is never imported from here.
"""
from .ctypes_defs import *  # noqa: F401,F403
from .host import HostScene, camera_ray_data, read_pfm, write_image  # noqa: F401

__version__ = "0.1.0"
