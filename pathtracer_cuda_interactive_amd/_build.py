"""In-tree builds of the native libraries (no JIT cache: the .so files travel with the repo).

  libpt_host.so  — g++   : host scene pipeline (csrc/host/*.cpp)
  libpt_hip.so   — hipcc : C ABI + HIP kernels for gfx950 (csrc/*.hip)

`-ffp-contract=off` on BOTH is part of the numerical contract:
the device kernels must round exactly like the CPU oracle.
"""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(REPO_DIR, "include")

HOST_LIB = os.path.join(PKG_DIR, "libpt_host.so")
HIP_LIB = os.environ.get("PT_HIP_LIB", os.path.join(PKG_DIR, "libpt_hip.so"))   # override: A/B builds in tools/

HOST_FLAGS = ["-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-ffp-contract=off", "-pthread"]
HIP_FLAGS = [
    "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-shared",
    "-ffp-contract=off",                      # no FMA contraction: bit-parity with the oracle
    "-fno-gpu-flush-denormals-to-zero",       # keep fp32 denormals (x86 does)
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
    # SLP vectorisation turns the scalar fp32 vector math into v_pk_add/mul_f32 plus ~20 v_mov shuffles per leaf test;
    # packed fp32 issues at half rate on gfx950, so the packing only adds instructions: cbox 4.21 -> 3.78 ms without it
    "-fno-slp-vectorize",
    "-pthread",                               # pt_tree_sweep.h builds big trees on a few host threads
]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _glob(d, exts):
    out = []
    for root, _, files in os.walk(d):
        for f in sorted(files):
            if f.endswith(exts):
                out.append(os.path.join(root, f))
    return out


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def build_host(force=False):
    srcs = _glob(os.path.join(CSRC, "host"), (".cpp",))
    deps = srcs + _glob(os.path.join(CSRC, "host"), (".h",)) + _glob(INCLUDE, (".h",))
    if force or _newer(HOST_LIB, deps):
        _run(["g++"] + HOST_FLAGS + ["-I", INCLUDE, "-o", HOST_LIB] + srcs)
    return HOST_LIB


def build_hip(force=False, extra_flags=()):
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + _glob(INCLUDE, (".h",))
    if force or _newer(HIP_LIB, deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc] + HIP_FLAGS + list(extra_flags) + ["-I", INCLUDE, "-I", CSRC, "-o", HIP_LIB] + srcs)
    return HIP_LIB


EXAMPLE_BIN = os.path.join(REPO_DIR, "examples", "render_main")


def build_example(force=False):
    """C++ front-end on the bare C ABI (examples/render_main.cpp): proves the boundary needs no Python."""
    src = os.path.join(REPO_DIR, "examples", "render_main.cpp")
    if force or _newer(EXAMPLE_BIN, [src, HOST_LIB, HIP_LIB] + _glob(INCLUDE, (".h",))):
        _run(["g++", "-std=c++17", "-O2", "-Wall", "-I", INCLUDE, src, "-L", PKG_DIR, "-lpt_host", "-lpt_hip",
              "-Wl,-rpath," + PKG_DIR, "-Wl,-rpath,$ORIGIN/../pathtracer_cuda_interactive_amd", "-Wl,-rpath,/opt/rocm/lib",
              "-o", EXAMPLE_BIN])
    return EXAMPLE_BIN


INTERACTIVE_BIN = os.path.join(REPO_DIR, "examples", "interactive_main")


def build_interactive_example(force=False):
    """The reference's interactive accumulation loop on the C ABI + the HIP runtime (examples/interactive_main.cpp)."""
    src = os.path.join(REPO_DIR, "examples", "interactive_main.cpp")
    if force or _newer(INTERACTIVE_BIN, [src, HOST_LIB, HIP_LIB] + _glob(INCLUDE, (".h",))):
        _run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-std=c++17", "-O2", "-Wall", "-I", INCLUDE, src, "-L", PKG_DIR, "-lpt_host", "-lpt_hip",
              "-Wl,-rpath," + PKG_DIR, "-Wl,-rpath,$ORIGIN/../pathtracer_cuda_interactive_amd", "-Wl,-rpath,/opt/rocm/lib",
              "-o", INTERACTIVE_BIN])
    return INTERACTIVE_BIN


def build_all(force=False):
    out = build_host(force), build_hip(force)
    build_example(force)
    build_interactive_example(force)
    return out


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
