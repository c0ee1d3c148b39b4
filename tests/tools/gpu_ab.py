"""A/B timing of differently-built libpt_hip.so files on the GPU box (build experiments that are compile-time switches).
Each build runs in its own child process (PT_HIP_LIB selects the library), builds alternate A B A B to cancel drift,
and every child first checks bit-exact parity against the oracle on a small frame.
Usage: python tests/tools/gpu_ab.py label=path/to/lib.so[:option=value,...] [label=path ...] -- scene[@spp] [scene[@spp] ...]"""
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CONFIGS = {"cbox": (640, 480, 64), "bunny": (640, 480, 64), "scene1": (640, 480, 16), "teapot": (640, 480, 16),
           "buddha_standin": (1280, 960, 16), "dragon_standin": (960, 540, 16), "scene4": (640, 480, 32),
           "scene1_phong": (640, 480, 16)}


def worker(scenes):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_binding as ob
    from pathtracer_cuda_interactive_amd import HostScene, PT_BVH_SORT_REFERENCE, standins
    from pathtracer_cuda_interactive_amd import device as dev
    sc = os.path.join(REPO, "tests", "golden", "scenes")
    out = {}
    for spec in scenes:
        name, _, spp = spec.partition("@")
        hs = standins.BUILDERS[name](sc) if name in standins.BUILDERS else HostScene.load(os.path.join(sc, name + ".pts"))
        d = hs.finalize(PT_BVH_SORT_REFERENCE)
        ds = dev.DeviceScene(d)
        for kv in filter(None, os.environ.get("PT_AB_OPTIONS", "").split(",")):
            k, v = kv.split("=")
            ds.set_option(k, int(v))
        small = hs.render_params(48, 32, 2, seed=5)
        want, _ = ob.render(d, small)
        ok = bool((ds.render(small).view(np.uint32) == want.view(np.uint32)).all())
        w, h, spp0 = CONFIGS[name]
        full = hs.render_params(w, h, int(spp) if spp else spp0)
        ts = []
        for _ in range(8 if spp else 6):
            ds.render(full)
            ts.append(ds.counters().kernel_ms)
        out[spec] = {"parity": ok, "median_ms": float(np.median(ts[1:])), "min_ms": float(min(ts[1:])), "vgprs": ds.info("vgprs"),
                     "occ": ds.info("occupancy"), "grid": ds.info("grid"), "lds": ds.info("lds_bytes")}
        ds.close()
    print("AB_RESULT " + json.dumps(out), flush=True)


def main():
    if sys.argv[1] == "--worker":
        return worker(sys.argv[2:])
    cut = sys.argv.index("--")
    libs = []
    opts = {}
    for a in sys.argv[1:cut]:
        label, rest = a.split("=", 1)
        path, _, o = rest.partition(":")
        libs.append((label, path))
        opts[label] = o
    scenes = sys.argv[cut + 1:]
    res = {label: [] for label, _ in libs}
    for rnd in range(int(os.environ.get("PT_AB_ROUNDS", "2"))):
        for label, path in libs:
            env = dict(os.environ, PT_HIP_LIB=os.path.abspath(path), PT_AB_OPTIONS=opts[label])
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"] + scenes, env=env, capture_output=True, text=True, timeout=900)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("AB_RESULT ")]
            if not line:
                print(f"{label}: FAILED\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}", flush=True)
                continue
            res[label].append(json.loads(line[0][10:]))
            print(f"round {rnd} {label}: " + "  ".join(f"{s} {v['median_ms']:.3f} ms" for s, v in res[label][-1].items()), flush=True)
    for s in scenes:
        for label, _ in libs:
            rs = [r[s] for r in res[label] if s in r]
            if rs:
                print(f"{s:15s} {label:18s} parity={'OK ' if all(r['parity'] for r in rs) else 'BAD'} vgpr={rs[0]['vgprs']:3d} occ={rs[0]['occ']} grid={rs[0]['grid']} lds={rs[0]['lds']} "
                      f"median {np.median([r['median_ms'] for r in rs]):9.3f} ms  min {min(r['min_ms'] for r in rs):9.3f} ms", flush=True)


if __name__ == "__main__":
    main()
