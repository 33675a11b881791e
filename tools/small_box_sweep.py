#!/usr/bin/env python3
"""Launch-geometry knobs at small box sizes (round 6, item 2): ms per graph-replayed step and per-kernel hipEvent times of the
Sedov run at n^3 for each setting of the CASTRO_AMD_* knobs (read when a context is created, so every variant builds its own).
usage: python tools/small_box_sweep.py [numerics] n [n ...] [-- NAME=VALUE[,NAME=VALUE] ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
import castro_amd

args = sys.argv[1:]
numerics = args.pop(0) if args and args[0] in ("contract", "exact") else "contract"
variants = None
if "--" in args:
    k = args.index("--")
    variants = [dict(kv.split("=") for kv in v.split(",")) for v in args[k + 1:]]
    args = args[:k]
sizes = [tuple(int(x) for x in a.split("x")) * (1 if "x" in a else 3) for a in args] or [(128,) * 3, (64,) * 3]
if variants is None:
    variants = [{"CASTRO_AMD_SIDE_STREAM": "1"},
                {"CASTRO_AMD_FUSED_WG": "64"}, {"CASTRO_AMD_FUSED_WG": "256"},
                {"CASTRO_AMD_FUSED_TILE_ROWS": "0"}, {"CASTRO_AMD_FUSED_TILE_ROWS": "8"}, {"CASTRO_AMD_FUSED_TILE_ROWS": "32"},
                {"CASTRO_AMD_WG": "128"}, {"CASTRO_AMD_FINAL_WG": "128"},
                {"CASTRO_AMD_TILE_ROWS": "0"}, {"CASTRO_AMD_TILE_ROWS": "16"}, {"CASTRO_AMD_TILE_ROWS": "64"},
                {"CASTRO_AMD_TRACE_TILE_ROWS": "0"}, {"CASTRO_AMD_TRACE_TILE_ROWS": "32"}, {"CASTRO_AMD_TRACE_TILE_ROWS": "128"},
                {"CASTRO_AMD_FOLD_TILE_ROWS": "16"}, {"CASTRO_AMD_FOLD_TILE_ROWS": "64"},
                {"CASTRO_AMD_FOLD_TILE": "0"}, {"CASTRO_AMD_FOLD_TILE": "1"}, {"CASTRO_AMD_FOLD_TILE": "2"}]
variants = [{}] + variants + [{}]


def measure(n, env):
    keep = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        c = castro_amd.Castro(n, numerics=numerics)
        c.initData("sedov")
        c.run_steps(10)
        c.prepare_step_graph()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            c.run_steps(40)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
        c.hydro.profile(True)
        c.hydro.profile_reset()
        for _ in range(5):
            c.step()
        torch.cuda.synchronize()
        prof = {a: ms / 5 for a, (ms, cnt) in c.hydro.profile_report().items()}
        c.hydro.profile(False)
        c.close()
        c.hydro.close()
        return best, prof
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        # the knobs are process-wide and re-read (unset = default) by the next context creation


for n in sizes:
    for env in (variants if len(sizes) <= 2 else [{}]):
        ms, prof = measure(n, env)
        tag = ",".join("%s=%s" % (k.replace("CASTRO_AMD_", ""), v) for k, v in env.items()) or "default"
        short = {k.replace("k_", ""): round(v * 1e3) for k, v in sorted(prof.items())}
        print("[%s] n=%s %-24s %.4f ms/step = %.3f G zones/s   us: %s" % (numerics, "x".join(map(str, n)), tag, ms, n[0] * n[1] * n[2] / ms / 1e6, short), flush=True)
