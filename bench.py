#!/usr/bin/env python3
"""bench.py -- cell-updates/s of the CTU hydro advance on the Sedov 3-D 256^3 single-level problem.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL halo exchange)

A "step" is one coarse time step of the level (Castro::advance: clean_state, FillPatch,
construct_ctu_hydro_source, clean_state, dt estimate), i.e. the reference's own FOM
("zones advanced per microsecond", Source/driver/Castro_advance.cpp:461-471).  Inputs are
synthetic (the Sedov initial data generated on the device) and resident in HBM before the
timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# contract traffic of construct_ctu_hydro_source per cell-update, 8 B x
#   (8 Sborder read + 8 S_new read + 8 S_new write + 3*8*2 fluxes read-modify-write + 3 mass_fluxes write) = 600 B.
# The default bench mode elides bytes the single-level driver does not need (declared, as 8(d) requires):
#   S_new = Sborder + ... (no S_new read) and fluxes[d] = ... instead of zero-fill + "+=" (no fluxes read):
#   8 B x (8 + 8 + 3*8 + 3) = 344 B.  --reference-contract runs the 600-B form.
PATH_BYTES_CONTRACT = 600.0
PATH_BYTES_ASSIGN = 344.0

# algorithmic (compulsory) bytes per processed unit of each hot-path kernel: every input and
# output array element exactly once (DESIGN.md "Kernels").  Unit = one zone/face of the kernel's box.
KERNEL_BYTES_PER_UNIT = {
    "k_ctoprim": 8 * (8 + 8),
    "k_divu": 8 * (3 + 1),
    "k_trace": 8 * (8 + 42 + 8),                          # + F1[x]: the first x Riemann solve is fused in
    "k_riemann1": 8 * (14 + 1 + 8),
    "k_trans1": 8 * (42 + 24 + 1 + 48),                   # all three normal directions in one launch
    "k_final": 8 * (14 + 16 + 1 + 1 + 8 + 9 + 17),        # fluxes read-modify-write (8 read + 8 write + mass)
    "k_final_assign": 8 * (14 + 16 + 1 + 1 + 8 + 9 + 9),  # fluxes written only
    "k_consup": 8 * (27 + 8 + 8),
    "k_consup_clean": 8 * (27 + 8 + 8),
    "k_clean_state": 8 * (8 + 8),
    "k_estdt": 8 * 5,
}


def kernel_units(name, n):
    nx, ny, nz = n
    return {
        "k_ctoprim": (nx + 8) * (ny + 8) * (nz + 8),
        "k_divu": (nx + 2) * (ny + 2) * (nz + 2),
        "k_trace": (nx + 2) * (ny + 2) * (nz + 2),
        "k_riemann1": ((nx + 1) * (ny + 2) * (nz + 2) + (nx + 2) * (ny + 1) * (nz + 2) + (nx + 2) * (ny + 2) * (nz + 1)) / 3.0,
        "k_trans1": (nx + 2) * (ny + 2) * (nz + 2),
        "k_final": ((nx + 1) * ny * nz + nx * (ny + 1) * nz + nx * ny * (nz + 1)) / 3.0,
        "k_consup": nx * ny * nz,
        "k_consup_clean": nx * ny * nz,
        "k_clean_state": nx * ny * nz,
        "k_estdt": nx * ny * nz,
    }.get(name, nx * ny * nz)


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the newest committed PMC pass (profiles/*_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 correction on FETCH_SIZE).
    Counters cannot be collected from inside this process, so None if no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        alias = {"k_trace": "k_trace_pair", "k_consup_clean": "k_consup", "k_divu": "k_divu_pair"}       # hipEvent label -> kernel symbol
        k = d["kernels"].get(kernel) or d["kernels"][alias[kernel]]
        return k["bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    except (KeyError, ValueError):
        return None, None


def pmc_traffic_per_step(prof, steps):
    """Sum over the hot-path kernels of (PMC bytes per launch from the committed profile) x (launches per step
    counted live): the L2->fabric traffic of one step.  None without a committed profile."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))["kernels"]
    alias = {"k_trace": "k_trace_pair", "k_consup_clean": "k_consup", "k_divu": "k_divu_pair"}
    tot = 0.0
    for name, (ms, launches) in prof.items():
        k = d.get(name) or d.get(alias.get(name, ""))
        if k is None:
            continue
        if name == "k_riemann1" and "k_riemann1_blockstart" in d:
            # two full launches + one block-start launch per step share the label
            per_step = 2 * k["bytes_per_launch"] + d["k_riemann1_blockstart"]["bytes_per_launch"]
            tot += per_step
            continue
        tot += k["bytes_per_launch"] * launches / max(steps, 1)
    return tot


def usable_cpus():
    """CPUs this process may actually use: affinity mask and cgroup CPU quota (the GPU boxes expose 256 logical CPUs
    behind a 16-CPU quota; oversubscribing them makes the OpenMP oracle several times slower)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())             # cgroup v1
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(ncell, steps):
    """The CPU oracle (a port organised like the reference's CPU path: ~75 sweeps per tile,
    tiles 1024x16x16, OpenMP over tiles) timed on this host on a bounded Sedov sample."""
    from oracle import oracle_lib as O
    n = (ncell, ncell, ncell)
    ntiles = max(1, (ncell // 16)) ** 2
    # one thread per usable CPU, one tile per thread at most (256 tiles at 256^3)
    threads = max(1, min(usable_cpus(), ntiles))
    lev = O.Level(n, O.make_geom(n), O.default_params(), nthreads=threads)
    lev.init_sedov()
    lev.step(0.01)                    # untimed first step (page faults, scratch allocation)
    t0 = time.time()
    for _ in range(steps):
        lev.step(0.01)
    wall = time.time() - t0
    lev.close()
    return {"value": ncell ** 3 * steps / wall, "unit": "cell-updates/s", "cores": threads, "kind": "port",
            "sample": "Sedov 3D %d^3 single level, %d coarse steps (whole advance), oracle/libcastro_oracle.so, "
                      "OpenMP over 1024x16x16 tiles" % (ncell, steps),
            "seconds": wall}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ncell", type=int, default=256, help="global zones per side (strong scaling: fixed as N grows)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: ncell^3 zones PER GPU (config 3: 512^3 on 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-ncell", type=int, default=256)
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--periodic", action="store_true", help="periodic instead of outflow boundaries (exercises the halo "
                    "pack/exchange/unpack path even on one GPU)")
    ap.add_argument("--force-overlap", action="store_true", help="staged overlap (ghost-free part of the update while the halo "
                    "exchange runs on the communication stream), as every rank of a multi-GPU run does")
    ap.add_argument("--overlap-tiles", action="store_true", help="the older interior tile + 6 boundary slabs form of the overlap")
    ap.add_argument("--reference-contract", action="store_true",
                    help="zero-fill + accumulate fluxes and run clean_state/estdt as separate passes (600 B/cell form)")
    args = ap.parse_args()

    import torch
    import castro_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    # one rank per GPU; the modulo only matters for the functional check of this script on a box with fewer GPUs than
    # ranks (CASTRO_AMD_BENCH_BACKEND=gloo, every rank on the same device -- timings are then meaningless)
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)

    comm = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CASTRO_AMD_BENCH_BACKEND", "nccl")         # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        comm = castro_amd.DistComm()
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    grid = castro_amd.default_grid(world)
    if args.weak:
        n_cell = tuple(args.ncell * grid[d] for d in range(3))
    else:
        n_cell = (args.ncell,) * 3

    contract = args.reference_contract
    bc = (0, 0, 0) if args.periodic else (2, 2, 2)
    c = castro_amd.Castro(n_cell, comm=comm, grid=grid, lo_bc=bc, hi_bc=bc,
                          overlap=(False if args.no_overlap else ("tiles" if args.overlap_tiles else
                                                                  (True if args.force_overlap else None))),
                          fuse_clean=not contract, flux_assign=not contract)
    PATH_BYTES_PER_CELL = PATH_BYTES_CONTRACT if contract else PATH_BYTES_ASSIGN
    c.initData("sedov")                      # synthetic input, generated on the device
    for _ in range(args.warmup):
        c.step()

    def sync():
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()
            torch.cuda.synchronize()

    c.hydro.profile(True)
    c.hydro.profile_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        c.step()
    sync()
    t1 = time.perf_counter()
    wall = t1 - t0
    if comm is not None:
        w = torch.tensor([wall], dtype=torch.float64, device="cpu" if comm.dist.get_backend() == "gloo" else "cuda")
        comm.dist.all_reduce(w, op=comm.dist.ReduceOp.MAX)
        wall = w.item()
    prof = c.hydro.profile_report()
    c.hydro.profile(False)

    total_cells = n_cell[0] * n_cell[1] * n_cell[2]
    value = total_cells * args.steps / wall

    # dominant kernel (largest total device time over the timed region), hipEvent-timed on its stream
    roof = None
    if prof:
        name, (tot_ms, launches) = max(prof.items(), key=lambda kv: kv[1][0])
        avg_s = tot_ms / launches / 1e3
        units = kernel_units(name, c.n)
        alg_bytes = KERNEL_BYTES_PER_UNIT.get(name + "_assign" if (name == "k_final" and not contract) else name, 0) * units
        achieved = alg_bytes / avg_s / 1e9
        traffic, traffic_src = pmc_traffic(name)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": name, "avg_launch_ms": avg_s * 1e3, "launches": launches,
                "algorithmic_bytes_per_launch": alg_bytes}
    hydro_ms = sum(v[0] for k, v in prof.items() if k in ("k_ctoprim", "k_divu", "k_trace", "k_riemann1", "k_trans1",
                                                          "k_final", "k_consup", "k_consup_clean")) / max(args.steps, 1)
    traffic_step = pmc_traffic_per_step(prof, args.steps) if prof else None
    path = {"bytes_per_cell_update": PATH_BYTES_PER_CELL,
            "achieved_GBs_per_gpu": value / world * PATH_BYTES_PER_CELL / 1e9,
            "frac_of_hbm_peak": value / world * PATH_BYTES_PER_CELL / 1e9 / HBM_PEAK_GBS,
            "hydro_kernels_ms_per_step": hydro_ms,
            # hardware utilisation: PMC traffic of one step (committed rocprofv3 profile of this command) over the
            # measured step time, against the 8 TB/s peak
            "pmc_traffic_bytes_per_step": traffic_step,
            "pmc_traffic_GBs": (traffic_step / (wall / args.steps) / 1e9) if traffic_step and world == 1 and c.n == (256, 256, 256) else None,
            "pmc_traffic_frac_of_hbm_peak": (traffic_step / (wall / args.steps) / 1e9 / HBM_PEAK_GBS) if traffic_step and world == 1 and c.n == (256, 256, 256) else None,
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in sorted(prof.items())}}

    out = {
        "metric": "cell-updates/sec, Sedov 3D 256\u00b3 single-level at 1/2/4/8 MI355X; % HBM roofline",
        "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if args.weak else "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "Sedov 3D %dx%dx%d single level, gamma-law EOS, PPM + CGF Riemann, CTU" % n_cell,
                   "rank_grid": "%dx%dx%d" % grid, "zones_per_gpu": c.n[0] * c.n[1] * c.n[2],
                   "overlap_halo": ("tiles" if c.overlap == "tiles" else bool(c.overlap and c._comm_stream is not None and c.neighbors)), "sim_time": c.time, "nstep": c.nstep,
                   "flux_mode": "accumulate (600 B/cell contract)" if contract else "assign (344 B/cell, declared)",
                   "fused_clean_state": not contract},
        "roofline": roof,
        "path_roofline": path,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_ncell, args.cpu_steps)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if comm is not None:
        comm.dist.destroy_process_group()


if __name__ == "__main__":
    main()
