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
#   The `contract` build with clean_state fused into the update (this script's default) reads neither the temperature nor the
#   species of the old state (k_ctoprim / k_finalx_consup: `lean_q` bit 1): 8 B x (6 + 8 + 3*8 + 3) = 328 B.
#   Only bytes that move are declared (tests/test_capi_symbols.py ties these figures to the kernels' read sets).
PATH_BYTES_CONTRACT = 600.0
STATE_PLANES_READ = {False: 8, True: 6}            # planes of the old state the path reads: all / without UTEMP, UFS (lean)
PATH_BYTES_ASSIGN = 8.0 * (STATE_PLANES_READ[False] + 8 + 3 * 8 + 3)          # 344
PATH_BYTES_ASSIGN_LEAN = 8.0 * (STATE_PLANES_READ[True] + 8 + 3 * 8 + 3)      # 328


def path_bytes(reference_contract, numerics, fused_clean=True):
    """declared bytes per cell-update of the mode that was timed"""
    if reference_contract:
        return PATH_BYTES_CONTRACT
    return PATH_BYTES_ASSIGN_LEAN if (numerics == "contract" and fused_clean) else PATH_BYTES_ASSIGN

# compulsory bytes per processed unit of each hot-path kernel: every input and output array element exactly once
# (DESIGN.md "Kernels").  Unit = one zone/face of the kernel's box.  These give the per-kernel HBM utilisation
# (roofline.kernel_utilisation); the roofline figure itself is the SURVEY 8(d) contract above.
KERNEL_BYTES_PER_UNIT = {
    "k_ctoprim": 8 * (STATE_PLANES_READ[False] + 8),
    "k_ctoprim_clean": 8 * (STATE_PLANES_READ[False] + 8),                       # + the pending clean_states; only changed components of U are written back
    "k_ctoprim_shell": 8 * (STATE_PLANES_READ[False] + 8),                       # the ghost shell after CASTRO_AMD_STAGE_VALID (same zone work)
    "k_ctoprim_bc": 8 * (8 + 8 + 8),                      # a boundary zone: the source zone's 8 components in, 8 out, its primitive record
    "k_divu": 8 * (3 + 1),
    "k_trace": 8 * (8 + 42 + 7),                          # + F1[x] (7-plane state form): the first x Riemann solve is fused in
    "k_riemann1": 8 * (14 + 1 + 8),
    "k_trans1": 8 * (42 + 24 + 1 + 48),                   # all three normal directions in one launch
    "k_trans1_fold": 8 * (42 + 7 + 1 + 42),               # + the first y / z solves: F1[y], F1[z] are neither written nor read;
                                                          #   F1[x] and F2 in the 7-plane state form
    "k_final_rmw": 8 * (14 + 14 + 1 + 1 + 8 + 9 + 17),    # fluxes read-modify-write (8 read + 8 write + mass); F2: 2 x 7 planes
    "k_final_assign": 8 * (14 + 14 + 1 + 1 + 8 + 9 + 9),  # fluxes written only
    "k_finalx_consup_rmw": 8 * (14 + 14 + 1 + 1 + 8 + 17 + 18 + 8),   # x faces + consup: + FL[y], FL[z] read, S_new written
    "k_finalx_consup_assign": 8 * (14 + 14 + 1 + 1 + 8 + 9 + 18 + 8),
    "k_consup": 8 * (27 + 8 + 8),
    "k_consup_clean": 8 * (27 + 8 + 8),
    "k_clean_state": 8 * (8 + 8),
    "k_estdt": 8 * 5,
}


# The `contract` build with the default options (gamma-law gas, one species, default solver) does not carry (rho e) or X
# through the edge states and X through the records (`gamma_law_edges`): its kernels have fewer planes.
KERNEL_BYTES_PER_UNIT_LEAN = {
    "k_ctoprim": 8 * (STATE_PLANES_READ[True] + 6),       # neither the temperature nor the species of the state is read
    "k_ctoprim_clean": 8 * (STATE_PLANES_READ[True] + 6),
    "k_ctoprim_shell": 8 * (STATE_PLANES_READ[True] + 6),
    "k_ctoprim_bc": 8 * (8 + 8 + 6),
    "k_trace": 8 * (6 + 30 + 5),                          # Q without (rho e), X; 5-plane edge states; 5-plane F1[x]
    "k_trans1_fold": 8 * (30 + 5 + 1 + 36),               # F2 in the 6-plane state form
    "k_final_rmw": 8 * (10 + 12 + 1 + 1 + 6 + 8 + 17),
    "k_final_assign": 8 * (10 + 12 + 1 + 1 + 6 + 8 + 9),  # FL without the species plane
    "k_finalx_consup_rmw": 8 * (10 + 12 + 1 + 1 + 6 + 17 + 16 + 8),
    "k_finalx_consup_assign": 8 * (10 + 12 + 1 + 1 + 6 + 9 + 16 + 8),
    # the whole final stage in one zone-centred launch (k_final_tile): edge states and F2 records of all three directions, the sound
    # speed, div(u) and Sborder once; the three flux arrays (+ mass fluxes) and S_new out; no FL
    "k_final_tile_rmw": 8 * (30 + 36 + 1 + 1 + 6 + 3 * 17 + 8),
    "k_final_tile_assign": 8 * (30 + 36 + 1 + 1 + 6 + 3 * 9 + 8),
}


def kernel_bytes_per_unit(name, contract, lean=False, fused_divu=False):
    """fused_divu: div(u) is computed inside the trace launch (round 6, `contract` build): one more plane written there, the three
    velocity planes it reads are planes the trace kernel reads anyway, and k_divu's launch (3 + 1 planes) is gone"""
    tab = dict(KERNEL_BYTES_PER_UNIT, **KERNEL_BYTES_PER_UNIT_LEAN) if lean else KERNEL_BYTES_PER_UNIT
    if name == "k_trace" and fused_divu:
        return tab["k_trace"] + 8
    if name in ("k_final_x", "k_final_y", "k_final_z"):
        return tab["k_final_rmw" if contract else "k_final_assign"]
    if name == "k_finalx_consup":
        return tab["k_finalx_consup_rmw" if contract else "k_finalx_consup_assign"]
    if name == "k_final_tile":
        return tab.get("k_final_tile_rmw" if contract else "k_final_tile_assign", 0)
    return tab.get(name, 0)


def kernel_units(name, n, bc_zones=0):
    """zones / faces one launch of `name` processes on an n box; bc_zones: the zones of grow(box, 4) outside the domain that the
    hydro call fills itself (k_ctoprim_bc: CASTRO_AMD_BC_FILL) -- k_ctoprim then covers the rest of the grown box only"""
    nx, ny, nz = n
    return {
        "k_ctoprim": (nx + 8) * (ny + 8) * (nz + 8) - bc_zones,
        "k_ctoprim_clean": (nx + 8) * (ny + 8) * (nz + 8) - bc_zones,
        "k_ctoprim_bc": bc_zones,
        "k_ctoprim_shell": (nx + 8) * (ny + 8) * (nz + 8) - bc_zones - nx * ny * nz,
        "k_divu": (nx + 2) * (ny + 2) * (nz + 2),
        "k_trace": (nx + 2) * (ny + 2) * (nz + 2),
        "k_riemann1": ((nx + 2) * (ny + 1) * (nz + 2) + (nx + 2) * (ny + 2) * (nz + 1)) / 2.0,     # y and z launches
        "k_trans1": (nx + 2) * (ny + 2) * (nz + 2),
        "k_trans1_fold": (nx + 2) * (ny + 2) * (nz + 2),
        "k_final_x": (nx + 1) * ny * nz,
        "k_final_y": nx * (ny + 1) * nz,
        "k_final_z": nx * ny * (nz + 1),
        "k_finalx_consup": (nx + 1) * ny * nz,
    }.get(name, nx * ny * nz)


def source_stamp():
    """sha1 over the kernel sources: ties a committed PMC profile to the code it was taken from"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "castro_amd", "csrc")
    for f in ("hydro_device.h", "ctu_kernels.h", "ctu_kernels.hip", "aux_kernels.hip", "capi.hip"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_per_step():
    """L2->fabric bytes of one 256^3 step from the newest committed PMC pass (profiles/*_traffic.json: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 correction on FETCH_SIZE, tools/pmc.sh).  Counters cannot be
    collected from inside this process; the profile is used only if it carries the stamp of the kernel sources this
    run was built from, otherwise (None, reason)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, "no committed profile"
    try:
        d = json.load(open(files[-1]))
    except ValueError:
        return None, "unreadable profile"
    if d.get("source_stamp") != source_stamp():
        return None, "%s was taken from other kernel sources (stamp %s, now %s)" % (
            os.path.relpath(files[-1], ROOT), d.get("source_stamp"), source_stamp())
    return d.get("bytes_per_step"), os.path.relpath(files[-1], ROOT)


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


def self_launch(nproc):
    """One rank per GPU of this node under torch.distributed.run (rendezvous on 127.0.0.1, a free port), as a child process."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if lines:
        print(lines[-1])
    else:
        sys.stderr.write(p.stdout[-4000:])
    sys.stdout.flush()
    return p.returncode if p.returncode else (0 if lines else 1)


def noise_state(torch, n, device):
    """A rough but physical conserved state (NUM_STATE, nz, ny, nx), generated on the device: a smooth background, an oblique
    density / pressure jump and zone-to-zone noise of +-20 % in rho and p -- every limiter, the flattening and all wave
    patterns of the Riemann solver are exercised in every zone (the form of tests/util.physical_state(smooth=False))."""
    nx, ny, nz = n
    g = torch.Generator(device=device)
    g.manual_seed(1234)
    f64 = dict(dtype=torch.float64, device=device)
    z = torch.arange(nz, **f64).view(nz, 1, 1)
    y = torch.arange(ny, **f64).view(1, ny, 1)
    x = torch.arange(nx, **f64).view(1, 1, nx)

    def rnd(lo, hi):
        return lo + (hi - lo) * torch.rand((nz, ny, nx), generator=g, **f64)
    rho = 1.0 + 0.3 * torch.sin(0.37 * x + 0.3) * torch.cos(0.29 * y + 1.1) + 0.2 * torch.sin(0.41 * z + 2.0)
    p = 1.0 + 0.4 * torch.cos(0.31 * x + 0.23 * y + 0.7) + 0.2 * torch.sin(0.33 * z + 1.9)
    s = (x - nx / 2) + 0.6 * (y - ny / 2) - 0.4 * (z - nz / 2)
    rho = torch.where(s > 0, rho * 0.2, rho) * rnd(0.8, 1.25)
    p = torch.where(s > 0, p * 0.05, p) * rnd(0.8, 1.25)
    u = 0.5 * torch.sin(0.21 * x + 0.4) + rnd(-0.05, 0.05)
    v = 0.4 * torch.cos(0.27 * y + 0.3) + rnd(-0.05, 0.05)
    w = 0.3 * torch.sin(0.19 * z + 1.1) + rnd(-0.05, 0.05)
    eint = p / 0.4
    U = torch.empty((8, nz, ny, nx), **f64)
    U[0], U[1], U[2], U[3] = rho, rho * u, rho * v, rho * w
    U[4] = eint + 0.5 * rho * (u * u + v * v + w * w)
    U[5], U[6], U[7] = eint, 1.0, rho * rnd(0.999, 1.0)
    return U


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
    ap.add_argument("--overlap-staged", action="store_true", help="the round-2 staged form of the overlap (ctoprim + tracing of the "
                    "inner zones beside the exchange, split trace launches); --force-overlap is the light split of round 6")
    ap.add_argument("--no-contract-leg", action="store_true", help="skip the 600-B contract leg of the default run")
    ap.add_argument("--stepwise", action="store_true", help="one host round trip per step (Castro.step) instead of the "
                    "host-free batch (Castro.run_steps)")
    ap.add_argument("--reference-contract", action="store_true",
                    help="zero-fill + accumulate fluxes and run clean_state/estdt as separate passes (600 B/cell form)")
    ap.add_argument("--numerics", choices=("exact", "contract"), default=os.environ.get("CASTRO_AMD_BENCH_NUMERICS", "contract"),
                    help="build of the kernel library the headline is measured on: `contract` (FMA contraction, reciprocal "
                         "division, rsq sqrt; rtol 1e-10 against the oracle on every plotfile field: tests/test_gpu_contract.py) "
                         "or `exact` (bit-identical to the oracle: tests/test_gpu_parity.py); the other one is timed as a leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra legs (developed Sedov state, noisy state, "
                    "per-step medians, rank proxies) of the default single-GPU run")
    ap.add_argument("--proxy-rank-of", type=int, nargs="*", default=None, metavar="N",
                    help="untimed extra (single GPU): ms/step of ONE rank's box of the N-rank strong-scaling run of this workload "
                         "with a full 26-region halo exchange through RCCL self-send (default with extras: 2 4 8); a PROJECTION, "
                         "reported in config.rank_proxies, never in value")
    ap.add_argument("--proxy-in-process", action="store_true", help="run the rank proxies in this process (default: in a child process, so "
                    "that nothing they do can take the bench line down with it)")
    ap.add_argument("--proxy-child", action="store_true", help=argparse.SUPPRESS)          # internal: this IS the child
    ap.add_argument("--proxy-value", type=float, default=0.0, help=argparse.SUPPRESS)       # internal: the parent's cell-updates/s
    args = ap.parse_args()

    # `python bench.py --gpus N` outside a launcher: start the N ranks as a CHILD torch.distributed.run (never an exec, and
    # before this process has imported torch or touched the GPU), relay rank 0's JSON line and exit with the child's code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    # ONE JSON line on stdout and nothing else: libraries that chat on file descriptor 1 (RCCL prints a version banner there when
    # a communicator is created) are sent to stderr for the whole run; the line goes to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

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
        ranks_seen, backend_name = dist.get_world_size(), dist.get_backend()
    else:
        ranks_seen, backend_name = 1, "none"
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d: launch with torch.distributed.run --nproc-per-node %d" % (
        world, args.gpus, args.gpus)
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        rccl = None

    grid = castro_amd.default_grid(world)
    if args.weak:
        n_cell = tuple(args.ncell * grid[d] for d in range(3))
    else:
        n_cell = (args.ncell,) * 3

    contract = args.reference_contract
    bc = (0, 0, 0) if args.periodic else (2, 2, 2)
    overlap = (False if args.no_overlap else ("tiles" if args.overlap_tiles else ("staged" if args.overlap_staged else
                                                                                   (True if args.force_overlap else None))))

    def sync():
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()
            torch.cuda.synchronize()

    def run_proxies(want, value):
        """Projection of the strong-scaling curve from ONE GPU (clearly not a measurement of N GPUs): the box one rank of the N-rank run
        owns (z, then y, then x halved), periodic in every direction so that all 26 neighbour regions exist, every region packed, sent to
        and received from this same rank through ncclSend / ncclRecv of the kernel library's communicator (CASTRO_AMD_HALO_SELF_SEND),
        unpacked, with the halo overlap exactly as a rank of that size would run it and the step graph of the RCCL form.  What it
        leaves out: xGMI instead of a loopback copy, and the waiting for the slowest rank."""
        proxies = {"note": "PROJECTED from one GPU: one rank's box of the N-rank strong-scaling run, 26-region exchange through RCCL self-send; "
                           "projected_value = N x zones of the box / time, an upper bound of what N GPUs can reach"}
        saved = {k: os.environ.get(k) for k in ("CASTRO_AMD_C_HALO", "CASTRO_AMD_HALO_SELF_SEND")}
        os.environ["CASTRO_AMD_C_HALO"], os.environ["CASTRO_AMD_HALO_SELF_SEND"] = "1", "1"
        try:
            for N in want:
                gN = castro_amd.default_grid(N)
                nb = tuple(n_cell[d] // gN[d] for d in range(3))
                try:
                    c = castro_amd.Castro(nb, lo_bc=(0, 0, 0), hi_bc=(0, 0, 0), overlap=overlap, numerics=args.numerics, proxy_ranks=N)
                    c.initData("sedov", r_init=0.01 * 256.0 / max(n_cell))
                    ks = max(10, min(args.steps, 30))
                    c.run_steps(6)
                    c.prepare_step_graph()
                    sync()
                    t0 = time.perf_counter()
                    c.run_steps(ks)
                    sync()
                    ms = (time.perf_counter() - t0) / ks * 1e3
                    hs = c.halo_stats()
                    proxies[str(N)] = {"box": list(nb), "ms_per_step": ms, "projected_value": N * nb[0] * nb[1] * nb[2] / ms * 1e3,
                                       "projected_efficiency_vs_1gpu": ((N * nb[0] * nb[1] * nb[2] / ms * 1e3) / (N * value)) if value > 0 else None,
                                       "overlap_halo": (c.overlap if c.overlap in ("tiles", "staged") else bool(c.overlap and c._comm_stream is not None and c.neighbors)),
                                       "step_graph": bool(getattr(c, "_graphs", None)), "regions": hs["regions"],
                                       "bytes_exchanged_per_step": sum(nbr["sbuf"].numel() * 8 for nbr in c.neighbors),
                                       "fillboundary_ms": hs["fillboundary_ms"], "issued_by": hs["issued_by"]}
                    c.close()
                    c.comm.close(c.hydro)          # the proxy's own one-rank RCCL communicator
                    del c
                    torch.cuda.empty_cache()
                except Exception as e:
                    proxies[str(N)] = {"box": list(nb), "error": "%s: %s" % (type(e).__name__, e)}
                    torch.cuda.synchronize()
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        return proxies

    if args.proxy_child:
        # the child of a default run: the proxies and nothing else, one JSON object on the real stdout
        out = run_proxies(args.proxy_rank_of or [2, 4, 8], args.proxy_value)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps({"rank_proxies": out}) + "\n").encode())
        return

    def run(contract_mode, steps, warmup, kernel_pass, stepwise=None, state="sedov", per_step=False, numerics=None, overlap_=None):
        """W untimed warm-up steps, K timed steps (no profiling events in the timed region), then -- untimed -- a second
        pass of min(K, 5) steps with hipEvents around every kernel on its launch stream for the per-kernel table."""
        if stepwise is None and (args.stepwise or world > 1):
            stepwise = bool(args.stepwise)
        if stepwise is None:
            try:
                return run(contract_mode, steps, warmup, kernel_pass, stepwise=False, state=state, per_step=per_step, numerics=numerics, overlap_=overlap_)
            except castro_amd.AdvanceFailure:
                raise
            except Exception as e:          # a box whose runtime refuses the graph capture: the stepwise form measures the same kernels
                print("bench.py: host-free batch failed (%s: %s); falling back to --stepwise" % (type(e).__name__, e), file=sys.stderr)
                torch.cuda.synchronize()
                return run(contract_mode, steps, warmup, kernel_pass, stepwise=True, state=state, per_step=per_step, numerics=numerics, overlap_=overlap_)
        c = castro_amd.Castro(n_cell, comm=comm, grid=grid, lo_bc=bc, hi_bc=bc, overlap=overlap if overlap_ is None else overlap_,
                              fuse_clean=not contract_mode, flux_assign=not contract_mode, numerics=numerics or args.numerics)
        if state == "noise":
            c.set_state(noise_state(torch, n_cell, torch.device("cuda", local_rank)))
        else:
            c.initData("sedov")                  # synthetic input, generated on the device
        if state == "developed":
            c.evolve(0.01)                       # untimed: the blast wave at the reference's stop_time (about 1000 steps at 256^3)
        # host-free stepping (Castro.run_steps): dt, time and the step checks stay on the device, one host
        # synchronisation per batch; on one rank a captured pair of steps is replayed as a hipGraph.  --stepwise keeps
        # the round-1/2 form (one allreduce + host read per step).  Same kernels, same dt, bit-identical states.
        host_free = (not stepwise) and c.host_free_ok()
        if host_free:
            c.run_steps(warmup)
            if warmup >= 2:
                c.prepare_step_graph()           # untimed: the graph for the roles the state buffers have now (per rank with RCCL)
        else:
            for _ in range(warmup):
                c.step()
        sync()
        t0 = time.perf_counter()
        per = []
        if per_step:
            # one synchronised wall time per step, for the median
            for _ in range(steps):
                t1 = time.perf_counter()
                c.step()
                sync()
                per.append(time.perf_counter() - t1)
        elif host_free:
            c.run_steps(steps)
        else:
            for _ in range(steps):
                c.step()
        sync()
        wall = time.perf_counter() - t0
        if comm is not None:
            w = torch.tensor([wall], dtype=torch.float64, device="cpu" if comm.dist.get_backend() == "gloo" else "cuda")
            comm.dist.all_reduce(w, op=comm.dist.ReduceOp.MAX)
            wall = w.item()
        prof, ksteps = {}, 0
        if kernel_pass:
            ksteps = min(steps, 5)
            c.hydro.profile(True)
            c.hydro.profile_reset()
            for _ in range(ksteps):
                c.step()
            sync()
            prof = c.hydro.profile_report()
            c.hydro.profile(False)
        info = {"zones_per_gpu": c.n[0] * c.n[1] * c.n[2], "sim_time": c.time, "nstep": c.nstep, "n": c.n,
                "overlap_halo": (c.overlap if c.overlap in ("tiles", "staged") else bool(c.overlap and c._comm_stream is not None and c.neighbors)),
                "halo": c.halo_stats() if hasattr(c, "halo_stats") else None, "host_free": host_free,
                "step_graph": bool(host_free and getattr(c, "_graphs", None)), "per_step": per,
                "numerics": c.hydro.numerics, "library": c.hydro.lib.castro_amd_version().decode()}
        c.close()                                # the C-ABI halo plans own device buffers outside torch's allocator
        del c
        torch.cuda.empty_cache()
        return wall, prof, ksteps, info

    wall, prof, ksteps, info = run(contract, args.steps, args.warmup, True)
    total_cells = n_cell[0] * n_cell[1] * n_cell[2]
    value = total_cells * args.steps / wall
    bytes_per_cell = path_bytes(contract, info["numerics"], fused_clean=not contract)

    # per-kernel HBM utilisation: compulsory bytes of a launch over its hipEvent-timed duration
    kutil = {}
    # one rank, no periodic direction: every zone of the ghost shell is a physical-boundary zone, filled by the hydro call
    nn = info["n"]
    bc_zones = ((nn[0] + 8) * (nn[1] + 8) * (nn[2] + 8) - nn[0] * nn[1] * nn[2]) if ("k_ctoprim_bc" in prof and world == 1 and not args.periodic) else 0
    for name, (tot_ms, launches) in sorted(prof.items()):
        bpu = kernel_bytes_per_unit(name, contract, lean=(info["numerics"] == "contract"), fused_divu="k_divu" not in prof)
        avg_ms = tot_ms / launches
        e = {"avg_launch_ms": avg_ms, "launches_per_step": launches / ksteps, "ms_per_step": tot_ms / ksteps}
        if bpu:
            nb = bpu * kernel_units(name, info["n"], bc_zones)
            e.update({"compulsory_bytes_per_launch": nb, "GBs": nb / avg_ms / 1e6, "frac_of_hbm_peak": nb / avg_ms / 1e6 / HBM_PEAK_GBS})
        kutil[name] = e
    traffic_step, traffic_src = pmc_traffic_per_step() if (world == 1 and info["n"] == (256, 256, 256) and not contract) else (None, "n/a for this configuration")

    # roofline = cell-updates/s x declared bytes per cell-update over the 8 TB/s HBM peak (per GPU)
    achieved = value / world * bytes_per_cell / 1e9
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic_step, "traffic_source": traffic_src,
            "bytes_per_cell_update": bytes_per_cell,
            "definition": "SURVEY 8(d): cell-updates/s x declared contract bytes per cell-update / 8 TB/s; traffic = L2->fabric "
                          "bytes of one step (PMC, committed profile of the same sources)",
            "kernel_utilisation": kutil}
    if traffic_step:
        roof["traffic_GBs"] = traffic_step / (wall / args.steps) / 1e9
        roof["traffic_over_algorithmic"] = traffic_step / (bytes_per_cell * total_cells)
    if not contract and world == 1 and not args.no_contract_leg:
        # the reference's full output contract (zero-filled fluxes accumulated with +=, S_new read-modify-write,
        # clean_state / estdt as separate passes) timed in the same run
        w6, _, _, _ = run(True, max(5, args.steps // 2), 2, False)
        v6 = total_cells * max(5, args.steps // 2) / w6
        roof["contract_600B"] = {"ms_per_step": w6 / max(5, args.steps // 2) * 1e3, "value": v6,
                                 "achieved": v6 * PATH_BYTES_CONTRACT / 1e9, "frac": v6 * PATH_BYTES_CONTRACT / 1e9 / HBM_PEAK_GBS}

    # the other build of the same sources, same workload, timed in the same process (both libraries are loaded side by side)
    other_leg = None
    if world == 1 and not args.no_extras:
        other = "exact" if args.numerics == "contract" else "contract"
        try:
            w_o, _, _, i_o = run(contract, args.steps, args.warmup, False, numerics=other)
            b_o = path_bytes(contract, i_o["numerics"], fused_clean=not contract)
            other_leg = {"numerics": i_o["numerics"], "ms_per_step": w_o / args.steps * 1e3, "value": total_cells * args.steps / w_o,
                         "bytes_per_cell_update": b_o, "frac": total_cells * args.steps / w_o * b_o / 1e9 / HBM_PEAK_GBS}
        except Exception as e:
            other_leg = {"numerics": other, "error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.synchronize()
    # a multi-GPU run: the same workload with the halo overlap switched the other way (castro.py OVERLAP_MIN_ZONES was set from
    # one GPU; this leg is the measurement on real links), untimed extra
    overlap_leg = None
    if world > 1 and not args.no_extras and overlap is None:
        try:
            # the other setting of the halo overlap (default since round 6: the light split, on) on the same links
            w_v, _, _, i_v = run(contract, args.steps, args.warmup, False, overlap_=not info["overlap_halo"])
            overlap_leg = {"ms_per_step": w_v / args.steps * 1e3, "value": total_cells * args.steps / w_v, "overlap_halo": i_v["overlap_halo"],
                           "host_free": i_v["host_free"], "step_graph": i_v["step_graph"]}
        except Exception as e:
            overlap_leg = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.synchronize()
    extras = None
    if world == 1 and not contract and not args.no_extras:
        # untimed extras (the headline above is unchanged): the same step on states that are not
        # 99.9 % ambient gas, and per-step medians as BASELINE.md section 3 words the protocol
        ks = max(10, min(args.steps, 50))
        extras = {"steps_each": ks}
        w_med, _, _, i_med = run(False, max(50, args.steps), 2, False, stepwise=True, per_step=True)
        ps = sorted(i_med["per_step"])
        extras["stepwise_median_ms"] = ps[len(ps) // 2] * 1e3
        extras["stepwise_mean_ms"] = sum(ps) / len(ps) * 1e3
        extras["stepwise_steps"] = len(ps)
        for st in ("developed", "noise"):
            try:
                w_x, _, _, i_x = run(False, ks, 4, False, state=st)
                extras[st + "_ms_per_step"] = w_x / ks * 1e3
                extras[st + "_sim_time"] = i_x["sim_time"]
                extras[st + "_nstep"] = i_x["nstep"]
            except Exception as e:
                extras[st + "_ms_per_step"] = None
                extras[st + "_error"] = "%s: %s" % (type(e).__name__, e)
                torch.cuda.synchronize()

    # Projection of the strong-scaling curve from ONE GPU (run_proxies above), by default in a CHILD process: the proxies drive RCCL
    # self-send inside captured two-stream graphs, and whatever could go wrong there (round 6 met a segmentation fault inside
    # hipStreamEndCapture with another stream arrangement) must not cost the measured line of this process
    proxies = None
    want = args.proxy_rank_of if args.proxy_rank_of is not None else ([2, 4, 8] if (world == 1 and not args.no_extras and not contract) else [])
    if world == 1 and want and args.proxy_in_process:
        proxies = run_proxies(want, value)
    elif world == 1 and want:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--proxy-child", "--proxy-value", repr(value), "--proxy-rank-of"] + [str(N) for N in want] + [
            "--ncell", str(args.ncell), "--steps", str(args.steps), "--numerics", args.numerics, "--no-cpu-baseline"]
        for flag, on in (("--no-overlap", args.no_overlap), ("--force-overlap", args.force_overlap), ("--overlap-tiles", args.overlap_tiles),
                         ("--overlap-staged", args.overlap_staged), ("--weak", args.weak)):
            if on:
                cmd.append(flag)
        try:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()                 # the child needs the memory of the legs that are done
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and "rank_proxies" in ln]
            if lines:
                proxies = json.loads(lines[-1])["rank_proxies"]
                proxies["run_in"] = "child process"
            else:
                proxies = {"error": "proxy child exited with code %d: %s" % (r.returncode, r.stderr[-400:])}
        except Exception as e:
            proxies = {"error": "%s: %s" % (type(e).__name__, e)}

    out = {
        "metric": "cell-updates/sec, Sedov 3D 256\u00b3 single-level at 1/2/4/8 MI355X; % HBM roofline",
        "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if args.weak else "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "Sedov 3D %dx%dx%d single level, gamma-law EOS, PPM + CGF Riemann, CTU" % n_cell,
                   "rank_grid": "%dx%dx%d" % grid, "zones_per_gpu": info["zones_per_gpu"],
                   "overlap_halo": info["overlap_halo"], "sim_time": info["sim_time"], "nstep": info["nstep"],
                   "flux_mode": "accumulate (600 B/cell contract)" if contract else
                                "assign (%d B/cell, declared%s)" % (bytes_per_cell, "; old temperature / species not read" if bytes_per_cell < PATH_BYTES_ASSIGN else ""),
                   "fused_clean_state": not contract, "halo": info["halo"],
                   "host_free_steps": info["host_free"], "step_graph": info["step_graph"],
                   "ranks_seen": ranks_seen, "backend": backend_name, "rccl_version": rccl,
                   "numerics": info["numerics"], "library": info["library"],
                   "numerics_parity": {"contract": "rtol 1e-10 on all 33 plotfile fields vs the CPU oracle (tests/test_gpu_contract.py)",
                                       "exact": "bit-identical to the CPU oracle (tests/test_gpu_parity.py)"}[info["numerics"]],
                   "other_numerics_leg": other_leg, "other_states": extras, "rank_proxies": proxies, "overlap_leg": overlap_leg},
        "roofline": roof,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_ncell, args.cpu_steps)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if comm is not None:
        try:
            from castro_amd.hydro import HipHydro
            hc = HipHydro(local_rank, numerics=args.numerics)
            comm.close(hc)                         # the kernel library's communicator, before the process group goes
            hc.close()
        except Exception:
            pass
        comm.dist.destroy_process_group()


if __name__ == "__main__":
    main()
