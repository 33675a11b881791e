"""Single-level driver around the hot path, mirroring the reference's `Castro` class.

Method names follow the reference so a reader of Source/driver can map one to the other:

  Castro.initData                 Source/driver/Castro.cpp:934   (+ Exec/<problem>/problem_initialize*.H)
  Castro.advance / do_advance_ctu Source/driver/Castro_advance.cpp:19, Castro_advance_ctu.cpp:15-397
  Castro.expand_state (FillPatch) Source/driver/Castro.cpp:4201-4209
  Castro.clean_state              Source/driver/Castro.cpp:4238-4278
  Castro.construct_ctu_hydro_source  Source/hydro/Castro_ctu_hydro.cpp:16
  Castro.estTimeStep / computeInitialDt / computeNewDt   Source/driver/Castro.cpp:1490-1866

Decomposition: one box per rank (one rank per GPU); the level-0 domain is cut into a
px x py x pz grid of equal boxes.  The FillPatch ghost exchange is point-to-point
(torch.distributed batch_isend_irecv == grouped ncclSend/ncclRecv on RCCL, one peer per
xGMI link) on a communication stream, overlapped with the hydro update of the interior
sub-box, which needs no remote data, on the compute stream.
"""
import itertools
import os

import time

import torch

from . import _lib as L

NUM_STATE, NUM_GROW = L.NUM_STATE, L.NUM_GROW


# --------------------------------------------------------------------------------------------
# communicators
# --------------------------------------------------------------------------------------------
def _use_c_halo(default):
    """CASTRO_AMD_C_HALO: 1 = the FillBoundary of the C ABI (castro_amd_fill_boundary: RCCL send / recv issued by the kernel
    library), 0 = torch.distributed.batch_isend_irecv; default: the C form whenever the backend is RCCL."""
    v = os.environ.get("CASTRO_AMD_C_HALO")
    return default if v is None else v not in ("0", "")


def _drop_graphs_of(users):
    """the step graphs of the Castro objects that captured collectives of a communicator that is about to go"""
    for u in list(users or ()):
        if getattr(u, "_graphs", None):
            u._graphs = {}


class SingleComm:
    rank, size = 0, 1
    device_side = True          # no collective ever touches the host

    def c_comm(self, hydro):
        """A one-rank RCCL communicator of the C ABI when CASTRO_AMD_C_HALO=1 (periodic wraps and the code path of a
        multi-rank run on one GPU), else None."""
        if not _use_c_halo(False):
            return None
        if getattr(self, "_ccomm", None) is None:
            self._ccomm = hydro.comm_create(1, 0, hydro.comm_unique_id())
        return self._ccomm

    def close(self, hydro=None):
        """destroy the C-ABI communicator (after every Castro object that used it has been closed)"""
        cc, self._ccomm = getattr(self, "_ccomm", None), None
        _drop_graphs_of(getattr(self, "_users", None))
        if cc and hydro is not None:
            hydro.comm_destroy(cc)

    def exchange(self, sends, recvs):
        assert not sends and not recvs

    def allreduce_min(self, t):
        return t

    def barrier(self):
        pass

    def gather_objects(self, obj):
        return [obj]


class DistComm:
    """torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" on CPU for tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        # RCCL collectives are enqueued on the device (the host does not wait for them); gloo moves host memory
        self.device_side = dist.get_backend(group) == "nccl"
        self._ccomm = None

    def c_comm(self, hydro):
        """The RCCL communicator of the C ABI over the same ranks (castro_amd_comm_create; the unique id travels through
        torch.distributed), or None when the backend is not RCCL / CASTRO_AMD_C_HALO=0."""
        if not _use_c_halo(self.device_side) or not self.device_side:
            return None
        if self._ccomm is None:
            # every rank takes part in every step below whatever happens on it, and the ranks agree on the outcome: either all
            # of them use the C path or none does (a rank on its own in ncclCommInitRank would hang the others)
            try:
                uid = hydro.comm_unique_id() if self.rank == 0 else None
            except Exception as e:                      # librccl could not be bound
                uid, err = None, e
            box = [uid]
            self.dist.broadcast_object_list(box, src=self.dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                            group=self.group)
            ok = torch.ones(1, dtype=torch.int32, device="cuda")
            cc = None
            if box[0] is None:
                ok.zero_()
            else:
                try:
                    cc = hydro.comm_create(self.size, self.rank, box[0])
                except Exception as e:
                    ok.zero_()
                    err = e
            self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN, group=self.group)
            if int(ok.item()) == 1:
                self._ccomm = cc
                self._chydro = hydro
            else:
                if cc is not None:
                    hydro.comm_destroy(cc)
                if self.rank == 0:
                    import sys
                    print("castro_amd: the C-level halo exchange is not available on every rank (%s); using torch.distributed"
                          % (locals().get("err", "another rank failed"),), file=sys.stderr)
                self._ccomm = False
        return self._ccomm or None

    def close(self, hydro=None):
        """destroy the C-ABI communicator (collective in spirit: every rank calls it, after closing its Castro objects)"""
        cc, self._ccomm = self._ccomm, None
        _drop_graphs_of(getattr(self, "_users", None))
        if cc and hydro is not None:
            hydro.comm_destroy(cc)

    def exchange(self, sends, recvs):
        """sends/recvs: lists of (peer_rank, tag, tensor).  Grouped point-to-point."""
        dist = self.dist
        if dist.get_backend(self.group) == "gloo" and any(b.is_cuda for _, _, b in list(sends) + list(recvs)):
            # test transport only (two ranks sharing one GPU): gloo moves host memory
            hs = [(p, t, b.cpu()) for p, t, b in sends]
            hr = [(p, t, torch.empty_like(b, device="cpu"), b) for p, t, b in recvs]
            self.exchange(hs, [(p, t, h) for p, t, h, _ in hr])
            for _, _, h, b in hr:
                b.copy_(h)
            return
        ops = []
        # post receives first, ordered by (peer, tag) on both sides so the grouped call matches up
        for peer, tag, buf in sorted(recvs, key=lambda x: (x[0], x[1])):
            ops.append(dist.P2POp(dist.irecv, buf, peer, group=self.group, tag=tag))
        for peer, tag, buf in sorted(sends, key=lambda x: (x[0], x[1])):
            ops.append(dist.P2POp(dist.isend, buf, peer, group=self.group, tag=tag))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def allreduce_min(self, t):
        if t.is_cuda and self._ccomm and os.environ.get("CASTRO_AMD_C_ALLREDUCE", "1") != "0":
            # the library's own RCCL communicator, on the current stream: ncclAllReduce(MIN) behind the kernels that produced t,
            # capturable into the step graph together with the grouped ncclSend / ncclRecv of castro_amd_fill_boundary
            self._chydro.allreduce_min_c(self._ccomm, t)
            return t
        if t.is_cuda and self.dist.get_backend(self.group) == "gloo":
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MIN, group=self.group)
            t.copy_(h)
            return t
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return t

    def barrier(self):
        self.dist.barrier(group=self.group)

    def gather_objects(self, obj):
        """every rank's (small, picklable) object, in rank order, on all ranks"""
        out = [None] * self.size
        self.dist.all_gather_object(out, obj, group=self.group)
        return out


def default_grid(size):
    """Split z first, then y, then x."""
    g = [1, 1, 1]
    d = 2
    while size > 1:
        assert size % 2 == 0, "rank count must be a power of two"
        g[d] *= 2
        size //= 2
        d = (d - 1) % 3
    return tuple(g)


def checked_estimate(est, empty_ok=False):
    """The CFL estimate of estTimeStep has no retry path behind it (Castro.cpp:1507-1626): a NaN zone is dropped by the
    minimum like in the reference, but an estimate that is not a positive finite number (every zone NaN, a negative sound
    speed sum, an empty level) must not become a time step."""
    if empty_ok and est == 1.e200:          # no zone contributed: the initial value of the reduction
        return est
    if not (est > 0.0) or est == float("inf") or est >= 1.e199:
        raise AdvanceFailure("estTimeStep: the CFL estimate is not a positive finite number (%r)" % (est,))
    return est


class AdvanceFailure(RuntimeError):
    pass


# smallest box side from which the halo overlap is on by default in a multi-rank run.  Round 6: the LIGHT split (ctoprim with the
# pending cleans on the valid zones beside the exchange, the ghost shell as one launch, everything downstream un-split) is host-free
# under castro.use_retry and lives inside the per-rank step graph, so it is compared graph against graph: with all 26 regions
# through RCCL self-send on one GPU it costs +0.09 / +0.03 / +0.03 / +0.01 ms at 256^3 / 256x256x128 / 256x128x128 / 128^3 per rank
# (0.4-1.2 %; profiles/r06b_overlap_light_split.txt: ctoprim over the valid rows + the shell launch against one contiguous ctoprim, and
# the loop-back copy competing with it for the same HBM; the kernel trace shows both running side by side) and it can hide up to the 0.05-0.31 ms of that ctoprim behind real links: on for every box.  (Round 5 had
# 384 here: the round-2 staged form -- split trace launches, not host-free under use_retry -- cost 0.2-0.8 ms; overlap="staged".)
OVERLAP_MIN_ZONES = 0


# --------------------------------------------------------------------------------------------
class Castro:
    def __init__(self, n_cell, prob_lo=(0., 0., 0.), prob_hi=(1., 1., 1.), lo_bc=(2, 2, 2), hi_bc=(2, 2, 2),
                 params=None, hydro=None, comm=None, grid=None, overlap=None, make_params=None, fuse_clean=True, flux_assign=True,
                 use_retry=True, retry_subcycle_factor=0.5, max_subcycles=10, dt_cutoff=1.e-12,
                 do_grav=False, const_grav=0.0, grav_source_type=4, box=None, rotation=None, fixed_dt=-1.0, initial_dt=-1.0, max_dt=1.e200,
                 alloc=True, numerics=None, proxy_ranks=1):
        """numerics: "exact" | "contract" for the HipHydro this object creates (castro_amd/_lib.py).  alloc=False: the geometry and bookkeeping of a box another rank owns (castro_amd/amr.py), no device memory."""
        self.n_cell = tuple(int(x) for x in n_cell)
        self.owned = bool(alloc)
        self.comm = comm if comm is not None else SingleComm()
        try:
            import weakref
            if getattr(self.comm, "_users", None) is None:
                self.comm._users = weakref.WeakSet()
            self.comm._users.add(self)              # comm.close() drops the step graphs that captured its collectives
        except (AttributeError, TypeError):
            pass
        if hydro is None:
            from .hydro import HipHydro
            dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
            hydro = HipHydro(dev, numerics=numerics)          # raises without a GPU: no CPU fallback
        self.hydro = hydro
        self.params = params if params is not None else (make_params() if make_params else L.default_params())
        self.geom = (hydro.make_geom if hasattr(hydro, "make_geom") else L.make_geom)(
            self.n_cell, prob_lo, prob_hi, lo_bc, hi_bc)
        self.lo_bc, self.hi_bc = tuple(lo_bc), tuple(hi_bc)
        self.periodic = tuple(lo_bc[d] == 0 and hi_bc[d] == 0 for d in range(3))

        # --- decomposition: one box per rank ---
        self.grid = tuple(grid) if grid is not None else default_grid(self.comm.size)
        assert self.grid[0] * self.grid[1] * self.grid[2] == self.comm.size
        r = self.comm.rank
        self.coords = (r % self.grid[0], (r // self.grid[0]) % self.grid[1], r // (self.grid[0] * self.grid[1]))
        self.lo, self.hi = [], []
        for d in range(3):
            assert self.n_cell[d] % self.grid[d] == 0, "domain must divide evenly over the rank grid"
            nloc = self.n_cell[d] // self.grid[d]
            assert nloc >= 2 * NUM_GROW or self.grid[d] == 1, "boxes must be at least 8 zones wide when decomposed"
            self.lo.append(self.coords[d] * nloc)
            self.hi.append(self.lo[d] + nloc - 1)
        self.lo, self.hi = tuple(self.lo), tuple(self.hi)
        if box is not None:
            # a single box that does not tile the domain (a refined patch, castro_amd/amr.py): one rank; its ghost zones
            # come from the coarser level and from the other boxes of its level (periodic images included)
            assert self.comm.size == 1
            self.lo, self.hi = tuple(int(x) for x in box[0]), tuple(int(x) for x in box[1])
        self.n = tuple(self.hi[d] - self.lo[d] + 1 for d in range(3))
        self.glo = tuple(x - NUM_GROW for x in self.lo)
        self.ghi = tuple(x + NUM_GROW for x in self.hi)
        self.gbox = (self.glo, self.ghi)
        self.bx = (self.lo, self.hi)

        # --- state: two bordered buffers (Sborder / S_new share storage layout; swap per step) ---
        if not alloc:
            hydro_alloc = lambda *a, **k: None
        else:
            hydro_alloc = hydro.alloc
        self.S_old_b = hydro_alloc(NUM_STATE, self.glo, self.ghi)
        self.S_new_b = hydro_alloc(NUM_STATE, self.glo, self.ghi)
        self.flux_boxes, self.fluxes, self.mass_fluxes = [], [], []
        for d in range(3):
            fhi = list(self.hi)
            fhi[d] += 1
            self.flux_boxes.append((self.lo, tuple(fhi)))
            self.fluxes.append(hydro_alloc(NUM_STATE, self.lo, fhi))
            self.mass_fluxes.append(hydro_alloc(1, self.lo, fhi))
        # [CFL estimate after the last clean_state, min rho, CFL estimate after the first clean_state]
        self.red = hydro.alloc(1, (0, 0, 0), (2, 0, 0)).reshape(3) if alloc else None

        self._plans = {}
        self.neighbors = self._build_neighbors() if box is None else []
        # Overlap of the halo exchange with compute (the forms are listed where self.overlap is set below): True = the light
        # split of round 6; "staged" runs ctoprim on the valid zones and the PPM tracing of the zones >= 3 from the box faces
        # while the exchange is in flight, then the rest -- no redundant work, split trace launches; the older interior-tile +
        # six-slab split ("tiles") re-does ctoprim/trace on 3x the slab volume (+18 % at 256^3, +44 % at 128^3 per rank).
        # Default: OVERLAP_MIN_ZONES above (round 6: the light split, on for every multi-rank box).
        # proxy_ranks > 1 (bench.py --proxy-rank-of): this single-rank object stands for one rank of such a run -- the defaults
        # that depend on the communicator size are taken as that rank would take them
        if overlap is None:
            overlap = max(self.comm.size, int(proxy_ranks)) > 1 and min(self.n) >= OVERLAP_MIN_ZONES
        # True: the light split of round 6 (ctoprim + the pending cleans on the valid zones beside the exchange, then the ghost
        # shell in one launch and the un-split update); "staged": the round-2 split (ctoprim + tracing of the inner zones beside
        # the exchange, split trace launches after it); "tiles": interior tile + six slabs
        self.overlap = overlap                                                   # True | "staged" | "tiles" | False
        # the hydro call can fill the physical-boundary zones of Sborder itself (CASTRO_AMD_BC_FILL: the boundary-zone mode of
        # k_ctoprim instead of k_bc_fill + the k_ctoprim pass over those zones); needs the image of a mirrored ghost layer inside
        # the box.  The light overlap needs it (a boundary fill between its two stages would hand clean zones to a pass that
        # cleans); the plain path keeps k_bc_fill + ONE k_ctoprim over the whole grown box, which is 0.05 ms faster per 256^3 step
        # (a launch over (n+8)^3 zones of an (n+8)-wide FAB is one contiguous stream, the valid rows alone are not:
        # profiles/r06h_*).  CASTRO_AMD_BC_IN_HYDRO=0: never; =2: in the plain path as well (A/B, tests).
        self.bc_in_hydro = (hasattr(self.hydro, "lib") and box is None and alloc and min(self.n) >= NUM_GROW
                            and os.environ.get("CASTRO_AMD_BC_IN_HYDRO", "1") != "0")
        self.bc_in_hydro_plain = self.bc_in_hydro and os.environ.get("CASTRO_AMD_BC_IN_HYDRO", "1") == "2"
        self.fuse_clean = bool(fuse_clean)
        self.fuse_post_clean = True        # post_timestep's clean_state may ride in the fused pass (a level of CastroAmr: no)
        # the clean_state sweeps in front of the hydro update ride inside k_ctoprim (castro_amd_ctu_hydro_fab_ex) when the
        # rank's box is updated by one un-staged call; CASTRO_AMD_FUSE_SBORDER_CLEAN=0 keeps the separate sweep
        self.fuse_sborder_clean = (self.fuse_clean and hasattr(self.hydro, "lib") and box is None
                                   and os.environ.get("CASTRO_AMD_FUSE_SBORDER_CLEAN", "1") != "0")
        self._pending_cleans, self._post_clean_done, self._whole_step = 2, False, False
        # one hydro call per step: "zero fluxes, then +=" (Castro_advance.cpp:391-394) is an assignment
        self.flux_assign = bool(flux_assign)
        self._flux_clear = False
        # castro.use_retry, retry_subcycle_factor, max_subcycles, dt_cutoff (_cpp_parameters:311-354)
        self.use_retry, self.retry_subcycle_factor = bool(use_retry), float(retry_subcycle_factor)
        self.max_subcycles, self.dt_cutoff = int(max_subcycles), float(dt_cutoff)
        self.nsubcycles, self.nretries, self.last_failure = 0, 0, ""
        # castro.fixed_dt / castro.initial_dt (_cpp_parameters; Castro.cpp:1490-1513, 1655)
        self.fixed_dt, self.initial_dt = float(fixed_dt), float(initial_dt)
        self.max_dt = float(max_dt)                             # castro.max_dt (Castro.cpp:1515-1530)
        # castro.do_grav with gravity.gravity_type = "ConstantGrav": g along the last dimension (Gravity.cpp:860-866)
        self.do_grav, self.grav, self.grav_source_type = bool(do_grav), (0.0, 0.0, float(const_grav)), int(grav_source_type)
        # castro.do_rotation: `rotation` = _lib.make_rotation(rotational_period, rot_axis, ...)
        self.rotation = rotation
        self.have_sources = self.do_grav or rotation is not None
        if self.have_sources:
            NSRC, NGS = 7, 3            # NSRC, NUM_GROW_SRC (Castro_setup.cpp:317-327)
            self.sbox = (tuple(x - NGS for x in self.lo), tuple(x + NGS for x in self.hi))
            self.old_source = hydro_alloc(NSRC, *self.sbox)
            self.new_source = hydro_alloc(NSRC, self.lo, self.hi)
            self.src_neighbors = self._build_neighbors(NGS, NSRC) if alloc else []
            # castro.source_term_predictor = 1: Castro::source_corrector and the bookkeeping of create_source_corrector
            self.source_corrector = hydro_alloc(NSRC, *self.sbox)
        self.lastDt, self._in_retry = 1.e200, False
        self._comm_stream = None
        if self.overlap and alloc and self.S_new_b.is_cuda:
            self._comm_stream = torch.cuda.Stream(device=self.S_new_b.device)

        self.time = 0.0
        self.dt = 0.0
        self.nstep = 0
        self.hydro_seconds = 0.0
        if hasattr(hydro, "reserve"):
            hydro.reserve(*self.n)

    # ----------------------------------------------------------------------------------------
    def close(self):
        """Give back what lives outside torch's allocator: the C-ABI halo plans of this object (two packed device buffers
        each).  The RCCL communicator of the C ABI belongs to the comm object (DistComm / SingleComm.close), which may serve
        several Castro objects.  Idempotent; never called while a stream is capturing (a free inside a capture aborts)."""
        plans = getattr(self, "_plans", {})
        # a captured step graph holds the plans' buffers (and the communicator's collectives) by address: never replay one
        # after they are gone -- the next run_steps captures afresh (or runs stream-ordered)
        if getattr(self, "_graphs", None):
            self._graphs = {}
        self._eager_done = False
        if not any("cplan" in plan for plan in plans.values()):
            return
        h = getattr(self, "hydro", None)
        if h is None or not getattr(h, "h", None):
            for plan in plans.values():
                plan.pop("cplan", None)
            return                                  # the context (and its library state) is gone already
        if torch.cuda.is_available():
            torch.cuda.synchronize()                # a plan is single-stream and may still be in flight
        for plan in plans.values():                 # the torch-side tables stay: expand_state keeps its one-launch pack / unpack
            cp = plan.pop("cplan", None)
            if cp is not None:
                h.halo_plan_destroy(cp)

    def __del__(self):
        try:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                return
            self.close()
        except Exception:
            pass

    def S_new(self):
        """Valid-region view of the new-time state, shape (NUM_STATE, nz, ny, nx)."""
        g = NUM_GROW
        return self.S_new_b[:, g:-g, g:-g, g:-g]

    # ----------------------------------------------------------------------------------------
    def _build_neighbors(self, ng=NUM_GROW, ncomp=NUM_STATE):
        """[(peer_rank, send_tag, recv_tag, send_box, recv_box)] for the up-to-26 neighbours (ng ghost layers of
        an ncomp-component FAB)."""
        out = []
        for off in itertools.product((-1, 0, 1), repeat=3):      # (ox, oy, oz)
            if off == (0, 0, 0):
                continue
            nb = []
            ok = True
            for d in range(3):
                c = self.coords[d] + off[d]
                if c < 0 or c >= self.grid[d]:
                    if self.periodic[d]:
                        c %= self.grid[d]
                    else:
                        ok = False
                        break
                nb.append(c)
            if not ok:
                continue
            peer = nb[0] + self.grid[0] * (nb[1] + self.grid[1] * nb[2])
            slo, shi, rlo, rhi = [], [], [], []
            for d in range(3):
                if off[d] == -1:
                    slo.append(self.lo[d]); shi.append(self.lo[d] + ng - 1)
                    rlo.append(self.lo[d] - ng); rhi.append(self.lo[d] - 1)
                elif off[d] == 1:
                    slo.append(self.hi[d] - ng + 1); shi.append(self.hi[d])
                    rlo.append(self.hi[d] + 1); rhi.append(self.hi[d] + ng)
                else:
                    slo.append(self.lo[d]); shi.append(self.hi[d])
                    rlo.append(self.lo[d]); rhi.append(self.hi[d])
            code = (off[0] + 1) + 3 * (off[1] + 1) + 9 * (off[2] + 1)
            rcode = (-off[0] + 1) + 3 * (-off[1] + 1) + 9 * (-off[2] + 1)
            # what I send towards `off` is what the peer receives from direction -off
            out.append(dict(peer=peer, send_tag=code, recv_tag=rcode, sbox=(tuple(slo), tuple(shi)),
                            rbox=(tuple(rlo), tuple(rhi)), off=off))
        # one contiguous send and one receive buffer, a slice per neighbour: all regions are packed / unpacked by one
        # launch each (castro_amd_pack_regions_fab); a periodic wrap onto this rank unpacks straight from the send buffer
        sizes = []
        for nbr in out:
            n = 1
            for d in range(3):
                n *= nbr["sbox"][1][d] - nbr["sbox"][0][d] + 1
            sizes.append(n * ncomp)
        total = max(sum(sizes), 1)
        sall = self.hydro.alloc(1, (0, 0, 0), (total - 1, 0, 0)).reshape(-1)
        rall = self.hydro.alloc(1, (0, 0, 0), (total - 1, 0, 0)).reshape(-1)
        off = 0
        for nbr, n in zip(out, sizes):
            nbr["off"] = off
            nbr["sbuf"], nbr["rbuf"] = sall[off:off + n], rall[off:off + n]
            off += n
        if out and hasattr(self.hydro, "region_table"):
            h = self.hydro
            local = [nb for nb in out if nb["peer"] == self.comm.rank]
            remote = [nb for nb in out if nb["peer"] != self.comm.rank]
            plan = dict(sall=sall, rall=rall, pack=h.region_table([nb["sbox"] for nb in out], [nb["off"] for nb in out]))
            # what I send towards `off` is what I receive from `-off`
            src = {nb["send_tag"]: nb for nb in local}
            plan["unpack_local"] = h.region_table([nb["rbox"] for nb in local], [src[nb["recv_tag"]]["off"] for nb in local]) if local else None
            plan["unpack_remote"] = h.region_table([nb["rbox"] for nb in remote], [nb["off"] for nb in remote]) if remote else None
            # the same exchange issued by the kernel library itself (castro_amd_fill_boundary) when the ranks talk RCCL
            ccomm = self.comm.c_comm(h) if hasattr(self.comm, "c_comm") and hasattr(h, "halo_plan") else None
            if ccomm is not None:
                plan["cplan"] = h.halo_plan(ccomm, [(nb["peer"], nb["sbox"], nb["rbox"], nb["send_tag"], nb["recv_tag"]) for nb in out], ncomp)
            self._plans[id(out)] = plan
        return out

    # ---- AmrLevel::FillPatch at a single level: same-level copy + physical BCs ----
    def expand_state(self, S, box=None, neighbors=None, bc=True, mark_packed=False):
        """box / neighbors default to the state's (NUM_GROW ghosts); the Source_Type FillPatch passes its own.
        bc=False: the same-level exchange only (the hydro call that follows fills the physical-boundary zones itself:
        CASTRO_AMD_BC_FILL).  mark_packed: an event is recorded behind the pack -- the last read of the valid zones -- for
        _wait_packed() on the stream that goes on to write them (the light overlap)."""
        h = self.hydro
        box = self.gbox if box is None else box
        neighbors = self.neighbors if neighbors is None else neighbors
        plan = self._plans.get(id(neighbors))
        if plan is not None and "cplan" in plan:
            if mark_packed:
                h.fill_boundary_ex(plan["cplan"], S, box, self.geom if bc else None)
                self._packed = ("c", plan["cplan"])
            else:
                h.fill_boundary(plan["cplan"], S, box, self.geom if bc else None)
            return

        def packed():
            if mark_packed:
                if getattr(self, "_ev_packed", None) is None:
                    self._ev_packed = torch.cuda.Event()
                self._ev_packed.record()
                self._packed = ("t", self._ev_packed)
        if plan is not None:
            h.pack_regions(S, box, plan["pack"], plan["sall"])
            packed()
            sends = [(nb["peer"], nb["send_tag"], nb["sbuf"]) for nb in neighbors if nb["peer"] != self.comm.rank]
            recvs = [(nb["peer"], nb["recv_tag"], nb["rbuf"]) for nb in neighbors if nb["peer"] != self.comm.rank]
            self.comm.exchange(sends, recvs)
            if plan["unpack_local"] is not None:
                h.unpack_regions(S, box, plan["unpack_local"], plan["sall"])
            if plan["unpack_remote"] is not None:
                h.unpack_regions(S, box, plan["unpack_remote"], plan["rall"])
            if bc:
                h.bc_fill(S, box, self.geom)
            return
        sends, recvs, local = [], [], []
        for nb in neighbors:
            h.pack(S, box, nb["sbox"][0], nb["sbox"][1], nb["sbuf"])
        packed()
        for nb in neighbors:
            if nb["peer"] == self.comm.rank:
                local.append(nb)
            else:
                sends.append((nb["peer"], nb["send_tag"], nb["sbuf"]))
                recvs.append((nb["peer"], nb["recv_tag"], nb["rbuf"]))
        # periodic wrap onto myself: the buffer I send towards `off` is the one I receive from `-off`
        for nb in local:
            src = next(x for x in local if x["send_tag"] == nb["recv_tag"])
            nb["rbuf"].copy_(src["sbuf"])
        self.comm.exchange(sends, recvs)
        for nb in neighbors:
            h.unpack(S, box, nb["rbox"][0], nb["rbox"][1], nb["rbuf"])
        if bc:
            h.bc_fill(S, box, self.geom)

    def _wait_packed(self):
        """the current stream waits for the pack of the last expand_state(mark_packed=True)"""
        kind, what = self._packed
        if kind == "c":
            self.hydro.halo_plan_wait_packed(what)
        else:
            torch.cuda.current_stream().wait_event(what)

    def _light_overlap(self):
        """the light split can carry the pending cleans inside its two ctoprim launches: overlap is True and the hydro call fills
        the physical-boundary zones itself (a boundary fill between the two stages would hand clean zones to a pass that cleans)"""
        return self.overlap is True and self.bc_in_hydro and not self.have_sources

    def halo_stats(self, repeats=5):
        """What one FillBoundary of the state costs this rank: bytes sent to other ranks per step, the number of
        neighbour regions, and the wall time of pack + exchange + unpack + BC fill measured on an idle stream
        (bench.py --gpus N reports it; Castro.cpp:4201-4209 expand_state)."""
        remote = [nb for nb in self.neighbors if nb["peer"] != self.comm.rank]
        nbytes = sum(nb["sbuf"].numel() * 8 for nb in remote)
        torch.cuda.synchronize()
        self.comm.barrier()
        t0 = time.perf_counter()
        for _ in range(repeats):
            self.expand_state(self.S_old_b)
        torch.cuda.synchronize()
        self.comm.barrier()
        ms = (time.perf_counter() - t0) / repeats * 1e3
        plan = self._plans.get(id(self.neighbors))
        return {"regions": len(self.neighbors), "remote_regions": len(remote), "bytes_sent_per_step": nbytes,
                "fillboundary_ms": ms,
                "issued_by": "castro_amd_fill_boundary (C ABI, %s)" % self.hydro.comm_version() if plan is not None and "cplan" in plan
                else "torch.distributed.batch_isend_irecv"}

    # ---- Castro::clean_state ---------------------------------------------------------------
    def clean_state(self, S, ntimes=1):
        self.hydro.clean_state(S, self.gbox, self.lo, self.hi, self.params, ntimes=ntimes)

    # ---- Castro::initData ------------------------------------------------------------------
    def initData(self, problem="sedov", **kw):
        h = self.hydro
        if problem == "sedov":
            h.sedov_init(self.S_new_b, self.gbox, self.lo, self.hi, self.geom, self.params, **kw)
        elif problem == "sod":
            h.sod_init(self.S_new_b, self.gbox, self.lo, self.hi, self.geom, self.params, **kw)
        else:
            raise ValueError(problem)
        self.clean_state(self.S_new_b, 1)      # Castro.cpp:1100-1160
        self.time, self.nstep, self.dt = 0.0, 0, 0.0

    def set_state(self, full_state):
        """Initial data from a host array (NUM_STATE, nz, ny, nx) covering the whole domain (a custom
        problem_initialize_state_data); followed by the post-init clean_state like initData."""
        g, lo, n = NUM_GROW, self.lo, self.n
        part = torch.as_tensor(full_state[:, lo[2]:lo[2] + n[2], lo[1]:lo[1] + n[1], lo[0]:lo[0] + n[0]])
        self.S_new_b[:, g:g + n[2], g:g + n[1], g:g + n[0]] = part.to(self.S_new_b.device, self.S_new_b.dtype)
        self.clean_state(self.S_new_b, 1)
        self.time, self.nstep, self.dt = 0.0, 0, 0.0

    # ---- Castro::estTimeStep (hydro limiter) -------------------------------------------------
    def _reduce(self):
        """[min dx/(c+|u|), min rho] over the whole level (device reduction + allreduce MIN)."""
        self.red.fill_(1.e200)
        self.hydro.estdt_cfl(self.S_new_b, self.gbox, self.lo, self.hi, self.geom, self.params, self.red)
        self.comm.allreduce_min(self.red)
        v = self.red.tolist()
        return v[0], v[1]

    def estTimeStep(self):
        if self.fixed_dt > 0.0:                                 # Castro.cpp:1511-1513
            return self.fixed_dt
        est, _ = self._reduce()
        return min(self.max_dt, checked_estimate(est) * self.params.cfl)

    def computeInitialDt(self, stop_time=-1.0):
        # Castro::initialTimeStep (Castro.cpp:1490-1504)
        dt_0 = self.initial_dt if self.initial_dt > 0.0 else self.params.init_shrink * self.estTimeStep()
        eps = 0.001 * dt_0
        if stop_time >= 0.0 and (self.time + dt_0) > (stop_time - eps):
            dt_0 = stop_time - self.time
        return dt_0

    def computeNewDt(self, dt_old, stop_time=-1.0, est=None):
        dt_0 = self.estTimeStep() if (est is None or self.fixed_dt > 0.0) else est
        if self.fixed_dt <= 0.0:                                # Castro.cpp:1655
            dt_0 = min(dt_0, self.params.change_max * dt_old)
        eps = 2.220446049250313e-16
        if stop_time >= 0.0 and (self.time + dt_0) >= (stop_time - eps):
            dt_0 = stop_time - self.time
        return dt_0

    # ---- Castro::construct_ctu_hydro_source over this rank's box ------------------------------
    def construct_ctu_hydro_source(self, time, dt, tiles=None, fuse_clean=False, src=None, stage=None, sborder_clean=0,
                                   d_dt=None, bc_fill=False):
        """fuse_clean: no new-time source follows the hydro update, so S_new.min(URHO), clean_state(S_new)
        and the CFL estimate run inside the update pass (castro_amd_ctu_hydro_clean_fab) and reduce into
        self.red, which the caller has initialised.  When the attempt covers the whole step of a single level,
        the clean_state of Castro::post_timestep (Castro.cpp:1909-1916) is applied in the same pass.  A second
        clean_state is not always the identity (the dual-energy reset can take its other branch once eden has been
        floored), so the pass reduces the CFL estimate twice: after the first application (red[2]: what the validity
        check of do_advance_ctu sees) and after the last (red[0]: what estTimeStep of the next coarse step sees); the
        density check uses the raw update (red[1])."""
        h = self.hydro
        post_clean = fuse_clean and self.fuse_post_clean and getattr(self, "_whole_step", False) and stage not in ("A", "valid")
        if post_clean:
            self._post_clean_done = True
        for bx in (tiles or [self.bx]):
            h.construct_ctu_hydro_source(bx, self.S_old_b, self.gbox, self.S_new_b, self.gbox, self.geom,
                                         self.params, time, dt, fluxes=self.fluxes, flux_boxes=self.flux_boxes,
                                         mass_fluxes=self.mass_fluxes, vbx=self.bx, update_from_sborder=src is None,
                                         src=src, src_box=self.sbox if src is not None else None,
                                         clean_ntimes=(2 if post_clean else 1) if fuse_clean else 0, red=self.red if fuse_clean else None,
                                         flux_assign=self.flux_assign and self._flux_clear, stage=stage,
                                         **({"sborder_clean": sborder_clean} if sborder_clean else {}),
                                         **({"d_dt": d_dt} if d_dt is not None else {}),
                                         **({"bc_fill": True} if bc_fill else {}))

    def _shell_tiles(self):
        """interior box (needs no ghost data) + 6 boundary slabs of thickness NUM_GROW."""
        g = NUM_GROW
        ilo = tuple(self.lo[d] + g for d in range(3))
        ihi = tuple(self.hi[d] - g for d in range(3))
        if any(ihi[d] < ilo[d] for d in range(3)):
            return None, [self.bx]
        shells = []
        lo, hi = list(self.lo), list(self.hi)
        # z slabs take the full x,y extent; y slabs the remaining z; x slabs the remaining y,z
        shells.append(((lo[0], lo[1], lo[2]), (hi[0], hi[1], ilo[2] - 1)))
        shells.append(((lo[0], lo[1], ihi[2] + 1), (hi[0], hi[1], hi[2])))
        shells.append(((lo[0], lo[1], ilo[2]), (hi[0], ilo[1] - 1, ihi[2])))
        shells.append(((lo[0], ihi[1] + 1, ilo[2]), (hi[0], hi[1], ihi[2])))
        shells.append(((lo[0], ilo[1], ilo[2]), (ilo[0] - 1, ihi[1], ihi[2])))
        shells.append(((ihi[0] + 1, ilo[1], ilo[2]), (hi[0], ihi[1], ihi[2])))
        return (ilo, ihi), shells

    # ---- Castro::do_advance_ctu (Source/driver/Castro_advance_ctu.cpp:15-397) -----------------
    def do_advance_ctu(self, time, dt):
        """One attempt at advancing S_old -> S_new by dt on the current time levels.
        Returns (success, reason, new_dt) like the reference's advance_status."""
        h = self.hydro
        S = self.S_old_b
        # clean_state(S_old) [Castro_advance.cpp:311] and clean_state(Sborder, 4 ghosts)
        # [Castro_advance.cpp:186] are both zone-local: on the valid zones they compose to
        # "clean twice"; ghost zones are copies (or sign-reflected copies) of twice-cleaned
        # valid zones, so they are filled AFTER the cleaning.  See DESIGN.md "clean_state order".
        # The first attempt of a step cleans twice (initialize_advance's clean_state(S_old) + this Sborder's); a later
        # subcycle starts from the previous subcycle's cleaned S_new (Sborder's clean only); a retry on the same old
        # data finds the valid zones of Sborder already in place.
        # With one box per rank, no sources and no staged overlap the pending cleans ride inside k_ctoprim instead of a
        # sweep of their own (castro_amd_hydro_opts.sborder_clean_ntimes): FillPatch copies the uncleaned zones and the
        # hydro call cleans valid and ghost zones alike -- the same zone-local function of the same values.
        use_overlap = self.overlap and self._comm_stream is not None and self.neighbors
        light = self._light_overlap()
        # the hydro call fills the physical-boundary zones: in the light overlap, or on request (see __init__)
        bc_h = (self.bc_in_hydro_plain or (self.bc_in_hydro and bool(use_overlap) and light)) and not self.have_sources
        sb_clean = 0
        if self._pending_cleans > 0:
            if self.fuse_sborder_clean and not self.have_sources and (not use_overlap or light):
                sb_clean = self._pending_cleans
            else:
                self.clean_state(S, self._pending_cleans)
        self._pending_cleans = 0

        # S_new.min(URHO) check (Castro_advance_ctu.cpp:168-216) on the un-cleaned update, clean_state(S_new)
        # (:221-225) and the estTimeStep validity check (:386-392) are fused into the update pass
        # (no new-time source terms on this path) + one 2-double allreduce
        fuse = self.fuse_clean and not self.have_sources
        self.red.fill_(1.e200)
        if self.have_sources:
            return self._do_advance_with_sources(time, dt, S)

        if use_overlap and self.overlap == "tiles":
            # interior tile + six boundary slabs (measured slower than the staged form, kept for comparison)
            interior, shells = self._shell_tiles()
            cur = torch.cuda.current_stream()
            self._comm_stream.wait_stream(cur)
            with torch.cuda.stream(self._comm_stream):
                self.expand_state(S)                                   # halo exchange + BC fill
            if interior is not None:
                self.construct_ctu_hydro_source(time, dt, tiles=[interior], fuse_clean=fuse)   # overlapped: needs no ghost data
            cur.wait_stream(self._comm_stream)
            self.construct_ctu_hydro_source(time, dt, tiles=shells, fuse_clean=fuse)
        elif use_overlap and self.overlap == "staged":
            # halo exchange + BC fill on the communication stream while the compute stream runs the part of the
            # update that reads no ghost zone (ctoprim on the valid zones, PPM tracing 3 zones inside the box)
            cur = torch.cuda.current_stream()
            self._comm_stream.wait_stream(cur)
            self.expand_state(S)                    # the exchange on the issuing stream, stage A on the side stream (see _advance_light_overlap)
            with torch.cuda.stream(self._comm_stream):
                self.construct_ctu_hydro_source(time, dt, stage="A")
            cur.wait_stream(self._comm_stream)
            self.construct_ctu_hydro_source(time, dt, fuse_clean=fuse, stage="B")
        elif use_overlap:
            self._advance_light_overlap(S, time, dt, fuse, sb_clean, bc_h)
        else:
            self.expand_state(S, bc=not bc_h)
            self.construct_ctu_hydro_source(time, dt, fuse_clean=fuse, sborder_clean=sb_clean, bc_fill=bc_h)
        self._flux_clear = False

        if not fuse:
            h.clean_state_reduce(self.S_new_b, self.gbox, self.lo, self.hi, self.geom, self.params, self.red, ntimes=1)
        self.comm.allreduce_min(self.red)
        est, rho_min, est1 = self.red.tolist()
        if rho_min < self.params.small_dens:
            # retry_small_density_cutoff keeps its default (-1e200): every such step is rejected
            return False, ("negative density" if rho_min < 0.0 else "small density") + " (density = %e)" % rho_min, None
        # the validity check sees S_new cleaned once (Castro_advance_ctu.cpp:221-225, 386-392); what is handed on as the
        # next step's estimate is the one of the state as the step leaves it (post_timestep's clean_state included when
        # it rode along), which is what estTimeStep would return at the start of the next coarse step
        chk_dt = self.fixed_dt if self.fixed_dt > 0.0 else min(self.max_dt, est1 * self.params.cfl)
        if self.params.change_max * chk_dt < dt:
            return False, "timestep validity check failed", None
        new_dt = self.fixed_dt if self.fixed_dt > 0.0 else min(self.max_dt, est * self.params.cfl)
        return True, "", new_dt

    def _advance_light_overlap(self, S, time, dt, fuse, sb_clean, bc_h, d_dt=None):
        """The light split (round 6).  Issuing stream: pack -> grouped send / recv -> unpack (-> BC fill when the hydro call does
        not do it).  Side stream, once the pack has read the valid zones: ctoprim with the pending clean_states on the valid
        zones (it writes them in place, and no ghost zone).  Then, both joined: the ghost shell in one launch (+ the
        physical-boundary zones) and the whole un-split update."""
        # The exchange stays on the stream the step is issued on and the ghost-free compute goes to the side stream: RCCL 2.26
        # captured on a stream that was forked INTO a capture takes the process down at hipStreamEndCapture (measured,
        # profiles/r06a_*); on the capture's origin stream its group is captured like any other node.
        cur = torch.cuda.current_stream()
        self.expand_state(S, bc=not bc_h, mark_packed=True)
        kw = {} if d_dt is None else {"d_dt": d_dt}
        with torch.cuda.stream(self._comm_stream):
            self._wait_packed()                     # joins the side stream to everything up to and including the pack
            self.construct_ctu_hydro_source(time, dt, stage="valid", sborder_clean=sb_clean, **kw)
        cur.wait_stream(self._comm_stream)
        self.construct_ctu_hydro_source(time, dt, fuse_clean=fuse, stage="rest", sborder_clean=sb_clean, bc_fill=bc_h, **kw)

    def _do_advance_with_sources(self, time, dt, S):
        """do_advance_ctu with old- and new-time gravity / rotation sources (Castro_advance_ctu.cpp:94-143, 156-274;
        construct_old_source / construct_new_source, Source/sources/Castro_sources.cpp:230-349)."""
        h = self.hydro
        lo, hi = self.lo, self.hi
        self.expand_state(S)
        if self.params.source_term_predictor == 1:
            # create_source_corrector (Castro_advance_ctu.cpp:60-62, Castro.cpp:3780-3818): not on the attempt that follows a
            # rejected one -- the data it is made from have been overwritten by then
            if not self._in_retry:
                self.create_source_corrector()
            h.set_source_corrector(self.source_corrector, self.sbox)
        # MultiFab::Copy(S_new, Sborder) (:94); do_old_sources (:127-131): construct at t^n, apply with the full dt,
        # clean_state -- the copy, the update and the cleaning in one pass where the backend has it
        fused = hasattr(h, "apply_source")
        # one pass (castro_amd_sources_mf, round 6): zero + gravity + rotation + apply + clean_state of a stage in one kernel
        # -- the separate calls below read and write the source and the state three to four times
        one_pass = hasattr(h, "sources_mf") and os.environ.get("CASTRO_AMD_SOURCES_ONE_PASS", "1") != "0"
        if one_pass:
            h.sources_mf(0, h.make_source_boxes([(lo, hi, (S, self.gbox), (self.S_new_b, self.gbox), (self.old_source, self.sbox),
                                                  self.mass_fluxes, self.flux_boxes)]),
                         self.grav if self.do_grav else None, self.grav_source_type if self.do_grav else 4, self.rotation, self.geom,
                         self.params, dt, ntimes=1)
        elif not fused:
            h.copy(self.S_new_b, self.gbox, S, self.gbox, lo, hi)
        if not one_pass:
            self.old_source.zero_()
            if self.do_grav:
                h.old_gravity_source(S, self.gbox, self.old_source, self.sbox, lo, hi, self.grav, self.grav_source_type, dt)
            if self.rotation is not None:
                h.old_rotation_source(S, self.gbox, self.old_source, self.sbox, lo, hi, self.rotation, self.geom, dt)
            if fused:
                h.apply_source(self.S_new_b, self.gbox, S, self.gbox, dt, self.old_source, self.sbox, 7, lo, hi, self.params, ntimes=1)
            else:
                h.saxpy(self.S_new_b, self.gbox, dt, self.old_source, self.sbox, 7, lo, hi)
                h.clean_state(self.S_new_b, self.gbox, lo, hi, self.params, ntimes=1)
        # FillPatch of the source for the tracing
        self.expand_state(self.old_source, self.sbox, self.src_neighbors)
        # hydro with the old source traced in the predictor; S_new += (it already holds the old source)
        try:
            self.construct_ctu_hydro_source(time, dt, src=self.old_source)
        finally:
            if self.params.source_term_predictor == 1:
                h.set_source_corrector(None, None)        # the context must not keep a pointer into this object's tensor
        self._flux_clear = False
        # S_new.min(URHO) (:168-216), clean_state(S_new) (:221-225)
        h.clean_state_reduce(self.S_new_b, self.gbox, lo, hi, self.geom, self.params, self.red, ntimes=1)
        self.comm.allreduce_min(self.red)
        _, rho_min, _ = self.red.tolist()
        if rho_min < self.params.small_dens:
            return False, ("negative density" if rho_min < 0.0 else "small density") + " (density = %e)" % rho_min, None
        # do_new_sources (:262-268): corrector from the new state, apply, clean_state
        if one_pass:
            h.sources_mf(1, h.make_source_boxes([(lo, hi, (S, self.gbox), (self.S_new_b, self.gbox), (self.new_source, (lo, hi)),
                                                  self.mass_fluxes, self.flux_boxes)]),
                         self.grav if self.do_grav else None, self.grav_source_type if self.do_grav else 4, self.rotation, self.geom,
                         self.params, dt, ntimes=1)
        else:
            self.new_source.zero_()
            if self.do_grav:
                h.new_gravity_source(S, self.gbox, self.S_new_b, self.gbox, self.new_source, (lo, hi), self.mass_fluxes,
                                     self.flux_boxes, lo, hi, self.grav, self.grav_source_type, dt, self.geom)
            if self.rotation is not None:
                h.new_rotation_source(S, self.gbox, self.S_new_b, self.gbox, self.new_source, (lo, hi), self.mass_fluxes,
                                      self.flux_boxes, lo, hi, self.rotation, self.geom, dt)
            if fused:
                h.apply_source(self.S_new_b, self.gbox, self.S_new_b, self.gbox, dt, self.new_source, (lo, hi), 7, lo, hi,
                               self.params, ntimes=1)
            else:
                h.saxpy(self.S_new_b, self.gbox, dt, self.new_source, (lo, hi), 7, lo, hi)
                h.clean_state(self.S_new_b, self.gbox, lo, hi, self.params, ntimes=1)
        # timestep validity check (:386-392)
        new_dt = self.estTimeStep()
        if self.params.change_max * new_dt < dt:
            return False, "timestep validity check failed", None
        return True, "", new_dt

    def create_source_corrector(self):
        """Castro::create_source_corrector (Castro.cpp:3780-3818): the lagged predictor dS/dt ~ 2 x (new-time corrector of
        the last advance) / lastDt for the momentum sources; the factor dt/2 is applied in src_to_prim.  The "old"
        Source_Type data after the swap are the corrector `new_source` of the last successful (sub)step."""
        c = self.source_corrector
        c.zero_()
        g = 3                                   # NUM_GROW_SRC
        n = self.n
        c[1:4, g:g + n[2], g:g + n[1], g:g + n[0]] = self.new_source[1:4]
        self.expand_state(c, self.sbox, self.src_neighbors)           # AmrLevel::FillPatch(Source_Type, UMX, 3)
        c.mul_(2.0 / self.lastDt)

    def _swap_state_time_levels(self):
        self.S_old_b, self.S_new_b = self.S_new_b, self.S_old_b

    def _save_old_state(self):
        return self.S_old_b.clone()

    def _restore_old_state(self, prev):
        self.S_old_b.copy_(prev)

    def _zero_fluxes(self):
        """fluxes[d].setVal(0) (Castro_advance.cpp:391-394, Castro_advance_ctu.cpp:455-461).  In flux-assign mode
        the fill is not executed: the next hydro call overwrites every face instead of accumulating."""
        self._flux_clear = True
        if not self.flux_assign:
            for d in range(3):
                self.fluxes[d].zero_()
                self.mass_fluxes[d].zero_()

    # ---- Castro::advance (Source/driver/Castro_advance.cpp:19-121) ------------------------------
    def advance(self, time, dt):
        # initialize_advance: swap_state_time_levels, zero the flux registers, dt_subcycle = 1e200
        self._swap_state_time_levels()
        self._zero_fluxes()
        self._pending_cleans = 2       # clean_state(S_old) here (Castro_advance.cpp:311) + clean_state(Sborder) (:186)
        self._post_clean_done = False
        self.nsubcycles, self.nretries = 1, 0
        if not self.use_retry:
            self._whole_step = True
            self._in_retry = False
            ok, reason, new_dt = self.do_advance_ctu(time, dt)
            if not ok:
                raise AdvanceFailure("Advance was unsuccessful: " + reason)      # amrex::Abort in the reference
            self.lastDt = dt
            return new_dt
        return self.subcycle_advance_ctu(time, dt)

    # ---- Castro::subcycle_advance_ctu + retry_advance_ctu (Castro_advance_ctu.cpp:403-768) ------
    def subcycle_advance_ctu(self, time, dt):
        dt_subcycle = 1.e200
        if dt_subcycle == 1.e200:
            dt_subcycle = dt
        subcycle_time = time
        sub_iteration = 0
        eps = 1.0e-14
        do_swap = False
        prev_old = None
        new_dt = None
        self.nretries = 0
        while subcycle_time < (1.0 - eps) * (time + dt):
            # shorten the last subcycle so that it lands on time + dt
            if subcycle_time + dt_subcycle > (1.0 - self.dt_cutoff) * (time + dt):
                dt_subcycle = (time + dt) - subcycle_time
            if dt_subcycle <= self.dt_cutoff * time:
                raise AdvanceFailure("Error: subcycled timesteps too short.")
            num_subcycles_remaining = int(round(((time + dt) - subcycle_time) / dt_subcycle))
            if num_subcycles_remaining > self.max_subcycles:
                raise AdvanceFailure("Error: too many subcycles.")
            if do_swap:
                self._swap_state_time_levels()
                self._pending_cleans = 1          # the old data of this subcycle are the cleaned S_new of the last one
            else:
                do_swap = True
            # an attempt at the whole step in one go may take post_timestep's clean_state into its fused pass
            self._whole_step = sub_iteration == 0 and self.nretries == 0
            ok, reason, new_dt = self.do_advance_ctu(subcycle_time, dt_subcycle)
            self._in_retry = False                                # Castro_advance_ctu.cpp:651-653
            if not ok:
                # retry_advance_ctu: halve the subcycle, keep the original old data, clear the fluxes
                dt_subcycle = min(dt_subcycle, dt_subcycle) * self.retry_subcycle_factor
                self._post_clean_done = False
                if prev_old is None:
                    prev_old = self._save_old_state()
                self._zero_fluxes()
                do_swap = False
                self.nretries += 1
                self.last_failure = reason
                self._in_retry = True
                continue
            subcycle_time += dt_subcycle
            sub_iteration += 1
            self.lastDt = dt_subcycle                             # :715
        if sub_iteration > 1 and prev_old is not None:
            self._restore_old_state(prev_old)     # state[k].replaceOldData(*prev_state[k])
        self.nsubcycles = sub_iteration
        return new_dt

    # ---- Castro::writePlotFile (Source/driver/Castro_io.cpp:853) ------------------------------
    def writePlotFile(self, dirname, derive=None):
        from .plotfile import write_plotfile
        return write_plotfile(dirname, self, derive=derive)

    # ---- Amr::coarseTimeStep loop -----------------------------------------------------------
    def step(self, stop_time=-1.0):
        if self.nstep == 0:
            self.dt = self.computeInitialDt(stop_time)
        else:
            self.dt = self.computeNewDt(self.dt, stop_time, est=self._next_est)
        self._next_est = self.advance(self.time, self.dt)
        if not self._post_clean_done:
            self.clean_state(self.S_new_b, 1)       # Castro::post_timestep: clean_state(S_new) on every level
            self._next_est = None                   # computeNewDt estimates from the state post_timestep leaves
        self.time += self.dt
        self.nstep += 1
        return self.dt

    # ---- host-free stepping: dt, time and the step checks stay on the device -----------------------------------
    def host_free_ok(self):
        """The device-resident form of step() covers the plain single-level advance: fused clean_state / CFL reduction, no
        source terms, device-side collectives.  A rejected step is latched on the device; with castro.use_retry the host
        then redoes that step with the reference's retry logic, without it the batch raises like step() does."""
        use_overlap = self.overlap and self._comm_stream is not None and self.neighbors
        # with castro.use_retry a rejected step is redone by the host from its old state (run_steps): every write to the
        # caller's arrays must then sit in a kernel that checks the latched status, i.e. the cleans ride in k_ctoprim
        retry_ok = (not self.use_retry) or (self.fuse_sborder_clean and (not use_overlap or self._light_overlap()))
        return (self.fuse_clean and self.fuse_post_clean and not self.have_sources and retry_ok
                and hasattr(self.hydro, "step_control") and getattr(self.comm, "device_side", False)
                and self.overlap != "tiles")

    def _step_device(self, stop_time):
        """One coarse step with every decision left on the device: advance (Castro_advance.cpp:19-121) with dt read from
        self._ctl, then castro_amd_step_control (checks of do_advance_ctu, time += dt, computeNewDt)."""
        self._swap_state_time_levels()
        self._zero_fluxes()
        self._post_clean_done, self._whole_step, self._in_retry = True, True, False
        S = self.S_old_b
        use_overlap = self.overlap and self._comm_stream is not None and self.neighbors
        light = self._light_overlap()
        bc_h = self.bc_in_hydro_plain or (self.bc_in_hydro and bool(use_overlap) and light)
        sb_clean = 2 if (self.fuse_sborder_clean and (not use_overlap or light)) else 0
        if not sb_clean:
            self.clean_state(S, 2)                  # clean_state(S_old) + clean_state(Sborder), see do_advance_ctu
        if use_overlap and self.overlap == "staged":
            cur = torch.cuda.current_stream()
            self._comm_stream.wait_stream(cur)
            self.expand_state(S)
            with torch.cuda.stream(self._comm_stream):
                self.construct_ctu_hydro_source(0.0, 0.0, stage="A", d_dt=self._ctl)
            cur.wait_stream(self._comm_stream)
            self.construct_ctu_hydro_source(0.0, 0.0, fuse_clean=True, stage="B", d_dt=self._ctl)
        elif use_overlap:
            self._advance_light_overlap(S, 0.0, 0.0, True, sb_clean, bc_h, d_dt=self._ctl)
        else:
            self.expand_state(S, bc=not bc_h)
            self.construct_ctu_hydro_source(0.0, 0.0, fuse_clean=True, sborder_clean=sb_clean, d_dt=self._ctl, bc_fill=bc_h)
        self._flux_clear = False
        self.comm.allreduce_min(self.red)
        self.hydro.step_control(self.red, self._ctl, self.params, self.max_dt, self.fixed_dt, stop_time, use_retry=self.use_retry)
        if not torch.cuda.is_current_stream_capturing():
            self._eager_done = True

    def _ensure_ctl(self):
        if getattr(self, "_ctl", None) is None:
            self._ctl = torch.zeros(L.CTL_SIZE, dtype=torch.float64, device=self.red.device)
            self._graphs, self._eager_done = {}, False

    def capture_step_graph(self, stop_time=-1.0):
        """The hipGraph of a PAIR of host-free steps (the state buffers swap roles every step) for the buffers' current
        roles; captured once per (stop_time, roles) and cached.  Capturing executes nothing and leaves this object as it was."""
        self._ensure_ctl()
        assert self._eager_done, "capture_step_graph: run at least one host-free step first (scratch is reserved lazily)"
        # everything the capture bakes into the launches by value belongs to the key: the parameter block and geometry, the
        # dt limits and the flags of the driver -- a run that changes one of them between two batches gets a new graph
        # instead of a replay of the old values
        baked = (bytes(self.params), bytes(self.geom), self.max_dt, self.fixed_dt, bool(self.use_retry), bool(self.flux_assign),
                 bool(self.fuse_clean), bool(self.fuse_post_clean), bool(self.fuse_sborder_clean), str(self.overlap))
        key = (float(stop_time), self.S_old_b.data_ptr(), self.S_new_b.data_ptr(), hash(baked))
        if key not in self._graphs:
            # No finaliser may run while the stream is capturing: a collected context, graph or event would call hipFree /
            # hipGraphExecDestroy in the middle of the capture, which the runtime answers with abort().  torch.cuda.graph
            # stopped collecting on entry (torch >= 2.9), so: collect now, and keep the cyclic collector off until the end.
            import gc
            torch.cuda.synchronize()
            gc.collect()
            gc_was_on = gc.isenabled()
            gc.disable()
            keep = self._host_step_state()
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    if os.environ.get("CASTRO_AMD_TEST_FAIL_CAPTURE") == "1":       # tests: a runtime that refuses the capture half way
                        self._step_device(stop_time)
                        raise RuntimeError("capture refused (CASTRO_AMD_TEST_FAIL_CAPTURE)")
                    self._step_device(stop_time)
                    self._step_device(stop_time)
            except BaseException:
                self._restore_host_step_state(keep)        # nothing was executed: the host-side roles go back too
                raise
            finally:
                if gc_was_on:
                    gc.enable()
            self._graphs[key] = g
        return self._graphs[key]

    def _rank_graph_ok(self):
        """A distributed run replays a per-rank hipGraph of a pair of steps -- pack, the grouped ncclSend / ncclRecv, unpack, BC
        fill, the hydro kernels, ncclAllReduce(MIN), k_step_control -- when every collective of the step is issued by the kernel
        library on its own RCCL communicator (castro_amd_fill_boundary / castro_amd_allreduce_min on the capturing stream) and
        no rank has failed a capture before.  CASTRO_AMD_STEP_GRAPH_RCCL=0 keeps the stream form."""
        if os.environ.get("CASTRO_AMD_STEP_GRAPH_RCCL", "1") == "0" or getattr(self, "_rank_graph_failed", False):
            return False
        plan = self._plans.get(id(self.neighbors))
        have_c = bool(getattr(self.comm, "_ccomm", None)) and os.environ.get("CASTRO_AMD_C_ALLREDUCE", "1") != "0"
        return bool(getattr(self.comm, "device_side", False) and have_c and (not self.neighbors or (plan is not None and "cplan" in plan)))

    _HOST_STEP_STATE = ("S_old_b", "S_new_b", "_flux_clear", "_post_clean_done", "_whole_step", "_in_retry", "_pending_cleans")

    def _host_step_state(self):
        return {k: getattr(self, k) for k in self._HOST_STEP_STATE}

    def _restore_host_step_state(self, keep):
        for k, v in keep.items():
            setattr(self, k, v)

    def _capture_rank_graph(self, stop_time):
        """capture_step_graph on every rank, then one all-reduce (torch.distributed, outside any capture) to agree on the
        outcome: the graph is used only if EVERY rank has one (a rank replaying a graph while another issues the stream form would
        still match call for call, but a rank whose runtime refused the capture should not be the only one on the slow path
        unnoticed).  Returns the graph or None (stream form from now on)."""
        g, ok = None, 1
        # _step_device changes host-side state while it captures (the roles of the two state buffers, the flux-assign and
        # post-clean flags): a capture that dies half way has executed nothing on the device, so the object must be put back
        # exactly as it was or the stream-ordered steps that follow would start from the wrong buffer
        keep = self._host_step_state()
        try:
            g = self.capture_step_graph(stop_time)
        except Exception as e:                      # the runtime or RCCL refused the capture
            ok = 0
            self._rank_graph_error = "%s: %s" % (type(e).__name__, e)
            self._restore_host_step_state(keep)
            torch.cuda.synchronize()
        flag = torch.tensor([ok], dtype=torch.int32, device=self.red.device)
        self.comm.dist.all_reduce(flag, op=self.comm.dist.ReduceOp.MIN, group=self.comm.group)
        if int(flag.item()) != 1:
            self._rank_graph_failed = True
            self._graphs = {}
            if self.comm.rank == 0:
                import sys
                print("castro_amd: per-rank step graph not available on every rank (%s); stream-ordered steps instead"
                      % getattr(self, "_rank_graph_error", "another rank failed"), file=sys.stderr)
            return None
        return g

    def prepare_step_graph(self, stop_time=-1.0):
        """Capture (untimed) the step graph run_steps would use for the buffers' current roles: one rank, or RCCL ranks
        (_rank_graph_ok; collective: every rank calls it).  Returns True if a graph is in place."""
        if os.environ.get("CASTRO_AMD_STEP_GRAPH", "1") == "0" or not self.host_free_ok():
            return False
        self._ensure_ctl()
        if not self._eager_done:
            return False
        if isinstance(self.comm, SingleComm):
            return self.capture_step_graph(stop_time) is not None
        return self._rank_graph_ok() and self._capture_rank_graph(stop_time) is not None

    def run_steps(self, nsteps, stop_time=-1.0, graph=None):
        """`nsteps` coarse steps with ONE host synchronisation at the end instead of one per step: the time step lives in a
        device vector (castro_amd_step_control), the kernels read it from there, a rejected step latches a status that
        is raised here.  With graph (default: at least 4 steps; a single rank, or RCCL ranks whose collectives the kernel library
        issues itself: _rank_graph_ok) a pair of steps -- the two roles of the ping-pong state buffers -- is captured in a
        hipGraph once, per rank, and replayed.  Falls back to step() when host_free_ok()
        is false.  Bit-identical to step(): the same kernels with the same dt, computed by the same expressions."""
        if nsteps <= 0:
            return
        if not self.host_free_ok():
            for _ in range(nsteps):
                self.step(stop_time)
            return
        dt0 = self.computeInitialDt(stop_time) if self.nstep == 0 else self.computeNewDt(self.dt, stop_time, est=self._next_est)
        self._ensure_ctl()
        head = torch.zeros(L.CTL_HIST, dtype=torch.float64)
        head[L.CTL_DT], head[L.CTL_TIME], head[L.CTL_NSTEP] = dt0, self.time, float(self.nstep)
        # with castro.use_retry the advance is subcycle_advance_ctu's single subcycle, (time + dt) - time
        head[L.CTL_DTHYDRO] = ((self.time + dt0) - self.time) if self.use_retry else dt0
        self._ctl[:L.CTL_HIST].copy_(head)
        self.red.fill_(1.e200)
        n0 = self.nstep
        if graph is None:
            graph = nsteps >= 4 and os.environ.get("CASTRO_AMD_STEP_GRAPH", "1") != "0" and (
                isinstance(self.comm, SingleComm) or self._rank_graph_ok())
        left = nsteps
        if graph:
            if not self._eager_done:
                self._step_device(stop_time)            # two eager steps first: every lazy allocation happens outside a capture
                self._step_device(stop_time)
                left -= 2
            if isinstance(self.comm, SingleComm):
                try:
                    g = self.capture_step_graph(stop_time)
                except AdvanceFailure:
                    raise
                except Exception as e:              # the runtime refused the capture: the object is as it was (capture_step_graph
                    g = None                        # restores the host-side roles), the steps go out stream-ordered
                    self._graph_error = "%s: %s" % (type(e).__name__, e)
                    torch.cuda.synchronize()
            else:
                g = self._capture_rank_graph(stop_time)
            while g is not None and left >= 2:
                g.replay()
                left -= 2
        for _ in range(left):
            self._step_device(stop_time)
        v = self._ctl.tolist()                      # the one synchronisation
        status, done = int(v[L.CTL_STATUS]), int(v[L.CTL_NSTEP]) - n0
        self.time, self.nstep = v[L.CTL_TIME], int(v[L.CTL_NSTEP])
        if done > 0:
            self.dt = v[L.CTL_HIST + (self.nstep - 1) % L.CTL_NHIST]
            self.lastDt = self.dt
            k = min(done, L.CTL_NHIST)
            self.dt_history = [v[L.CTL_HIST + (self.nstep - k + m) % L.CTL_NHIST] for m in range(k)]
        self._next_est, self._device_next_dt = None, v[L.CTL_DT]
        if status:
            # The launches after the rejected step wrote nothing (they check the latched status): its old state is intact
            # in whichever buffer held it.  `nsteps - done` steps were issued from the rejected one on, each swapping roles.
            if (nsteps - done) % 2 == 0:
                self._swap_state_time_levels()
            dt_failed = v[L.CTL_DT]                 # not advanced on a rejected step
            rho_min = v[L.CTL_RHOMIN]
            why = ("negative density" if rho_min < 0.0 else "small density") + " (density = %e)" % rho_min if status & 1 \
                else "timestep validity check failed"
            if not self.use_retry:
                raise AdvanceFailure("Advance was unsuccessful: %s (step %d of a host-free batch)" % (why, self.nstep + 1))
            # Castro::retry_advance_ctu on the host: the step again from its (already cleaned) old state; the first
            # attempt repeats the rejected one, then the reference's subcycling takes over
            self.dt = dt_failed
            self._zero_fluxes()
            self._pending_cleans, self._post_clean_done = 0, False
            self.nsubcycles, self.nretries = 1, 0
            self._next_est = self.subcycle_advance_ctu(self.time, self.dt)
            if not self._post_clean_done:
                self.clean_state(self.S_new_b, 1)
                self._next_est = None
            self.time += self.dt
            self.nstep += 1
            self.run_steps(nsteps - done - 1, stop_time, graph=graph)

    def evolve(self, stop_time, max_step=10 ** 9, host_free=None):
        """Amr::coarseTimeStep until stop_time (or max_step).  host_free (default: whenever host_free_ok()): the steps go out
        in batches through run_steps, each batch as long as the steps are certain not to reach stop_time -- dt grows by at
        most change_max per step, so k steps cover at most dt * cm * (cm^k - 1) / (cm - 1) -- with one host synchronisation per
        batch; the last steps, where the clipping to stop_time decides, go one by one.  Same states, same dt sequence."""
        import math
        eps = 2.220446049250313e-16
        if host_free is None:
            host_free = self.host_free_ok()
        while self.nstep < max_step and self.time < stop_time - eps:
            k = 0
            if host_free and self.nstep > 0 and self.fixed_dt <= 0.0 and self.dt > 0.0:
                cm = self.params.change_max
                remaining = stop_time - self.time
                if cm > 1.0:
                    # largest k with dt * cm * (cm^k - 1) / (cm - 1) < 0.99 * remaining
                    x = 1.0 + 0.99 * remaining * (cm - 1.0) / (self.dt * cm)
                    k = int(math.floor(math.log(x) / math.log(cm))) if x > 1.0 else 0
                else:
                    k = int(0.99 * remaining / self.dt)
                k = min(k, max_step - self.nstep, 4 * L.CTL_NHIST)
            if k >= 2:
                self.run_steps(k, stop_time)
            else:
                self.step(stop_time)
        return self.nstep
