"""AMR with subcycling: a coarse level covering the domain and any number of refined levels (ratio 2), each a list of
boxes.

SURVEY.md 8 f-3.  What the reference does through AMReX's Amr / AmrLevel / FluxRegister / Interpolater / grid
generation classes [3P] is orchestrated here on top of `Castro` objects, one per box:

  Amr::timeStep recursion with subcycling        coarse advance, then two fine advances of dt/2, down the hierarchy
  AmrLevel::FillPatch (fine level)               cell_cons_interp of the time-interpolated coarse state
                                                 (Castro_setup.cpp:352-364, Castro.cpp:4201-4209), clean_state of those
                                                 ghost zones (Castro_advance.cpp:186), then valid data of the other
                                                 boxes of the level, then the physical boundary fill
  Castro::FluxRegCrseInit / FluxRegFineAdd       Castro.cpp:2487-2545 -- one register per face of every fine box
  Castro::post_timestep: reflux, avgDown,        Castro.cpp:2549-2700, 3096-3113, post_timestep (:2140-2260); a coarse
      clean_state                                zone under another fine box is refluxed too and then overwritten by
                                                 avgDown, like in AMReX
  Castro::computeNewDt / computeInitialDt        Castro.cpp:1629-1866 over all levels (n_cycle = 1, 2, 2, ...)
  Castro::errorEst + Amr::regrid / grid_places   amr.refinement_indicators (AMRErrorTag [3P] restated) on every level
                                                 below max_level, whenever a level has taken regrid_int steps since
                                                 its last regrid (Amr::level_count; grids above it): tags from the finest
                                                 level down, buffered by n_error_buf, united with the (coarsened,
                                                 buffered) new boxes of the level above (proper nesting), clustered
                                                 into boxes (castro_amd/cluster.py: Berger-Rigoutsos with grid_eff,
                                                 blocking_factor, max_grid_size; `cluster=False`: one bounding box per
                                                 level).  New boxes take the data of the old boxes of their level
                                                 where they overlap and interpolated coarse data elsewhere; a regrid
                                                 adds at most one level; Amr::bldFineLevels at start-up.

Fixed hierarchies: `patches=[...]`, one entry per refined level: a box (lo, hi) or a list of boxes, in the zones of
the level below.  Level 0 is one box (per rank; this driver is single-rank).
Periodic domains: the periodic images of the boxes of a level enter the same-level copies and the reflux as boxes shifted
by the domain extent.  Constant gravity and rotation sources on every level (the Source_Type FillPatch of a refined level
interpolates the coarse sources like the state).  Not provided: multi-rank AMR.  The interpolation,
flux-register and clustering arithmetic is AMReX's, restated from its published description
(include/castro_hydro_amd.h, castro_amd/cluster.py): parity with an AMReX build is unpinned.
"""
import itertools
import os

import numpy as np
import torch

from . import _lib as L
from . import cluster as CL
from .castro import Castro, NUM_GROW, NUM_STATE, checked_estimate

NSRC = 7                     # components of Source_Type (Castro_setup.cpp:317-327)


def _coarsen(i):
    return i // 2            # floor division is AMReX's coarsen() for negative indices too


def _shift(box, sh):
    """The box (lo, hi) moved by sh zones.  A FAB described with a shifted box is read at index - sh: this is how the
    periodic images of a box enter the same-index copy / reflux operations without any kernel knowing about them."""
    return tuple(box[0][d] + sh[d] for d in range(3)), tuple(box[1][d] + sh[d] for d in range(3))


def _is_box(x):
    return len(x) == 2 and len(x[0]) == 3 and not hasattr(x[0][0], "__len__")


class _Patch(Castro):
    """One box of a refined level: FillPatch takes ghost zones from the coarser level and from its siblings."""

    def bind(self, level, pbox):
        self.level, self.pbox = level, pbox
        if pbox is None:                       # a box of a level-0 that is cut into several boxes: nothing coarser
            self.cbox, self.ctmp, self.shell, self.regs = None, None, [], {}
            if self.have_sources:
                self.new_source_g = self.hydro.alloc(NSRC, *self.sbox) if self.owned else None
            return
        # coarse zones under the grown fine box, grown by one for the slopes
        self.cbox = (tuple(_coarsen(self.glo[d]) - 1 for d in range(3)), tuple(_coarsen(self.ghi[d]) + 1 for d in range(3)))
        self.ctmp = self.hydro.alloc(NUM_STATE, *self.cbox) if self.owned else None
        lo, hi = self.lo, self.hi
        glo, ghi = self.glo, self.ghi
        self.shell = [((glo[0], glo[1], glo[2]), (ghi[0], ghi[1], lo[2] - 1)), ((glo[0], glo[1], hi[2] + 1), (ghi[0], ghi[1], ghi[2])),
                      ((glo[0], glo[1], lo[2]), (ghi[0], lo[1] - 1, hi[2])), ((glo[0], hi[1] + 1, lo[2]), (ghi[0], ghi[1], hi[2])),
                      ((glo[0], lo[1], lo[2]), (lo[0] - 1, hi[1], hi[2])), ((hi[0] + 1, lo[1], lo[2]), (ghi[0], hi[1], hi[2]))]
        if self.have_sources:
            self._bind_sources()
        # flux registers: the coarse faces on the six sides of this box
        plo, phi = pbox
        self.regs = {}
        for d in range(3):
            for side in (0, 1):
                rlo, rhi = list(plo), list(phi)
                rlo[d] = rhi[d] = (plo[d] if side == 0 else phi[d] + 1)
                self.regs[(d, side)] = (self.hydro.alloc(NUM_STATE, rlo, rhi) if self.owned else None, (tuple(rlo), tuple(rhi)))

    def _bind_sources(self):
        """Source_Type data (NUM_GROW_SRC ghost zones): the coarse zones under the grown box, and the ghost shell."""
        slo, shi = self.sbox
        self.scbox = (tuple(_coarsen(slo[d]) - 1 for d in range(3)), tuple(_coarsen(shi[d]) + 1 for d in range(3)))
        self.stmp = self.hydro.alloc(NSRC, *self.scbox) if self.owned else None
        self.new_source_g = self.hydro.alloc(NSRC, *self.sbox) if self.owned else None
        lo, hi = self.lo, self.hi
        self.sshell = [((slo[0], slo[1], slo[2]), (shi[0], shi[1], lo[2] - 1)), ((slo[0], slo[1], hi[2] + 1), (shi[0], shi[1], shi[2])),
                       ((slo[0], slo[1], lo[2]), (shi[0], lo[1] - 1, hi[2])), ((slo[0], hi[1] + 1, lo[2]), (shi[0], shi[1], hi[2])),
                       ((slo[0], lo[1], lo[2]), (lo[0] - 1, hi[1], hi[2])), ((hi[0] + 1, lo[1], lo[2]), (shi[0], hi[1], hi[2]))]

    def expand_state(self, S, box=None, neighbors=None):
        assert box is None, "the Source_Type FillPatch of a refined level is level-wide: _Level.fill_source"
        self.level.fill_box(self, S)


class _Level:
    """The boxes of one level and the level-wide parts of Castro::advance."""

    def __init__(self, amr, l, boxes):
        self.amr, self.l, self.boxes = amr, l, list(boxes)
        self.mine = [b for b in self.boxes if b.owned]        # the boxes this rank advances (all of them on one rank)
        self.alpha = 0.0          # (t_level - t_parent_old) / dt_parent of the FillPatch being prepared
        b0 = self.boxes[0]
        self.hydro, self.params, self.geom = b0.hydro, b0.params, b0.geom
        self.red = self.mine[0].red if self.mine else self.hydro.alloc(1, (0, 0, 0), (2, 0, 0)).reshape(3)
        for b in self.mine:
            b.red = self.red      # one [min dt, min rho] pair for the level: every box reduces into it
            b.fuse_post_clean = False      # post_timestep's clean_state comes after reflux and avgDown (_time_step)
        self._pending_cleans, self._post_clean_done, self._whole_step = 2, False, False
        for k in ("use_retry", "retry_subcycle_factor", "max_subcycles", "dt_cutoff", "max_dt", "fixed_dt", "have_sources"):
            setattr(self, k, getattr(b0, k))
        self.fuse_clean = b0.fuse_clean
        self.nsubcycles, self.nretries, self.last_failure = 0, 0, ""
        self.time, self.nstep = 0.0, 0
        # castro.source_term_predictor: Castro::lastDt of this level (1e200 in every constructor, Castro.cpp:906 -- a level made
        # by a regrid starts with it again) and the in_retry flag of subcycle_advance_ctu
        self.lastDt, self._in_retry = 1.e200, False

    def __getattr__(self, name):
        # a level of one box answers for its box (S_new(), lo, hi, n, gbox, S_new_b, ...)
        boxes = self.__dict__.get("boxes", ())
        if len(boxes) == 1 and boxes[0].owned:
            return getattr(boxes[0], name)
        raise AttributeError("%s (level %d has %d boxes)" % (name, self.__dict__.get("l", -1), len(boxes)))

    def box_list(self):
        return [b.bx for b in self.boxes]

    # ---- static overlap tables (rebuilt at every regrid) -----------------------------------------
    def bind(self):
        self.plain_base = self.l == 0 and len(self.boxes) == 1 and not isinstance(self.boxes[0], _Patch)
        if self.plain_base:                    # the whole domain in one Castro box: it fills its own ghost zones
            return
        parents = self.amr.lev[self.l - 1].boxes if self.l > 0 else []
        # periodic images: shifts by the domain extent (in zones of this level / of the parent level)
        per = self.amr.periodic
        ext = [(2 ** self.l) * self.amr.n_cell[d] for d in range(3)]
        shifts = list(itertools.product(*[(-ext[d], 0, ext[d]) if per[d] else (0,) for d in range(3)]))
        cshifts = [tuple(x // 2 for x in sh) for sh in shifts]
        for b in self.boxes:
            b.sib = [(s, it, sh) for s in self.boxes for sh in shifts if not (s is b and sh == (0, 0, 0))
                     for it in [CL.intersect(b.gbox, _shift(s.bx, sh))] if it]
            b.at_domain_edge = any(b.glo[d] < b.geom.domlo[d] or b.ghi[d] > b.geom.domhi[d] for d in range(3))
            if self.l == 0:                    # no coarser level: no coarse data, no flux registers, nothing to average onto
                b.csrc, b.csrc_valid, b.crse_init, b.reflux_to, b.avg_to = [], [], {}, {}, []
                if self.have_sources:
                    b.ssib = [(sb, it, sh) for sb in self.boxes for sh in shifts if not (sb is b and sh == (0, 0, 0))
                              for it in [CL.intersect(b.sbox, _shift(sb.bx, sh))] if it]
                    b.ssrc, b.ssrc_valid, b.sshell = [], [], []
                continue
            b.csrc = [(p, it) for p in parents for it in [CL.intersect(b.cbox, p.gbox)] if it]
            if len(b.csrc) == 1:
                assert b.csrc[0][1] == b.cbox, "box not properly nested: its ghost zones need parent data beyond the parent's own ghost zones"
                b.csrc_valid = []
            else:
                # several parents: ghost zones first, valid zones last, so that valid data wins
                b.csrc_valid = [(p, it) for p in parents for it in [CL.intersect(b.cbox, p.bx)] if it]
                cov = np.zeros(tuple(b.cbox[1][d] - b.cbox[0][d] + 1 for d in (2, 1, 0)), dtype=bool)
                for _, (lo, hi) in b.csrc:
                    o = b.cbox[0]
                    cov[lo[2] - o[2]:hi[2] - o[2] + 1, lo[1] - o[1]:hi[1] - o[1] + 1, lo[0] - o[0]:hi[0] - o[0] + 1] = True
                assert cov.all(), "box not properly nested in the union of its parents"
            # FluxRegCrseInit sources and reflux targets per register
            b.crse_init, b.reflux_to = {}, {}
            for (d, side), (reg, rbox) in b.regs.items():
                b.crse_init[(d, side)] = [(p, it) for p in parents for it in [CL.intersect(rbox, p.flux_boxes[d])] if it]
                sh = -1 if side == 0 else 0           # the coarse zone outside the fine box: face - 1 (low side) or face
                zlo, zhi = list(rbox[0]), list(rbox[1])
                zlo[d] += sh; zhi[d] += sh
                tg = []
                for p in parents:
                    for csh in cshifts:                  # a coarse zone beyond a periodic boundary is its image inside
                        it = CL.intersect((tuple(zlo), tuple(zhi)), _shift(p.bx, csh))
                        if it:
                            flo, fhi = list(it[0]), list(it[1])
                            flo[d] -= sh; fhi[d] -= sh
                            tg.append((p, (tuple(flo), tuple(fhi)), csh))
                b.reflux_to[(d, side)] = tg
                # proper nesting (AMReX's Amr::grid_places guarantees it): every parent-level zone next to a face of
                # this box is a valid zone of the parent level (or lies outside a non-periodic domain); a face that
                # touched a still coarser level directly would lose its flux correction
                cdom = [(2 ** (self.l - 1)) * self.amr.n_cell[x] for x in range(3)]
                if not per[d]:
                    zlo[d], zhi[d] = max(zlo[d], 0), min(zhi[d], cdom[d] - 1)
                want = int(np.prod([max(zhi[x] - zlo[x] + 1, 0) for x in range(3)]))
                have = sum(int(np.prod([fhi[x] - flo[x] + 1 for x in range(3)])) for _, (flo, fhi), _ in tg)
                assert have == want, "box %s of level %d not properly nested: zones of level %d next to its %s face " \
                    "belong to no box of that level" % (b.bx, self.l, self.l - 1, "xyz"[d] + "-+"[side])
            b.avg_to = [(p, it) for p in parents for it in [CL.intersect(b.pbox, p.bx)] if it]
            if self.have_sources:                                # the same tables for the Source_Type FillPatch
                b.ssib = [(sb, it, sh) for sb in self.boxes for sh in shifts if not (sb is b and sh == (0, 0, 0))
                          for it in [CL.intersect(b.sbox, _shift(sb.bx, sh))] if it]
                b.ssrc = [(p, it) for p in parents for it in [CL.intersect(b.scbox, p.sbox)] if it]
                b.ssrc_valid = [(p, it) for p in parents for it in [CL.intersect(b.scbox, p.bx)] if it]
        self._op_cache = {}
        self.batched = hasattr(self.hydro, "make_ops") and self.amr.nranks == 1
        if self.batched and self.l > 0:
            # flux-register operations never change between regrids (registers and flux FABs keep their storage)
            mk = self.hydro.make_ops
            self.ops_fine_add = mk([(L.OP_FLUXREG_FINE_ADD, d, NUM_STATE, rbox[0], rbox[1], 1.0, 0.0, (reg, rbox),
                                     (b.fluxes[d], b.flux_boxes[d]), None)
                                    for b in self.boxes for (d, side), (reg, rbox) in b.regs.items()])
            self.ops_crse_init = mk([(L.OP_FLUXREG_CRSE_INIT, d, NUM_STATE, lo, hi, -1.0, 0.0, (reg, rbox),
                                      (p.fluxes[d], p.flux_boxes[d]), None)
                                     for b in self.boxes for (d, side), (reg, rbox) in b.regs.items()
                                     for p, (lo, hi) in b.crse_init[(d, side)]])

    # ---- AmrLevel::FillPatch ---------------------------------------------------------------------
    def _interp_ghosts(self, b, S):
        h, a = self.hydro, self.alpha
        # StateData time interpolation of the coarse data: (1 - a) old + a new
        for p, (lo, hi) in b.csrc + b.csrc_valid:
            h.lincomb(b.ctmp, b.cbox, 1.0 - a, p.S_old_b, p.gbox, a, p.S_new_b, p.gbox, NUM_STATE, lo, hi)
        if hasattr(h, "fillpatch_shell"):                       # both passes over the six slabs in one launch
            h.fillpatch_shell(b.ctmp, b.cbox, S, b.gbox, b.lo, b.hi, NUM_GROW, b.params, ntimes=1)
            return
        for lo, hi in b.shell:
            h.cc_interp(b.ctmp, b.cbox, S, b.gbox, lo, hi, NUM_STATE)
        for lo, hi in b.shell:                                 # clean_state(Sborder) reaches the ghost zones too
            h.clean_state(S, b.gbox, lo, hi, b.params, ntimes=1)

    def _copy_siblings(self, b, S, which):
        h = self.hydro
        for s, (lo, hi), sh in b.sib:
            h.copy(S, b.gbox, getattr(s, which), _shift(s.gbox, sh), lo, hi)
        h.bc_fill(S, b.gbox, b.geom)                           # fine zones outside the domain

    def _cached_ops(self, key, ptrs, build):
        """Operation tables hold raw pointers: S_old_b / S_new_b swap at every advance, so a table is kept per
        pointer configuration (two in the steady state)."""
        ent = self._op_cache.get((key, ptrs))
        if ent is None:
            if len(self._op_cache) > 256:
                self._op_cache.clear()
            ent = build()
            self._op_cache[(key, ptrs)] = ent
        return ent

    def fill(self, which):
        """Ghost zones of S_old_b / S_new_b (`which`) of every box of the level."""
        if self.plain_base:
            for b in self.mine:
                b.expand_state(getattr(b, which))
            return
        if self.amr.nranks > 1:
            return self._fill_ranks(which)
        if not self.batched:
            for b in self.boxes:
                if self.l > 0:
                    self._interp_ghosts(b, getattr(b, which))
            for b in self.boxes:                               # valid zones are final only after every box's clean pass
                self._copy_siblings(b, getattr(b, which), which)
            return
        h, a = self.hydro, self.alpha
        parents = self.amr.lev[self.l - 1].boxes if self.l > 0 else []
        # 1. time-interpolated coarse data under every box, all boxes in a few launches.  Where a box has several parents
        #    their grown boxes overlap: the ghost-zone pass goes first (overlapping entries carry identical values: a
        #    ghost zone of one parent is a copy of the valid zone of another or the same interpolation), the valid-zone
        #    pass second, as separate launches
        pp = tuple(t.data_ptr() for p in parents for t in (p.S_old_b, p.S_new_b))
        for key, attr in ((("lincomb_ghost", "csrc"), ("lincomb_valid", "csrc_valid")) if self.l > 0 else ()):
            ops = self._cached_ops((key,), pp, lambda attr=attr: h.make_ops(
                [(L.OP_LINCOMB, 0, NUM_STATE, lo, hi, 0.0, 0.0, (b.ctmp, b.cbox), (p.S_old_b, p.gbox), (p.S_new_b, p.gbox))
                 for b in self.boxes for p, (lo, hi) in getattr(b, attr)]))
            arr, n = ops
            for i in range(n):
                arr[i].a, arr[i].b = 1.0 - a, a
            h.fab_ops(ops)
        # 2. interpolation + clean_state of the ghost shells: one launch for the level (six slab operations per box)
        if self.l > 0 and self._level_calls():
            sp2 = tuple(getattr(b, which).data_ptr() for b in self.boxes)
            h.fab_ops(self._cached_ops(("shell", which), sp2, lambda: h.make_ops(
                [(L.OP_INTERP_CLEAN, 0, NUM_STATE, lo, hi, 1.0, 0.0, (getattr(b, which), b.gbox), (b.ctmp, b.cbox), None)
                 for b in self.boxes for lo, hi in b.shell if all(hi[d] >= lo[d] for d in range(3))])), params=self.params)
        else:
            for b in (self.boxes if self.l > 0 else ()):
                h.fillpatch_shell(b.ctmp, b.cbox, getattr(b, which), b.gbox, b.lo, b.hi, NUM_GROW, b.params, ntimes=1)
        # 3. valid zones of the siblings (final only after every box's clean pass), all boxes in a few launches
        sp = tuple(getattr(b, which).data_ptr() for b in self.boxes)
        h.fab_ops(self._cached_ops(("sib", which), sp, lambda: h.make_ops(
            [(L.OP_COPY, 0, NUM_STATE, lo, hi, 0.0, 0.0, (getattr(b, which), b.gbox), (getattr(sb, which), _shift(sb.gbox, sh)), None)
             for b in self.boxes for sb, (lo, hi), sh in b.sib])))
        # 4. physical boundaries
        for b in self.boxes:
            if b.at_domain_edge:
                h.bc_fill(getattr(b, which), b.gbox, b.geom)

    def _fill_ranks(self, which):
        """fill() with the boxes of this level and of the parent level spread over ranks: the same four passes, every
        box-to-box transfer through CastroAmr._xrun (local where both boxes are here, one message otherwise)."""
        h, a, X = self.hydro, self.alpha, self.amr._xrun
        X([("lincomb", b, p, lo, hi, a) for b in self.boxes for p, (lo, hi) in b.csrc])           # ghost zones of the parents first,
        X([("lincomb", b, p, lo, hi, a) for b in self.boxes for p, (lo, hi) in b.csrc_valid])     # valid zones last
        for b in (self.mine if self.l > 0 else ()):
            h.fillpatch_shell(b.ctmp, b.cbox, getattr(b, which), b.gbox, b.lo, b.hi, NUM_GROW, b.params, ntimes=1) \
                if hasattr(h, "fillpatch_shell") else self._interp_shell(b, getattr(b, which))
        X([("copy", b, sb, lo, hi, (which, sh)) for b in self.boxes for sb, (lo, hi), sh in b.sib])
        for b in self.mine:
            h.bc_fill(getattr(b, which), b.gbox, b.geom)

    def _interp_shell(self, b, S):
        for lo, hi in b.shell:
            self.hydro.cc_interp(b.ctmp, b.cbox, S, b.gbox, lo, hi, NUM_STATE)
        for lo, hi in b.shell:
            self.hydro.clean_state(S, b.gbox, lo, hi, b.params, ntimes=1)

    def fill_box(self, b, S):
        which = "S_new_b" if S is b.S_new_b else "S_old_b"
        assert S is getattr(b, which)
        if self.l > 0:
            self._interp_ghosts(b, S)
        self._copy_siblings(b, S, which)

    # ---- AmrLevel::FillPatch of Source_Type (Castro_advance_ctu.cpp:138-140) ------------------------------
    def fill_source(self, name):
        """Ghost zones of the Source_Type data `name` (old_source / new_source_g) of every box: coarse Source_Type data
        interpolated in time ((1 - alpha) old + alpha new, StateData's rule) and space (cell_cons_interp,
        Castro_setup.cpp:317-327), valid data of the other boxes of the level, physical boundaries."""
        if self.plain_base:
            for b in self.mine:
                b.expand_state(getattr(b, name), b.sbox, b.src_neighbors)
            return
        h, a = self.hydro, self.alpha
        if self.amr.nranks > 1:
            X = self.amr._xrun
            X([("src_lincomb", b, p, lo, hi, a) for b in self.boxes for p, (lo, hi) in b.ssrc + b.ssrc_valid], NSRC)
            for b in self.mine:
                for lo, hi in b.sshell:
                    h.cc_interp(b.stmp, b.scbox, getattr(b, name), b.sbox, lo, hi, NSRC)
            X([("src_copy", b, sb, lo, hi, (name, sh)) for b in self.boxes for sb, (lo, hi), sh in b.ssib], NSRC)
            for b in self.mine:
                h.bc_fill(getattr(b, name), b.sbox, b.geom)
            return
        if self.batched and self._level_calls():
            # The same four passes with every box of the level in one launch each (as fill() does for the state): a level of
            # 56 boxes made ~1000 launches here per advance.  Source_Type tensors keep their storage between regrids: one table
            # per (pass, name), the time-interpolation weights rewritten per call.
            mk, P = h.make_ops, self.params
            for key, attr in ((("src_lincomb_ghost", "ssrc"), ("src_lincomb_valid", "ssrc_valid")) if self.l > 0 else ()):
                ops = self._cached_ops((key,), (), lambda attr=attr: mk(
                    [(L.OP_LINCOMB, 0, NSRC, lo, hi, 0.0, 0.0, (b.stmp, b.scbox), (p.old_source, p.sbox), (p.new_source_g, p.sbox))
                     for b in self.boxes for p, (lo, hi) in getattr(b, attr)]))
                arr, n = ops
                for i in range(n):
                    arr[i].a, arr[i].b = 1.0 - a, a
                h.fab_ops(ops, params=P)
            if self.l > 0:
                h.fab_ops(self._cached_ops(("src_shell", name), (), lambda: mk(
                    [(L.OP_INTERP, 0, NSRC, lo, hi, 0.0, 0.0, (getattr(b, name), b.sbox), (b.stmp, b.scbox), None)
                     for b in self.boxes for lo, hi in b.sshell if all(hi[d] >= lo[d] for d in range(3))])), params=P)
            h.fab_ops(self._cached_ops(("src_sib", name), (), lambda: mk(
                [(L.OP_COPY, 0, NSRC, lo, hi, 0.0, 0.0, (getattr(b, name), b.sbox), (getattr(sb, name), _shift(sb.sbox, sh)), None)
                 for b in self.boxes for sb, (lo, hi), sh in b.ssib])), params=P)
            for b in self.boxes:
                if b.at_domain_edge:
                    h.bc_fill(getattr(b, name), b.sbox, b.geom)
            return
        for b in self.boxes:
            for p, (lo, hi) in b.ssrc + b.ssrc_valid:           # ghost zones of the parents first, valid zones last
                h.lincomb(b.stmp, b.scbox, 1.0 - a, p.old_source, p.sbox, a, p.new_source_g, p.sbox, NSRC, lo, hi)
            for lo, hi in b.sshell:
                h.cc_interp(b.stmp, b.scbox, getattr(b, name), b.sbox, lo, hi, NSRC)
        for b in self.boxes:
            for sb, (lo, hi), sh in b.ssib:
                h.copy(getattr(b, name), b.sbox, getattr(sb, name), _shift(sb.sbox, sh), lo, hi)
            h.bc_fill(getattr(b, name), b.sbox, b.geom)

    def fill_new_source(self):
        """The new-time Source_Type data with ghost zones, for the FillPatch of the next finer level."""
        h = self.hydro
        for b in self.mine:
            h.copy(b.new_source_g, b.sbox, b.new_source, b.bx, b.lo, b.hi)
        self.fill_source("new_source_g")

    def create_source_corrector(self):
        """Castro::create_source_corrector on a level of boxes (Castro.cpp:3780-3818): AmrLevel::FillPatch(Source_Type at its old
        time -- after the swap that is the new-time corrector of the LAST advance --, components UMX .. UMZ, NUM_GROW_SRC ghost
        zones): the valid zones of every box from its own data, ghost zones from the other boxes of the level, from the coarser
        level's Source_Type data interpolated in time and space (the FillPatch of fill_source) and from the physical boundaries;
        then x 2 / lastDt.  The factor dt / 2 is applied in src_to_prim (Castro_ctu.cpp:493-497)."""
        g = 3                                   # NUM_GROW_SRC
        for b in self.mine:
            c, n = b.source_corrector, b.n
            c.zero_()
            c[1:4, g:g + n[2], g:g + n[1], g:g + n[0]] = b.new_source[1:4]
        self.fill_source("source_corrector")
        for b in self.mine:
            c = b.source_corrector
            c[0].zero_()                        # the FillPatch of the reference moves the three momentum components only
            c[4:].zero_()
            c.mul_(2.0 / self.lastDt)

    def _advance_with_sources(self, time, dt):
        if self.params.source_term_predictor == 1 and not self._in_retry:
            # not on the attempt that follows a rejected one: the data it is made from are gone by then (Castro_advance_ctu.cpp:60-62)
            self.create_source_corrector()
        return self._advance_with_sources_impl(time, dt)

    def _advance_with_sources_impl(self, time, dt):
        """do_advance_ctu with old- and new-time gravity / rotation sources (Castro_advance_ctu.cpp:94-143, 156-274),
        stage by stage over the boxes of the level; the per-box arithmetic is Castro._do_advance_with_sources'."""
        h = self.hydro
        fused = hasattr(h, "apply_source")
        lvl = self._source_level_calls()        # the per-box stages below as one library call per level (castro_amd_sources_mf)
        if lvl is not None:
            sp = tuple(t.data_ptr() for b in self.mine for t in (b.S_old_b, b.S_new_b))
            h.sources_mf(0, self._cached_ops(("src_old",), sp, lambda: h.make_source_boxes(
                [(b.lo, b.hi, (b.S_old_b, b.gbox), (b.S_new_b, b.gbox), (b.old_source, b.sbox), b.mass_fluxes, b.flux_boxes)
                 for b in self.mine])), lvl.grav if lvl.do_grav else None, lvl.grav_source_type if lvl.do_grav else 4, lvl.rotation,
                         lvl.geom, self.params, dt, ntimes=1)
        for b in (self.mine if lvl is None else ()):
            S, lo, hi = b.S_old_b, b.lo, b.hi
            b.old_source.zero_()
            if b.do_grav:
                h.old_gravity_source(S, b.gbox, b.old_source, b.sbox, lo, hi, b.grav, b.grav_source_type, dt)
            if b.rotation is not None:
                h.old_rotation_source(S, b.gbox, b.old_source, b.sbox, lo, hi, b.rotation, b.geom, dt)
            if fused:
                h.apply_source(b.S_new_b, b.gbox, S, b.gbox, dt, b.old_source, b.sbox, NSRC, lo, hi, b.params, ntimes=1)
            else:
                h.copy(b.S_new_b, b.gbox, S, b.gbox, lo, hi)
                h.saxpy(b.S_new_b, b.gbox, dt, b.old_source, b.sbox, NSRC, lo, hi)
                h.clean_state(b.S_new_b, b.gbox, lo, hi, b.params, ntimes=1)
        self.fill_source("old_source")
        predictor = self.params.source_term_predictor == 1
        def hydro(b):
            # b.hydro is the context this box's call runs on (one of the stream pool's, _hydro_calls)
            if predictor:
                b.hydro.set_source_corrector(b.source_corrector, b.sbox)
            try:
                b.construct_ctu_hydro_source(time, dt, src=b.old_source)
            finally:
                if predictor:
                    b.hydro.set_source_corrector(None, None)      # the context must not keep a pointer into this box's tensor
            b._flux_clear = False
        if predictor or not self._hydro_level(time, dt, with_src=True):
            self._hydro_calls(hydro)
        self._clean_reduce_new()
        self.amr.comm.allreduce_min(self.red)
        _, rho_min, _ = self.red.tolist()
        if rho_min < self.params.small_dens:
            return False, ("negative density" if rho_min < 0.0 else "small density") + " (density = %e)" % rho_min, None
        if lvl is not None:
            sp = tuple(t.data_ptr() for b in self.mine for t in (b.S_old_b, b.S_new_b))
            h.sources_mf(1, self._cached_ops(("src_new",), sp, lambda: h.make_source_boxes(
                [(b.lo, b.hi, (b.S_old_b, b.gbox), (b.S_new_b, b.gbox), (b.new_source, (b.lo, b.hi)), b.mass_fluxes, b.flux_boxes)
                 for b in self.mine])), lvl.grav if lvl.do_grav else None, lvl.grav_source_type if lvl.do_grav else 4, lvl.rotation,
                         lvl.geom, self.params, dt, ntimes=1)
        for b in (self.mine if lvl is None else ()):
            S, lo, hi = b.S_old_b, b.lo, b.hi
            b.new_source.zero_()
            if b.do_grav:
                h.new_gravity_source(S, b.gbox, b.S_new_b, b.gbox, b.new_source, (lo, hi), b.mass_fluxes, b.flux_boxes,
                                     lo, hi, b.grav, b.grav_source_type, dt, b.geom)
            if b.rotation is not None:
                h.new_rotation_source(S, b.gbox, b.S_new_b, b.gbox, b.new_source, (lo, hi), b.mass_fluxes, b.flux_boxes,
                                      lo, hi, b.rotation, b.geom, dt)
            if fused:
                h.apply_source(b.S_new_b, b.gbox, b.S_new_b, b.gbox, dt, b.new_source, (lo, hi), NSRC, lo, hi, b.params, ntimes=1)
            else:
                h.saxpy(b.S_new_b, b.gbox, dt, b.new_source, (lo, hi), NSRC, lo, hi)
                h.clean_state(b.S_new_b, b.gbox, lo, hi, b.params, ntimes=1)
        new_dt = self.estTimeStep()
        if self.params.change_max * new_dt < dt:
            return False, "timestep validity check failed", None
        return True, "", new_dt

    def _source_level_calls(self):
        """The box whose gravity / rotation settings stand for the level when the source stages go out as one library call per
        level (device backend, every box on this rank, the same settings in every box -- they come from one set of inputs);
        None: box by box."""
        if not (self._level_calls() and self.mine and hasattr(self.hydro, "sources_mf")):
            return None
        b0 = self.mine[0]
        for b in self.mine:
            if (b.do_grav != b0.do_grav or (b.do_grav and (tuple(b.grav) != tuple(b0.grav) or b.grav_source_type != b0.grav_source_type))
                    or b.rotation is not b0.rotation):
                return None
        return b0

    def _state_boxes(self, which):
        sp = tuple(getattr(b, which).data_ptr() for b in self.mine)
        return self._cached_ops(("state_boxes", which), sp, lambda: self.hydro.make_state_boxes(
            [(b.lo, b.hi, (getattr(b, which), b.gbox)) for b in self.mine]))

    def _clean_reduce_new(self):
        """clean_state_reduce of S_new of every box into the level's reduction vector."""
        h = self.hydro
        if self._level_calls() and len(self.mine) > 1 and hasattr(h, "clean_state_reduce_mf"):
            h.clean_state_reduce_mf(self._state_boxes("S_new_b"), self.mine[0].geom, self.params, self.red, ntimes=1)
            return
        for b in self.mine:
            h.clean_state_reduce(b.S_new_b, b.gbox, b.lo, b.hi, b.geom, b.params, self.red, ntimes=1)

    # ---- Castro::advance over the boxes of the level (Castro_advance.cpp:19-121) ----------------------
    def _swap_state_time_levels(self):
        for b in self.mine:
            b._swap_state_time_levels()

    def _zero_fluxes(self):
        for b in self.mine:
            b._zero_fluxes()

    def _save_old_state(self):
        return [b.S_old_b.clone() for b in self.mine]

    def _restore_old_state(self, prev):
        for b, p in zip(self.mine, prev):
            b.S_old_b.copy_(p)

    advance = Castro.advance
    subcycle_advance_ctu = Castro.subcycle_advance_ctu

    def _hydro_calls(self, fn):
        """fn(b) -- the hydro update of one box -- for every box of this rank.  The boxes of a level are independent
        there, and one small box fills a fraction of the chip (a 32^3 box is 64 workgroups of 256 CUs' worth), so with
        the device backend up to `box_streams` boxes are in flight at once, each on its own stream with its own scratch
        context; the level's reductions are atomic minima into one pair, the streams are joined before anything else runs."""
        pool = self.amr._stream_pool(self.l) if len(self.mine) > 1 else None
        if not pool:
            for b in self.mine:
                fn(b)
            return
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        used = pool[:min(len(pool), len(self.mine))]
        for _, st in used:
            st.wait_event(ev)
        for i, b in enumerate(self.mine):
            ctx, st = pool[i % len(pool)]
            keep, b.hydro = b.hydro, ctx
            try:
                with torch.cuda.stream(st):
                    fn(b)
            finally:
                b.hydro = keep
        for _, st in used:
            main.wait_stream(st)

    def do_advance_ctu(self, time, dt):
        """Castro::do_advance_ctu (Castro_advance_ctu.cpp:15-397) with every stage done for all boxes before the next."""
        if self.l > 0 and time != self._t0:
            # a later subcycle of a retried step: the coarse data are interpolated to ITS old time, not the step's
            self.alpha = self._alpha0 + (time - self._t0) / self._dt_parent
        if self._pending_cleans > 0:           # see Castro.do_advance_ctu
            self._clean_boxes("S_old_b", self._pending_cleans)
        self._pending_cleans = 0
        self._cached_est = None
        self.red.fill_(1.e200)
        self.fill("S_old_b")
        if self.have_sources:
            return self._advance_with_sources(time, dt)
        if not self._hydro_level(time, dt):
            def hydro(b):
                b.construct_ctu_hydro_source(time, dt, fuse_clean=self.fuse_clean)
                b._flux_clear = False
            self._hydro_calls(hydro)
        if not self.fuse_clean:
            self._clean_reduce_new()
        self.amr.comm.allreduce_min(self.red)                 # the level's minima over the ranks that hold its boxes
        est_last, rho_min, est = self.red.tolist()    # [2]: the estimate after the first clean_state (the only one here
                                                      # unless post_timestep's rode along: then [0] is the one after it)
        if self._post_clean_done:
            # the state leaves this advance as post_timestep will leave it (finest level, whole-step attempt): its CFL
            # estimate is what estTimeStep would reduce again from the same zones (Castro.step: _next_est)
            self._cached_est = est_last
        if rho_min < self.params.small_dens:
            return False, ("negative density" if rho_min < 0.0 else "small density") + " (density = %e)" % rho_min, None
        new_dt = self.fixed_dt if self.fixed_dt > 0.0 else min(self.max_dt, est * self.params.cfl)
        if self.params.change_max * new_dt < dt:
            return False, "timestep validity check failed", None
        return True, "", new_dt

    def invalidate_estimate(self):
        """S_new of this level has been written outside its own advance (avgDown from a finer level, reflux, initData, a
        regrid that filled it, a user's edit of b.S_new()): the CFL estimate cached from the fused post-update pass no longer
        describes it and estTimeStep reduces over the zones again, as the reference always does (Castro.cpp:1507-1626)."""
        self._cached_est = None

    def estTimeStep(self):
        if getattr(self, "_cached_est", None) is not None and self._post_clean_done and self.boxes:
            return min(self.max_dt, checked_estimate(self._cached_est) * self.params.cfl)
        self.red.fill_(1.e200)
        if self._level_calls() and len(self.mine) > 1 and hasattr(self.hydro, "estdt_cfl_mf"):
            self.hydro.estdt_cfl_mf(self._state_boxes("S_new_b"), self.mine[0].geom, self.params, self.red)
        else:
            for b in self.mine:
                self.hydro.estdt_cfl(b.S_new_b, b.gbox, b.lo, b.hi, b.geom, b.params, self.red)
        self.amr.comm.allreduce_min(self.red)
        return min(self.max_dt, checked_estimate(self.red.tolist()[0], empty_ok=not self.boxes) * self.params.cfl)

    def clean_new(self):
        self._clean_boxes("S_new_b", 1)

    def _level_calls(self):
        """One library call per level instead of one per box (castro_amd_ctu_hydro_mf, castro_amd_fab_ops_p): the device
        backend with every box on this rank; CASTRO_AMD_LEVEL_CALLS=0 keeps the per-box calls."""
        return (getattr(self, "batched", False) and hasattr(self.hydro, "construct_ctu_hydro_source_mf")
                and os.environ.get("CASTRO_AMD_LEVEL_CALLS", "1") != "0")

    def _clean_boxes(self, which, ntimes):
        """Castro::clean_state x ntimes on the valid zones of S_old_b / S_new_b (`which`) of every box of this rank."""
        if not self._level_calls() or len(self.mine) < 2:
            for b in self.mine:
                b.clean_state(getattr(b, which), ntimes)
            return
        sp = tuple(getattr(b, which).data_ptr() for b in self.mine)
        ops = self._cached_ops(("clean", which, ntimes), sp, lambda: self.hydro.make_ops(
            [(L.OP_CLEAN, 0, NUM_STATE, b.lo, b.hi, float(ntimes), 0.0, (getattr(b, which), b.gbox), (getattr(b, which), b.gbox), None)
             for b in self.mine]))
        self.hydro.fab_ops(ops, params=self.params)

    def _hydro_level(self, time, dt, with_src=False):
        """The hydro update of every box of the level through ONE library call (the MFIter loop of
        construct_ctu_hydro_source in C++: boxes dealt round robin to the contexts / streams of the box-stream pool, forked
        from and joined to the current stream inside the call).  False: not applicable, the caller loops over the boxes.
        with_src (_advance_with_sources_impl): every box hands its old-time source FAB to the call, which traces it and adds
        the hydro update to the S_new the caller has prepared; the new-time sources follow, so nothing is fused behind it."""
        if not self._level_calls() or not self.mine:
            return False
        fa = [bool(b.flux_assign and b._flux_clear) for b in self.mine]
        if any(x != fa[0] for x in fa) or any(b.fuse_post_clean for b in self.mine) or any(b.have_sources != with_src for b in self.mine):
            return False
        sp = tuple(t.data_ptr() for b in self.mine for t in (b.S_old_b, b.S_new_b))
        if with_src:
            boxes = self._cached_ops(("hydro_mf_src",), sp, lambda: self.hydro.make_hydro_boxes(
                [(b.bx, b.bx, (b.S_old_b, b.gbox), (b.S_new_b, b.gbox), b.fluxes, b.flux_boxes, b.mass_fluxes, (b.old_source, b.sbox))
                 for b in self.mine]))
            pool = self.amr._stream_pool(self.l) if len(self.mine) > 1 else None
            self.hydro.construct_ctu_hydro_source_mf(pool, boxes, self.mine[0].geom, self.params, time, dt, update_from_sborder=False,
                                                     flux_assign=fa[0], clean_ntimes=0, red=None)
            for b in self.mine:
                b._flux_clear = False
            return True
        boxes = self._cached_ops(("hydro_mf",), sp, lambda: self.hydro.make_hydro_boxes(
            [(b.bx, b.bx, (b.S_old_b, b.gbox), (b.S_new_b, b.gbox), b.fluxes, b.flux_boxes, b.mass_fluxes) for b in self.mine]))
        pool = self.amr._stream_pool(self.l) if len(self.mine) > 1 else None
        # On the finest level nothing touches S_new between the update and post_timestep's clean_state (no reflux, no
        # avgDown into it; FluxRegFineAdd only reads the fluxes): an attempt at the whole step takes that clean_state into the
        # fused pass, like the single-level driver (Castro.construct_ctu_hydro_source), instead of a sweep of its own
        post = bool(self.fuse_clean and getattr(self, "fuse_post_level", False) and getattr(self, "_whole_step", False))
        if post:
            self._post_clean_done = True
        self.hydro.construct_ctu_hydro_source_mf(pool, boxes, self.mine[0].geom, self.params, time, dt, update_from_sborder=True,
                                                 flux_assign=fa[0], clean_ntimes=(2 if post else 1) if self.fuse_clean else 0,
                                                 red=self.red if self.fuse_clean else None)
        for b in self.mine:
            b._flux_clear = False
        return True


_TAG_KINDS = {"value_greater": 0, "value_less": 1, "gradient": 2, "relative_gradient": 3}
_FIELDS = {"density": 0, "xmom": 1, "ymom": 2, "zmom": 3, "rho_E": 4, "rho_e": 5, "Temp": 6, "rho_X": 7}


class CastroAmr:
    def __init__(self, n_cell, patch_crse=None, prob_lo=(0., 0., 0.), prob_hi=(1., 1., 1.), lo_bc=(2, 2, 2), hi_bc=(2, 2, 2),
                 params=None, make_hydro=None, make_params=None, refine=None, regrid_int=2, n_error_buf=1,
                 blocking_factor=8, patches=None, max_level=1, cluster=False, grid_eff=0.7, max_grid_size=128,
                 do_grav=False, const_grav=0.0, grav_source_type=4, rotation=None, comm=None, base_grid=None, box_streams=4):
        """base_grid = (gx, gy, gz): level 0 as gx x gy x gz equal boxes instead of one (amr.max_grid_size on the base level);
        with `comm` they are dealt over the ranks like the boxes of the refined levels.
        comm: a castro_amd.DistComm to spread the boxes of every refined level over its ranks (box i of level l on rank
        (i + l) mod size, level 0 on rank 0 unless base_grid cuts it up): every rank builds the same hierarchy, holds the
        memory of its own boxes only and moves box-to-box data (coarse data under fine ghost shells, sibling ghost zones,
        coarse fluxes for the registers, registers for the reflux, averaged-down zones) with one grouped RCCL
        point-to-point exchange per pass.
        patch_crse = (lo, hi): the coarse zones covered by a FIXED refined box;
        patches = [entry, ...]: one entry per refined level, a box (lo, hi) or a list of boxes in the zones of the
        level below it (amr.max_level = len(patches)); or
        refine = [(field, kind, value), ...] like amr.refinement_indicators (field: a state name or a pointwise
        derived field such as pressure, kind:
        value_greater | value_less | gradient | relative_gradient) for boxes that follow the tags up to
        amr.max_level = max_level: one bounding box per level, or with cluster=True the Berger-Rigoutsos boxes
        (amr.grid_eff, amr.blocking_factor and amr.max_grid_size in zones of the new level)."""
        if patch_crse is not None:
            assert patches is None
            patches = [patch_crse]
        assert (patches is None) != (refine is None), "give either fixed patches or refinement indicators"
        from .castro import SingleComm
        self.comm = comm if comm is not None else SingleComm()
        self.rank, self.nranks = self.comm.rank, self.comm.size
        self._mk = (lambda: None) if make_hydro is None else make_hydro
        self.params = params if params is not None else (make_params() if make_params else L.default_params())
        self._kw = dict(prob_lo=prob_lo, prob_hi=prob_hi, lo_bc=lo_bc, hi_bc=hi_bc, params=self.params, overlap=False,
                        do_grav=do_grav, const_grav=const_grav, grav_source_type=grav_source_type, rotation=rotation)
        self.n_cell = tuple(n_cell)
        self.periodic = tuple(lo_bc[d] == 0 and hi_bc[d] == 0 for d in range(3))
        # boxes of a level whose hydro updates may be in flight at once (device backend); CASTRO_AMD_BOX_STREAMS overrides
        self.box_streams = int(os.environ.get("CASTRO_AMD_BOX_STREAMS", box_streams))
        self._hydros = []
        if base_grid is None or tuple(base_grid) == (1, 1, 1):
            base = Castro(n_cell, hydro=self._hydro_for(0), alloc=(self.rank == 0), **self._kw)
            base.owner = 0
            if base.have_sources and base.owned:
                base.new_source_g = base.hydro.alloc(NSRC, *base.sbox)
            self.lev = [_Level(self, 0, [base])]                      # lev[0] covers the domain
            self.lev[0].bind()
        else:
            gx = tuple(int(x) for x in base_grid)
            assert all(self.n_cell[d] % gx[d] == 0 and self.n_cell[d] // gx[d] >= 2 * NUM_GROW for d in range(3)), \
                "base_grid must divide the domain into boxes of at least %d zones a side" % (2 * NUM_GROW)
            nb = tuple(self.n_cell[d] // gx[d] for d in range(3))
            self.lev = []
            boxes = []
            for i, (kz, jy, ix) in enumerate(itertools.product(range(gx[2]), range(gx[1]), range(gx[0]))):
                lo = (ix * nb[0], jy * nb[1], kz * nb[2])
                hi = tuple(lo[d] + nb[d] - 1 for d in range(3))
                owner = i % self.nranks
                b = _Patch(self.n_cell, hydro=self._hydro_for(0), box=(lo, hi), alloc=(owner == self.rank), **self._kw)
                b.owner = owner
                boxes.append(b)
            lev0 = _Level(self, 0, boxes)
            for b in boxes:
                b.bind(lev0, None)
            self.lev.append(lev0)
            lev0.bind()
        self.refine = refine
        self.regrid_int, self.n_error_buf, self.blocking_factor = int(regrid_int), int(n_error_buf), int(blocking_factor)
        self.cluster = bool(cluster)
        self.grid_eff = float(grid_eff) if cluster else 0.0           # 0: the bounding box of the tags is accepted as it is
        self.max_grid_size = int(max_grid_size) if cluster else None
        self.nregrid = 0
        self.max_level = int(max_level) if refine is not None else len(patches or [])
        for pb in (patches or []):
            self._push_level([tuple(map(tuple, pb))] if _is_box(pb) else [tuple(map(tuple, b)) for b in pb])
        self.time, self.nstep = 0.0, 0
        self.dt_level = [0.0] * 16
        self.level_count = [0] * 16                                   # Amr::level_count: steps of a level since its last regrid

    # views used by the tests and the plotfile writer
    crse = property(lambda self: self.lev[0])
    fine = property(lambda self: self.lev[1] if len(self.lev) > 1 else None)
    levels = property(lambda self: list(self.lev))
    boxes = property(lambda self: [None] + [[b.pbox for b in lev.boxes] for lev in self.lev[1:]])
    # pbox[l]: the box of level l in level l-1 zones when the level has one box, else the list of its boxes
    pbox = property(lambda self: [None] + [bl[0] if len(bl) == 1 else bl for bl in self.boxes[1:]])
    plo = property(lambda self: self.pbox[1][0] if len(self.lev) > 1 else None)
    phi = property(lambda self: self.pbox[1][1] if len(self.lev) > 1 else None)

    def _stream_pool(self, l):
        """[(scratch context, stream)] for the concurrent hydro calls of the boxes of level l, or None (one stream)"""
        if self.box_streams <= 1 or not torch.cuda.is_available() or not hasattr(self._hydro_for(l), "device") \
                or not hasattr(self._hydro_for(l), "lib"):
            return None
        pools = self.__dict__.setdefault("_pools", {})
        if l not in pools:
            from .hydro import HipHydro
            dev = self._hydro_for(l).device
            # the other contexts of a level run the same build of the kernel library as its first one
            pools[l] = [(self._hydro_for(l) if k == 0 else HipHydro(dev.index, numerics=getattr(self._hydro_for(l), "numerics", None)),
                         torch.cuda.Stream(device=dev)) for k in range(self.box_streams)]
        return pools[l]

    def all_hydros(self):
        """Every scratch context this hierarchy launches kernels through: the per-level ones and those of the box-stream
        pools (their latched device status and their kernel profilers are otherwise invisible to the caller)."""
        seen, out = set(), []
        for h in list(self._hydros) + [h for pool in self.__dict__.get("_pools", {}).values() for h, _ in pool]:
            if id(h) not in seen:
                seen.add(id(h))
                out.append(h)
        return out

    def status(self):
        """OR of the latched device status words (rho <= 0 met in ctoprim) of all contexts; clears them."""
        st = 0
        for h in self.all_hydros():
            if hasattr(h, "status"):
                st |= int(h.status())
        return st

    def _hydro_for(self, l):
        while len(self._hydros) <= l:
            h = self._mk()
            if h is None:
                from .hydro import HipHydro
                h = HipHydro(torch.cuda.current_device())
            self._hydros.append(h)
        return self._hydros[l]

    def _make_level(self, l, pboxes):
        """Level l >= 1 covering the boxes `pboxes` (zones of level l-1), with its flux registers."""
        boxes = []
        for i, (plo, phi) in enumerate(pboxes):
            flo = tuple(2 * x for x in plo)
            fhi = tuple(2 * x + 1 for x in phi)
            owner = (i + l) % self.nranks             # level 0 is box 0 of rank 0; one-box levels go round the ranks
            b = _Patch(tuple((2 ** l) * x for x in self.n_cell), hydro=self._hydro_for(l), box=(flo, fhi),
                       alloc=(owner == self.rank), **self._kw)
            b.owner = owner
            boxes.append(b)
        lev = _Level(self, l, boxes)
        for b, pb in zip(boxes, pboxes):
            b.bind(lev, pb)
        return lev

    # ---- box-to-box operations with the boxes spread over ranks ----------------------------------------------
    def _xrun(self, ops, ncomp=NUM_STATE):
        """ops = [(kind, D, S, lo, hi, extra)]: region [lo, hi] of box D from box S.  Both here: the operation itself.
        S here and D elsewhere: the source's contribution is staged in a buffer of the region's shape and sent; D here
        and S elsewhere: it is received and applied, in list order together with the local ones (later entries overwrite
        earlier ones exactly like on one rank).  Every rank walks the same list, so the messages pair up by position."""
        me = self.rank
        sends, recvs, bufs = [], [], {}
        for n, (kind, D, S, lo, hi, extra) in enumerate(ops):
            if D.owner == S.owner:
                continue
            if S.owner == me:
                t = self._xstage(kind, D, S, lo, hi, extra)
                sends.append((D.owner, n, t))
            elif D.owner == me:
                h = D.hydro
                t = h.alloc(ncomp, lo, hi)
                bufs[n] = t
                recvs.append((S.owner, n, t))
        self.comm.exchange(sends, recvs)
        for n, (kind, D, S, lo, hi, extra) in enumerate(ops):
            if D.owner != me:
                continue
            self._xapply(kind, D, S, lo, hi, extra, bufs.get(n))

    def _xstage(self, kind, D, S, lo, hi, extra):
        h = S.hydro
        t = h.alloc(NSRC if kind.startswith("src_") else NUM_STATE, lo, hi)
        box = (tuple(lo), tuple(hi))
        if kind == "src_lincomb":              # Source_Type data of a parent, interpolated in time
            h.lincomb(t, box, 1.0 - extra, S.old_source, S.sbox, extra, S.new_source_g, S.sbox, NSRC, lo, hi)
            return t
        if kind == "src_copy":
            name, sh = extra
            h.copy(t, box, getattr(S, name), _shift(S.sbox, sh), lo, hi)
            return t
        if kind == "lincomb":
            h.lincomb(t, box, 1.0 - extra, S.S_old_b, S.gbox, extra, S.S_new_b, S.gbox, NUM_STATE, lo, hi)
        elif kind == "crse_new":
            h.lincomb(t, box, 0.0, S.S_new_b, S.gbox, 1.0, S.S_new_b, S.gbox, NUM_STATE, lo, hi)
        elif kind == "copy":
            which, sh = extra
            h.copy(t, box, getattr(S, which), _shift(S.gbox, sh), lo, hi)
        elif kind == "crse_init":
            d, side = extra
            h.copy(t, box, S.fluxes[d], S.flux_boxes[d], lo, hi)
        elif kind == "reflux":
            d, side, vol, csh = extra
            reg, rbox = S.regs[(d, side)]
            h.copy(t, box, reg, rbox, lo, hi)
        elif kind == "avgdown":
            h.avgdown(S.S_new_b, S.gbox, t, box, lo, hi, NUM_STATE)
        else:
            raise ValueError(kind)
        return t

    def _xapply(self, kind, D, S, lo, hi, extra, buf):
        """the operation on this rank's box D; `buf` holds the staged source when S lives elsewhere"""
        h = D.hydro
        box = (tuple(lo), tuple(hi))
        if kind == "src_lincomb":
            if buf is None:
                h.lincomb(D.stmp, D.scbox, 1.0 - extra, S.old_source, S.sbox, extra, S.new_source_g, S.sbox, NSRC, lo, hi)
            else:
                h.copy(D.stmp, D.scbox, buf, box, lo, hi)
        elif kind == "src_copy":
            name, sh = extra
            if buf is None:
                h.copy(getattr(D, name), D.sbox, getattr(S, name), _shift(S.sbox, sh), lo, hi)
            else:
                h.copy(getattr(D, name), D.sbox, buf, box, lo, hi)
        elif kind == "lincomb":
            if buf is None:
                h.lincomb(D.ctmp, D.cbox, 1.0 - extra, S.S_old_b, S.gbox, extra, S.S_new_b, S.gbox, NUM_STATE, lo, hi)
            else:
                h.copy(D.ctmp, D.cbox, buf, box, lo, hi)
        elif kind == "crse_new":                                # FillCoarsePatch of a regrid: the parents' new-time data
            if buf is None:
                h.lincomb(D.ctmp, D.cbox, 0.0, S.S_new_b, S.gbox, 1.0, S.S_new_b, S.gbox, NUM_STATE, lo, hi)
            else:
                h.copy(D.ctmp, D.cbox, buf, box, lo, hi)
        elif kind == "copy":
            which, sh = extra
            if buf is None:
                h.copy(getattr(D, which), D.gbox, getattr(S, which), _shift(S.gbox, sh), lo, hi)
            else:
                h.copy(getattr(D, which), D.gbox, buf, box, lo, hi)
        elif kind == "crse_init":
            d, side = extra
            reg, rbox = D.regs[(d, side)]
            src, sbox = (S.fluxes[d], S.flux_boxes[d]) if buf is None else (buf, box)
            h.fluxreg_crse_init(reg, rbox, src, sbox, lo, hi, NUM_STATE, -1.0)
        elif kind == "reflux":
            d, side, vol, csh = extra
            reg, rbox = S.regs[(d, side)] if buf is None else (buf, box)
            h.reflux(D.S_new_b, _shift(D.gbox, csh), reg, rbox, lo, hi, d, side, NUM_STATE, vol)
        elif kind == "avgdown":
            if buf is None:
                h.avgdown(S.S_new_b, S.gbox, D.S_new_b, D.gbox, lo, hi, NUM_STATE)
            else:
                h.copy(D.S_new_b, D.gbox, buf, box, lo, hi)
        else:
            raise ValueError(kind)

    def gather_level(self, l):
        """[(box, new-time state as a numpy array)] of level l on rank 0 (None elsewhere): for tests and plotfiles"""
        mine = [(b.bx, b.S_new().cpu().numpy()) for b in self.lev[l].mine]
        allv = self.comm.gather_objects(mine)
        if self.rank != 0:
            return None
        got = {bx: a for part in allv for bx, a in part}
        return [(b.bx, got[b.bx]) for b in self.lev[l].boxes]

    def _push_level(self, pboxes):
        lev = self._make_level(len(self.lev), pboxes)
        self.lev.append(lev)
        lev.bind()

    def _drop_fine(self):
        del self.lev[1:]

    # ---- Castro::errorEst (Castro.cpp:3131-3164) ---------------------------------------------------
    def _fill_ghosts_new(self, upto, lbase=0, alpha=1.0):
        """Ghost zones of the new-time data of levels lbase..upto (each FillPatch reads the level below it).  Level
        lbase sits at `alpha` inside its parent's [old, new] interval, the finer ones are synchronised with it."""
        for l in range(lbase, upto + 1):
            self.lev[l].alpha = alpha if l == lbase else 1.0
            self.lev[l].fill("S_new_b")

    def _tags(self, l):
        """(tags, mask, origin): tagged zones and valid zones of level l as 0/1 tensors (on the level's device) over the
        bounding region of its boxes."""
        lev = self.lev[l]
        bl = lev.box_list()
        olo = tuple(min(b[0][d] for b in bl) for d in range(3))
        ohi = tuple(max(b[1][d] for b in bl) for d in range(3))
        h = lev.hydro
        tags, mask = h.alloc(1, olo, ohi)[0], h.alloc(1, olo, ohi)[0]
        for b in lev.boxes:
            sl = tuple(slice(b.lo[d] - olo[d], b.hi[d] - olo[d] + 1) for d in (2, 1, 0))
            mask[sl] = 1.0
            if not b.owned:                                     # its owner tags it; tag_boxes() combines the ranks' tags
                continue
            t = h.alloc(1, b.lo, b.hi)
            for field, kind, value in self.refine:
                if field in _FIELDS:
                    h.error_tag(b.S_new_b, b.gbox, _FIELDS[field], t, (b.lo, b.hi), b.lo, b.hi, _TAG_KINDS[kind], value)
                else:
                    # a derived field (amr.refine.<name>.field_name = pressure, ...): evaluated with one ghost zone
                    # for the gradient kinds, like the reference's derive on a grown box
                    g1 = (tuple(x - 1 for x in b.lo), tuple(x + 1 for x in b.hi))
                    d = h.alloc(1, *g1)
                    center = [0.5 * (b.geom.problo[dd] + b.geom.probhi[dd]) for dd in range(3)]
                    h.derive(field, b.S_new_b, b.gbox, d, g1, 0, g1[0], g1[1], b.geom, b.params, center)
                    h.error_tag(d, g1, 0, t, (b.lo, b.hi), b.lo, b.hi, _TAG_KINDS[kind], value)
            tags[sl] = (t[0] > 0.5).to(tags.dtype)
        return tags, mask, olo

    def _cell_images(self, l, a):
        """Shifts (in blocking cells of level l) to the periodic images of the domain."""
        ext = [(2 ** l) * self.n_cell[d] // a for d in range(3)]
        return list(itertools.product(*[(-ext[d], 0, ext[d]) if self.periodic[d] else (0,) for d in range(3)]))

    def _erode_cells(self, p, o, l, a):
        """p (bool, [z, y, x], origin o in cells of a zones of level l) with every cell removed that has a neighbour
        outside p.  Beyond a physical boundary everything counts as inside; beyond a periodic one lies the other end
        of p when p spans the domain."""
        for d in range(3):
            ax = 2 - d
            n = p.shape[ax]
            nd = (2 ** l) * self.n_cell[d] // a
            at_lo, at_hi = o[d] == 0, o[d] + n == nd
            first, last = np.take(p, [0], axis=ax), np.take(p, [n - 1], axis=ax)
            if self.periodic[d]:
                wrap = at_lo and at_hi
                below, above = (last, first) if wrap else (np.zeros_like(first), np.zeros_like(first))
            else:
                below = np.ones_like(first) if at_lo else np.zeros_like(first)
                above = np.ones_like(first) if at_hi else np.zeros_like(first)
            q = np.concatenate([below, p, above], axis=ax)
            sl = lambda a0, a1: tuple(slice(a0, a1) if x == ax else slice(None) for x in range(3))
            p = q[sl(0, n)] & q[sl(1, n + 1)] & q[sl(2, n + 2)]
        return p

    def _nesting_cells(self, l, lbase, o, shape, a):
        """Amr::grid_places' proper nesting domain [3P] for the tags of level l when level lbase and everything below
        it keep their boxes: the region of level lbase shrunk by one blocking cell, refined and shrunk by another cell
        for every level up to l -- so that each new level fits around the next with a cell to spare.  Returned over
        the cells (of a zones of level l) of the region (o, shape); None when there is nothing to respect (level 0
        covers the domain)."""
        if lbase == 0:
            return None
        assert all(((2 ** lbase) * self.n_cell[d]) % a == 0 for d in range(3))
        bl = self.lev[lbase].box_list()
        olo = [min(b[0][d] for b in bl) // a for d in range(3)]
        ohi = [max(b[1][d] for b in bl) // a for d in range(3)]
        p = np.zeros(tuple(ohi[d] - olo[d] + 1 for d in (2, 1, 0)), dtype=bool)
        for lo, hi in bl:
            p[tuple(slice(lo[d] // a - olo[d], hi[d] // a - olo[d] + 1) for d in (2, 1, 0))] = True
        p = self._erode_cells(p, olo, lbase, a)
        for i in range(lbase + 1, l + 1):
            p = p.repeat(2, axis=0).repeat(2, axis=1).repeat(2, axis=2)
            olo = [2 * x for x in olo]
            p = self._erode_cells(p, olo, i, a)
        out = np.zeros(shape, dtype=bool)
        src = ((tuple(olo), tuple(olo[d] + p.shape[2 - d] - 1 for d in range(3))))
        it = CL.intersect(src, (tuple(o), tuple(o[d] + shape[2 - d] - 1 for d in range(3))))
        if it:
            out[tuple(slice(it[0][d] - o[d], it[1][d] - o[d] + 1) for d in (2, 1, 0))] = \
                p[tuple(slice(it[0][d] - olo[d], it[1][d] - olo[d] + 1) for d in (2, 1, 0))]
        return out

    def tag_boxes(self, l=0, ghosts_filled=False, cover=(), lbase=None, nest=()):
        """New boxes of level l+1 in level-l zones: clustered, buffered tags of level l, also covering `cover`
        (boxes in level-l zones that the new boxes must contain: the footprint of the level above) and the blocking
        cells of and around `nest` (that footprint without buffer: proper nesting), all inside the proper nesting
        domain of a regrid that leaves levels <= lbase (default l) alone.  Buffering and the reduction to
        blocking_factor/2-sized cells run where the tags are (max-pooling); only the reduced arrays go to the host
        for the clustering."""
        import torch.nn.functional as F
        if not ghosts_filled:
            self._fill_ghosts_new(l)
        tags, mask, o = self._tags(l)
        n = self.n_error_buf
        if n > 0:                                               # amr.n_error_buf: a (2n+1)^3 maximum
            tags = F.max_pool3d(tags[None, None], kernel_size=2 * n + 1, stride=1, padding=n)[0, 0]
        shape = tags.shape
        for lo, hi in cover:
            it = CL.intersect((lo, hi), (o, tuple(o[d] + shape[2 - d] - 1 for d in range(3))))
            if it:
                tags[it[0][2] - o[2]:it[1][2] - o[2] + 1, it[0][1] - o[1]:it[1][1] - o[1] + 1, it[0][0] - o[0]:it[1][0] - o[0] + 1] = 1.0
        tags = tags * mask
        a = max(self.blocking_factor // 2, 1)                   # blocking_factor is in zones of level l+1
        assert all(x % a == 0 for x in o) and all(s % a == 0 for s in shape), "level boxes must be multiples of blocking_factor/2"
        if a > 1:                                               # any zone tagged / all zones valid per a^3 cell
            ct = F.max_pool3d(tags[None, None], kernel_size=a, stride=a)[0, 0]
            cm = -F.max_pool3d(-mask[None, None], kernel_size=a, stride=a)[0, 0]
        else:
            ct, cm = tags, mask
        ct = (ct > 0.5).cpu().numpy()
        if self.nranks > 1:
            # buffering and pooling are maxima, so the maximum over the ranks' partial tags commutes with them: only the
            # reduced arrays travel
            ct = np.logical_or.reduce(self.comm.gather_objects(ct))
        oc = tuple(x // a for x in o)
        for lo, hi in nest:                                     # one cell around the footprint of level l+2
            for sh in self._cell_images(l, a):
                it = CL.intersect((tuple(lo[d] // a - 1 + sh[d] for d in range(3)), tuple(hi[d] // a + 1 + sh[d] for d in range(3))),
                                  (oc, tuple(oc[d] + ct.shape[2 - d] - 1 for d in range(3))))
                if it:
                    ct[tuple(slice(it[0][d] - oc[d], it[1][d] - oc[d] + 1) for d in (2, 1, 0))] = True
        if not ct.any():
            return []
        cm = (cm > 0.5).cpu().numpy()
        pn = self._nesting_cells(l, l if lbase is None else lbase, oc, ct.shape, a)
        if pn is not None:
            cm &= pn
            if not (ct & cm).any():
                return []
        # The clustering is a pure function of the reduced arrays: a regrid that finds the tags of the last one (a front that has
        # not crossed a blocking cell since) takes the boxes of the last one instead of 1-2 ms of Berger-Rigoutsos on the host
        max_size = None if self.max_grid_size is None else max(self.max_grid_size // 2, a)
        key = (ct.shape, ct.tobytes(), cm.tobytes(), tuple(o), a, self.grid_eff, max_size)
        memo = self.__dict__.setdefault("_cluster_memo", {})
        hit = memo.get(l)
        if hit is not None and hit[0] == key:
            return list(hit[1])
        out = CL.boxes_from_coarse(ct, cm, o, a, grid_eff=self.grid_eff, max_size=max_size)
        memo[l] = (key, list(out))
        return out

    def tag_box(self, l=0):
        """The one-box form: (lo, hi) in level-l zones or None."""
        bl = self.tag_boxes(l)
        return bl[0] if bl else None

    # ---- Amr::grid_places: new box lists for levels lbase+1.. (at most one level more than now) ------
    def _grid_places(self, lbase=0, alpha=1.0):
        finest = len(self.lev) - 1
        top = min(finest, self.max_level - 1)                   # the finest level that may carry tags
        self._fill_ghosts_new(top, lbase, alpha)
        new = {l: self.boxes[l] for l in range(1, lbase + 1)}   # levels up to lbase keep their boxes
        for l in range(top, lbase - 1, -1):
            cover, nest = [], []
            for lo, hi in new.get(l + 2, []):                   # proper nesting: contain the level above + a buffer
                nest.append((tuple(_coarsen(lo[d]) for d in range(3)), tuple(_coarsen(hi[d]) for d in range(3))))
                cover.append((tuple(_coarsen(lo[d]) - self.n_error_buf for d in range(3)),
                              tuple(_coarsen(hi[d]) + self.n_error_buf for d in range(3))))
            new[l + 1] = self.tag_boxes(l, ghosts_filled=True, cover=cover, lbase=lbase, nest=nest)
        out = []
        for l in range(1, top + 2):
            if not new.get(l):
                break
            out.append(new[l])
        # boxes of level l+1 were placed on the OLD level-l boxes: keep what lies inside the NEW ones
        for i in range(max(lbase, 1), len(out)):
            kept = []
            for b in out[i]:
                for plo, phi in out[i - 1]:
                    it = CL.intersect(b, (tuple(2 * x for x in plo), tuple(2 * x + 1 for x in phi)))
                    if it:
                        kept.append(it)
            if not kept:
                del out[i:]
                break
            out[i] = sorted(kept, key=lambda b: (b[0][2], b[0][1], b[0][0]))
        return out

    # ---- Amr::regrid(lbase): new grids above level lbase, data from the old boxes of a level where they exist,
    #      else interpolated ---------------------------------------------------------------------------------
    def regrid(self, lbase=0, alpha=1.0):
        new = self._grid_places(lbase, alpha)
        if new == self.boxes[1:]:
            return False
        old_lev = list(self.lev)
        keep = 1                                                # levels below the first changed box list stay as they are
        while keep <= min(len(new), len(self.lev) - 1) and new[keep - 1] == self.boxes[keep]:
            keep += 1
        del self.lev[keep:]
        for l in range(keep, len(new) + 1):
            parent = self.lev[l - 1]
            if l - 1 >= keep:                                   # a level made in this regrid: its ghost zones are not filled yet
                parent.alpha = 1.0
                parent.fill("S_new_b")
            self._push_level(new[l - 1])
            lev = self.lev[l]
            h = lev.hydro
            if self.nranks > 1:
                self._xrun([("crse_new", b, p, lo, hi, None) for b in lev.boxes for p, (lo, hi) in b.csrc + b.csrc_valid])
                for b in lev.mine:
                    h.cc_interp(b.ctmp, b.cbox, b.S_new_b, b.gbox, b.lo, b.hi, NUM_STATE)
                if l < len(old_lev):
                    ops = []
                    for b in lev.boxes:
                        for ob in old_lev[l].boxes:
                            it = CL.intersect(ob.bx, b.bx)
                            if it:
                                ops.append(("copy", b, ob, it[0], it[1], ("S_new_b", (0, 0, 0))))
                    self._xrun(ops)
                lev.time, lev.nstep = self.time, self.nstep
                continue
            for b in lev.boxes:
                # FillCoarsePatch: cell-conservative interpolation of the (ghost-filled) coarse data over the whole new box
                for p, (lo, hi) in b.csrc + b.csrc_valid:
                    h.lincomb(b.ctmp, b.cbox, 0.0, p.S_new_b, p.gbox, 1.0, p.S_new_b, p.gbox, NUM_STATE, lo, hi)
                h.cc_interp(b.ctmp, b.cbox, b.S_new_b, b.gbox, b.lo, b.hi, NUM_STATE)
                if l < len(old_lev):
                    for ob in old_lev[l].boxes:
                        it = CL.intersect(ob.bx, b.bx)
                        if it:
                            h.copy(b.S_new_b, b.gbox, ob.S_new_b, ob.gbox, it[0], it[1])
            lev.time, lev.nstep = self.time, self.nstep
        self.nregrid += 1
        return True

    # ---- Amr::init (bldFineLevels: one new level per pass) / Castro::post_init ---------------------------
    def initData(self, problem="sedov", **kw):
        self.invalidate_estimates()                     # also on an existing hierarchy: no estimate survives new data
        for b in self.lev[0].mine:
            b.initData(problem, **kw)
        if self.refine is not None:
            self._drop_fine()
            while len(self.lev) - 1 < self.max_level:
                bl = self.tag_boxes(len(self.lev) - 1)
                if not bl:
                    break
                self._push_level(bl)
                for b in self.lev[-1].mine:
                    b.initData(problem, **kw)                   # fine levels start from the problem initialiser
            # Amr::bldFineLevels' closing loop [3P]: each level above was placed inside the one before it; regrid from
            # level 0 (initial data from the initialiser again) until the coarser levels have grown around the finer
            # ones and the box lists stay as they are
            for _ in range(4):
                if len(self.lev) == 1:
                    break
                for l in range(len(self.lev) - 1, 0, -1):
                    self.avgDown(l)
                new = self._grid_places(0)
                if new == self.boxes[1:]:
                    break
                self._drop_fine()
                for bl in new:
                    self._push_level(bl)
                    for b in self.lev[-1].mine:
                        b.initData(problem, **kw)
        else:
            for lev in self.lev[1:]:
                for b in lev.mine:
                    b.initData(problem, **kw)
        # Castro::post_init (Castro.cpp:2220-2235): average down from the finest level, nothing else -- the averaged zones
        # are first cleaned by initialize_advance
        for l in range(len(self.lev) - 1, 0, -1):
            self.avgDown(l)
        self.time, self.nstep = 0.0, 0
        self.level_count = [0] * 16

    # ---- Castro::avgDown (Castro.cpp:3096-3113): level l onto level l-1 ------------------------------
    def invalidate_estimates(self):
        """call after writing any level's S_new from outside (user edits between coarse steps): see _Level.invalidate_estimate"""
        for lev in self.lev:
            lev.invalidate_estimate()

    def avgDown(self, l=1):
        self.lev[l - 1].invalidate_estimate()           # the coarser level's S_new is overwritten under the fine boxes
        if self.nranks > 1:
            return self._xrun([("avgdown", p, b, lo, hi, None) for b in self.lev[l].boxes for p, (lo, hi) in b.avg_to])
        fine = self.lev[l]
        h = fine.hydro
        if fine._level_calls():
            # the whole level in one launch (CASTRO_AMD_OP_AVGDOWN): the regions are disjoint zones of the coarse level
            pp = tuple(b.S_new_b.data_ptr() for b in fine.boxes) + tuple(p.S_new_b.data_ptr() for p in self.lev[l - 1].boxes)
            h.fab_ops(fine._cached_ops(("avgdown",), pp, lambda: h.make_ops(
                [(L.OP_AVGDOWN, 0, NUM_STATE, lo, hi, 0.0, 0.0, (p.S_new_b, p.gbox), (b.S_new_b, b.gbox), None)
                 for b in fine.boxes for p, (lo, hi) in b.avg_to])), params=fine.params)
            return
        for b in fine.boxes:
            for p, (lo, hi) in b.avg_to:
                h.avgdown(b.S_new_b, b.gbox, p.S_new_b, p.gbox, lo, hi, NUM_STATE)

    # ---- Castro::computeInitialDt / computeNewDt over the hierarchy ----------------------------------
    def _dt0(self, stop_time, initial):
        P = self.params
        n_factor, dt_0 = 1, 1.e100
        for l, lev in enumerate(self.lev):
            n_factor *= (1 if l == 0 else 2)
            dt = lev.estTimeStep()
            if initial:
                dt *= P.init_shrink
            else:
                dt = min(dt, P.change_max * self.dt_level[l])
            dt_0 = min(dt_0, n_factor * dt)
        if initial:
            eps = 0.001 * dt_0
            if stop_time >= 0.0 and (self.time + dt_0) > (stop_time - eps):
                dt_0 = stop_time - self.time
        else:
            eps = 2.220446049250313e-16
            if stop_time >= 0.0 and (self.time + dt_0) >= (stop_time - eps):
                dt_0 = stop_time - self.time
        return dt_0

    # ---- Amr::timeStep: advance level l, then recursively twice the next finer level, then post_timestep -----
    def _time_step(self, l, t, dt, alpha):
        """alpha: position of this level's old time inside the parent's [old, new] interval (0 or 1/2)."""
        # Amr::timeStep: every level i >= l that has taken regrid_int steps since its last regrid gets new grids above
        # it (all those levels are synchronised at time t here)
        if self.refine is not None and self.regrid_int > 0:
            i = l
            while i <= min(len(self.lev) - 1, self.max_level - 1):
                if self.level_count[i] >= self.regrid_int:
                    self.regrid(i, alpha if i == l else 1.0)
                    for k in range(i, len(self.level_count)):
                        self.level_count[k] = 0
                i += 1
        lev, finest = self.lev[l], len(self.lev) - 1
        h = lev.hydro
        lev.alpha = alpha
        lev._t0, lev._alpha0, lev._dt_parent = t, alpha, 2.0 * dt
        lev.fuse_post_level = l == finest and os.environ.get("CASTRO_AMD_FUSE_POST_FINEST", "1") != "0"
        lev.advance(t, dt)
        self.level_count[l] += 1
        if l > 0:
            # FluxRegFineAdd: + this level's fluxes (already dt x area) summed over the 4 fine faces
            if lev.batched:
                h.fab_ops(lev.ops_fine_add)
            else:
                for b in lev.mine:
                    for (d, side), (reg, rbox) in b.regs.items():
                        h.fluxreg_fine_add(reg, rbox, b.fluxes[d], b.flux_boxes[d], rbox[0], rbox[1], d, NUM_STATE, 1.0)
        if l < finest:
            fine = self.lev[l + 1]
            # ghost zones of the new data of this level, for the FillPatch of the next finer one
            lev.alpha = alpha + 0.5
            lev.fill("S_new_b")
            if lev.have_sources:
                lev.fill_new_source()
            # FluxRegCrseInit: -1 x this level's fluxes through the faces of the finer boxes
            if self.nranks > 1:
                self._xrun([("crse_init", b, p, lo, hi, (d, side)) for b in fine.boxes for (d, side) in b.regs
                            for p, (lo, hi) in b.crse_init[(d, side)]])
            elif fine.batched:
                h.fab_ops(fine.ops_crse_init)
            else:
                for b in fine.boxes:
                    for (d, side), (reg, rbox) in b.regs.items():
                        for p, (lo, hi) in b.crse_init[(d, side)]:
                            h.fluxreg_crse_init(reg, rbox, p.fluxes[d], p.flux_boxes[d], lo, hi, NUM_STATE, -1.0)
            for it in range(2):
                self._time_step(l + 1, t + it * (dt / 2), dt / 2, 0.5 * it)
            # post_timestep: reflux, avgDown, clean_state
            vol = lev.geom.dx[0] * lev.geom.dx[1] * lev.geom.dx[2]
            # face orientation by face orientation, like FluxRegister::Reflux's OrientationIter [3P]: the registers of one
            # orientation touch disjoint coarse zones (two boxes with a common outside neighbour on the same side would
            # overlap), so they can share launches; a zone next to several boxes gets its contributions in this order
            if self.nranks > 1:
                self._xrun([("reflux", p, b, lo, hi, (d, side, vol, csh)) for d in range(3) for side in (0, 1)
                            for b in fine.boxes for p, (lo, hi), csh in b.reflux_to[(d, side)]])
            for d in (range(3) if self.nranks == 1 else ()):
                for side in (0, 1):
                    if fine.batched:
                        pp = tuple(p.S_new_b.data_ptr() for p in lev.boxes)
                        h.fab_ops(fine._cached_ops(("reflux", d, side), pp, lambda d=d, side=side: h.make_ops(
                            [(L.OP_REFLUX, (d, side), NUM_STATE, lo, hi, vol, 0.0, (p.S_new_b, _shift(p.gbox, csh)), b.regs[(d, side)], None)
                             for b in fine.boxes for p, (lo, hi), csh in b.reflux_to[(d, side)]])))
                    else:
                        for b in fine.boxes:
                            reg, rbox = b.regs[(d, side)]
                            for p, (lo, hi), csh in b.reflux_to[(d, side)]:
                                h.reflux(p.S_new_b, _shift(p.gbox, csh), reg, rbox, lo, hi, d, side, NUM_STATE, vol)
            lev.invalidate_estimate()                   # reflux has corrected this level's S_new
            self.avgDown(l + 1)
        # Castro::post_timestep ends with clean_state(S_new) on EVERY level (Castro.cpp:1909-1916), the finest included
        # (there it may have ridden in the fused pass of the update: _hydro_level)
        if not (l == finest and lev._post_clean_done):
            lev.clean_new()

    # ---- Amr::coarseTimeStep ---------------------------------------------------------------------
    def step(self, stop_time=-1.0):
        dt0 = self._dt0(stop_time, self.nstep == 0)
        for l in range(len(self.dt_level)):
            self.dt_level[l] = dt0 / (2 ** l)
        t = self.time
        self._time_step(0, t, dt0, 0.0)
        self.time = t + dt0
        self.nstep += 1
        for lev in self.lev:
            lev.time, lev.nstep = self.time, self.nstep
        return dt0

    def evolve(self, stop_time, max_step=10 ** 9):
        eps = 2.220446049250313e-16
        while self.nstep < max_step and self.time < stop_time - eps:
            self.step(stop_time)
        return self.nstep

    # ---- diagnostics -------------------------------------------------------------------------
    def composite_sum(self, comp):
        """Volume integral of a conserved component over the composite grid (finest data wherever refined)."""
        tot = 0.0
        for l, lev in enumerate(self.lev):
            v = lev.geom.dx[0] * lev.geom.dx[1] * lev.geom.dx[2]
            for b in lev.boxes:
                S = b.S_new()[comp]
                if l + 1 < len(self.lev):
                    S = S.clone()
                    o = b.lo
                    for f in self.lev[l + 1].boxes:
                        it = CL.intersect(f.pbox, b.bx)
                        if it:
                            (p, q) = it
                            S[p[2] - o[2]:q[2] - o[2] + 1, p[1] - o[1]:q[1] - o[1] + 1, p[0] - o[0]:q[0] - o[0] + 1] = 0.0
                tot += S.sum().item() * v
        return tot

    def zones_advanced_per_coarse_step(self):
        """Zone updates of one coarse step (level l advances 2^l times): the reference's FOM numerator."""
        return sum((2 ** l) * sum(b.n[0] * b.n[1] * b.n[2] for b in lev.boxes) for l, lev in enumerate(self.lev))
