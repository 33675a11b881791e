"""AMR with subcycling: a coarse level covering the domain and one refined patch per finer level (ratio 2).

First slice of SURVEY.md 8 f-3.  What the reference does through AMReX's Amr / AmrLevel / FluxRegister /
Interpolater classes [3P] is orchestrated here on top of two `Castro` level objects:

  Amr::timeStep recursion with subcycling        coarse advance, then two fine advances of dt/2
  AmrLevel::FillPatch (fine level)               own valid data + cell_cons_interp of the time-interpolated coarse
                                                 state (Castro_setup.cpp:352-364, Castro.cpp:4201-4209), then
                                                 clean_state of the ghost zones (Castro_advance.cpp:186)
  Castro::FluxRegCrseInit / FluxRegFineAdd       Castro.cpp:2487-2545
  Castro::post_timestep: reflux, avgDown,        Castro.cpp:2549-2700, 3096-3113, post_timestep (:2140-2260)
      clean_state
  Castro::computeNewDt / computeInitialDt        Castro.cpp:1629-1866 over both levels (n_cycle = 1, 2)

  Castro::errorEst + Amr::regrid                 amr.refinement_indicators (AMRErrorTag [3P] restated) on the coarse level
                                                 every regrid_int steps; the refined region is ONE box, the bounding
                                                 box of the tags grown by n_error_buf and aligned to blocking_factor
                                                 (AMReX clusters tags into many boxes with Berger-Rigoutsos [3P])

Any number of levels, either fixed, properly nested patches (`patches=[...]`) or tag-driven (`refine=[...]`,
`max_level`): Amr::grid_places restated for one box per level -- tags are evaluated from the finest level down,
each new box is the aligned, buffered bounding box of the tags of the level below united with the (coarsened,
buffered) new box of the level above, so the hierarchy stays properly nested; a regrid adds at most one level.
Not provided: Berger-Rigoutsos clustering, more than one patch per level, multi-rank AMR, gravity on AMR levels.  The interpolation and flux-register arithmetic is AMReX's, restated
(include/castro_hydro_amd.h): parity with an AMReX build is unpinned.
"""
import torch

from . import _lib as L
from .castro import Castro, NUM_GROW, NUM_STATE


def _coarsen(i):
    return i // 2            # floor division is AMReX's coarsen() for negative indices too


class _FineLevel(Castro):
    """The refined patch: FillPatch takes the ghost zones from the coarse level."""

    def bind(self, crse):
        self.crse = crse
        self.alpha = 0.0          # (t_fine - t_crse_old) / dt_crse of the advance being prepared
        # coarse zones under the grown fine box, grown by one for the slopes
        self.cbox = (tuple(_coarsen(self.glo[d]) - 1 for d in range(3)), tuple(_coarsen(self.ghi[d]) + 1 for d in range(3)))
        for d in range(3):
            assert self.cbox[0][d] >= crse.glo[d] and self.cbox[1][d] <= crse.ghi[d], \
                "patch not properly nested: its ghost zones need parent data beyond the parent's own ghost zones"
        self.ctmp = self.hydro.alloc(NUM_STATE, *self.cbox)
        lo, hi, g = self.lo, self.hi, NUM_GROW
        glo, ghi = self.glo, self.ghi
        self.shell = [((glo[0], glo[1], glo[2]), (ghi[0], ghi[1], lo[2] - 1)), ((glo[0], glo[1], hi[2] + 1), (ghi[0], ghi[1], ghi[2])),
                      ((glo[0], glo[1], lo[2]), (ghi[0], lo[1] - 1, hi[2])), ((glo[0], hi[1] + 1, lo[2]), (ghi[0], ghi[1], hi[2])),
                      ((glo[0], lo[1], lo[2]), (lo[0] - 1, hi[1], hi[2])), ((hi[0] + 1, lo[1], lo[2]), (ghi[0], hi[1], hi[2]))]

    def expand_state(self, S, box=None, neighbors=None):
        assert box is None, "the refined patch carries no Source_Type data"
        h, c = self.hydro, self.crse
        a = self.alpha
        # StateData time interpolation of the coarse data: (1 - a) old + a new
        h.lincomb(self.ctmp, self.cbox, 1.0 - a, c.S_old_b, c.gbox, a, c.S_new_b, c.gbox, NUM_STATE, *self.cbox)
        for lo, hi in self.shell:
            h.cc_interp(self.ctmp, self.cbox, S, self.gbox, lo, hi, NUM_STATE)
        h.bc_fill(S, self.gbox, self.geom)                     # fine zones outside the domain (none for an interior patch)
        for lo, hi in self.shell:                              # clean_state(Sborder) reaches the ghost zones too
            h.clean_state(S, self.gbox, lo, hi, self.params, ntimes=1)


_TAG_KINDS = {"value_greater": 0, "value_less": 1, "gradient": 2, "relative_gradient": 3}
_FIELDS = {"density": 0, "xmom": 1, "ymom": 2, "zmom": 3, "rho_E": 4, "rho_e": 5, "Temp": 6, "rho_X": 7}


class CastroAmr:
    def __init__(self, n_cell, patch_crse=None, prob_lo=(0., 0., 0.), prob_hi=(1., 1., 1.), lo_bc=(2, 2, 2), hi_bc=(2, 2, 2),
                 params=None, make_hydro=None, make_params=None, refine=None, regrid_int=2, n_error_buf=1,
                 blocking_factor=8, patches=None, max_level=1):
        """patch_crse = (lo, hi): the coarse zones covered by a FIXED refined patch;
        patches = [(lo, hi), ...]: one fixed patch per finer level, each in the index space of the level below it
        (amr.max_level = len(patches)); or
        refine = [(field, kind, value), ...] like amr.refinement_indicators (field: a state name, kind:
        value_greater | value_less | gradient | relative_gradient) for patches that follow the tags, one per level up
        to amr.max_level = max_level."""
        if patch_crse is not None:
            assert patches is None
            patches = [patch_crse]
        assert (patches is None) != (refine is None), "give either fixed patches or refinement indicators"
        self._mk = (lambda: None) if make_hydro is None else make_hydro
        self.params = params if params is not None else (make_params() if make_params else L.default_params())
        self._kw = dict(prob_lo=prob_lo, prob_hi=prob_hi, lo_bc=lo_bc, hi_bc=hi_bc, params=self.params, overlap=False)
        self.n_cell = tuple(n_cell)
        self.lev = [Castro(n_cell, hydro=self._mk(), **self._kw)]      # lev[0] covers the domain
        self.pbox = [None]                                            # pbox[l]: patch of level l in level l-1 zones
        self.regs = [None]                                            # regs[l]: flux register around that patch
        self._hydros = [self.lev[0].hydro]
        self.refine = refine
        self.regrid_int, self.n_error_buf, self.blocking_factor = int(regrid_int), int(n_error_buf), int(blocking_factor)
        self.nregrid = 0
        self.max_level = int(max_level) if refine is not None else len(patches or [])
        for pb in (patches or []):
            self._push_level(tuple(pb[0]), tuple(pb[1]))
        self.time, self.nstep = 0.0, 0
        self.dt_level = [0.0] * 8

    # two-level views used by the tag-driven mode and the tests
    crse = property(lambda self: self.lev[0])
    fine = property(lambda self: self.lev[1] if len(self.lev) > 1 else None)
    plo = property(lambda self: self.pbox[1][0] if len(self.lev) > 1 else None)
    phi = property(lambda self: self.pbox[1][1] if len(self.lev) > 1 else None)
    levels = property(lambda self: list(self.lev))

    def _hydro_for(self, l):
        while len(self._hydros) <= l:
            h = self._mk()
            if h is None:
                from .hydro import HipHydro
                h = HipHydro(torch.cuda.current_device())
            self._hydros.append(h)
        return self._hydros[l]

    def _make_level(self, l, plo, phi):
        """Level l >= 1 covering the zones [plo, phi] of level l-1, and the flux register around it."""
        parent = self.lev[l - 1]
        flo = tuple(2 * x for x in plo)
        fhi = tuple(2 * x + 1 for x in phi)
        fine = _FineLevel(tuple((2 ** l) * x for x in self.n_cell), hydro=self._hydro_for(l), box=(flo, fhi), **self._kw)
        fine.bind(parent)
        h = parent.hydro
        reg = {}
        for d in range(3):
            for side in (0, 1):
                lo, hi = list(plo), list(phi)
                lo[d] = hi[d] = (plo[d] if side == 0 else phi[d] + 1)
                reg[(d, side)] = (h.alloc(NUM_STATE, lo, hi), (tuple(lo), tuple(hi)))
        return fine, reg

    def _push_level(self, plo, phi):
        fine, reg = self._make_level(len(self.lev), plo, phi)
        self.lev.append(fine); self.pbox.append((plo, phi)); self.regs.append(reg)

    def _drop_fine(self):
        del self.lev[1:], self.pbox[1:], self.regs[1:]

    # ---- Castro::errorEst (Castro.cpp:3131-3164) + the one-box stand-in for the grid generator ------
    def _fill_ghosts_new(self, upto):
        """Ghost zones of the new-time data of levels 0..upto (each FillPatch reads the level below it)."""
        for l in range(upto + 1):
            lev = self.lev[l]
            if l > 0:
                lev.alpha = 1.0
            lev.expand_state(lev.S_new_b)

    def _align(self, lo, hi, l):
        """Grow [lo, hi] (level-l zones) to multiples of blocking_factor/2 and clip it to the level-l domain."""
        a = max(self.blocking_factor // 2, 1)                   # blocking_factor is in zones of level l+1
        olo, ohi = [], []
        for d in range(3):
            olo.append(max((lo[d] // a) * a, 0))
            ohi.append(min(-((-(hi[d] + 1)) // a) * a - 1, (2 ** l) * self.n_cell[d] - 1))
        return tuple(olo), tuple(ohi)

    def tag_box(self, l=0, ghosts_filled=False):
        """Bounding box (level-l zones) of the tagged zones of level l, buffered and aligned; None if nothing is
        tagged.  The box stays inside level l's own box."""
        c = self.lev[l]
        h = c.hydro
        if not ghosts_filled:
            self._fill_ghosts_new(l)
        tags = h.alloc(1, c.lo, c.hi)
        for field, kind, value in self.refine:
            h.error_tag(c.S_new_b, c.gbox, _FIELDS[field], tags, (c.lo, c.hi), c.lo, c.hi, _TAG_KINDS[kind], value)
        nz = torch.nonzero(tags[0] > 0.5)                       # (k, j, i) triples relative to c.lo
        if nz.numel() == 0:
            return None
        mn, mx = nz.min(dim=0).values.tolist(), nz.max(dim=0).values.tolist()
        lo = tuple(max(c.lo[d] + mn[2 - d] - self.n_error_buf, c.lo[d]) for d in range(3))
        hi = tuple(min(c.lo[d] + mx[2 - d] + self.n_error_buf, c.hi[d]) for d in range(3))
        return self._align(lo, hi, l)

    def _grid_places(self):
        """boxes[l] (l >= 1, level l-1 zones) of the new hierarchy, or None from the first level that disappears."""
        finest = len(self.lev) - 1
        top = min(finest, self.max_level - 1)                   # the finest level that may carry tags
        self._fill_ghosts_new(top)
        boxes = {}
        for l in range(top, -1, -1):
            b = self.tag_box(l, ghosts_filled=True)
            up = boxes.get(l + 2)
            if up is not None:                                  # proper nesting: cover the level above + a buffer
                ulo = tuple(_coarsen(up[0][d]) - self.n_error_buf for d in range(3))
                uhi = tuple(_coarsen(up[1][d]) + self.n_error_buf for d in range(3))
                c = self.lev[l]
                ulo = tuple(max(ulo[d], c.lo[d]) for d in range(3))
                uhi = tuple(min(uhi[d], c.hi[d]) for d in range(3))
                if b is not None:
                    ulo = tuple(min(ulo[d], b[0][d]) for d in range(3))
                    uhi = tuple(max(uhi[d], b[1][d]) for d in range(3))
                b = self._align(ulo, uhi, l)
            boxes[l + 1] = b
        out = []
        for l in range(1, top + 2):
            if boxes.get(l) is None:
                break
            out.append(boxes[l])
        # a box of level l+1 must lie inside the NEW box of level l (tags were taken on the old one)
        for i in range(1, len(out)):
            plo, phi = out[i - 1]
            lo = tuple(max(out[i][0][d], 2 * plo[d]) for d in range(3))
            hi = tuple(min(out[i][1][d], 2 * phi[d] + 1) for d in range(3))
            if any(lo[d] > hi[d] for d in range(3)):
                del out[i:]
                break
            out[i] = (lo, hi)
        return out

    # ---- Amr::regrid: new fine grids, data from the old fine level where it exists, else interpolated ---
    def regrid(self):
        boxes = self._grid_places()
        if boxes == self.pbox[1:]:
            return False
        old_lev, old_S = list(self.lev), [lev.S_new_b for lev in self.lev]
        keep = 1                                                # levels below the first changed box are kept as they are
        while keep <= min(len(boxes), len(self.lev) - 1) and boxes[keep - 1] == self.pbox[keep]:
            keep += 1
        del self.lev[keep:], self.pbox[keep:], self.regs[keep:]
        for l in range(keep, len(boxes) + 1):
            self._push_level(*boxes[l - 1])
            new, c = self.lev[l], self.lev[l - 1]
            h = c.hydro
            # FillCoarsePatch: cell-conservative interpolation of the (ghost-filled) coarse data over the whole new box
            if l - 1 >= keep:                                   # a level made in this regrid: its ghost zones are not filled yet
                c.alpha = 1.0
                c.expand_state(c.S_new_b)
            h.lincomb(new.ctmp, new.cbox, 0.0, c.S_new_b, c.gbox, 1.0, c.S_new_b, c.gbox, NUM_STATE, *new.cbox)
            h.cc_interp(new.ctmp, new.cbox, new.S_new_b, new.gbox, new.lo, new.hi, NUM_STATE)
            if l < len(old_lev):
                old = old_lev[l]
                olo = tuple(max(old.lo[d], new.lo[d]) for d in range(3))
                ohi = tuple(min(old.hi[d], new.hi[d]) for d in range(3))
                if all(olo[d] <= ohi[d] for d in range(3)):
                    h.copy(new.S_new_b, new.gbox, old_S[l], old.gbox, olo, ohi)
            new.time, new.nstep = self.time, self.nstep
        self.nregrid += 1
        return True

    # ---- Amr::init (bldFineLevels: one new level per pass) / Castro::post_init ---------------------------
    def initData(self, problem="sedov", **kw):
        self.crse.initData(problem, **kw)
        if self.refine is not None:
            self._drop_fine()
            while len(self.lev) - 1 < self.max_level:
                box = self.tag_box(len(self.lev) - 1)
                if box is None:
                    break
                self._push_level(*box)
                self.lev[-1].initData(problem, **kw)            # fine levels start from the problem initialiser
        else:
            for lev in self.lev[1:]:
                lev.initData(problem, **kw)
        for l in range(len(self.lev) - 1, 0, -1):
            self.avgDown(l)
            self.lev[l - 1].clean_state(self.lev[l - 1].S_new_b, 1)
        self.time, self.nstep = 0.0, 0

    # ---- Castro::avgDown (Castro.cpp:3096-3113): level l onto level l-1 ------------------------------
    def avgDown(self, l=1):
        c, f = self.lev[l - 1], self.lev[l]
        c.hydro.avgdown(f.S_new_b, f.gbox, c.S_new_b, c.gbox, self.pbox[l][0], self.pbox[l][1], NUM_STATE)

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
        lev, finest = self.lev[l], len(self.lev) - 1
        if l > 0:
            lev.alpha = alpha
        lev.advance(t, dt)
        if l > 0:
            # FluxRegFineAdd: + this level's fluxes (already dt x area) summed over the 4 fine faces
            h = self.lev[l - 1].hydro
            for (d, side), (reg, rbox) in self.regs[l].items():
                h.fluxreg_fine_add(reg, rbox, lev.fluxes[d], lev.flux_boxes[d], rbox[0], rbox[1], d, NUM_STATE, 1.0)
        if l < finest:
            h = lev.hydro
            # ghost zones of the new data of this level, for the FillPatch of the next finer one
            if l > 0:
                lev.alpha = alpha + 0.5
            lev.expand_state(lev.S_new_b)
            # FluxRegCrseInit: -1 x this level's fluxes through the faces of the finer patch
            for (d, side), (reg, rbox) in self.regs[l + 1].items():
                h.fluxreg_crse_init(reg, rbox, lev.fluxes[d], lev.flux_boxes[d], rbox[0], rbox[1], NUM_STATE, -1.0)
            for it in range(2):
                self._time_step(l + 1, t + it * (dt / 2), dt / 2, 0.5 * it)
            # post_timestep: reflux, avgDown, clean_state
            vol = lev.geom.dx[0] * lev.geom.dx[1] * lev.geom.dx[2]
            for (d, side), (reg, rbox) in self.regs[l + 1].items():
                h.reflux(lev.S_new_b, lev.gbox, reg, rbox, rbox[0], rbox[1], d, side, NUM_STATE, vol)
            self.avgDown(l + 1)
            lev.clean_state(lev.S_new_b, 1)

    # ---- Amr::coarseTimeStep ---------------------------------------------------------------------
    def step(self, stop_time=-1.0):
        if self.refine is not None and self.regrid_int > 0 and self.nstep > 0 and self.nstep % self.regrid_int == 0:
            self.regrid()
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
            S = lev.S_new()[comp]
            if l + 1 < len(self.lev):
                S = S.clone()
                p, q = self.pbox[l + 1]
                o = lev.lo
                S[p[2] - o[2]:q[2] - o[2] + 1, p[1] - o[1]:q[1] - o[1] + 1, p[0] - o[0]:q[0] - o[0] + 1] = 0.0
            tot += S.sum().item() * v
        return tot
