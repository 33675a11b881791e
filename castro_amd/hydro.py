"""Host-side mirror of the reference's hydro interface over the C ABI.

`HipHydro` is the per-(device, stream) object whose methods carry the reference's names
(`construct_ctu_hydro_source`, `clean_state`, `estdt_cfl`, ...) and take FABs as torch CUDA
tensors shaped (ncomp, nz, ny, nx) plus their index box.  torch is used for device
memory and streams only; every operation is a call into libcastro_hydro_amd.so.
"""
import ctypes as C

import torch

from . import _lib as L


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(stream):
    """hipStream_t of `stream`, or of torch's current stream on the current device (the raw-handle query is ~20x cheaper
    than torch.cuda.current_stream(), which matters when a level of small boxes issues thousands of calls per step)."""
    if stream is None:
        if _raw_stream is not None:
            return C.c_void_p(_raw_stream(torch.cuda.current_device()))
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


class HipHydro:
    """One castro_amd_ctx.  No CPU fallback: constructing it without a HIP device raises."""

    name = "hip"

    def __init__(self, device=0, numerics=None):
        """numerics: "exact" | "contract" (None: CASTRO_AMD_NUMERICS, default exact) -- which build of the kernel library
        this context runs (castro_amd/_lib.py)."""
        if not torch.cuda.is_available():
            raise RuntimeError("castro_amd.HipHydro needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.lib = L.load(numerics)
        self.numerics = L.numerics_of(self.lib)
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        L.check(self.lib.castro_amd_ctx_create(C.byref(h), int(device)), "ctx_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.castro_amd_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            # a finaliser that runs in the middle of a stream capture must not free device memory (the runtime aborts):
            # leave the context to the end of the process instead
            if torch.cuda.is_current_stream_capturing():
                return
            self.close()
        except Exception:
            pass

    # ---- memory -------------------------------------------------------------------------
    def alloc(self, ncomp, lo, hi, fill=0.0):
        """A FAB for box [lo,hi] as a torch tensor (ncomp, nz, ny, nx), FP64, on the device."""
        shape = (ncomp, hi[2] - lo[2] + 1, hi[1] - lo[1] + 1, hi[0] - lo[0] + 1)
        return torch.full(shape, fill, dtype=torch.float64, device=self.device)

    def reserve(self, nx, ny, nz):
        L.check(self.lib.castro_amd_ctx_reserve(self.h, nx, ny, nz), "ctx_reserve")

    def scratch_bytes(self):
        return int(self.lib.castro_amd_ctx_scratch_bytes(self.h))

    def status(self, stream=None):
        return int(self.lib.castro_amd_ctx_status(self.h, _stream_ptr(stream)))

    def set_source_corrector(self, corr, box):
        """Castro::source_corrector for the hydro calls that follow (params.source_term_predictor = 1); None clears it."""
        fab = L.fab_of(corr, *box) if corr is not None else L.fab_desc(None, (0, 0, 0), (0, 0, 0), 0)
        L.check(self.lib.castro_amd_ctx_set_source_corrector(self.h, C.byref(fab)), "ctx_set_source_corrector")

    def poison_scratch(self, stream=None):
        """NaN-fill the scratch arena (tests: a call must not read what an earlier call left there)."""
        L.check(self.lib.castro_amd_ctx_poison_scratch(self.h, _stream_ptr(stream)), "ctx_poison_scratch")

    # ---- pointwise forms (known-answer vectors at the level of the reference's own functions) ----
    def cmpflx_points(self, idir, qm, qp, cl, cr, params, bnd_fac=None, is_shock=None):
        """Castro::cmpflx_plus_godunov's body on n interfaces: qm, qp (7, n); returns (11, n)."""
        n = qm.shape[1]
        out = torch.empty((11, n), dtype=torch.float64, device=self.device)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        L.check(self.lib.castro_amd_cmpflx_points(n, int(idir), ptr(qm), ptr(qp), ptr(cl), ptr(cr), ptr(bnd_fac), ptr(is_shock),
                                                  C.byref(params), ptr(out), _stream_ptr(None)), "cmpflx_points")
        return out

    def ppm_points(self, s, flatn, u, c, dtdx):
        """ppm_reconstruct + ppm_int_profile: s (5, n) -> (sm, sp, Ip[3], Im[3]) as (8, n)."""
        n = s.shape[1]
        out = torch.empty((8, n), dtype=torch.float64, device=self.device)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        L.check(self.lib.castro_amd_ppm_points(n, ptr(s), ptr(flatn), ptr(u), ptr(c), float(dtdx), ptr(out), _stream_ptr(None)),
                "ppm_points")
        return out

    def flatten_points(self, p7, u5):
        n = p7.shape[1]
        out = torch.empty((n,), dtype=torch.float64, device=self.device)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        L.check(self.lib.castro_amd_flatten_points(n, ptr(p7), ptr(u5), ptr(out), _stream_ptr(None)), "flatten_points")
        return out

    def trans_points(self, q, f1r, f1l, cdtdx1, params, tdir=0, f2r=None, f2l=None, cdtdx2=0.0, fe=None):
        """actual_trans_single (f2r is None) / actual_trans_final: q (7, n), flux records (8, n) -> (7, n);
        fe: (2 or 4, n) (rho e) fluxes for transverse_reset_rhoe = 1."""
        n = q.shape[1]
        out = torch.empty((7, n), dtype=torch.float64, device=self.device)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        L.check(self.lib.castro_amd_trans_points(n, 2 if f2r is not None else 1, int(tdir), ptr(q), ptr(f1r), ptr(f1l), ptr(f2r),
                                                 ptr(f2l), ptr(fe), float(cdtdx1), float(cdtdx2), C.byref(params), ptr(out),
                                                 _stream_ptr(None)), "trans_points")
        return out

    # ---- the hot path: Castro::construct_ctu_hydro_source, one FAB/tile --------------------
    def construct_ctu_hydro_source(self, bx, Sborder, sb_box, S_new, snew_box, geom, params, time, dt,
                                   fluxes=None, flux_boxes=None, mass_fluxes=None, qe=None, vbx=None,
                                   update_from_sborder=False, src=None, src_box=None, stream=None,
                                   clean_ntimes=0, red=None, flux_assign=False, stage=None, sborder_clean=0, d_dt=None, bc_fill=False):
        """clean_ntimes > 0 selects castro_amd_ctu_hydro_clean_fab: the update is followed, in the same
        pass, by S_new.min(URHO), clean_state x clean_ntimes and the CFL estimate, reduced into `red`.
        sborder_clean > 0 (castro_amd_ctu_hydro_fab_ex): clean_state that many times on every zone of Sborder, in place,
        inside the pass that reads it.  d_dt: a device tensor whose first element is the time step (host-free stepping;
        `dt` is ignored)."""
        bxlo, bxhi = bx
        vlo, vhi = vbx if vbx is not None else bx
        fb = (L.Fab * 3)()
        mb = (L.Fab * 3)()
        qb = (L.Fab * 3)()
        for d in range(3):
            if flux_boxes is not None:
                flo, fhi = flux_boxes[d]
            else:
                flo, fhi = list(vlo), list(vhi)
                fhi[d] += 1
            fb[d] = L.fab_of(fluxes[d] if fluxes is not None else None, flo, fhi)
            mb[d] = L.fab_of(mass_fluxes[d] if mass_fluxes is not None else None, flo, fhi)
            qb[d] = L.fab_of(qe[d] if qe is not None else None, flo, fhi)
        sfab = L.fab_of(src, *src_box) if src is not None else L.fab_desc(None, bxlo, bxhi, 0)
        flags = L.UPDATE_FROM_SBORDER if update_from_sborder else L.UPDATE_ADD
        if flux_assign:
            flags |= L.FLUX_ASSIGN
        if stage is not None:         # "A": the ghost-free part (overlaps the halo exchange); "B": the rest -- the round-2 split;
            #                           "valid": ctoprim (+ the cleans) on the valid zones only; "rest": everything else (round 6)
            flags |= {"A": L.STAGE_A, "B": L.STAGE_B, "valid": L.STAGE_VALID, "rest": L.STAGE_REST}[stage]
        if bc_fill:                   # the call fills the physical-boundary zones of Sborder itself (CASTRO_AMD_BC_FILL)
            flags |= L.BC_FILL
        if sborder_clean > 0 or d_dt is not None:
            o = L.HydroOpts(flags, int(clean_ntimes), red.data_ptr() if red is not None else None, int(sborder_clean),
                            d_dt.data_ptr() if d_dt is not None else None)
            rc = self.lib.castro_amd_ctu_hydro_fab_ex(
                self.h, L.i3(bxlo), L.i3(bxhi), L.i3(vlo), L.i3(vhi),
                C.byref(L.fab_of(Sborder, *sb_box)), C.byref(sfab), C.byref(L.fab_of(S_new, *snew_box)),
                fb, mb, qb, C.byref(geom), C.byref(params), float(time), float(dt), C.byref(o), _stream_ptr(stream))
            L.check(rc, "ctu_hydro_fab_ex")
            return
        if clean_ntimes > 0:
            rc = self.lib.castro_amd_ctu_hydro_clean_fab(
                self.h, L.i3(bxlo), L.i3(bxhi), L.i3(vlo), L.i3(vhi),
                C.byref(L.fab_of(Sborder, *sb_box)), C.byref(sfab), C.byref(L.fab_of(S_new, *snew_box)),
                fb, mb, qb, C.byref(geom), C.byref(params), float(time), float(dt), flags,
                int(clean_ntimes), C.c_void_p(red.data_ptr()) if red is not None else None, _stream_ptr(stream))
            L.check(rc, "ctu_hydro_clean_fab")
            return
        rc = self.lib.castro_amd_ctu_hydro_fab(
            self.h, L.i3(bxlo), L.i3(bxhi), L.i3(vlo), L.i3(vhi),
            C.byref(L.fab_of(Sborder, *sb_box)), C.byref(sfab), C.byref(L.fab_of(S_new, *snew_box)),
            fb, mb, qb, C.byref(geom), C.byref(params), float(time), float(dt), flags, _stream_ptr(stream))
        L.check(rc, "ctu_hydro_fab")

    # ---- gravity source terms (Source/gravity/Castro_gravity.cpp) and apply_source_to_state ----------
    def old_gravity_source(self, state, box, source, src_box, lo, hi, grav, grav_source_type, dt, stream=None):
        g = (C.c_double * 3)(*[float(x) for x in grav])
        L.check(self.lib.castro_amd_old_gravity_source_fab(self.h, C.byref(L.fab_of(state, *box)), C.byref(L.fab_of(source, *src_box)),
                                                           L.i3(lo), L.i3(hi), C.byref(g), int(grav_source_type), float(dt),
                                                           _stream_ptr(stream)), "old_gravity_source_fab")

    def new_gravity_source(self, state_old, old_box, state_new, new_box, source, src_box, mass_fluxes, flux_boxes, lo, hi,
                           grav, grav_source_type, dt, geom, stream=None):
        g = (C.c_double * 3)(*[float(x) for x in grav])
        mb = (L.Fab * 3)()
        for d in range(3):
            mb[d] = L.fab_of(mass_fluxes[d], *flux_boxes[d])
        L.check(self.lib.castro_amd_new_gravity_source_fab(self.h, C.byref(L.fab_of(state_old, *old_box)),
                                                           C.byref(L.fab_of(state_new, *new_box)),
                                                           C.byref(L.fab_of(source, *src_box)), mb, L.i3(lo), L.i3(hi), C.byref(g),
                                                           int(grav_source_type), float(dt), C.byref(geom), _stream_ptr(stream)),
                "new_gravity_source_fab")

    def old_rotation_source(self, state, box, source, src_box, lo, hi, rot, geom, dt, stream=None):
        L.check(self.lib.castro_amd_old_rotation_source_fab(self.h, C.byref(L.fab_of(state, *box)), C.byref(L.fab_of(source, *src_box)),
                                                            L.i3(lo), L.i3(hi), C.byref(rot), C.byref(geom), float(dt),
                                                            _stream_ptr(stream)), "old_rotation_source_fab")

    def new_rotation_source(self, state_old, old_box, state_new, new_box, source, src_box, mass_fluxes, flux_boxes, lo, hi,
                            rot, geom, dt, stream=None):
        mb = (L.Fab * 3)()
        for d in range(3):
            mb[d] = L.fab_of(mass_fluxes[d], *flux_boxes[d])
        L.check(self.lib.castro_amd_new_rotation_source_fab(self.h, C.byref(L.fab_of(state_old, *old_box)),
                                                            C.byref(L.fab_of(state_new, *new_box)),
                                                            C.byref(L.fab_of(source, *src_box)), mb, L.i3(lo), L.i3(hi),
                                                            C.byref(rot), C.byref(geom), float(dt), _stream_ptr(stream)),
                "new_rotation_source_fab")

    def saxpy(self, dst, dst_box, a, src, src_box, ncomp, lo, hi, stream=None):
        """dst[:ncomp] += a * src[:ncomp] on [lo,hi] (Castro::apply_source_to_state)."""
        L.check(self.lib.castro_amd_saxpy_fab(self.h, C.byref(L.fab_of(dst, *dst_box)), float(a), C.byref(L.fab_of(src, *src_box)),
                                              int(ncomp), L.i3(lo), L.i3(hi), _stream_ptr(stream)), "saxpy_fab")

    def apply_source(self, dst, dst_box, base, base_box, a, src, src_box, nsrc, lo, hi, params, ntimes=1, stream=None):
        """dst = base + a * src[:nsrc] (other components copied) followed by clean_state x ntimes, one pass."""
        L.check(self.lib.castro_amd_apply_source_fab(self.h, C.byref(L.fab_of(dst, *dst_box)), C.byref(L.fab_of(base, *base_box)),
                                                     float(a), C.byref(L.fab_of(src, *src_box)), int(nsrc), L.i3(lo), L.i3(hi),
                                                     C.byref(params), int(ntimes), _stream_ptr(stream)), "apply_source_fab")

    # ---- two-level AMR building blocks (include/castro_hydro_amd.h) ------------------------------
    def error_tag(self, field, field_box, comp, tags, tags_box, lo, hi, kind, value, stream=None):
        """kind: 0 value_greater, 1 value_less, 2 gradient, 3 relative_gradient (AMRErrorTag)."""
        L.check(self.lib.castro_amd_error_tag_fab(self.h, C.byref(L.fab_of(field, *field_box)), int(comp),
                                                  C.byref(L.fab_of(tags, *tags_box)), L.i3(lo), L.i3(hi), int(kind),
                                                  float(value), _stream_ptr(stream)), "error_tag_fab")

    def cc_interp(self, crse, crse_box, fine, fine_box, lo, hi, ncomp, stream=None):
        L.check(self.lib.castro_amd_cc_interp_fab(self.h, C.byref(L.fab_of(crse, *crse_box)), C.byref(L.fab_of(fine, *fine_box)),
                                                  L.i3(lo), L.i3(hi), int(ncomp), _stream_ptr(stream)), "cc_interp_fab")

    def fillpatch_shell(self, crse, crse_box, fine, fine_box, vlo, vhi, ngrow, params, ntimes=1, stream=None):
        """cc_interp + clean_state x ntimes on grow([vlo, vhi], ngrow) minus [vlo, vhi], one launch."""
        L.check(self.lib.castro_amd_fillpatch_shell_fab(self.h, C.byref(L.fab_of(crse, *crse_box)), C.byref(L.fab_of(fine, *fine_box)),
                                                        L.i3(vlo), L.i3(vhi), int(ngrow), C.byref(params), int(ntimes),
                                                        _stream_ptr(stream)), "fillpatch_shell_fab")

    @staticmethod
    def make_ops(specs):
        """ctypes array of castro_amd_fab_op from (kind, dir, ncomp, lo, hi, a, b, (dst, box), (src, box), (src2, box) | None);
        for OP_REFLUX `dir` is the pair (dir, side) and `a` the zone volume."""
        arr = (L.FabOp * max(len(specs), 1))()
        for o, (kind, dir_, ncomp, lo, hi, a, b, dst, src, src2) in zip(arr, specs):
            if isinstance(dir_, tuple):
                dir_, o.side = dir_
            o.kind, o.dir, o.ncomp, o.a, o.b = kind, dir_, ncomp, a, b
            for d in range(3):
                o.lo[d], o.hi[d] = lo[d], hi[d]
            o.dst, o.src = L.fab_of(dst[0], *dst[1]), L.fab_of(src[0], *src[1])
            o.src2 = L.fab_of(src2[0], *src2[1]) if src2 is not None else o.src
        return arr, len(specs)

    def fab_ops(self, ops, stream=None, params=None):
        """Independent FAB-to-FAB region operations (make_ops): one launch for up to sixteen, and with `params`
        (castro_amd_fab_ops_p: needed by OP_CLEAN / OP_INTERP_CLEAN) one launch for any number."""
        arr, n = ops
        if n:
            if params is not None:
                L.check(self.lib.castro_amd_fab_ops_p(self.h, n, arr, C.byref(params), _stream_ptr(stream)), "fab_ops_p")
            else:
                L.check(self.lib.castro_amd_fab_ops(self.h, n, arr, _stream_ptr(stream)), "fab_ops")

    @staticmethod
    def make_hydro_boxes(specs):
        """ctypes array of castro_amd_hydro_box from (bx, vbx, (Sborder, box), (S_new, box), fluxes, flux_boxes, mass_fluxes
        [, (src, box)]) -- src: the old-time source FAB of the box (Source_Type data with ghost zones), traced by the call."""
        arr = (L.HydroBox * max(len(specs), 1))()
        for hb, spec in zip(arr, specs):
            bx, vbx, sb, sn, fluxes, flux_boxes, mass_fluxes = spec[:7]
            src = spec[7] if len(spec) > 7 else None
            for d in range(3):
                hb.bxlo[d], hb.bxhi[d], hb.vbxlo[d], hb.vbxhi[d] = bx[0][d], bx[1][d], vbx[0][d], vbx[1][d]
                hb.flux[d] = L.fab_of(fluxes[d], *flux_boxes[d])
                hb.mass_flux[d] = L.fab_of(mass_fluxes[d], *flux_boxes[d])
                hb.qe[d] = L.fab_of(None, *flux_boxes[d])
            hb.Sborder, hb.S_new = L.fab_of(sb[0], *sb[1]), L.fab_of(sn[0], *sn[1])
            hb.src = L.fab_of(src[0], *src[1]) if src is not None else L.fab_desc(None, bx[0], bx[1], 0)
        return arr, len(specs)

    def construct_ctu_hydro_source_mf(self, pool, boxes, geom, params, time, dt, update_from_sborder=True, flux_assign=False,
                                      clean_ntimes=0, red=None, stream=None):
        """castro_amd_ctu_hydro_mf: the hydro update of every box of a level in one call (the MFIter loop).
        pool: [(HipHydro, torch stream)] the boxes are dealt to round robin, or None for this context on the current stream."""
        arr, n = boxes
        if not n:
            return
        flags = L.UPDATE_FROM_SBORDER if update_from_sborder else L.UPDATE_ADD
        if flux_assign:
            flags |= L.FLUX_ASSIGN
        o = L.HydroOpts(flags, int(clean_ntimes), red.data_ptr() if red is not None else None, 0, None)
        main = _stream_ptr(stream)
        if pool:
            ctxs = (C.c_void_p * len(pool))(*[h.h for h, _ in pool])
            sts = (C.c_void_p * len(pool))(*[C.c_void_p(st.cuda_stream) for _, st in pool])
            k = len(pool)
        else:
            ctxs, sts, k = (C.c_void_p * 1)(self.h), (C.c_void_p * 1)(main), 1
        L.check(self.lib.castro_amd_ctu_hydro_mf(ctxs, sts, k, arr, n, C.byref(geom), C.byref(params), float(time), float(dt),
                                                 C.byref(o), main), "ctu_hydro_mf")

    # ---- the per-box stages around the hydro update, for every box of a level in one call ----------------
    @staticmethod
    def make_source_boxes(specs):
        """ctypes array of castro_amd_source_box from (lo, hi, (S_old, box), (S_new, box), (source, box), mass_fluxes, flux_boxes)."""
        arr = (L.SourceBox * max(len(specs), 1))()
        for sb, (lo, hi, so, sn, src, mass_fluxes, flux_boxes) in zip(arr, specs):
            for d in range(3):
                sb.lo[d], sb.hi[d] = lo[d], hi[d]
                sb.mass_flux[d] = L.fab_of(mass_fluxes[d], *flux_boxes[d])
            sb.S_old, sb.S_new, sb.source = L.fab_of(so[0], *so[1]), L.fab_of(sn[0], *sn[1]), L.fab_of(src[0], *src[1])
        return arr, len(specs)

    def sources_mf(self, stage, boxes, grav, grav_source_type, rot, geom, params, dt, ntimes=1, stream=None):
        """castro_amd_sources_mf: stage 0 = old-time sources + S_new = S_old + dt * source + clean_state, stage 1 = new-time
        sources + S_new += dt * source + clean_state, for every box of `boxes` (make_source_boxes)."""
        arr, n = boxes
        if n:
            g = (C.c_double * 3)(*[float(x) for x in grav]) if grav is not None else None
            L.check(self.lib.castro_amd_sources_mf(self.h, int(stage), n, arr, g, int(grav_source_type),
                                                   C.byref(rot) if rot is not None else None, C.byref(geom), C.byref(params),
                                                   float(dt), int(ntimes), _stream_ptr(stream)), "sources_mf")

    @staticmethod
    def make_state_boxes(specs):
        """ctypes array of castro_amd_state_box from (lo, hi, (state, box))."""
        arr = (L.StateBox * max(len(specs), 1))()
        for sb, (lo, hi, st) in zip(arr, specs):
            for d in range(3):
                sb.lo[d], sb.hi[d] = lo[d], hi[d]
            sb.state = L.fab_of(st[0], *st[1])
        return arr, len(specs)

    def clean_state_reduce_mf(self, boxes, geom, params, out, ntimes=1, stream=None):
        arr, n = boxes
        if n:
            L.check(self.lib.castro_amd_clean_state_reduce_mf(self.h, n, arr, C.byref(geom), C.byref(params), int(ntimes),
                                                              C.c_void_p(out.data_ptr()), _stream_ptr(stream)), "clean_state_reduce_mf")

    def estdt_cfl_mf(self, boxes, geom, params, out, stream=None):
        arr, n = boxes
        if n:
            L.check(self.lib.castro_amd_estdt_mf(self.h, n, arr, C.byref(geom), C.byref(params), C.c_void_p(out.data_ptr()),
                                                 _stream_ptr(stream)), "estdt_mf")

    def lincomb(self, dst, dst_box, a, x, x_box, b, y, y_box, ncomp, lo, hi, stream=None):
        L.check(self.lib.castro_amd_lincomb_fab(self.h, C.byref(L.fab_of(dst, *dst_box)), float(a), C.byref(L.fab_of(x, *x_box)),
                                                float(b), C.byref(L.fab_of(y, *y_box)), int(ncomp), L.i3(lo), L.i3(hi),
                                                _stream_ptr(stream)), "lincomb_fab")

    def avgdown(self, fine, fine_box, crse, crse_box, lo, hi, ncomp, stream=None):
        L.check(self.lib.castro_amd_avgdown_fab(self.h, C.byref(L.fab_of(fine, *fine_box)), C.byref(L.fab_of(crse, *crse_box)),
                                                L.i3(lo), L.i3(hi), int(ncomp), _stream_ptr(stream)), "avgdown_fab")

    def fluxreg_crse_init(self, reg, reg_box, cflux, cflux_box, lo, hi, ncomp, mult, stream=None):
        L.check(self.lib.castro_amd_fluxreg_crse_init_fab(self.h, C.byref(L.fab_of(reg, *reg_box)), C.byref(L.fab_of(cflux, *cflux_box)),
                                                          L.i3(lo), L.i3(hi), int(ncomp), float(mult), _stream_ptr(stream)),
                "fluxreg_crse_init_fab")

    def fluxreg_fine_add(self, reg, reg_box, fflux, fflux_box, lo, hi, dir, ncomp, mult, stream=None):
        L.check(self.lib.castro_amd_fluxreg_fine_add_fab(self.h, C.byref(L.fab_of(reg, *reg_box)), C.byref(L.fab_of(fflux, *fflux_box)),
                                                         L.i3(lo), L.i3(hi), int(dir), int(ncomp), float(mult), _stream_ptr(stream)),
                "fluxreg_fine_add_fab")

    def reflux(self, state, state_box, reg, reg_box, lo, hi, dir, side, ncomp, vol, stream=None):
        L.check(self.lib.castro_amd_reflux_fab(self.h, C.byref(L.fab_of(state, *state_box)), C.byref(L.fab_of(reg, *reg_box)),
                                               L.i3(lo), L.i3(hi), int(dir), int(side), int(ncomp), float(vol), _stream_ptr(stream)),
                "reflux_fab")

    # ---- derived plotfile fields (Source/driver/Derive.cpp) ---------------------------------
    def derive(self, name, state, box, der, der_box, dcomp, lo, hi, geom, params, center, stream=None):
        ctr = (C.c_double * 3)(*[float(x) for x in center])
        L.check(self.lib.castro_amd_derive_fab(self.h, L.DERIVE_IDS[name], C.byref(L.fab_of(state, *box)),
                                               C.byref(L.fab_of(der, *der_box)), int(dcomp), L.i3(lo), L.i3(hi),
                                               C.byref(geom), C.byref(params), C.byref(ctr), _stream_ptr(stream)),
                "derive_fab")

    def step_control(self, red, ctl, params, max_dt=1.e200, fixed_dt=-1.0, stop_time=-1.0, use_retry=False, stream=None):
        """castro_amd_step_control: do_advance_ctu's checks, time += dt and computeNewDt on the device (one thread)."""
        L.check(self.lib.castro_amd_step_control(self.h, C.c_void_p(red.data_ptr()), C.c_void_p(ctl.data_ptr()), C.byref(params),
                                                 float(max_dt), float(fixed_dt), float(stop_time), 1 if use_retry else 0,
                                                 _stream_ptr(stream)),
                "step_control")

    # ---- Castro::clean_state ---------------------------------------------------------------
    def clean_state(self, state, box, lo, hi, params, ntimes=1, stream=None):
        L.check(self.lib.castro_amd_clean_state_fab(self.h, C.byref(L.fab_of(state, *box)), L.i3(lo), L.i3(hi),
                                                    C.byref(params), int(ntimes), _stream_ptr(stream)), "clean_state_fab")

    def clean_state_reduce(self, state, box, lo, hi, geom, params, out, ntimes=1, stream=None):
        """min-density check + clean_state + estdt in one pass (see castro_amd_clean_state_reduce_fab)."""
        L.check(self.lib.castro_amd_clean_state_reduce_fab(self.h, C.byref(L.fab_of(state, *box)), L.i3(lo), L.i3(hi),
                                                           C.byref(geom), C.byref(params), int(ntimes),
                                                           C.c_void_p(out.data_ptr()), _stream_ptr(stream)),
                "clean_state_reduce_fab")

    # ---- Castro::estdt_cfl + S_new.min(URHO) -------------------------------------------------
    def estdt_cfl(self, state, box, lo, hi, geom, params, out, stream=None):
        """Reduces into `out` (device tensor of 2 doubles: [min dx/(c+|u|), min rho]); the caller
        initialises it (e.g. out.fill_(1e200))."""
        L.check(self.lib.castro_amd_estdt_fab(self.h, C.byref(L.fab_of(state, *box)), L.i3(lo), L.i3(hi),
                                              C.byref(geom), C.byref(params), C.c_void_p(out.data_ptr()),
                                              _stream_ptr(stream)), "estdt_fab")

    # ---- FillPatch pieces --------------------------------------------------------------------
    def bc_fill(self, state, box, geom, stream=None):
        L.check(self.lib.castro_amd_bc_fill_fab(self.h, C.byref(L.fab_of(state, *box)), C.byref(geom),
                                                _stream_ptr(stream)), "bc_fill_fab")

    def copy(self, dst, dst_box, src, src_box, lo, hi, stream=None):
        L.check(self.lib.castro_amd_copy_fab(self.h, C.byref(L.fab_of(dst, *dst_box)), C.byref(L.fab_of(src, *src_box)),
                                             L.i3(lo), L.i3(hi), _stream_ptr(stream)), "copy_fab")

    def pack(self, state, box, lo, hi, buf, stream=None):
        L.check(self.lib.castro_amd_pack_fab(self.h, C.byref(L.fab_of(state, *box)), L.i3(lo), L.i3(hi),
                                             C.c_void_p(buf.data_ptr()), _stream_ptr(stream)), "pack_fab")

    def unpack(self, state, box, lo, hi, buf, stream=None):
        L.check(self.lib.castro_amd_unpack_fab(self.h, C.byref(L.fab_of(state, *box)), L.i3(lo), L.i3(hi),
                                               C.c_void_p(buf.data_ptr()), _stream_ptr(stream)), "unpack_fab")

    # ---- problem setups ----------------------------------------------------------------------
    @staticmethod
    def region_table(boxes, offsets):
        """ctypes arrays for pack_regions / unpack_regions: boxes [(lo, hi)], offsets in doubles."""
        n = len(boxes)
        lo = (C.c_int * (3 * n))(*[x for b in boxes for x in b[0]])
        hi = (C.c_int * (3 * n))(*[x for b in boxes for x in b[1]])
        off = (C.c_longlong * n)(*offsets)
        return n, lo, hi, off

    def pack_regions(self, state, box, table, buf, stream=None):
        """Every region of `table` (region_table) of the FAB into its slice of `buf`, one launch."""
        n, lo, hi, off = table
        L.check(self.lib.castro_amd_pack_regions_fab(self.h, C.byref(L.fab_of(state, *box)), n, lo, hi, off,
                                                     C.c_void_p(buf.data_ptr()), _stream_ptr(stream)), "pack_regions_fab")

    def unpack_regions(self, state, box, table, buf, stream=None):
        n, lo, hi, off = table
        L.check(self.lib.castro_amd_unpack_regions_fab(self.h, C.byref(L.fab_of(state, *box)), n, lo, hi, off,
                                                       C.c_void_p(buf.data_ptr()), _stream_ptr(stream)), "unpack_regions_fab")

    # ---- FillBoundary on RCCL behind the C ABI (castro_amd/csrc/halo_rccl.hip) -----------------------------------
    def comm_version(self):
        return self.lib.castro_amd_comm_version().decode()

    def comm_unique_id(self):
        """ncclGetUniqueId: 128 bytes for the host to hand to every rank"""
        buf = C.create_string_buffer(128)
        L.check(self.lib.castro_amd_comm_unique_id(buf), "comm_unique_id")
        return bytes(buf.raw)

    def comm_create(self, nranks, rank, unique_id):
        """ncclCommInitRank (collective over the ranks): an opaque castro_amd_comm handle"""
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        L.check(self.lib.castro_amd_comm_create(C.byref(h), int(nranks), int(rank), buf, int(self.device.index or 0)), "comm_create")
        return h

    def comm_destroy(self, comm):
        if comm:
            self.lib.castro_amd_comm_destroy(comm)

    def halo_plan(self, comm, regions, ncomp):
        """regions: [(peer, sbox (lo, hi), rbox (lo, hi), send_tag, recv_tag)] -> castro_amd_halo_plan handle"""
        n = len(regions)
        arr = (L.HaloRegion * max(n, 1))()
        for r, (peer, sbox, rbox, stag, rtag) in zip(arr, regions):
            r.peer, r.send_tag, r.recv_tag = int(peer), int(stag), int(rtag)
            for d in range(3):
                r.sbox_lo[d], r.sbox_hi[d] = int(sbox[0][d]), int(sbox[1][d])
                r.rbox_lo[d], r.rbox_hi[d] = int(rbox[0][d]), int(rbox[1][d])
        h = C.c_void_p()
        L.check(self.lib.castro_amd_halo_plan_create(C.byref(h), comm, n, arr, int(ncomp)), "halo_plan_create")
        return h

    def halo_plan_destroy(self, plan):
        if plan:
            self.lib.castro_amd_halo_plan_destroy(plan)

    def halo_plan_bytes_sent(self, plan):
        return int(self.lib.castro_amd_halo_plan_bytes_sent(plan))

    def fill_boundary(self, plan, state, box, geom=None, stream=None):
        """pack -> grouped ncclSend / ncclRecv -> unpack -> physical BC fill, enqueued on the stream"""
        L.check(self.lib.castro_amd_fill_boundary(self.h, plan, C.byref(L.fab_of(state, *box)),
                                                  C.byref(geom) if geom is not None else None, _stream_ptr(stream)), "fill_boundary")

    def fill_boundary_ex(self, plan, state, box, geom=None, stream=None):
        """fill_boundary + the plan's "packed" event recorded behind the pack launch (the last read of the valid zones)"""
        L.check(self.lib.castro_amd_fill_boundary_ex(self.h, plan, C.byref(L.fab_of(state, *box)),
                                                     C.byref(geom) if geom is not None else None, 0, _stream_ptr(stream)), "fill_boundary_ex")

    def halo_plan_wait_packed(self, plan, stream=None):
        """`stream` (default: the current one) waits until the last fill_boundary_ex of `plan` has packed"""
        L.check(self.lib.castro_amd_halo_plan_wait_packed(plan, _stream_ptr(stream)), "halo_plan_wait_packed")

    # ---- several boxes per rank: one grouped exchange for all local FABs of a level (castro_amd_halo_group_*) ----------------
    def halo_group(self, comm, nfabs, sends, recvs, ncomp):
        """sends / recvs: [(fab, peer, (lo, hi), tag)] as castro_amd.halo.level_messages derives them -> castro_amd_halo_group handle"""
        def arr(msgs):
            a = (L.HaloMsg * max(len(msgs), 1))()
            for m, (fab, peer, box, tag) in zip(a, msgs):
                m.fab, m.peer, m.tag = int(fab), int(peer), int(tag)
                for d in range(3):
                    m.lo[d], m.hi[d] = int(box[0][d]), int(box[1][d])
            return a
        h = C.c_void_p()
        L.check(self.lib.castro_amd_halo_group_create(C.byref(h), comm, int(nfabs), len(sends), arr(sends), len(recvs), arr(recvs), int(ncomp)),
                "halo_group_create")
        return h

    def halo_group_destroy(self, group):
        if group:
            self.lib.castro_amd_halo_group_destroy(group)

    def halo_group_bytes_sent(self, group):
        return int(self.lib.castro_amd_halo_group_bytes_sent(group))

    def fill_boundary_group(self, group, states, boxes, geom=None, stream=None):
        """states[f], boxes[f]: tensor and index box of local FAB f"""
        fabs = (L.Fab * len(states))(*[L.fab_of(t, *b) for t, b in zip(states, boxes)])
        L.check(self.lib.castro_amd_fill_boundary_group(self.h, group, fabs, C.byref(geom) if geom is not None else None,
                                                        _stream_ptr(stream)), "fill_boundary_group")

    def fill_boundary_group_ex(self, group, states, boxes, geom=None, stream=None):
        """fill_boundary_group + the group's "packed" event recorded behind the last pack launch"""
        fabs = (L.Fab * len(states))(*[L.fab_of(t, *b) for t, b in zip(states, boxes)])
        L.check(self.lib.castro_amd_fill_boundary_group_ex(self.h, group, fabs, C.byref(geom) if geom is not None else None, 0,
                                                           _stream_ptr(stream)), "fill_boundary_group_ex")

    def halo_group_wait_packed(self, group, stream=None):
        L.check(self.lib.castro_amd_halo_group_wait_packed(group, _stream_ptr(stream)), "halo_group_wait_packed")

    def allreduce_min_c(self, comm, t, stream=None):
        L.check(self.lib.castro_amd_allreduce_min(comm, C.c_void_p(t.data_ptr()), int(t.numel()), _stream_ptr(stream)), "allreduce_min")

    def sedov_init(self, state, box, lo, hi, geom, params, r_init=0.01, p_ambient=1.e-5, exp_energy=1.0,
                   dens_ambient=1.0, nsub=10, stream=None):
        L.check(self.lib.castro_amd_sedov_init_fab(self.h, C.byref(L.fab_of(state, *box)), L.i3(lo), L.i3(hi),
                                                   C.byref(geom), C.byref(params), r_init, p_ambient, exp_energy,
                                                   dens_ambient, int(nsub), _stream_ptr(stream)), "sedov_init_fab")

    def sod_init(self, state, box, lo, hi, geom, params, rho_l, u_l, p_l, rho_r, u_r, p_r, idir=1, frac=0.5,
                 stream=None):
        L.check(self.lib.castro_amd_sod_init_fab(self.h, C.byref(L.fab_of(state, *box)), L.i3(lo), L.i3(hi),
                                                 C.byref(geom), C.byref(params), rho_l, u_l, p_l, rho_r, u_r, p_r,
                                                 int(idir), float(frac), _stream_ptr(stream)), "sod_init_fab")

    # ---- profiling ---------------------------------------------------------------------------
    def profile(self, enable=True):
        self.lib.castro_amd_ctx_profile(self.h, 1 if enable else 0)

    def profile_reset(self):
        self.lib.castro_amd_ctx_profile_reset(self.h)

    def profile_report(self):
        """{kernel name: (total_ms, launches)} measured with hipEvents on the launch stream."""
        out = {}
        n = self.lib.castro_amd_ctx_profile_count(self.h)
        for i in range(n):
            name = C.create_string_buffer(64)
            ms = C.c_double()
            cnt = C.c_longlong()
            self.lib.castro_amd_ctx_profile_get(self.h, i, name, 64, C.byref(ms), C.byref(cnt))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out
