"""CPU stand-in for castro_amd.hydro.HipHydro, used ONLY by the tests.

It exposes the same method set as HipHydro but works on CPU torch tensors and calls the
oracle, so the driver logic of castro_amd.Castro (decomposition, FillPatch halo exchange,
dt control, tile/shell splitting) can be exercised without a GPU and with the gloo backend.
The product never imports this file.
"""
import ctypes as C

import numpy as np
import torch

from oracle import oracle_lib as O


class OracleBackend:
    name = "oracle"

    def __init__(self, nthreads=1):
        self.nthreads = nthreads
        self.device = torch.device("cpu")

    # geometry/params of the oracle flavour (same field layout for the shared fields)
    make_geom = staticmethod(lambda n, plo, phi, lbc, hbc: O.make_geom(n, plo, phi, lbc, hbc))

    def alloc(self, ncomp, lo, hi, fill=0.0):
        shape = (ncomp, hi[2] - lo[2] + 1, hi[1] - lo[1] + 1, hi[0] - lo[0] + 1)
        return torch.full(shape, fill, dtype=torch.float64)

    @staticmethod
    def _a4(t, box):
        if t is None:
            return O.a4(None, box[0], box[1])
        return O.a4(t.numpy(), box[0], box[1])

    def construct_ctu_hydro_source(self, bx, Sborder, sb_box, S_new, snew_box, geom, params, time, dt,
                                   fluxes=None, flux_boxes=None, mass_fluxes=None, qe=None, vbx=None,
                                   update_from_sborder=False, src=None, src_box=None, stream=None,
                                   clean_ntimes=0, red=None, flux_assign=False, stage=None):
        if stage == "A":
            return              # the oracle has no staged form: everything happens in stage B
        L = O.lib()
        vlo, vhi = vbx if vbx is not None else bx
        fa, ma, qa = (O.A4 * 3)(), (O.A4 * 3)(), (O.A4 * 3)()
        for d in range(3):
            fbox = flux_boxes[d] if flux_boxes is not None else None
            fa[d] = self._a4(fluxes[d] if fluxes is not None else None, fbox or bx)
            ma[d] = self._a4(mass_fluxes[d] if mass_fluxes is not None else None, fbox or bx)
            qa[d] = self._a4(qe[d] if qe is not None else None, fbox or bx)
        sb, sn = self._a4(Sborder, sb_box), self._a4(S_new, snew_box)
        if flux_assign:           # the oracle accumulates: zero this tile's faces first
            vh = list(vhi)
            for d in range(3):
                hi = list(bx[1])
                hi[d] += 1 if bx[1][d] == vh[d] else 0
                for arr in ((fluxes[d] if fluxes is not None else None), (mass_fluxes[d] if mass_fluxes is not None else None)):
                    if arr is not None:
                        arr[self._slices(flux_boxes[d] if flux_boxes is not None else bx, bx[0], hi)] = 0.0
        if update_from_sborder:
            L.ora_fill_interior_copy(sn, sb, O.i3(bx[0]), O.i3(bx[1]))
        sa = self._a4(src, src_box) if src is not None else O.a4(None, bx[0], bx[1])
        st = L.ora_ctu_hydro_tile(O.i3(bx[0]), O.i3(bx[1]), O.i3(vlo), O.i3(vhi), sb, sa, sn,
                                  fa, ma, qa, C.byref(geom), C.byref(params), float(dt))
        assert st == 0
        if clean_ntimes > 0:      # castro_amd_ctu_hydro_clean_fab == the update followed by the separate pass
            self.clean_state_reduce(S_new, snew_box, bx[0], bx[1], geom, params, red, ntimes=clean_ntimes)

    def old_gravity_source(self, state, box, source, src_box, lo, hi, grav, grav_source_type, dt, stream=None):
        g = (C.c_double * 3)(*[float(x) for x in grav])
        O.lib().ora_old_gravity_source(O.i3(lo), O.i3(hi), self._a4(state, box), self._a4(source, src_box), C.byref(g),
                                       int(grav_source_type), float(dt))

    def new_gravity_source(self, state_old, old_box, state_new, new_box, source, src_box, mass_fluxes, flux_boxes, lo, hi,
                           grav, grav_source_type, dt, geom, stream=None):
        g = (C.c_double * 3)(*[float(x) for x in grav])
        dx = (C.c_double * 3)(*[geom.dx[d] for d in range(3)])
        mf = (O.A4 * 3)()
        for d in range(3):
            mf[d] = self._a4(mass_fluxes[d], flux_boxes[d])
        O.lib().ora_new_gravity_source(O.i3(lo), O.i3(hi), self._a4(state_old, old_box), self._a4(state_new, new_box),
                                       self._a4(source, src_box), mf, C.byref(g), int(grav_source_type), float(dt), C.byref(dx))

    @staticmethod
    def _rot(rot):
        R = O.Rotation()
        for f, _ in O.Rotation._fields_:
            setattr(R, f, getattr(rot, f))
        return R

    def old_rotation_source(self, state, box, source, src_box, lo, hi, rot, geom, dt, stream=None):
        O.lib().ora_old_rotation_source(O.i3(lo), O.i3(hi), self._a4(state, box), self._a4(source, src_box),
                                        C.byref(self._rot(rot)), C.byref(geom), float(dt))

    def new_rotation_source(self, state_old, old_box, state_new, new_box, source, src_box, mass_fluxes, flux_boxes, lo, hi,
                            rot, geom, dt, stream=None):
        mf = (O.A4 * 3)()
        for d in range(3):
            mf[d] = self._a4(mass_fluxes[d], flux_boxes[d])
        O.lib().ora_new_rotation_source(O.i3(lo), O.i3(hi), self._a4(state_old, old_box), self._a4(state_new, new_box),
                                        self._a4(source, src_box), mf, C.byref(self._rot(rot)), C.byref(geom), float(dt))

    def saxpy(self, dst, dst_box, a, src, src_box, ncomp, lo, hi, stream=None):
        O.lib().ora_saxpy(O.i3(lo), O.i3(hi), self._a4(dst, dst_box), float(a), self._a4(src, src_box), int(ncomp))

    def error_tag(self, field, field_box, comp, tags, tags_box, lo, hi, kind, value, stream=None):
        O.lib().ora_error_tag(O.i3(lo), O.i3(hi), self._a4(field, field_box), int(comp), self._a4(tags, tags_box), int(kind),
                              float(value))

    def cc_interp(self, crse, crse_box, fine, fine_box, lo, hi, ncomp, stream=None):
        O.lib().ora_cc_interp(O.i3(lo), O.i3(hi), self._a4(crse, crse_box), self._a4(fine, fine_box), int(ncomp))

    def lincomb(self, dst, dst_box, a, x, x_box, b, y, y_box, ncomp, lo, hi, stream=None):
        O.lib().ora_lincomb(O.i3(lo), O.i3(hi), self._a4(dst, dst_box), float(a), self._a4(x, x_box), float(b),
                            self._a4(y, y_box), int(ncomp))

    def avgdown(self, fine, fine_box, crse, crse_box, lo, hi, ncomp, stream=None):
        O.lib().ora_avgdown(O.i3(lo), O.i3(hi), self._a4(fine, fine_box), self._a4(crse, crse_box), int(ncomp))

    def fluxreg_crse_init(self, reg, reg_box, cflux, cflux_box, lo, hi, ncomp, mult, stream=None):
        O.lib().ora_reg_crse_init(O.i3(lo), O.i3(hi), self._a4(reg, reg_box), self._a4(cflux, cflux_box), int(ncomp), float(mult))

    def fluxreg_fine_add(self, reg, reg_box, fflux, fflux_box, lo, hi, dir, ncomp, mult, stream=None):
        O.lib().ora_reg_fine_add(O.i3(lo), O.i3(hi), self._a4(reg, reg_box), self._a4(fflux, fflux_box), int(dir), int(ncomp),
                                 float(mult))

    def reflux(self, state, state_box, reg, reg_box, lo, hi, dir, side, ncomp, vol, stream=None):
        O.lib().ora_reflux(O.i3(lo), O.i3(hi), self._a4(state, state_box), self._a4(reg, reg_box), int(dir), int(side),
                           int(ncomp), float(vol))

    def derive(self, name, state, box, der, der_box, dcomp, lo, hi, geom, params, center, stream=None):
        from castro_amd._lib import DERIVE_IDS
        ctr = (C.c_double * 3)(*[float(x) for x in center])
        d = der[dcomp:dcomp + 1]
        rc = O.lib().ora_derive(DERIVE_IDS[name], O.i3(lo), O.i3(hi), self._a4(state, box), self._a4(d, der_box),
                                C.byref(geom), C.byref(params), C.byref(ctr))
        assert rc == 0

    def clean_state(self, state, box, lo, hi, params, ntimes=1, stream=None):
        for _ in range(ntimes):
            O.lib().ora_clean_state(O.i3(lo), O.i3(hi), self._a4(state, box), C.byref(params))

    def set_source_corrector(self, corr, box):
        if corr is None:
            O.lib().ora_set_source_corrector(None)
        else:
            self._corr = self._a4(corr, box)           # kept alive: the library stores the descriptor
            O.lib().ora_set_source_corrector(C.byref(self._corr))

    def clean_state_reduce(self, state, box, lo, hi, geom, params, out, ntimes=1, stream=None):
        a = self._a4(state, box)
        r = O.lib().ora_min_density(O.i3(lo), O.i3(hi), a)
        # [2]: the estimate after the first clean_state (the validity check of do_advance_ctu), [0]: after the last
        self.clean_state(state, box, lo, hi, params, ntimes=1)
        e1 = e = O.lib().ora_estdt_cfl_guarded(O.i3(lo), O.i3(hi), a, C.byref(geom), C.byref(params))
        if ntimes > 1:
            self.clean_state(state, box, lo, hi, params, ntimes=ntimes - 1)
            e = O.lib().ora_estdt_cfl_guarded(O.i3(lo), O.i3(hi), a, C.byref(geom), C.byref(params))
        out[0] = min(out[0].item(), e)
        out[1] = min(out[1].item(), r)
        if out.numel() > 2:
            out[2] = min(out[2].item(), e1)

    def estdt_cfl(self, state, box, lo, hi, geom, params, out, stream=None):
        a = self._a4(state, box)
        e = O.lib().ora_estdt_cfl(O.i3(lo), O.i3(hi), a, C.byref(geom), C.byref(params))
        r = O.lib().ora_min_density(O.i3(lo), O.i3(hi), a)
        out[0] = min(out[0].item(), e)
        out[1] = min(out[1].item(), r)

    def bc_fill(self, state, box, geom, stream=None):
        O.lib().ora_bc_fill(self._a4(state, box), C.byref(geom))

    @staticmethod
    def _slices(box, lo, hi):
        return (slice(None),) + tuple(slice(lo[2 - a] - box[0][2 - a], hi[2 - a] - box[0][2 - a] + 1) for a in range(3))

    def pack(self, state, box, lo, hi, buf, stream=None):
        buf.copy_(state[self._slices(box, lo, hi)].reshape(-1))

    def unpack(self, state, box, lo, hi, buf, stream=None):
        sl = self._slices(box, lo, hi)
        state[sl] = buf.reshape(state[sl].shape)

    # the one-launch forms of the device path (castro_amd_pack_regions_fab): same table / offset bookkeeping, so the gloo
    # tests on CPU go through the same exchange plan as a multi-GPU run
    @staticmethod
    def region_table(boxes, offsets):
        return [(tuple(b[0]), tuple(b[1]), int(o)) for b, o in zip(boxes, offsets)]

    def pack_regions(self, state, box, table, buf, stream=None):
        for lo, hi, off in table:
            sl = state[self._slices(box, lo, hi)]
            buf[off:off + sl.numel()] = sl.reshape(-1)

    def unpack_regions(self, state, box, table, buf, stream=None):
        for lo, hi, off in table:
            sl = self._slices(box, lo, hi)
            n = state[sl].numel()
            state[sl] = buf[off:off + n].reshape(state[sl].shape)

    def copy(self, dst, dst_box, src, src_box, lo, hi, stream=None):
        dst[self._slices(dst_box, lo, hi)] = src[self._slices(src_box, lo, hi)]

    def sedov_init(self, state, box, lo, hi, geom, params, r_init=0.01, p_ambient=1.e-5, exp_energy=1.0,
                   dens_ambient=1.0, nsub=10, stream=None):
        O.lib().ora_sedov_init(O.i3(lo), O.i3(hi), self._a4(state, box), C.byref(geom), C.byref(params),
                               r_init, p_ambient, exp_energy, dens_ambient, nsub)

    def sod_init(self, state, box, lo, hi, geom, params, rho_l, u_l, p_l, rho_r, u_r, p_r, idir=1, frac=0.5,
                 stream=None):
        O.lib().ora_sod_init(O.i3(lo), O.i3(hi), self._a4(state, box), C.byref(geom), C.byref(params),
                             rho_l, u_l, p_l, rho_r, u_r, p_r, idir, frac)

    def profile(self, enable=True):
        pass

    def profile_reset(self):
        pass

    def profile_report(self):
        return {}
