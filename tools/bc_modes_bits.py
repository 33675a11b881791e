import os, sys
sys.path.insert(0, ".")
os.environ["CASTRO_AMD_HALO_SELF_SEND"] = "1"
import numpy as np, torch
import castro_amd
from castro_amd import halo
from castro_amd.hydro import HipHydro
from tests.util import physical_state
numerics = "contract"
n, ng = (32, 16, 12), 4
lo_bc, hi_bc = (2, 3, 2), (2, 2, 4)
G = castro_amd.make_geom(n, (0., 0., 0.), (1., 0.5, 0.375), lo_bc, hi_bc)
rng = np.random.default_rng(17)
dom = ((0, 0, 0), tuple(x - 1 for x in n))
gdom = (tuple(x - ng for x in dom[0]), tuple(x + ng for x in dom[1]))
U0 = physical_state(rng, gdom[0], gdom[1], smooth=False, vel=1.0)
U0[7] = U0[0] * rng.uniform(0.9, 1.0, size=U0[0].shape)
dt = 5e-4
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
def outputs(h, bx):
    Sn = h.alloc(8, *bx); fl, ms, fb = [], [], []
    for d in range(3):
        fhi = list(bx[1]); fhi[d] += 1
        fb.append((bx[0], tuple(fhi))); fl.append(h.alloc(8, bx[0], fhi)); ms.append(h.alloc(1, bx[0], fhi))
    return Sn, fl, ms, fb
for small_dens, sbc in ((0.5, 2), (1e-100, 2), (0.5, 0)):
    P = castro_amd.default_params(small_dens=small_dens) if small_dens > 1e-50 else castro_amd.default_params()
    res = {}
    for form in ("bcfill_then_call", "call_fills", "valid_rest"):
        h = HipHydro(0, numerics=numerics)
        Ud = dev(U0); Sn, fl, ms, fb = outputs(h, dom)
        kw = dict(fluxes=fl, flux_boxes=fb, mass_fluxes=ms, update_from_sborder=True, flux_assign=True, sborder_clean=sbc)
        if form == "bcfill_then_call":
            h.bc_fill(Ud, gdom, G); h.construct_ctu_hydro_source(dom, Ud, gdom, Sn, dom, G, P, 0.0, dt, **kw)
        elif form == "call_fills":
            h.construct_ctu_hydro_source(dom, Ud, gdom, Sn, dom, G, P, 0.0, dt, bc_fill=True, **kw)
        else:
            h.construct_ctu_hydro_source(dom, Ud, gdom, Sn, dom, G, P, 0.0, dt, stage="valid", **kw)
            h.construct_ctu_hydro_source(dom, Ud, gdom, Sn, dom, G, P, 0.0, dt, stage="rest", bc_fill=True, **kw)
        torch.cuda.synchronize()
        res[form] = (Sn.cpu().numpy(), Ud.cpu().numpy())
        h.close()
    for form in ("call_fills", "valid_rest"):
        a, b = res["bcfill_then_call"], res[form]
        d = np.abs(a[0] - b[0]); dS = np.abs(a[1] - b[1])
        print("small_dens %g sb_clean %d %-12s: S_new differing %d (max rel %.2e), Sborder differing %d (max %.2e)" % (
            small_dens, sbc, form, int((d > 0).sum()), (d / np.maximum(np.abs(a[0]), 1e-300)).max(), int((dS > 0).sum()), dS.max()))
