"""Diagnostic: the light overlap through the C-ABI halo path with RCCL self-send, stepwise and as a captured graph."""
import faulthandler
import os
import sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CASTRO_AMD_C_HALO"] = os.environ.get("C_HALO", "1")
os.environ["CASTRO_AMD_HALO_SELF_SEND"] = os.environ.get("SELF_SEND", "1")
import torch
import castro_amd

def say(*a):
    print(*a, flush=True)

n = (48, 40, 32)
for numerics in ("exact", "contract"):
    kw = dict(lo_bc=(0, 0, 0), hi_bc=(0, 0, 0), numerics=numerics)
    ref = castro_amd.Castro(n, overlap=False, **kw)
    ref.initData("sedov", r_init=0.1, nsub=4)
    a = castro_amd.Castro(n, overlap=True, **kw)
    a.initData("sedov", r_init=0.1, nsub=4)
    say(numerics, "light:", a._light_overlap(), "host_free:", a.host_free_ok(), "plans:", [("cplan" in p) for p in a._plans.values()])

    def cmp(tag):
        torch.cuda.synchronize()
        d = (a.S_new_b != ref.S_new_b)
        g = 4
        dv = (a.S_new() != ref.S_new())
        say("  %-28s nstep %d/%d dt equal %s; differing: all %d, valid %d, per comp %s" % (
            tag, a.nstep, ref.nstep, a.dt == ref.dt, int(d.sum()), int(dv.sum()), [int(x.sum()) for x in dv]))
    for _ in range(2):
        a.step(); ref.step()
    cmp("stepwise x2")
    a.run_steps(2, graph=False); ref.step(); ref.step()
    cmp("host-free eager x2")
    a.run_steps(4); [ref.step() for _ in range(4)]
    cmp("graph x4 (%s)" % bool(a._graphs))
    a.close(); ref.close()
