import sys, os, torch
sys.path.insert(0, '.')
import castro_amd
n = int(sys.argv[1]); steps = int(sys.argv[2])
c = castro_amd.Castro((n, n, n), flux_assign=True)
c.initData("sedov")
g = 4
v = lambda b: b[:, g:-g, g:-g, g:-g]
rho0 = v(c.S_new_b)[0]
print(n, "init asym", [(rho0 - rho0.flip((d,))).abs().max().item() for d in range(3)], "E asym", [(v(c.S_new_b)[4] - v(c.S_new_b)[4].flip((d,))).abs().max().item() for d in range(3)])
for s in range(steps):
    c.step(0.01)
    torch.cuda.synchronize()
    S = v(c.S_new_b)
    print(n, "step", s, "dt", c.dt, "asym z,y,x", [(S[0] - S[0].flip((d,))).abs().max().item() for d in range(3)], "max drho", (S[0]-1).abs().max().item(), "status", c.hydro.status())
