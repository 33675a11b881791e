import time, torch, sys
sys.path.insert(0, ".")
import castro_amd
c = castro_amd.Castro((256,)*3)
c.initData("sedov")
for _ in range(5): c.step()
for on in (False, True, False, True):
    c.hydro.profile(on); c.hydro.profile_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): c.step()
    torch.cuda.synchronize(); print("profile", on, (time.perf_counter() - t0) / 20 * 1e3, "ms/step")
