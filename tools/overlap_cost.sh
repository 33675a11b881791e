#!/bin/bash
# NUMERICS (environment, default contract): the build of the kernel library, pinned on every bench.py line
# Per-rank cost of the halo exchange and of the staged overlap, sized on ONE GPU: the box one rank of the 2-, 4- and 8-rank
# strong-scaling run of the 256^3 problem owns (256x256x128, 256x128x128, 128^3) and the 256^3 box of the weak run, periodic in
# every direction, all 26 regions through ncclSend / ncclRecv to the own rank (bench.py --proxy-rank-of), without and with the
# overlap: --force-overlap = the light split of round 6 (ctoprim on the valid zones beside the exchange; host-free and graph-replayed
# like the plain form: `graph True` on both legs), --overlap-staged = the round-2 form.  Sets castro_amd/castro.py: OVERLAP_MIN_ZONES.
N=${NUMERICS:-contract}
for mode in "--no-overlap" "--force-overlap" "--overlap-staged"; do
  python bench.py --numerics $N --steps 20 --warmup 3 --no-cpu-baseline --no-contract-leg --no-extras --proxy-rank-of 1 2 4 8 $mode 2> gpurun_out/ov$mode.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{') and 'metric' in l][-1]); p=d['config']['rank_proxies']
for k in ('1','2','4','8'):
    e=p[k]
    print('[$N] $mode N=%s box %s: %.3f ms/step (graph %s, overlap %s, fillboundary alone %.3f ms, %.1f MB exchanged)' % (k, e['box'], e['ms_per_step'], e.get('step_graph'), e.get('overlap_halo'), e.get('fillboundary_ms', 0), e.get('bytes_exchanged_per_step', 0)/1e6) if 'error' not in e else (k, e))
" || tail -30 gpurun_out/ov$mode.err
done
