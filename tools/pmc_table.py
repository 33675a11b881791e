#!/usr/bin/env python3
"""Derived per-kernel table from a tools_pmc.sh summary (gpurun_out/<tag>_pmc_summary.txt)."""
import re
import sys
txt = open(sys.argv[1]).read()
blocks = re.split(r'\n(?=\S)', txt)
print("%-30s %7s %6s %6s %6s %6s %8s %8s %8s %7s" % ("kernel", "ms", "VALU%", "wait%", "stall%", "act%", "inst/wv", "RD_GB", "WR_GB", "L2hit%"))
for b in blocks:
    lines = b.strip().split('\n')
    name = lines[0].strip()
    d = {}
    for l in lines[1:]:
        p = l.split()
        d[p[0]] = float(p[1])
    if 'GRBM_GUI_ACTIVE' not in d or 'SQ_WAVE_CYCLES' not in d:
        continue
    cyc = d['GRBM_GUI_ACTIVE'] / 8.0
    ms = cyc / 2.1e9 * 1e3      # assumes ~2.1 GHz under load; use the stats pass for wall time
    valu = d['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc) * 100
    wc = d['SQ_WAVE_CYCLES']
    print("%-30s %7.2f %6.1f %6.1f %6.1f %6.1f %8.0f %8.2f %8.2f %7.1f" % (
        name[:30], ms, valu, d['SQ_WAIT_ANY'] / wc * 100, d['SQ_WAIT_INST_ANY'] / wc * 100,
        d['SQ_ACTIVE_INST_ANY'] / wc * 100, d['SQ_INSTS_VALU'] / d['SQ_WAVES'],
        d['TCC_EA0_RDREQ_sum'] * 128 / 1e9, d['TCC_EA0_WRREQ_sum'] * 64 / 1e9,
        d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum']) * 100))
