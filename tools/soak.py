#!/usr/bin/env python3
"""Create / run / destroy soak: device memory must return to the same level after every cycle.  Stages isolate which
call leaves memory behind (create only, + uploads, + single solve, + batch solve, + source promotion)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import velo_amd
from velo_amd import api, synth

d = synth.scan_pair(n_beams=32, n_azimuth=600)

def cycle(stage):
    ctxs = [api.Context(0, icp_skip=1) for _ in range(4)]
    if stage >= 1:
        for c in ctxs:
            c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
    if stage >= 2:
        ctxs[0].frame_to_frame(d["x0"])
    if stage >= 3:
        for _ in range(3):
            api.frame_to_frame_batch(ctxs, [d["x0"]] * 4)
    if stage >= 4:
        ctxs[0].source_to_target(); ctxs[0].set_source(d["src_xyz"], d["src_off"]); ctxs[0].frame_to_frame(d["x0"])
    for c in ctxs: c.close()

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = 0.0
for stage, name in enumerate(["create/destroy", "+uploads", "+single solve", "+batch solve", "+source promotion"]):
    for _ in range(3): cycle(stage)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    series = []
    for rep in range(cycles):
        cycle(stage)
        series.append((free0 - torch.cuda.mem_get_info(0)[0]) / 1e6)
    worst = max(worst, series[-1])
    print("%-18s drift MB after 1/10/20/%d cycles: %.1f %.1f %.1f %.1f" % (name, cycles, series[0], series[min(9, cycles - 1)], series[min(19, cycles - 1)], series[-1]), flush=True)
print("SOAK", "OK" if worst < 8.0 else "LEAK", "worst drift %.1f MB" % worst)
