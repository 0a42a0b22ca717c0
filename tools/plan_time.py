"""What a job pays between registration and its first fusion launch: the plan from the registered rectangles to a table in
device memory -- host planner + upload against host sweep + expansion on the device (csrc/plan_expand.hip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from image_stitcher_amd import native, placement
T = 2048
have_gpu = torch.cuda.is_available()
for g in (16, 32):
    sh = placement.Shifts((3, -244), (-244, -2))
    wc, hc = placement.canvas_size(g, g, T, T, use_registration=True, shifts=sh)
    t = time.perf_counter()
    rects = placement.grid_rects(g, g, T, T, sh)
    t_rects = time.perf_counter() - t
    for on_device in ((False, True) if have_gpu else (False,)):
        for rep in range(4):
            if have_gpu:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan = native.FusePlan(rects, T, T, hc, wc, native.SQ_FUSE_OVERWRITE, expand_on_device=on_device)
            t1 = time.perf_counter()
            if have_gpu:
                plan.device_table('cuda:0')
                torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f'{g}x{g} grid ({plan.n_items} items, table {plan.table_bytes / 1e6:.1f} MB): rects {t_rects * 1e3:.2f} ms, '
                  f'{"host sweep" if on_device else "host plan "} {(t1 - t0) * 1e3:6.2f} ms, '
                  f'{"expand on device" if on_device else "upload          "} {(t2 - t1) * 1e3:6.2f} ms, together {(t2 - t0) * 1e3:6.2f} ms', flush=True)
            del plan
