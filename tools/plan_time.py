import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
from image_stitcher_amd import native, placement
T=2048
for g in (16,32):
    sh=placement.Shifts((3,-244),(-244,-2))
    wc,hc=placement.canvas_size(g,g,T,T,use_registration=True,shifts=sh)
    t=time.perf_counter()
    rects=placement.grid_rects(g,g,T,T,sh)
    t1=time.perf_counter()
    for _ in range(3):
        t2=time.perf_counter()
        plan=native.FusePlan(rects,T,T,hc,wc,native.SQ_FUSE_OVERWRITE)
        t3=time.perf_counter()
        print(g,'rects %.2f ms plan %.2f ms'%((t1-t)*1e3,(t3-t2)*1e3), plan.n_items if hasattr(plan,'n_items') else '')
        del plan
