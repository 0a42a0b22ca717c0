#!/usr/bin/env python3
"""Bench of the registration-and-fusion hot path on MI355X (contract: see the task brief).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (default ``cfg3`` = BASELINE.json configs[2], the largest single-GPU configuration;
the metric's own 32x32x4cx50z configuration is 1.6 TiB of tiles and does not fit one GPU):
a 16x16 grid of 2048x2048 uint16 tiles, 4 channels x 10 z, ``-r -ff`` (registration +
float32 flatfield), synthetic tiles generated on the device and resident in HBM before the
timed region.  With N GPUs every rank stitches one such region (weak scaling: regions/wells
are independent, like planes); the only collective is the all-gather of the shift table.

One step = one pass of the hot path over the resident region(s): per-tile min/max + centre-pair
phase cross-correlation on the registration plane (device), all-gather of the shift rows
(N > 1), host integer geometry + span plan, then ONE fusion launch over all 40 planes.

Printed JSON (rank 0, one line): whole-job Mvoxel/s, plus
  roofline     fusion kernel, algorithmic bytes / HIP-event launch time vs 8 TB/s HBM peak
  cpu_baseline the numpy oracle (oracle/stitch_oracle.py) timed on this box's host on one
               (c, z) plane of the same workload (N = 1, rank 0 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: grid, channels, z, flatfield, description
    'cfg2': dict(grid=8, channels=1, nz=1, flat=False,
                 desc='8x8 grid of 2048x2048 uint16 tiles, 1 ch, 1 z, -r (BASELINE configs[1])'),
    'cfg3': dict(grid=16, channels=4, nz=10, flat=True,
                 desc='16x16 grid of 2048x2048 uint16 tiles, 4 ch, 10 z, -r -ff (BASELINE configs[2])'),
    'cfg4shard': dict(grid=32, channels=4, nz=2, flat=True,
                      desc='32x32 grid of 2048x2048 uint16 tiles, 4 ch, 2 of 50 z resident per GPU, -r -ff '
                           '(BASELINE configs[3] shard)'),
}
TILE, OVERLAP, DRIFT = 2048, 244, (3, -2)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='cfg3')
    ap.add_argument('--planes', type=int, default=0, help='override the number of resident (c,z) planes')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--traffic-bytes', type=float, default=None,
                    help='HBM bytes per fusion launch from a separate rocprofv3 --pmc pass (else null)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from image_stitcher_amd import native, placement, registration, sharding, synth

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    backend = os.environ.get('SQ_DIST_BACKEND', 'nccl')   # 'gloo' only to rehearse N > 1 on a 1-GPU box
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    coll_dev = dev if backend == 'nccl' else None

    wl = WORKLOADS[args.workload]
    g, C, Z = wl['grid'], wl['channels'], wl['nz']
    n_planes = C * Z
    spec = synth.GridSpec(rows=g, cols=g, tile_h=TILE, tile_w=TILE, ov_y=OVERLAP, ov_x=OVERLAP,
                          jy=DRIFT[0], jx=DRIFT[1], channels=synth.DEFAULT_CHANNELS[:C], nz=Z,
                          seed=1000 * 3 + rank * 100)
    truth = placement.Shifts((DRIFT[0], -OVERLAP), (-OVERLAP, DRIFT[1]))
    wc, hc = placement.canvas_size(g, g, TILE, TILE, use_registration=True, shifts=truth)
    plane_in = g * g * TILE * TILE * 2
    plane_out = hc * wc * 2
    free, total = torch.cuda.mem_get_info(dev)
    fit = int((free - (6 << 30)) // (plane_in + plane_out))
    if args.planes:
        n_planes = args.planes
    if fit < n_planes:
        if rank == 0:
            print(f"[bench] only {fit} of {n_planes} planes fit in {free / 2**30:.0f} GiB free HBM", file=sys.stderr)
        n_planes = max(1, fit)
    Z_eff = max(1, n_planes // C) if n_planes >= C else 1
    C_eff = min(C, n_planes)
    n_planes = C_eff * Z_eff

    # ---- resident inputs: tiles generated on the device (not timed) -------------------------
    tiles = torch.empty((n_planes, g * g, TILE, TILE), dtype=torch.uint16, device=dev)
    for p in range(n_planes):
        c, z = divmod(p, Z_eff)
        desc = np.zeros(g * g, dtype=native.SYNTH_DTYPE)
        for r in range(g):
            for col in range(g):
                oy, ox = spec.origin(r, col)
                desc[r * g + col] = (spec.scene_seed(0, 0, z, c) % 2 ** 64,
                                     spec.noise_seed(0, 0, z, c, spec.fov_index(r, col)) % 2 ** 64, oy, ox)
        native.synth_tiles(desc, TILE, TILE, spec.noise, 'uint16', dev, out=tiles[p])
    flat_list = None
    if wl['flat']:
        ffs = [torch.from_numpy(synth.synthetic_flatfield(TILE, TILE, np.float32) * np.float32(1 + 0.03125 * c)).to(dev)
               for c in range(C_eff)]
        flat_list = [ffs[p // Z_eff] for p in range(n_planes)]
    flat_ptrs = native.pointer_table(flat_list, dev) if flat_list else None
    canvas = torch.empty((n_planes, hc, wc), dtype=torch.uint16, device=dev)
    xs = [spec.stage_mm(0, c)[0] for c in range(g)]
    ys = [spec.stage_mm(r, 0)[1] for r in range(g)]
    # write order inside a plane = sorted file names of the fov numbers (stitcher.py:168)
    order = placement.filename_order([spec.fov_index(r, c) for r in range(g) for c in range(g)])
    order_rc = [divmod(i, g) for i in order]
    tile_order = torch.tensor(order, dtype=torch.int64)
    # tile pointer table in write order (plane-major), so rect i <-> pointer i
    esz = TILE * TILE * 2
    base = tiles.data_ptr()
    ptrs = (base + (torch.arange(n_planes, dtype=torch.int64)[:, None] * (g * g) + tile_order[None, :]) * esz)
    ptrs = ptrs.reshape(-1).to(dev)
    reg_plane = tiles[0]   # registration channel = first channel, z level 0 (CLI defaults)
    torch.cuda.synchronize()

    fuse_events = []
    state = {}
    gathers = []

    lap = {}

    def tick(name, t_prev):
        t = time.perf_counter()
        lap[name] = lap.get(name, 0.0) + (t - t_prev)
        return t

    serial = bool(os.environ.get('SQ_BENCH_SERIAL'))

    def start_registration():
        # registration: centre pairs on the registration plane (stitcher.py:422-498), enqueued only
        return registration.register_grid_center_async(reg_plane, g, g, xs, ys, spec.pixel_size_um,
                                                       spec.pixel_binning, normalization='phase')

    def step(record, pending, more):
        """One region: collect its registration, all-gather, geometry + span plan, one fusion launch.  Regions
        are independent, so the NEXT region's registration is enqueued ahead of this region's fusion launch:
        its kernels run first, and its read-back, planning and launch happen while this fusion runs (the
        device never waits for the host's millisecond of planning).  SQ_BENCH_SERIAL=1 keeps the steps apart."""
        if os.environ.get('SQ_BENCH_BREAKDOWN') == '2':
            torch.cuda.synchronize()      # diagnostic only: lets the host timings below exclude GPU back-pressure
        t = time.perf_counter()
        shifts = (pending or start_registration()).result()
        t = tick('register', t)
        # the shift table is all-gathered (RCCL) beside the fusion launch: a rank fuses with its own
        # region's shifts, the other rows are only needed when the table is written out
        row = sharding.shifts_to_row(shifts)
        gathers.append((row, sharding.all_gather_shift_table_async(row[None], device=coll_dev)))
        mine = sharding.row_to_shifts(row)
        t = tick('allgather', t)
        # host integer geometry + span plan (rebuilt every step: it depends on the shifts)
        rects = placement.grid_rects(g, g, TILE, TILE, mine, order=order_rc)
        w_px, h_px = placement.canvas_size(g, g, TILE, TILE, use_registration=True, shifts=mine)
        if (w_px, h_px) != (wc, hc):
            raise RuntimeError(f"registration returned {mine}, canvas {h_px}x{w_px} != planned {hc}x{wc}")
        t = tick('rects', t)
        plan = native.FusePlan(rects, TILE, TILE, hc, wc, native.SQ_FUSE_OVERWRITE)
        t = tick('plan', t)
        nxt = start_registration() if (more and not serial) else None
        t = tick('register_next', t)
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        native.fuse_planes(plan, None, canvas, flat_list, tile_ptrs=ptrs, flat_ptrs=flat_ptrs)
        if record:
            e1.record()
            fuse_events.append((e0, e1))
        state['plan'], state['shifts'] = plan, mine
        tick('fuse_launch', t)
        return nxt

    def run_steps(n, record):
        pending = None
        for k in range(n):
            pending = step(record, pending, k + 1 < n)

    def finish_gathers():
        # inside the timed region: every step's table has arrived and holds this rank's row
        for row, pending in gathers:
            table = pending.result()
            assert table.shape == (world, sharding.SHIFT_ROW) and (table[rank] == row).all(), "shift table all-gather"
        gathers.clear()

    def fence():
        finish_gathers()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run_steps(args.warmup, False)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps, True)      # K registrations, K plans, K fusion launches, all inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev or 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    plan = state['plan']
    assert tuple(state['shifts'].h_shift) == truth.h_shift and tuple(state['shifts'].v_shift) == truth.v_shift, \
        f"registration did not recover the planted drift: {state['shifts']}"
    voxels_per_step = world * n_planes * hc * wc
    value = voxels_per_step * args.steps / elapsed / 1e6
    fuse_ms = float(np.mean([a.elapsed_time(b) for a, b in fuse_events]))
    covered = plan.covered_voxels
    # 4 B per covered voxel (2 B read + 2 B write), 2 B per uncovered voxel (zero write),
    # + the float32 flatfield once per plane
    alg_bytes = n_planes * (covered * 4 + (hc * wc - covered) * 2 + (TILE * TILE * 4 if wl['flat'] else 0))
    achieved = alg_bytes / (fuse_ms * 1e-3) / 1e9
    traffic = args.traffic_bytes
    if traffic is None:
        # PMC counters need their own rocprofv3 passes; the last committed measurement of this exact
        # launch (same workload, same plane count) is reported, else null
        try:
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic_latest.json')) as fh:
                pm = json.load(fh)
            if pm.get('workload') == args.workload and pm.get('planes') == n_planes:
                traffic = pm['traffic_bytes_per_launch']
        except (OSError, ValueError, KeyError):
            traffic = None

    out = {
        'metric': 'stitched Mvoxels/s', 'value': round(value, 1), 'unit': 'Mvoxel/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'u16', 'data': 'synthetic',
        'config': {'workload': wl['desc'], 'planes_resident_per_gpu': n_planes,
                   'canvas': [hc, wc], 'tiles_per_plane': g * g,
                   'parallelism': f'one region per GPU x{world}, shift-table all-gather' if world > 1 else 'single GPU',
                   'shifts': {'h': list(state['shifts'].h_shift), 'v': list(state['shifts'].v_shift)},
                   'step': 'minmax + centre-pair PCC + span plan + one fusion launch over all planes; the next '
                           "region's registration is enqueued ahead of the fusion launch (regions are independent)"},
        'roofline': {'bound': 'hbm', 'kernel': 'fuse_overwrite_kernel<u16,f32 flat>' if wl['flat'] else 'fuse_overwrite_kernel<u16>',
                     'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                     'algorithmic_bytes_per_launch': int(alg_bytes), 'launch_ms': round(fuse_ms, 4)},
    }

    if rank == 0 and os.environ.get('SQ_BENCH_BREAKDOWN'):
        n = args.steps + args.warmup
        print('[bench] host ms per step: ' + ', '.join(f'{k} {v / n * 1e3:.2f}' for k, v in lap.items()), file=sys.stderr)
    # accuracy side of the metric (BASELINE.json: shift RMSE <= 0.5 px, fused max-rel-err <= 1e-5)
    sh = state['shifts']
    err = np.array([sh.h_shift[0] - truth.h_shift[0], sh.h_shift[1] - truth.h_shift[1],
                    sh.v_shift[0] - truth.v_shift[0], sh.v_shift[1] - truth.v_shift[1]], dtype=np.float64)
    out['parity'] = {'shift_rmse_px': float(np.sqrt((err ** 2).mean())), 'shift_reference': 'planted drift'}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'], check = cpu_baseline(tiles, flat_list, order, order_rc, spec, xs, ys, g, hc, wc, truth, canvas)
        out['parity'].update(check)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(tiles, flat_list, order, order_rc, spec, xs, ys, g, hc, wc, truth, canvas, max_planes=16):
    """The oracle (a numpy port of the reference path) on a bounded sample of the same workload:
    registration of the two centre pairs once (stitcher.py:1244-1246), then overwrite fusion of
    up to ``max_planes`` (c, z) planes, ~10-30 s of single-core work."""
    from image_stitcher_amd import placement
    from oracle import stitch_oracle as O
    n = min(max_planes, tiles.shape[0])
    host = tiles[:n].cpu().numpy()
    flats = [flat_list[p].cpu().numpy() if flat_list else None for p in range(n)]
    t0 = time.perf_counter()
    mx, my = O.max_overlaps(xs, ys, TILE, TILE, spec.pixel_size_um, spec.pixel_binning)
    ci = ri = (g - 1) // 2
    h = O.calculate_horizontal_shift(host[0, ri * g + ci], host[0, ri * g + ci + 1], mx, np.uint16, 'phase')
    v = O.calculate_vertical_shift(host[0, ri * g + ci], host[0, (ri + 1) * g + ci], my, np.uint16, 'phase')
    assert (tuple(h), tuple(v)) == (truth.h_shift, truth.v_shift)
    rects = placement.grid_rects(g, g, TILE, TILE, placement.Shifts(tuple(h), tuple(v)), order=order_rc)
    voxels = 0
    dt_check = 0.0
    mismatched = 0
    for p in range(n):
        plane = O.fuse_plane_overwrite([host[p, i] for i in order], rects, hc, wc, flats[p])
        voxels += plane.size
        tc = time.perf_counter()   # the comparison with the GPU canvas is not part of the CPU timing
        mismatched += int(np.count_nonzero(canvas[p].cpu().numpy() != plane))
        dt_check += time.perf_counter() - tc
        del plane
    dt = time.perf_counter() - t0 - dt_check
    base = {'value': round(voxels / dt / 1e6, 1), 'unit': 'Mvoxel/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n} (c,z) planes of the workload ({g}x{g} tiles -> {hc}x{wc} canvas each): registration of '
                      f'the 2 centre pairs once + fusion, numpy oracle, {dt:.1f} s, 1 of {os.cpu_count()} host cores used'}
    check = {'fused_max_rel_err': 0.0 if mismatched == 0 else None, 'fused_mismatched_voxels': mismatched,
             'fused_checked': f'{n} of the timed launch\'s canvas planes compared voxel by voxel with the oracle'}
    return base, check


if __name__ == '__main__':
    main()
