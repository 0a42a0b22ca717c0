#!/usr/bin/env python3
"""Bench of the registration-and-fusion hot path on MI355X (contract: see the task brief).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

N = 1 (default workload ``cfg3`` = BASELINE.json configs[2], the largest configuration that is resident on one
GPU): a 16x16 grid of 2048x2048 uint16 tiles, 4 channels x 10 z, ``-r -ff`` (registration + float32 flatfield),
synthetic tiles generated on the device and resident in HBM before the timed region.  One step = per-tile
min/max + centre-pair phase cross-correlation on the registration plane (device), host integer geometry + span
plan, then ONE fusion launch over all 40 planes.

N > 1 (default workload ``cfg4`` = BASELINE.json configs[3], the configuration the metric is quoted on): the
32x32 grid of 2048x2048 tiles, 4 channels x 50 z = 200 (c, z) planes = 1.6 TiB of tiles, ``-r -ff``.  STRONG
scaling: every rank takes ONE contiguous run of the 200 planes (sharding.contiguous_blocks: a channel's z planes
stay together and go through the kernel in full groups), no image data is
exchanged.  Registration is the north star's: ALL 1 984 adjacent pairs of the registration plane, dealt over the ranks in
contiguous runs of the tile-row-ordered pair list (a rank holds only the tiles its pairs touch), the [n_pairs, 3]
float64 table {dy, dx, err} all-gathered over RCCL once per step, per-axis median on every rank (--centre-pairs: the
reference's two centre pairs on rank 0 and an 8-int32 row instead).  A rank walks its planes in HBM-resident batches
(tiles + canvases of a batch fill the card); a batch's tiles are synthesised on the device BEFORE that batch's launch,
outside its event pair (inputs resident when timing starts, as the contract says; the generator is not part of the
hot path).  One step = the whole 200-plane job; the timed seconds of a rank are registration + all-gather + plan
(built inside the first timed job -- host sweep into spans, work list expanded on the device -- and re-used while the
shifts stay) + the fusion launches of all its batches, and the job time is the MAX over ranks.  ``--workload cfg4``
runs the same job on one GPU.

Printed JSON (rank 0, one line): whole-job Mvoxel/s, plus
  roofline     fusion kernel, algorithmic bytes / HIP-event launch time vs 8 TB/s HBM peak (rank 0's launches);
               traffic = HBM bytes per launch from the PMC counters: at N = 1 measured by this run itself (two child passes
               under rocprofv3 --pmc after the timed region, live_traffic), else the last committed measurement, labelled
  cpu_baseline the numpy oracle (oracle/stitch_oracle.py) timed on this box's host on a bounded sample of
               the same workload (N = 1, rank 0 only), and the genuine reference's timing from the authoring
               container for the record
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
    'cfg4': dict(grid=32, channels=4, nz=50, flat=True, job=True,
                 desc='32x32 grid of 2048x2048 uint16 tiles, 4 ch, 50 z, -r -ff, 200 planes dealt over the GPUs '
                      '(BASELINE configs[3], the configuration the metric is quoted on)'),
}
TILE, OVERLAP, DRIFT = 2048, 244, (3, -2)
HBM_PEAK_GBS = 8000.0
GROUP = 5            # planes the fusion kernel carries through an item together (ZB in csrc/fuse.hip)
# the genuine reference (unmodified /root/reference/stitcher.py, scikit-image 0.18.3, dask's threaded scheduler)
# cannot travel to the GPU box; its timing is from the authoring container (tools/time_reference.py)
REFERENCE_TIMING = {'value': 18.6, 'unit': 'Mvoxel/s', 'cores': 8, 'kind': 'reference',
                    'sample': 'one (c,z) plane of config 3 (16x16 x 2048^2, -r -ff): stitch_region(...).compute() 57.2 s',
                    'where': 'authoring container, 8 host cores (not this box); tools/time_reference.py'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default=None,
                    help='default: cfg3 on one GPU, cfg4 (the headline job, strong scaling) on several')
    ap.add_argument('--planes', type=int, default=0, help='override the number of (c,z) planes (resident planes / job size)')
    ap.add_argument('--batch', type=int, default=0, help='cfg4: planes per resident batch (default: what fits, at most 10, a multiple of 5)')
    ap.add_argument('--centre-pairs', action='store_true',
                    help="cfg4: register like the reference (the centre tile's two pairs, on rank 0; stitcher.py:455-485) instead of "
                         'ALL 1 984 adjacent pairs sharded over the ranks (the default: the north star\'s registration)')
    ap.add_argument('--canvas-order', choices=['spread', 'plane'], default='spread',
                    help="region workloads: where a (c, z) plane's canvas sits in the canvas allocation.  'spread' (default): "
                         "z-major, so that the z planes of a channel -- which go through the kernel together -- lie a channel "
                         "count of planes apart, spread over the whole allocation; 'plane': plane p = c * Z + z at slot p")
    ap.add_argument('--layout', choices=['mixed', 'arena', 'separate'], default='mixed',
                    help="'mixed' (default): the canvas in a native.DeviceArena -- physical slices classified by a probe and mapped "
                         "round-robin over the card's three memory classes (csrc/arena.hip), tiles in a plain allocation: what "
                         "Stitcher.stitch_region does; 'arena': canvas slots and tile stacks interleaved in ONE plain allocation "
                         "(round 3); 'separate': one plain allocation each (round 2)")
    ap.add_argument('--registration-stream', choices=['side', 'main'], default='main',
                    help="region workloads: the next region's centre-pair registration on the launch stream ahead of the fusion "
                         "launch (default), or on a stream of its own beside the launch in flight (measured: the 0.3 ms it saves "
                         "per step come back as a slower fusion launch, profiles/r03_exp_registration_stream.log)")
    ap.add_argument('--host-plan', action='store_true',
                    help="headline job: build the whole fusion plan on the host and upload it (round 2) instead of expanding the "
                         "spans into the work list on the device")
    ap.add_argument('--fusion-mode', choices=['overwrite', 'feather'], default='overwrite',
                    help="region workloads: 'feather' adds the distance-weighted blend (the north star's fusion; an extension, parity "
                         "against this build's own definition oracle.fuse_plane_feather) as a `feather` object to the line: the same "
                         "planes, tiles and gains fused by fuse_feather_zg_kernel into the uint16 canvas and into a float32 one, "
                         "2 rho + 2 (+ 2) bytes per voxel, HIP-event launch times, windows of one plane per channel compared with "
                         "the oracle.  The default N = 1 run includes it (--no-feather skips it); `value` stays the overwrite "
                         "fusion, the reference's own")
    ap.add_argument('--no-feather', action='store_true')
    ap.add_argument('--weak', action='store_true', help='N > 1 with a region workload: one region per rank (weak scaling)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--sha-out', default=None,
                    help='cfg4: write {plane: position-weighted digest of its fused canvas} of the last step to this JSON '
                         'file (one file per rank, ".rankR" appended); for the N-rank == 1-rank check, not for timing')
    ap.add_argument('--traffic-bytes', type=float, default=None,
                    help='HBM bytes per fusion launch from a separate rocprofv3 --pmc pass (else measured live, see --no-live-traffic)')
    ap.add_argument('--no-live-traffic', action='store_true',
                    help="N = 1 default workload: do not run the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this launch "
                         "after the timed region; roofline.traffic then carries the last committed measurement, labelled as such")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has not touched the GPU (torch is not
        # even imported yet) and never will: the N ranks are children, their rank 0 prints the JSON line.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from image_stitcher_amd import sharding

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus disagree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    check_device_count(world, torch.cuda.device_count())
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    backend = os.environ.get('SQ_DIST_BACKEND', 'nccl')   # 'gloo' only to rehearse N > 1 on a 1-GPU box
    # SQ_BENCH_FORCE_DIST=1: open the process group (and run the collectives) with ONE rank too -- the only way to
    # take the RCCL branch on a one-GPU box (torchrun --nproc-per-node 1, or plain python with the env below)
    force_dist = bool(os.environ.get('SQ_BENCH_FORCE_DIST'))
    if world > 1 or force_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if force_dist and 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
            os.environ['SQ_DIST_FORCE_COLLECTIVE'] = '1'
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    coll_dev = dev if backend == 'nccl' else None
    name = args.workload or ('cfg3' if world == 1 else 'cfg4')
    wl = dict(WORKLOADS[name], name=name)
    ctx = dict(args=args, rank=rank, world=world, dev=dev, coll_dev=coll_dev, wl=wl, backend=backend)
    if wl.get('job'):
        out = run_job(ctx)
    else:
        if world > 1 and not args.weak:
            raise SystemExit(f"workload {name} on {world} GPUs is a weak-scaling run (one region per rank): pass --weak, "
                             "or use the default workload (cfg4, strong scaling)")
        out = run_region(ctx)
        if world == 1 and args.workload is None and not args.planes and args.traffic_bytes is None and not args.no_live_traffic:
            while ARENAS:                     # the passes are child processes of their own: they need the HBM this one held
                ARENAS.pop().close()
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            try:
                live = live_traffic(n_planes=wl['channels'] * wl['nz'])
            except Exception as exc:      # whatever goes wrong around the profiler must not cost the line its other numbers
                print(f'[bench] live traffic: {type(exc).__name__}: {exc}; keeping the committed measurement', file=sys.stderr)
                live = None
            if live is not None:
                out['roofline']['traffic'], out['roofline']['traffic_source'] = live
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------
# roofline.traffic, measured: HBM bytes of the fusion launch from the PMC counters
# ------------------------------------------------------------------------------------------------------------
def live_traffic(n_planes, timeout_s=120):
    """HBM bytes per fusion launch of THIS workload on THIS box: two short child runs of this script (2 steps after 1
    warm-up, no CPU leg, no headline job) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, the
    program directly after `--`, no trace domains beside the counters, as the guide's HBM section prescribes -- after the
    timed region, outside every timing.  traffic = (2 * FETCH_SIZE + WRITE_SIZE) KB per launch: gfx950 reports half the
    bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM / rocprofv3).  Returns (bytes, source) or None when the
    profiler is missing, the process is itself being profiled, or a pass fails (the caller keeps the committed number)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if prof is None or os.environ.get('SQ_BENCH_PMC_CHILD') or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ):
        return None
    env = dict(os.environ, SQ_BENCH_PMC_CHILD='1', SQ_BENCH_NO_REFERENCE_JOB='1', TMPDIR='/tmp')
    child = [sys.executable, os.path.abspath(__file__), '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-live-traffic']
    kb = {}
    with tempfile.TemporaryDirectory(prefix='sq_pmc_', dir='/tmp') as tmp:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out_dir = os.path.join(tmp, counter)
            cmd = [prof, '--pmc', counter, '--output-format', 'csv', '-d', out_dir, '-o', 'run', '--'] + child
            # the pass runs in a process group of its own: if it overruns, the profiler AND the program under it are ended
            # (this exact group, nothing else), so that no copy of the bench keeps the GPU's memory
            try:
                proc = subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            except OSError as exc:
                print(f'[bench] live traffic: cannot start the {counter} pass ({exc}); keeping the committed measurement', file=sys.stderr)
                return None
            try:
                _, err = proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                print(f'[bench] live traffic: {counter} pass did not finish in {timeout_s} s; keeping the committed measurement', file=sys.stderr)
                return None
            files = glob.glob(os.path.join(out_dir, '**', '*counter_collection.csv'), recursive=True)
            if proc.returncode != 0 or not files:
                # the program's own words, not the profiler's closing log lines
                said = [l for l in err.decode(errors='replace').splitlines() if not l.startswith(('W2', 'I2', 'E2')) and 'amdgpu.ids' not in l]
                print(f'[bench] live traffic: {counter} pass failed (rc {proc.returncode}); keeping the committed measurement: '
                      + ' | '.join(said[-12:])[-1500:], file=sys.stderr)
                return None
            with open(files[0]) as fh:
                vals = [float(r['Counter_Value']) for r in csv.DictReader(fh)
                        if 'fuse_overwrite' in r['Kernel_Name'] and r['Counter_Name'] == counter]
            if not vals:
                return None
            kb[counter] = (sum(vals) / len(vals), len(vals))
    traffic = (2.0 * kb['FETCH_SIZE'][0] + kb['WRITE_SIZE'][0]) * 1024.0
    return traffic, (f"measured after the timed region on this box: two child runs of this command (--steps 2 --warmup 1) under rocprofv3 "
                     f"--pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean over {kb['FETCH_SIZE'][1]} / {kb['WRITE_SIZE'][1]} launches of "
                     f"{n_planes} planes: (2 x {kb['FETCH_SIZE'][0]:.0f} + {kb['WRITE_SIZE'][0]:.0f}) KB")


# ------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own N ranks (one process per GPU)
# ------------------------------------------------------------------------------------------------------------
def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_argv(n, bench_args, port):
    """The command line of the N-rank run: torch.distributed.run on this one node, rendezvous on 127.0.0.1 (the
    container's host name may not resolve), this script with the caller's own arguments."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(bench_args)


def launch_ranks(n, bench_args):
    """Start the ranks as CHILD processes and return their exit status (non-zero if any rank failed; nothing is
    restarted).  stdout / stderr are inherited, so rank 0's JSON line is this command's JSON line."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL needs across processes on these hosts
    env.setdefault('OMP_NUM_THREADS', '4')
    proc = subprocess.Popen(launch_argv(n, bench_args, free_port()), env=env)
    try:
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        return proc.wait() or 130


# ------------------------------------------------------------------------------------------------------------
# shared pieces
# ------------------------------------------------------------------------------------------------------------
def check_device_count(world, n_devices):
    """One process per GPU: a world of N ranks needs N devices on this node (SQ_DIST_BACKEND=gloo rehearses N ranks on
    fewer cards and says so itself)."""
    if n_devices < world and os.environ.get('SQ_DIST_BACKEND', 'nccl') == 'nccl':
        raise SystemExit(f"bench.py --gpus {world}: this node shows {n_devices} GPU(s) to torch.cuda.device_count(); one rank per GPU "
                         f"needs {world} (rehearse with SQ_DIST_BACKEND=gloo on fewer cards)")


def parallelism_text(world, backend, all_pairs):
    """config.parallelism of the headline job, for the registration mode that actually ran."""
    coll = 'RCCL' if backend == 'nccl' else backend
    head = f"planes dealt over {world} GPU{'s' if world != 1 else ''} in contiguous runs (a channel's z planes stay together)"
    if all_pairs:
        return (f"{head}; registration pairs dealt over the ranks in contiguous runs of the tile-row-ordered pair list, every rank "
                f"registers its run, [n_pairs, 3] float64 pair table all-gathered over {coll}, per-axis median on every rank; "
                "no image data exchanged")
    return f"{head}; rank 0 registers the reference's centre pairs, 8-int32 shift row all-gathered over {coll}; no image data exchanged"


def with_scale_keys(out, job):
    """Every bench line carries the point of the strong-scaling curve under ONE key pair, whatever its own `value` measures:
    `scale_workload` / `scale_value` = the headline job (BASELINE configs[3], 200 planes of the 32x32 grid) on this run's GPUs.
    A job line is that point itself; the N = 1 region line (value = config 3, the largest resident configuration) takes it
    from the job it runs afterwards on the same GPU (`headline_job_on_this_gpu`); a line without such a job says null."""
    if job is None:
        out['scale_workload'], out['scale_value'], out['scale_unit'] = None, None, out.get('unit')
        out['scale_note'] = 'no headline job in this run (a --workload / --planes override, or SQ_BENCH_NO_REFERENCE_JOB)'
    else:
        out['scale_workload'], out['scale_value'], out['scale_unit'] = job['workload'], job['value'], job['unit']
        out['scale_ms_per_job'], out['scale_wall_ms_per_job'] = job.get('ms_per_step'), job.get('wall_ms_per_job')
        out['scale_note'] = ('the strong-scaling curve is drawn from scale_value at every N (same workload at every N); `value` of the '
                             'N = 1 line is config 3, the largest configuration resident on one GPU')
    return out


def grid_setup(g, rank_seed):
    from image_stitcher_amd import placement, synth
    spec = synth.GridSpec(rows=g, cols=g, tile_h=TILE, tile_w=TILE, ov_y=OVERLAP, ov_x=OVERLAP,
                          jy=DRIFT[0], jx=DRIFT[1], channels=synth.DEFAULT_CHANNELS[:4], nz=50, seed=rank_seed)
    truth = placement.Shifts((DRIFT[0], -OVERLAP), (-OVERLAP, DRIFT[1]))
    wc, hc = placement.canvas_size(g, g, TILE, TILE, use_registration=True, shifts=truth)
    xs = [spec.stage_mm(0, c)[0] for c in range(g)]
    ys = [spec.stage_mm(r, 0)[1] for r in range(g)]
    # write order inside a plane = sorted file names of the fov numbers (stitcher.py:168)
    order = placement.filename_order([spec.fov_index(r, c) for r in range(g) for c in range(g)])
    order_rc = [divmod(i, g) for i in order]
    return spec, truth, wc, hc, xs, ys, order, order_rc


def plane_desc(spec, g, c, z):
    """Generator descriptors of the g x g tiles of plane (c, z), storage (row-major) order."""
    from image_stitcher_amd import native
    desc = np.zeros(g * g, dtype=native.SYNTH_DTYPE)
    scene = spec.scene_seed(0, 0, z, c) % 2 ** 64
    for r in range(g):
        for col in range(g):
            oy, ox = spec.origin(r, col)
            desc[r * g + col] = (scene, spec.noise_seed(0, 0, z, c, spec.fov_index(r, col)) % 2 ** 64, oy, ox)
    return desc


def plane_digest(plane, rows_per_chunk=2048):
    """Order-sensitive 61-bit digest of a uint16 device plane, computed on the device chunk by chunk."""
    import torch
    mod = (1 << 61) - 1
    h, w = plane.shape
    weights = (torch.arange(rows_per_chunk * w, device=plane.device, dtype=torch.int64) % 65521) + 1
    d = 0
    for r0 in range(0, h, rows_per_chunk):
        chunk = plane[r0:r0 + rows_per_chunk].contiguous().view(torch.int16).to(torch.int64).flatten() & 0xFFFF
        s = int((chunk * weights[:chunk.numel()]).sum().item())
        d = (d * 1000003 + s) % mod
    return d


ARENAS = []      # the DeviceArenas of this process's canvases (alloc_planes): their probe results go into the line


def memory_info():
    """What the line says about the memory under the canvas: the card's partition modes (sysfs) and the arena's classes."""
    import glob
    modes = {}
    for key in ('current_memory_partition', 'current_compute_partition'):
        vals = set()
        for f in glob.glob(f'/sys/class/drm/card*/device/{key}'):
            try:
                with open(f) as fh:
                    vals.add(fh.read().strip())
            except OSError:
                pass
        modes[key.replace('current_', '')] = sorted(vals) or None
    return dict(modes, canvas_arena=ARENAS[-1].info if ARENAS else None)


def alloc_planes(n, g, hc, wc, dev, layout):
    """Tile stacks [n, g*g, TILE, TILE] and canvas slots [n, hc, wc] (uint16) for n planes.
    'mixed': the canvas in a native.DeviceArena (every stretch of it lies over all three memory classes of the card, so the
    fusion kernel's row-segment writes run at the spread-out rate wherever a plane starts; csrc/arena.hip, DESIGN.md 5.1
    point 10), the tiles in a plain allocation.
    'arena': ONE allocation [canvas slot 0 | tile stack 0 | canvas slot 1 | tile stack 1 | ...]: the canvas slots then lie
    (canvas plane + tile stack) bytes apart, spread over all the memory the job holds, instead of side by side in an
    allocation of their own -- the planes of a group write fastest when they lie in different stretches of device memory
    (DESIGN.md 5.1 point 9), and this doubles the distance between them.  Both are views the library takes as they are:
    a canvas at any plane stride (a multiple of 128 bytes keeps the plane groups), tiles by pointer table.
    'separate': round 2's two allocations."""
    import torch
    from image_stitcher_amd import native
    if layout == 'mixed':
        # the canvas FIRST, while the card is empty: the arena takes candidate memory chunk by chunk, classifies it, keeps a third
        # per memory class and gives the rest back (native.DeviceArena); the tiles then come from a plain allocation -- reads
        # do not depend on the class
        need = native.canvas_bytes(n, hc, wc, torch.uint16)
        # ranks that SHARE a card (the gloo rehearsal of N ranks on one GPU) must not each reach for all of its memory while
        # they look for a balanced arena: they take what they need and no more (the classes then are whatever comes)
        shared_card = int(os.environ.get('WORLD_SIZE', '1')) > max(1, torch.cuda.device_count())
        # ... and so does the counter pass (live_traffic): under rocprofv3 the card does not get the unchosen slices back when they
        # are released (5.9 of 288 GiB reported free 10 s later; without the profiler at once) and the tiles would find no room --
        # bytes per launch do not depend on where the canvas lies
        exact = shared_card or bool(os.environ.get('SQ_BENCH_PMC_CHILD'))
        arena = native.DeviceArena(need, dev, candidate_bytes=need if exact else None)
        canvas = native.empty_canvas(n, hc, wc, torch.uint16, dev, arena=arena)
        ARENAS.append(arena)
        return torch.empty((n, g * g, TILE, TILE), dtype=torch.uint16, device=dev), canvas
    if layout != 'arena':
        return torch.empty((n, g * g, TILE, TILE), dtype=torch.uint16, device=dev), native.empty_canvas(n, hc, wc, torch.uint16, dev)
    cplane = -(-(hc * wc * 2) // 4096) * 4096
    tplane = g * g * TILE * TILE * 2
    arena = torch.empty(n * (cplane + tplane), dtype=torch.uint8, device=dev).view(torch.uint16)
    stride = (cplane + tplane) // 2
    canvas = arena.as_strided((n, hc, wc), (stride, wc, 1))
    tiles = arena.as_strided((n, g * g, TILE, TILE), (stride, TILE * TILE, TILE, 1), storage_offset=cplane // 2)
    return tiles, canvas


def tile_pointer_table(tiles, planes, tile_order, dev):
    """int64 device table [len(planes) * tiles per plane]: pointer of tile tile_order[i] of tile stack planes[k] at k * T + i."""
    import torch
    esz = TILE * TILE * 2
    ptrs = tiles.data_ptr() + torch.as_tensor(list(planes), dtype=torch.int64)[:, None] * (tiles.stride(0) * 2) + tile_order[None, :] * esz
    return ptrs.reshape(-1).to(dev)


def algorithmic_bytes(n_planes, covered, hc, wc, flat):
    # 4 B per covered voxel (2 B read + 2 B write), 2 B per uncovered voxel (zero write),
    # + the float32 flatfield once per plane
    return n_planes * (covered * 4 + (hc * wc - covered) * 2 + (TILE * TILE * 4 if flat else 0))


def committed_traffic(workload, n_planes):
    """PMC counters need their own rocprofv3 passes (the guide's HBM section: separate --pmc runs); the bench line
    carries the last committed measurement of this exact launch, and says so."""
    for name in ('pmc_traffic_latest.json', 'pmc_traffic_cfg4_batch.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as fh:
                pm = json.load(fh)
            if pm.get('workload') == workload and pm.get('planes') == n_planes:
                return pm['traffic_bytes_per_launch'], f'profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of ' \
                    'this launch on another box; not measured in this run)'
            if pm.get('workload') == workload and pm.get('planes') and n_planes:
                # the same geometry with another number of planes per launch (the last batch of a rank's run is shorter):
                # every plane moves the same bytes
                return pm['traffic_bytes_per_launch'] * n_planes / pm['planes'], \
                    f'profiles/{name} (rocprofv3 --pmc passes of a {pm["planes"]}-plane launch of this geometry on another box, ' \
                    f'scaled to the {n_planes:g} planes of this run\'s mean launch; not measured in this run)'
        except (OSError, ValueError, KeyError):
            pass
    return None, 'not measured (no committed PMC pass for this workload / plane count)'


# ------------------------------------------------------------------------------------------------------------
# region workloads (cfg2 / cfg3 / cfg4shard): everything resident, one launch per step
# ------------------------------------------------------------------------------------------------------------
def run_region(ctx):
    import torch
    import torch.distributed as dist
    from image_stitcher_amd import native, placement, registration, sharding, synth
    args, rank, world, dev, coll_dev, wl = (ctx[k] for k in ('args', 'rank', 'world', 'dev', 'coll_dev', 'wl'))
    g, C, Z = wl['grid'], wl['channels'], wl['nz']
    n_planes = C * Z
    spec, truth, wc, hc, xs, ys, order, order_rc = grid_setup(g, 1000 * 3 + rank * 100)
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
    tiles, canvas = alloc_planes(n_planes, g, hc, wc, dev, args.layout)
    for p in range(n_planes):
        c, z = divmod(p, Z_eff)
        native.synth_tiles(plane_desc(spec, g, c, z), TILE, TILE, spec.noise, 'uint16', dev, out=tiles[p])
    flat_list = None
    if wl['flat']:
        ffs = [torch.from_numpy(synth.synthetic_flatfield(TILE, TILE, np.float32) * np.float32(1 + 0.03125 * c)).to(dev)
               for c in range(C_eff)]
        flat_list = [ffs[p // Z_eff] for p in range(n_planes)]
    # Canvas SLOTS.  The library takes the canvas as base + slot * stride and everything else per slot (tile pointers,
    # gain pointers), so which (c, z) plane a slot holds is the caller's choice.  The planes that share a gain image go
    # through the kernel together, and that group writes fastest when its planes lie far apart in device memory
    # (DESIGN.md 5.1 point 8: stretches of tens of GiB behave like separate resources for this write pattern; the kernel
    # deals a key's planes to its groups round-robin for the same reason).  'spread': slot s holds channel s % C,
    # z = s // C -- a channel's planes are C slots apart, over the whole allocation.
    if args.canvas_order == 'spread':
        plane_of_slot = [(s % C_eff) * Z_eff + s // C_eff for s in range(n_planes)]
    else:
        plane_of_slot = list(range(n_planes))
    slot_of_plane = {p: s for s, p in enumerate(plane_of_slot)}
    slot_flats = [flat_list[p] for p in plane_of_slot] if flat_list else None
    flat_ptrs = native.pointer_table(slot_flats, dev) if slot_flats else None
    # canvas: dense rows like the reference's array; every plane starts on a 128-byte line (alloc_planes)
    canvas_of_plane = [canvas[slot_of_plane[p]] for p in range(n_planes)]
    tile_order = torch.tensor(order, dtype=torch.int64)
    # tile pointer table in write order (slot-major), so rect i <-> pointer i
    ptrs = tile_pointer_table(tiles, plane_of_slot, tile_order, dev)
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

    # The next region's registration is a chain of small latency-bound kernels on three tiles: on the launch stream it
    # sits between two fusion launches with the chip idle around it (~0.45 ms per step); on a stream of its own its few
    # workgroups run beside the fusion launch in flight (regions are independent) -- which then takes 0.03-0.28 ms longer:
    # 26.59 / 26.85 against 26.86 / 26.88 ms per step on one box.  Kept as an option; the default leaves the launch alone.
    reg_stream = torch.cuda.Stream(device=dev) if args.registration_stream == 'side' else None

    def start_registration():
        # registration: centre pairs on the registration plane (stitcher.py:422-498), enqueued only
        if reg_stream is None:
            return registration.register_grid_center_async(reg_plane, g, g, xs, ys, spec.pixel_size_um,
                                                           spec.pixel_binning, normalization='phase')
        with torch.cuda.stream(reg_stream):
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
        # host integer geometry + span plan (rebuilt every step: every region is registered on its own tiles and
        # its plan follows from ITS shifts -- no cache across regions here; cfg4 below is one region per job)
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
        native.fuse_planes(plan, None, canvas, slot_flats, tile_ptrs=ptrs, flat_ptrs=flat_ptrs)
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
    alg_bytes = algorithmic_bytes(n_planes, plan.covered_voxels, hc, wc, wl['flat'])
    achieved = alg_bytes / (fuse_ms * 1e-3) / 1e9
    traffic, traffic_source = (args.traffic_bytes, 'rocprofv3 --pmc passes of this command (--traffic-bytes)') \
        if args.traffic_bytes is not None else committed_traffic(wl['name'], n_planes)

    out = {
        'metric': 'stitched Mvoxels/s', 'value': round(value, 1), 'unit': 'Mvoxel/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
        # N = 1: the single-GPU point of the strong-scaling curve the N > 1 runs draw (a fixed job divided over the
        # GPUs); --weak with N > 1: one region per rank
        'scaling': 'weak' if (world > 1 and args.weak) else 'strong',
        'vs_baseline': None, 'dtype': 'u16', 'data': 'synthetic',
        'config': {'workload': wl['desc'], 'planes_resident_per_gpu': n_planes,
                   'canvas': [hc, wc], 'tiles_per_plane': g * g, 'canvas_order': args.canvas_order, 'layout': args.layout,
                   'memory': memory_info(),
                   'parallelism': f'one region per GPU x{world}, shift-table all-gather' if world > 1 else 'single GPU',
                   'shifts': {'h': list(state['shifts'].h_shift), 'v': list(state['shifts'].v_shift)},
                   'step': 'minmax + centre-pair PCC + span plan + one fusion launch over all planes; the next '
                           "region's registration is enqueued ahead of the fusion launch (regions are independent)"},
        'roofline': {'bound': 'hbm', 'kernel': 'fuse_overwrite_zg_kernel (u16, f32 gains, plane groups)' if wl['flat'] and n_planes > 1
                     else ('fuse_overwrite_kernel<u16,f32 flat>' if wl['flat'] else 'fuse_overwrite_kernel<u16>'),
                     'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_source,
                     'algorithmic_bytes_per_launch': int(alg_bytes), 'launch_ms': round(fuse_ms, 4)},
    }

    if rank == 0 and os.environ.get('SQ_BENCH_BREAKDOWN'):
        n = args.steps + args.warmup
        print('[bench] host ms per step: ' + ', '.join(f'{k} {v / n * 1e3:.2f}' for k, v in lap.items()), file=sys.stderr)
    # accuracy side of the metric (BASELINE.json: shift RMSE <= 0.5 px, fused max-rel-err <= 1e-5)
    out['parity'] = shift_parity(state['shifts'], truth)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'], check = cpu_baseline(tiles, flat_list, order, spec, xs, ys, g, hc, wc, truth, canvas_of_plane)
        out['cpu_baseline_reference'] = REFERENCE_TIMING
        out['parity'].update(check)
    # (one GPU only: under several ranks the others would sit in the closing collectives while rank 0 blends)
    want_feather = world == 1 and (args.fusion_mode == 'feather' or (args.workload is None and not args.planes and not args.no_feather
                                                                     and not os.environ.get('SQ_BENCH_PMC_CHILD')))
    if want_feather and rank == 0:
        out['feather'] = feather_leg(args, dev, g, hc, wc, n_planes, order, order_rc, state['shifts'], tiles, canvas, canvas_of_plane,
                                     plane_of_slot, ptrs, flat_list, slot_flats, flat_ptrs, check=not args.no_cpu_baseline)
    if world == 1 and args.workload is None and not args.planes and not os.environ.get('SQ_BENCH_NO_REFERENCE_JOB'):
        # The N > 1 runs of this script measure the headline job (cfg4, strong scaling).  For the record, the SAME job on
        # this one GPU (one step after one warm-up, ~3 s): the single-GPU point of that scaling curve, next to the config-3
        # number above, which is this run's `value`.
        del tiles, canvas, canvas_of_plane, ptrs, flat_list, slot_flats, flat_ptrs, reg_plane
        state.clear()
        fuse_events.clear()
        import gc
        gc.collect()
        while ARENAS:      # the region's canvas arena goes back to the driver before the job builds its own
            ARENAS.pop().close()
        torch.cuda.empty_cache()
        ref_args = argparse.Namespace(**dict(vars(args), steps=1, warmup=1, no_cpu_baseline=True, planes=0, batch=0, sha_out=None))
        job = run_job(dict(ctx, args=ref_args, wl=dict(WORKLOADS['cfg4'], name='cfg4')))
        out['headline_job_on_this_gpu'] = {
            'workload': job['config']['workload'], 'value': job['value'], 'unit': job['unit'], 'ms_per_step': job['ms_per_step'],
            'wall_ms_per_job': job['wall_ms_per_job'],
            'steps': 1, 'warmup': 1, 'scaling_note': 'the N = 1 point of the strong-scaling curve that bench.py --gpus N measures',
            'resident_batches': job['config']['resident_batches_per_gpu'], 'roofline_frac': job['roofline']['frac'],
            'launch_ms': job['roofline']['launch_ms'], 'host_ms_per_job': job['host_ms_per_job_rank0'],
            'registration': job['config']['registration']}
    return with_scale_keys(out, out.get('headline_job_on_this_gpu'))


def feather_window_rects(rects, y0, y1, x0, x1):
    """The rectangles cut to the canvas window [y0, y1) x [x0, x1), in the window's coordinates (geometry only: a rectangle's
    source origin moves with the cut, so the tile pixels and their feather weights stay the ones of the whole canvas)."""
    out = []
    for sy, sx, h, w, dy, dx in np.asarray(rects, dtype=np.int64):
        ya, yb, xa, xb = max(dy, y0), min(dy + h, y1), max(dx, x0), min(dx + w, x1)
        out.append((sy + ya - dy, sx + xa - dx, max(0, yb - ya), max(0, xb - xa), max(0, ya - y0), max(0, xa - x0)))
    return np.array(out, dtype=np.int64)


def feather_leg(args, dev, g, hc, wc, n_planes, order, order_rc, shifts, tiles, canvas, canvas_of_plane, plane_of_slot, ptrs, flat_list,
                slot_flats, flat_ptrs, check=True):
    """The distance-weighted blend on the resident planes of the region workload: `fuse_feather_zg_kernel` into the uint16
    canvas (with the workload's gains), then into a float32 canvas of half the planes (the arena's bytes).  Algorithmic
    bytes: 2 rho per voxel read (rho = tile pixels / canvas voxels: every tile pixel once) + the voxel written
    (+ the gain image once per plane).  Parity: canvas windows of one plane per channel -- a band over a tile-row seam with
    its four-tile corners, and the canvas' top-left corner -- against oracle.fuse_plane_feather, this build's own definition
    of the blend (the reference has none: PARITY UNPINNED): uint16 voxels equal, float32 voxels within 1e-5 relative."""
    import torch
    from image_stitcher_amd import native, placement
    rects = placement.grid_rects(g, g, TILE, TILE, shifts, order=order_rc, crop=False)
    plan = native.FusePlan(rects, TILE, TILE, hc, wc, native.SQ_FUSE_FEATHER)
    rho = g * g * TILE * TILE / (hc * wc)
    steps = max(3, min(args.steps, 10))

    def timed(cv, n, flats, fp):
        ms = []
        for k in range(2 + steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.fuse_planes(plan, None, cv, flats, tile_ptrs=ptrs[:n * g * g], flat_ptrs=fp)
            e1.record()
            torch.cuda.synchronize()
            if k >= 2:
                ms.append(e0.elapsed_time(e1))
        return float(np.mean(ms))

    def entry(ms, n, out_bytes, flat):
        alg = n * (hc * wc * (2 * rho + out_bytes) + (TILE * TILE * 4 if flat else 0))
        return {'launch_ms': round(ms, 4), 'planes': n, 'bytes_per_voxel': round(2 * rho + out_bytes, 4), 'algorithmic_bytes_per_launch': int(alg),
                'achieved': round(alg / ms / 1e6, 1), 'unit': 'GB/s', 'frac': round(alg / ms / 1e6 / HBM_PEAK_GBS, 4),
                'value': round(n * hc * wc / ms / 1e3, 1), 'value_unit': 'Mvoxel/s'}

    def compare(cv_of_plane, out_dtype, flat):
        """Windows of one plane per channel (z 0) against the oracle's definition."""
        from oracle import stitch_oracle as O
        step = TILE - OVERLAP
        windows = [(7 * step - 40, 7 * step + OVERLAP + 120, 0, wc), (0, 2300, 0, 2300)]
        per = max(1, n_planes // max(1, len({id(f) for f in flat_list}) if flat_list else 1))
        planes = sorted({c * per for c in range(max(1, n_planes // per))})[:4]
        worst, bad, voxels = 0.0, 0, 0
        t0 = time.perf_counter()
        for p in planes:
            if p not in cv_of_plane:
                continue
            host_tiles = [tiles[p, i].cpu().numpy() for i in order]
            fl = flat_list[p].cpu().numpy() if (flat and flat_list) else None
            for y0, y1, x0, x1 in windows:
                y1, x1 = min(y1, hc), min(x1, wc)
                want = O.fuse_plane_feather(host_tiles, feather_window_rects(rects, y0, y1, x0, x1), y1 - y0, x1 - x0, fl, out_dtype)
                got = cv_of_plane[p][y0:y1, x0:x1].cpu().numpy()
                voxels += want.size
                if np.issubdtype(np.dtype(out_dtype), np.integer):
                    bad += int(np.count_nonzero(got != want))
                else:
                    with np.errstate(all='ignore'):
                        rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-30)
                    rel[(got == want)] = 0.0
                    worst = max(worst, float(np.nanmax(rel)))
                    bad += int(np.count_nonzero(rel > 1e-5))
        return {'planes': planes, 'windows': [list(w) for w in windows], 'voxels': voxels, 'mismatched_voxels': bad,
                'max_rel_err': worst if not np.issubdtype(np.dtype(out_dtype), np.integer) else (0.0 if bad == 0 else None),
                'oracle_s': round(time.perf_counter() - t0, 1)}

    out = {'definition': "oracle.fuse_plane_feather (this build's own: the reference has no blend) -- PARITY UNPINNED",
           'kernel': 'fuse_feather_zg_kernel (plane groups; one-tile items through the pipelined row copy / divide, two-tile strips through '
                     'the grouped blend, corners plane by plane)',
           'rho': round(rho, 4), 'plan_items': int(plan.n_items), 'steps': steps}
    flat = bool(slot_flats)
    out['u16'] = entry(timed(canvas, n_planes, slot_flats, flat_ptrs), n_planes, 2, flat)
    out['u16']['canvas'] = 'uint16 (rounded half to even, clipped)' + (', float32 gains' if flat else '')
    if check:
        out['u16']['parity'] = compare({p: canvas_of_plane[p] for p in range(n_planes)}, np.uint16, flat)
    if ARENAS and n_planes >= 2:
        # the float32 canvas of half the planes in the same arena bytes (the uint16 canvas is done with)
        nf = n_planes // 2
        arena = ARENAS[-1]
        arena.reset()
        cf = native.empty_canvas(nf, hc, wc, torch.float32, dev, arena=arena)
        fpf = native.pointer_table(slot_flats[:nf], dev) if flat else None
        out['f32'] = entry(timed(cf, nf, slot_flats[:nf] if flat else None, fpf), nf, 4, flat)
        out['f32']['canvas'] = 'float32 (the blended value as it is)' + (', float32 gains' if flat else '')
        if check:
            out['f32']['parity'] = compare({plane_of_slot[s]: cf[s] for s in range(nf)}, np.float32, flat)
            out['f32']['parity']['tolerance'] = 'fused float voxels within 1e-5 relative (BASELINE.json north_star)'
        del cf
    return out


def shift_parity(sh, truth):
    err = np.array([sh.h_shift[0] - truth.h_shift[0], sh.h_shift[1] - truth.h_shift[1],
                    sh.v_shift[0] - truth.v_shift[0], sh.v_shift[1] - truth.v_shift[1]], dtype=np.float64)
    return {'shift_rmse_px': float(np.sqrt((err ** 2).mean())), 'shift_reference': 'planted drift'}


def oracle_rects(O, g, xs, ys, h, v, order, spec, hc, wc):
    """The write-ordered rectangles from the ORACLE's own restatement of the reference's placement and crop
    (oracle.tile_rect, stitcher.py:656-679 / :570-587) -- nothing of the product's geometry code."""
    rects = []
    for i in order:
        r, c = divmod(i, g)
        info = {'x': xs[c], 'y': ys[r]}
        x_px, y_px, top, bottom, left, right = O.tile_rect(info, list(xs), list(ys), TILE, TILE, spec.pixel_size_um, True,
                                                           tuple(h), tuple(v), None, 0, wc, hc)
        rects.append((top, left, TILE - top - bottom, TILE - left - right, y_px + top, x_px + left))
    return np.array(rects, dtype=np.int64)


def cpu_baseline(tiles, flat_list, order, spec, xs, ys, g, hc, wc, truth, canvas, max_planes=16):
    """The oracle (a numpy port of the reference path) on a bounded sample of the same workload:
    registration of the two centre pairs once (stitcher.py:1244-1246), then overwrite fusion of
    up to ``max_planes`` (c, z) planes, ~10-30 s of single-core work.  Geometry, shifts and voxels all come
    from oracle code; the GPU canvas is compared with the result voxel by voxel."""
    from oracle import stitch_oracle as O
    total = tiles.shape[0]
    n = min(max_planes, total)
    # the sample is spread over the launch: with 4 channels x 10 z and 16 planes, z 0-3 of EVERY channel (each channel
    # has its own gain image), not the first 16 planes
    n_groups = len({id(f) for f in flat_list}) if flat_list else 1
    per = max(1, total // n_groups)
    take = -(-n // n_groups)
    sample = sorted({gi * per + k for gi in range(n_groups) for k in range(min(take, per))})[:n] if n < total else list(range(total))
    n = len(sample)
    host = np.stack([tiles[p].cpu().numpy() for p in sample])      # (torch has no uint16 fancy indexing on the device)
    flats = [flat_list[p].cpu().numpy() if flat_list else None for p in sample]
    t0 = time.perf_counter()
    mx, my = O.max_overlaps(xs, ys, TILE, TILE, spec.pixel_size_um, spec.pixel_binning)
    ci = ri = (g - 1) // 2
    h = O.calculate_horizontal_shift(host[0, ri * g + ci], host[0, ri * g + ci + 1], mx, np.uint16, 'phase')
    v = O.calculate_vertical_shift(host[0, ri * g + ci], host[0, (ri + 1) * g + ci], my, np.uint16, 'phase')
    assert (tuple(h), tuple(v)) == (truth.h_shift, truth.v_shift)
    ow, oh, _ = O.output_dimensions(list(xs), list(ys), TILE, TILE, spec.pixel_size_um, True, tuple(h), tuple(v), None, 1)
    assert (oh, ow) == (hc, wc), f"oracle canvas {oh}x{ow} != device canvas {hc}x{wc}"
    rects = oracle_rects(O, g, xs, ys, h, v, order, spec, hc, wc)
    voxels = 0
    dt_check = 0.0
    mismatched = 0
    for k, p in enumerate(sample):
        plane = O.fuse_plane_overwrite([host[k, i] for i in order], rects, hc, wc, flats[k])
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
             'fused_checked': f'{n} of the timed launch\'s canvas planes (planes {sample}: every channel / gain image) compared voxel '
                              'by voxel with the oracle (shifts, canvas size and rectangles from the oracle\'s own geometry)'}
    return base, check


# ------------------------------------------------------------------------------------------------------------
# the headline job (cfg4): 200 planes of a 32x32 grid dealt over the ranks, walked in resident batches
# ------------------------------------------------------------------------------------------------------------
def run_job(ctx):
    import torch
    import torch.distributed as dist
    from image_stitcher_amd import native, placement, registration, sharding, synth
    args, rank, world, dev, coll_dev, wl = (ctx[k] for k in ('args', 'rank', 'world', 'dev', 'coll_dev', 'wl'))
    g, C, Z = wl['grid'], wl['channels'], wl['nz']
    total_planes = args.planes or C * Z
    spec, truth, wc, hc, xs, ys, order, order_rc = grid_setup(g, 1000 * 4)     # every rank: the SAME acquisition
    # plane p = c * Z + z; every rank ONE contiguous run of planes (the z planes of a channel share a gain image: kept
    # together they go through the kernel in groups of 5, csrc/fuse.hip), cut into resident batches of a multiple of 5
    mine = sharding.contiguous_blocks(total_planes, rank, world)
    plane_in, plane_out = g * g * TILE * TILE * 2, hc * wc * 2
    free, _ = torch.cuda.mem_get_info(dev)
    fit = int((free - (8 << 30)) // (plane_in + plane_out))
    if fit < 1:
        raise SystemExit(f"[bench] not even one plane of the 32x32 grid fits in {free / 2**30:.0f} GiB of free HBM")
    cap = args.batch or max(1, min(fit, 2 * GROUP) // GROUP * GROUP or min(fit, GROUP - 1))
    batches = [mine[i:i + cap] for i in range(0, len(mine), cap)]
    bmax = max((len(b) for b in batches), default=1)

    tiles, canvas = alloc_planes(bmax, g, hc, wc, dev, args.layout)
    ffs = [torch.from_numpy(synth.synthetic_flatfield(TILE, TILE, np.float32) * np.float32(1 + 0.03125 * c)).to(dev)
           for c in range(C)]
    descs = {p: plane_desc(spec, g, p // Z, p % Z) for p in mine}
    tile_order = torch.tensor(order, dtype=torch.int64)
    ptrs_all = tile_pointer_table(tiles, range(bmax), tile_order, dev)
    flat_tables = [native.pointer_table([ffs[p // Z] for p in b], dev) for b in batches]
    # Registration inputs, resident before the timed region like every other input.
    #   default: ALL adjacent pairs of the registration plane (channel 0, z 0), dealt over the ranks in contiguous runs
    #            of the tile-row-ordered pair list (registration.pairs_of_rank); a rank holds only the tiles ITS pairs
    #            touch (its band of tile rows + the row below);
    #   --centre-pairs: the reference's scheme, the centre tile and its right / lower neighbour on rank 0
    #            (stitcher.py:455-485).
    all_pairs = not args.centre_pairs
    ci = ri = (g - 1) // 2
    d0 = plane_desc(spec, g, 0, 0)
    mx, my = placement.registration_crop_widths(xs, ys, TILE, TILE, spec.pixel_size_um, spec.pixel_binning)
    if all_pairs:
        pairs = registration.grid_pair_list(g, g)
        my_pairs = registration.pairs_of_rank(len(pairs), rank, world)
        reg_cells = registration.cells_of_pairs(pairs, my_pairs)
    else:
        pairs, my_pairs = [], []
        reg_cells = [(ri, ci), (ri, ci + 1), (ri + 1, ci)] if rank == 0 else []
    reg_tiles = native.synth_tiles(d0[[r * g + c for r, c in reg_cells]], TILE, TILE, spec.noise, 'uint16', dev) \
        if reg_cells else None
    reg_index = {rc: i for i, rc in enumerate(reg_cells)}
    torch.cuda.synchronize()

    plans = {}
    fuse_events, state, lap = [], {}, {}

    def plan_for(shifts):
        """Host geometry + span plan, kept across steps while the shifts do not change (the plan depends on
        nothing else); one upload per plan.  The cache is emptied after the warm-up, so the FIRST timed job builds
        and uploads its plan inside its timed seconds, as a real run does once."""
        rects = placement.grid_rects(g, g, TILE, TILE, shifts, order=order_rc)
        key = rects.tobytes()
        if key not in plans:
            w_px, h_px = placement.canvas_size(g, g, TILE, TILE, use_registration=True, shifts=shifts)
            if (w_px, h_px) != (wc, hc):
                raise RuntimeError(f"registration returned {shifts}, canvas {h_px}x{w_px} != planned {hc}x{wc}")
            plans.clear()
            # host: the sweep into spans; items, seam owners and their order are produced on the device (the GPU is idle
            # here: registration has just been read back) -- 1.5 instead of 5.9 ms for this grid, csrc/plan_expand.hip
            plans[key] = native.FusePlan(rects, TILE, TILE, hc, wc, native.SQ_FUSE_OVERWRITE, expand_on_device=not args.host_plan)
            plans[key].device_table(dev)          # the expansion (or upload) belongs to the job that needed the plan
        return plans[key]

    multi = world > 1 or dist.is_initialized()

    def sync_all():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    def register():
        """-> (Shifts, seconds of registration, seconds of the collective)."""
        t0 = time.perf_counter()
        if all_pairs:
            local = registration.register_pair_subset(reg_tiles, reg_index, pairs, my_pairs, TILE, TILE, mx, my, 'phase')
            t1 = time.perf_counter()
            table = sharding.all_gather_pair_table(local, len(pairs), rank, world, device=coll_dev)      # RCCL
            t2 = time.perf_counter()
            state['pair_table'] = table
            return registration.shifts_from_pair_table(pairs, table, TILE, TILE, mx, my, g), t1 - t0, t2 - t1
        row = sharding.shifts_to_row(None)
        if rank == 0:    # the reference registers once, on the first region's centre tiles (stitcher.py:1244-1246)
            sh = registration.register_grid_center(reg_tiles, g, g, xs, ys, spec.pixel_size_um, spec.pixel_binning,
                                                   normalization='phase', tile_index=lambda r, c: reg_index[(r, c)])
            row = sharding.shifts_to_row(sh)
        t1 = time.perf_counter()
        shifts = sharding.first_valid(sharding.all_gather_shift_table(row[None], device=coll_dev))   # RCCL
        return shifts, t1 - t0, time.perf_counter() - t1

    def synth_batch(b):
        for k, p in enumerate(b):      # this batch's tiles -> HBM (not timed: inputs are resident when timing starts)
            native.synth_tiles(descs[p], TILE, TILE, spec.noise, 'uint16', dev, out=tiles[k])

    def job(record):
        """One pass over the 200 planes.  Returns this rank's timed seconds = the host-clocked head (registration,
        all-gather, geometry + plan [+ its upload when it is new], closed by a synchronisation) + the fusion launches,
        each clocked by a pair of events on the launch stream.  The generator runs of the next batch are enqueued
        on the same stream between the launches (after the closing event of one, before the opening event of the
        next), so there is ONE host synchronisation per job, not two per batch."""
        if batches:
            synth_batch(batches[0])
        sync_all()
        t0 = time.perf_counter()
        shifts, t_reg, t_gather = register()
        t2 = time.perf_counter()
        plan = plan_for(shifts)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for k, v in (('register', t_reg), ('allgather', t_gather), ('plan', t3 - t2)):
            lap.setdefault(k, []).append(v)
        events = []
        for bi, b in enumerate(batches):
            if bi:
                synth_batch(b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.fuse_planes(plan, None, canvas[:len(b)], [ffs[p // Z] for p in b], tile_ptrs=ptrs_all[:len(b) * g * g],
                               flat_ptrs=flat_tables[bi])
            e1.record()
            events.append((e0, e1, len(b)))
            if args.sha_out and record == 'last':
                for k, p in enumerate(b):
                    state.setdefault('sha', {})[int(p)] = plane_digest(canvas[k])
        torch.cuda.synchronize()
        timed = (t3 - t0) + sum(a.elapsed_time(c) for a, c, _ in events) * 1e-3
        if record:
            fuse_events.extend(events)
        state['plan'], state['shifts'] = plan, shifts
        return timed

    for _ in range(args.warmup):
        job(False)
    plans.clear()          # the first timed job plans and uploads like the first job of a real run
    sync_all()
    w0 = time.perf_counter()
    per_job = [job('last' if k + 1 == args.steps else True) for k in range(args.steps)]
    seconds = float(sum(per_job))
    sync_all()
    # what a wall clock sees of the same K jobs: the timed pieces PLUS the device synthesis of every batch's tiles (the
    # stand-in for reading them; 1.6 TiB cannot be resident) and the gaps between launches -- reported beside ms_per_step,
    # never as `value` (--sha-out's digests would sit in it too: that option is for the N-rank == 1-rank check)
    wall = time.perf_counter() - w0
    if args.sha_out:
        with open(f'{args.sha_out}.rank{rank}', 'w') as fh:
            json.dump(state.get('sha', {}), fh)
    mine_s = seconds
    first_job = per_job[0]
    steady = float(np.mean(per_job[1:])) if len(per_job) > 1 else None
    if multi:      # the job's time is the slowest rank's: MAX over ranks of the sum and of the first job
        t = torch.tensor([seconds, first_job, steady if steady is not None else 0.0, wall], dtype=torch.float64, device=coll_dev or 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seconds, first_job, wall = float(t[0].item()), float(t[1].item()), float(t[3].item())
        steady = float(t[2].item()) if steady is not None else None

    plan, shifts = state['plan'], state['shifts']
    assert tuple(shifts.h_shift) == truth.h_shift and tuple(shifts.v_shift) == truth.v_shift, \
        f"registration did not recover the planted drift: {shifts}"
    value = total_planes * hc * wc * args.steps / seconds / 1e6
    ms = np.array([a.elapsed_time(b) for a, b, _ in fuse_events]) if fuse_events else np.zeros(1)
    npl = np.array([n for _, _, n in fuse_events]) if fuse_events else np.ones(1)
    alg = algorithmic_bytes(1, plan.covered_voxels, hc, wc, True)
    achieved = alg * npl.sum() / (ms.sum() * 1e-3) / 1e9 if ms.sum() > 0 else 0.0
    fracs = [achieved / HBM_PEAK_GBS]
    if multi:      # every rank's fraction, for the record
        t = torch.tensor([achieved / HBM_PEAK_GBS], dtype=torch.float64, device=coll_dev or 'cpu')
        allf = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allf, t)
        fracs = [float(x.item()) for x in allf]
    out = {
        'metric': 'stitched Mvoxels/s', 'value': round(value, 1), 'unit': 'Mvoxel/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(seconds / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'u16', 'data': 'synthetic',
        'wall_ms_per_job': round(wall / args.steps * 1e3, 3),
        'wall_note': 'perf_counter around the K timed jobs, MAX over ranks: ms_per_step plus the device synthesis of every batch\'s tiles '
                     '(stand-in for reading them: 1.6 TiB cannot be resident) and the gaps between launches',
        'first_job_ms': round(first_job * 1e3, 3), 'steady_job_ms': None if steady is None else round(steady * 1e3, 3),
        'first_job_note': 'the first timed job builds the fusion plan (cache emptied after the warm-up: host sweep into spans + work list '
                          'expanded on the device, or --host-plan: host planner + upload); the later ones re-use it while the shifts '
                          'stay; ms_per_step is the mean over ALL timed jobs, the first included',
        'host_ms_per_job_rank0': {k: round(float(np.mean(v[args.warmup:])) * 1e3, 3) for k, v in lap.items()},
        'config': {'workload': wl['desc'], 'planes_total': total_planes, 'planes_per_gpu': len(mine),
                   'registration': (f'all {len(pairs)} adjacent pairs of the registration plane, dealt over the ranks in contiguous runs '
                                    f'({len(my_pairs)} pairs / {len(reg_cells)} resident tiles on rank 0), [n_pairs, 3] float64 table '
                                    'all-gathered, per-axis median') if all_pairs else
                                   "the reference's centre pairs on rank 0 (stitcher.py:455-485), 8-int32 row all-gathered",
                   'resident_batches_per_gpu': [len(b) for b in batches], 'canvas': [hc, wc], 'tiles_per_plane': g * g, 'layout': args.layout,
                   'memory': memory_info(),
                   'parallelism': parallelism_text(world, ctx['backend'], all_pairs),
                   'shifts': {'h': list(shifts.h_shift), 'v': list(shifts.v_shift)},
                   'step': 'the whole job: registration + all-gather + span plan (kept while the shifts stay), then per '
                           'resident batch one fusion launch; a batch\'s tiles are synthesised on the device before its launch, '
                           'outside its event pair (inputs resident when timing starts); timed = MAX over ranks of host-clocked '
                           'head + event-clocked launches'},
        'roofline': {'bound': 'hbm', 'kernel': 'fuse_overwrite_zg_kernel (u16, f32 gains, plane groups)', 'achieved': round(achieved, 1),
                     'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                     'traffic': committed_traffic(wl['name'], float(npl.mean()) if float(npl.mean()) % 1 else int(npl.mean()))[0],
                     'traffic_source': committed_traffic(wl['name'], float(npl.mean()) if float(npl.mean()) % 1 else int(npl.mean()))[1],
                     'algorithmic_bytes_per_launch': int(alg * npl.mean()), 'launch_ms': round(float(ms.mean()), 4),
                     'frac_per_rank': [round(f, 4) for f in fracs], 'of_rank': 0},
        'parity': shift_parity(shifts, truth),
    }
    if rank == 0 and os.environ.get('SQ_BENCH_BREAKDOWN'):
        print(f'[bench] rank 0 timed {mine_s / args.steps * 1e3:.2f} ms/job; host ms per job (every job, warm-up first): ' +
              '; '.join(f'{k} ' + ' '.join(f'{x * 1e3:.2f}' for x in v) for k, v in lap.items()), file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the last batch is still resident: the oracle on a bounded sample of it (2 planes of the 32x32 grid)
        b = batches[-1]
        flat_list = [ffs[p // Z] for p in b]
        base, check = cpu_baseline_fusion_only(tiles[:len(b)], flat_list, order, spec, xs, ys, g, hc, wc, truth, canvas[:len(b)])
        out['cpu_baseline'], out['cpu_baseline_reference'] = base, REFERENCE_TIMING
        out['parity'].update(check)
    return with_scale_keys(out, {'workload': wl['desc'] + ('' if total_planes == C * Z else f' [cut to {total_planes} of {C * Z} planes]'), 'value': out['value'], 'unit': out['unit'], 'ms_per_step': out['ms_per_step'],
                                 'wall_ms_per_job': out['wall_ms_per_job']})


def cpu_baseline_fusion_only(tiles, flat_list, order, spec, xs, ys, g, hc, wc, truth, canvas, max_planes=2):
    """cfg4's resident batch does not hold the registration plane: the oracle's fusion (with the planted shifts the
    device registration was checked against) on a bounded sample, compared with the GPU canvas."""
    from oracle import stitch_oracle as O
    n = min(max_planes, tiles.shape[0])
    host = np.stack([tiles[p].cpu().numpy() for p in range(n)])      # plane by plane: the stacks may be views of an arena
    t0 = time.perf_counter()
    rects = oracle_rects(O, g, xs, ys, truth.h_shift, truth.v_shift, order, spec, hc, wc)
    voxels = mismatched = 0
    dt_check = 0.0
    for p in range(n):
        plane = O.fuse_plane_overwrite([host[p, i] for i in order], rects, hc, wc, flat_list[p].cpu().numpy())
        voxels += plane.size
        tc = time.perf_counter()
        mismatched += int(np.count_nonzero(canvas[p].cpu().numpy() != plane))
        dt_check += time.perf_counter() - tc
        del plane
    dt = time.perf_counter() - t0 - dt_check
    base = {'value': round(voxels / dt / 1e6, 1), 'unit': 'Mvoxel/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n} (c,z) planes of the workload ({g}x{g} tiles -> {hc}x{wc} canvas each): fusion only, numpy oracle, '
                      f'{dt:.1f} s, 1 of {os.cpu_count()} host cores used'}
    check = {'fused_max_rel_err': 0.0 if mismatched == 0 else None, 'fused_mismatched_voxels': mismatched,
             'fused_checked': f'{n} planes of the last resident batch compared voxel by voxel with the oracle'}
    return base, check


if __name__ == '__main__':
    main()
