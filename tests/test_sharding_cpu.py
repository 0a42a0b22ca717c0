"""The N > 1 path on CPU: world_size-2 gloo processes exchange the shift table exactly as the
GPU ranks do over RCCL, and the plane / pair partitions cover the work exactly once."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from image_stitcher_amd import sharding
from image_stitcher_amd.placement import Shifts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_cyclic_covers_everything_once():
    for n in (0, 1, 7, 40, 200):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in sharding.block_cyclic(n, r, world))
            assert seen == list(range(n))
            blocks = [sharding.contiguous_blocks(n, r, world) for r in range(world)]
            assert [i for b in blocks for i in b] == list(range(n))                      # runs, in rank order
            assert max(map(len, blocks)) - min(map(len, blocks)) <= 1
            sizes = [len(sharding.block_cyclic(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.block_cyclic(4, 2, 2)


def test_shift_rows_roundtrip():
    for s in (Shifts((3, -244), (-244, -2)), Shifts((-2, -48), (-31, 2), (3, -44), 1), Shifts()):
        row = sharding.shifts_to_row(s)
        assert row.dtype == np.int32 and row.shape == (sharding.SHIFT_ROW,)
        assert sharding.row_to_shifts(row) == s
    assert sharding.row_to_shifts(sharding.shifts_to_row(None)) is None
    assert sharding.first_valid(np.stack([sharding.shifts_to_row(None), sharding.shifts_to_row(Shifts((1, 2), (3, 4)))])) \
        == Shifts((1, 2), (3, 4))
    # single process: the gather is the identity
    rows = sharding.shifts_to_row(Shifts((1, 2), (3, 4)))[None]
    np.testing.assert_array_equal(sharding.all_gather_shift_table(rows), rows)
    np.testing.assert_array_equal(sharding.all_gather_shift_table_async(rows).result(), rows)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from image_stitcher_amd import sharding as sh
        from image_stitcher_amd.placement import Shifts as S
        # parity mode: only the rank that owns the registration plane has a result (stitcher.py:1244-1246)
        mine = S((3, -244), (-244, -2)) if rank == 1 else None
        table = sh.all_gather_shift_table(sh.shifts_to_row(mine)[None])
        assert table.shape == (world, sh.SHIFT_ROW)
        assert sh.first_valid(table) == S((3, -244), (-244, -2))
        # per-region mode (one region per rank): every rank contributes its own row, rank-major order
        own = S((rank, -100 - rank), (-200 - rank, rank), (7, 8) if rank == 0 else None, rank)
        table = sh.all_gather_shift_table(sh.shifts_to_row(own)[None])
        for r in range(world):
            want = S((r, -100 - r), (-200 - r, r), (7, 8) if r == 0 else None, r if r == 0 else 0)
            got = sh.row_to_shifts(table[r])
            assert got.h_shift == want.h_shift and got.v_shift == want.v_shift and got.h_shift_rev == want.h_shift_rev
        # several gathers in flight (the bench overlaps them with the fusion launches), collected later
        rows = [np.full((1, sh.SHIFT_ROW), 10 * k + rank, dtype=np.int32) for k in range(3)]
        pending = [sh.all_gather_shift_table_async(r) for r in rows]
        for k, pnd in enumerate(pending):
            got = pnd.result()
            assert got.shape == (world, sh.SHIFT_ROW) and [int(v) for v in got[:, 0]] == [10 * k + r for r in range(world)]
            assert pnd.result() is got          # idempotent
        # planes dealt block-cyclically: the union over ranks is every plane exactly once
        planes = torch.zeros(40, dtype=torch.int32)
        planes[sh.block_cyclic(40, rank, world)] = 1
        dist.all_reduce(planes)
        assert bool((planes == 1).all())
        np.save(os.path.join(out_dir, f'ok{rank}.npy'), table)
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / 'ok0.npy'), np.load(tmp_path / 'ok1.npy')
    np.testing.assert_array_equal(a, b)      # every rank ends with the same table


def _shift_table_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from image_stitcher_amd import sharding as sh
        from image_stitcher_amd.placement import Shifts as S
        from image_stitcher_amd.stitcher import Stitcher
        st = object.__new__(Stitcher)       # only the writer's inputs; no dataset, no device
        st.timepoints, st.regions = [0], ['A1', 'A2']
        st.per_region_registration = True
        st.output_folder = out_dir
        st.scan_pattern = 'Unidirectional'
        # fewer units (2) than ranks (3), shared by planes: rank 0 measured BOTH units, the others none
        rows = {i: sh.shifts_to_row(S((i, -100 - i), (-200 - i, i))) for i in range(2)} if rank == 0 else {}
        st._write_shift_table(2, rows, rank, world, None, shared=True)
        # units dealt block-cyclically (5 units, 3 ranks): ceil(5/3) = 2 rows per rank
        st.regions = ['A1', 'A2', 'A3', 'A4', 'A5']
        st.output_folder = os.path.join(out_dir, 'dealt')
        if rank == 0:
            os.makedirs(st.output_folder, exist_ok=True)
        dist.barrier()
        rows = {i: sh.shifts_to_row(S((i, -100 - i), (-200 - i, i))) for i in sh.block_cyclic(5, rank, world)}
        st._write_shift_table(5, rows, rank, world, None, shared=False)
    finally:
        dist.destroy_process_group()


def test_shift_table_json_shared_region_three_ranks(tmp_path):
    """2 units on 3 ranks (regions shared by planes) overflowed rank 0's gather buffer."""
    import json
    port = _free_port()
    mp.spawn(_shift_table_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    for folder, n in ((tmp_path, 2), (tmp_path / 'dealt', 5)):
        with open(folder / 'shift_table.json') as fh:
            doc = json.load(fh)
        assert doc['per_region_registration'] is True and len(doc['shifts']) == n
        for i, e in enumerate(doc['shifts']):
            assert e['h_shift'] == [i, -100 - i] and e['v_shift'] == [-200 - i, i]


# ---------------------------------------------------------------------------------------------------------------
# all-pairs registration sharded by pair (north star; SURVEY.md 8e: n_pairs x {dy, dx, err} float64 all-gather)
# ---------------------------------------------------------------------------------------------------------------
def test_pair_list_is_in_tile_row_order_and_runs_touch_row_bands():
    from image_stitcher_amd import registration as R
    pairs = R.grid_pair_list(32, 32)
    assert len(pairs) == 2 * 32 * 31 == 1984                                   # SURVEY 8e's figure
    assert sum(p[0] == R.PAIR_H for p in pairs) == sum(p[0] == R.PAIR_V for p in pairs) == 992
    assert len(set(pairs)) == len(pairs)
    for kind, a, b in pairs:
        assert b == ((a[0], a[1] + 1) if kind == R.PAIR_H else (a[0] + 1, a[1]))
    assert [p[1][0] for p in pairs] == sorted(p[1][0] for p in pairs)          # reference cells in tile-row order
    covered = []
    for world in (1, 2, 3, 8):
        most = 0
        for r in range(world):
            mine = R.pairs_of_rank(len(pairs), r, world)
            covered.extend(mine)
            cells = R.cells_of_pairs(pairs, mine)
            rows = {c[0] for c in cells}
            assert rows == set(range(min(rows), max(rows) + 1))               # one band of tile rows
            most = max(most, len(cells))
        assert sorted(covered) == list(range(len(pairs)))
        covered.clear()
        # a rank reads its band + at most two boundary rows, not the plane: 8 ranks -> <= (4 + 2) rows of 32
        assert most <= (32 // world + 2) * 32 or world == 1
    # a missing tile removes exactly the pairs it is part of
    present = {(r, c) for r in range(3) for c in range(3)} - {(1, 1)}
    left = R.grid_pair_list(3, 3, present)
    assert len(left) == 12 - 4 and all((1, 1) not in p[1:] for p in left)
    assert R.grid_pair_list(1, 1) == [] and len(R.grid_pair_list(1, 4)) == 3 and len(R.grid_pair_list(4, 1)) == 3


def test_pair_table_medians_convert_like_the_reference_and_take_the_lower_median():
    from image_stitcher_amd import registration as R
    pairs = R.grid_pair_list(3, 3)
    n0h, n1h = 1024, 256          # 2048^2 tiles, overlap crop 256: horizontal crop 1024 x 256, vertical 256 x 1024
    table = np.full((len(pairs), 3), 0.0)
    for i, (kind, a, b) in enumerate(pairs):
        table[i, :2] = (3.0, 12.0) if kind == R.PAIR_H else (12.0, -2.0)     # raw skimage shifts of the crops
    table[0, :2] = (40.0, 90.0)             # one derailed pair: the median ignores it
    table[3, :2] = (np.nan, np.nan)         # a pair that produced nothing is left out
    med = R.pair_table_medians(pairs, table, 2048, 2048, 256, 256, 3)
    assert med == {'h_shift': (3, 12 - n1h), 'v_shift': (12 - 256, -2)}      # round(s0), round(s1 - w) / round(s0 - w), round(s1)
    s = R.shifts_from_pair_table(pairs, table, 2048, 2048, 256, 256, 3)
    assert s == Shifts((3, -244), (-244, -2))
    # python round = banker's rounding on the float64 (stitcher.py:511): 2.5 -> 2, 3.5 -> 4
    t2 = table.copy()
    t2[[i for i, p in enumerate(pairs) if p[0] == R.PAIR_H], 0] = 2.5
    assert R.pair_table_medians(pairs, t2, 2048, 2048, 256, 256, 3)['h_shift'][0] == 2
    # S-Pattern: rows of the centre row's parity -> h_shift, the others -> h_shift_rev (stitcher.py:486-496)
    t3 = table.copy()
    for i, (kind, a, b) in enumerate(pairs):
        if kind == R.PAIR_H and a[0] % 2 == 0:
            t3[i, :2] = (-1.0, 20.0)
    med = R.pair_table_medians(pairs, t3, 2048, 2048, 256, 256, 3, 'S-Pattern')
    assert med['h_shift'] == (3, -244) and med['h_shift_rev'] == (-1, 20 - 256) and med['h_shift_rev_odd'] == 0
    # even count: the LOWER median, so the result is one of the measured integers
    two = R.grid_pair_list(1, 3)
    assert R.pair_table_medians(two, np.array([[1.0, 10.0, 0], [5.0, 30.0, 0]]), 2048, 2048, 256, 256, 1)['h_shift'] == (1, 10 - 256)
    # a single row has no vertical pairs: v_shift is not reported (the caller keeps the reference's default)
    assert 'v_shift' not in R.pair_table_medians(two, np.zeros((2, 3)), 2048, 2048, 256, 256, 1)


def _pair_table_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from image_stitcher_amd import sharding as sh
        for n_pairs in (1984, 7, 2, 0):       # 1984 = the 32x32 grid; 7 and 2: uneven runs, ranks with nothing
            full = np.arange(n_pairs * 3, dtype=np.float64).reshape(n_pairs, 3) * 0.1 - 5.0
            mine = sh.contiguous_blocks(n_pairs, rank, world)
            table = sh.all_gather_pair_table(full[mine], n_pairs, rank, world)
            assert table.dtype == np.float64 and table.shape == (n_pairs, 3)
            np.testing.assert_array_equal(table, full)                      # bit for bit, pair order, on every rank
        with pytest.raises(ValueError):
            sh.all_gather_pair_table(np.zeros((1, 3)), 1984, rank, world)   # not this rank's run
        # the whole sharded registration with the device step replaced by a stand-in (no GPU here): every rank loads
        # exactly the cells its run of pairs touches, registers exactly its run, and ends with the complete table
        from image_stitcher_amd import registration as R
        pairs = R.grid_pair_list(5, 4)
        seen = {}

        def fake_subset(tiles, local_index, prs, indices, h, w, mx, my, normalization='phase', minmax=None):
            assert prs is pairs and list(indices) == R.pairs_of_rank(len(pairs), rank, world)
            assert set(local_index) == set(seen['cells']) and len(tiles) == len(seen['cells'])
            return np.array([[i + 0.5, -i, 0.001 * i] for i in indices], dtype=np.float64).reshape(-1, 3)

        def load_cells(cells):
            seen['cells'] = list(cells)
            return list(cells)
        real = R.register_pair_subset
        R.register_pair_subset = fake_subset
        try:
            table = R.register_all_pairs_sharded(pairs, load_cells, 64, 64, 16, 16, 'phase', rank=rank, world=world)
        finally:
            R.register_pair_subset = real
        np.testing.assert_array_equal(table, np.array([[i + 0.5, -i, 0.001 * i] for i in range(len(pairs))]))
        assert seen['cells'] == R.cells_of_pairs(pairs, R.pairs_of_rank(len(pairs), rank, world)) and len(seen['cells']) < 20
        # a rank of this N-rank group that registers a region ON ITS OWN (per-region all-pairs registration with
        # the regions dealt to the ranks: Stitcher.run calls calculate_shifts without _pair_ranks, i.e. rank 0 of a world
        # of 1) takes no part in a collective -- the other ranks are busy with THEIR regions -- and gets its own rows back.
        # Every rank does it with a different table; a collective here would hang or mix them.
        own = np.arange(21, dtype=np.float64).reshape(7, 3) + 100.0 * rank
        np.testing.assert_array_equal(sh.all_gather_pair_table(own, 7, 0, 1), own)

        def fake_all(tiles, local_index, prs, indices, h, w, mx, my, normalization='phase', minmax=None):
            assert list(indices) == list(range(len(pairs)))          # the whole list: this rank alone registers its region
            return np.array([[i + rank, -i, 0.5] for i in indices], dtype=np.float64).reshape(-1, 3)
        R.register_pair_subset = fake_all
        try:
            alone = R.register_all_pairs_sharded(pairs, load_cells, 64, 64, 16, 16, 'phase', rank=0, world=1)
        finally:
            R.register_pair_subset = real
        np.testing.assert_array_equal(alone, np.array([[i + rank, -i, 0.5] for i in range(len(pairs))]))
        open(os.path.join(out_dir, f'pairs_ok{rank}'), 'w').close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_pair_table_all_gather_gloo(tmp_path, world):
    mp.spawn(_pair_table_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f'pairs_ok{r}').exists() for r in range(world))
    # single process: identity, and a world without a process group is an error rather than a silent partial table
    full = np.arange(12, dtype=np.float64).reshape(4, 3)
    np.testing.assert_array_equal(sharding.all_gather_pair_table(full, 4, 0, 1), full)
    with pytest.raises(RuntimeError):
        sharding.all_gather_pair_table(full[:2], 4, 0, 2)


def test_bench_launcher_argv(monkeypatch):
    """`python bench.py --gpus N` starts its own ranks: the launcher's command line, without a GPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_for_test', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv = bench.launch_argv(4, ['--gpus', '4', '--steps', '3', '--warmup', '1'], 29555)
    assert argv[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in argv and '--nproc-per-node=4' in argv
    assert argv[argv.index('--master-addr') + 1] == '127.0.0.1' and argv[argv.index('--master-port') + 1] == '29555'
    k = argv.index(os.path.join(ROOT, 'bench.py'))
    assert argv[k + 1:] == ['--gpus', '4', '--steps', '3', '--warmup', '1']
    assert 1024 < bench.free_port() < 65536
    # the launcher relays the children's status and never imports torch.cuda in the parent
    calls = {}

    class FakeProc:
        def __init__(self, cmd, env=None):
            calls['cmd'], calls['env'] = cmd, env

        def wait(self):
            return 3
    import subprocess
    monkeypatch.setattr(subprocess, 'Popen', FakeProc)
    assert bench.launch_ranks(2, ['--gpus', '2']) == 3
    assert calls['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and '--nproc-per-node=2' in calls['cmd']
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '1'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3 and calls['cmd'][-4:] == ['--gpus', '2', '--steps', '1']


def test_bench_arena_layout_views_do_not_alias():
    """bench.alloc_planes('arena'): canvas slots and tile stacks interleaved in ONE allocation -- both are plain views (a
    canvas at a plane stride that is a multiple of 128 bytes, dense tile stacks reachable through the pointer table) and
    no canvas voxel shares a byte with a tile."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_for_test2', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n, g, hc, wc = 3, 1, 101, 203
    tiles, canvas = bench.alloc_planes(n, g, hc, wc, 'cpu', 'arena')
    assert tuple(tiles.shape) == (n, 1, bench.TILE, bench.TILE) and tuple(canvas.shape) == (n, hc, wc)
    assert canvas.stride(0) == tiles.stride(0) and (canvas.stride(0) * 2) % 128 == 0 and canvas.stride(1) == wc
    assert all(tiles[p].is_contiguous() and canvas[p].is_contiguous() for p in range(n))
    canvas.view(torch.int16).fill_(-1)                  # 0xFFFF everywhere in the canvas slots
    tiles.view(torch.int16).fill_(7)
    assert bool((canvas.view(torch.int16) == -1).all()) and bool((tiles.view(torch.int16) == 7).all())
    # slot k's tile stack follows slot k's canvas, slot k + 1's canvas follows that
    for k in range(n):
        assert tiles[k].data_ptr() - canvas[k].data_ptr() == -(-(hc * wc * 2) // 4096) * 4096
        if k:
            assert canvas[k].data_ptr() - tiles[k - 1].data_ptr() == bench.TILE * bench.TILE * 2
    order = torch.tensor([0], dtype=torch.int64)
    table = bench.tile_pointer_table(tiles, [2, 0, 1], order, 'cpu')
    assert table.tolist() == [tiles[2].data_ptr(), tiles[0].data_ptr(), tiles[1].data_ptr()]
    # 'separate': two dense allocations, the same shapes
    t2, c2 = bench.alloc_planes(n, g, hc, wc, 'cpu', 'separate')
    assert t2.is_contiguous() and tuple(t2.shape) == tuple(tiles.shape) and tuple(c2.shape) == tuple(canvas.shape)
    assert bench.tile_pointer_table(t2, range(n), order, 'cpu').tolist() == [t2[k].data_ptr() for k in range(n)]


def test_bench_live_traffic_reads_the_pmc_passes(monkeypatch, tmp_path):
    """bench.py's roofline.traffic: two child passes under `rocprofv3 --pmc` (one counter each, the program directly
    after `--`), bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB averaged over the fusion launches.  Driven here with a stand-in
    profiler that writes the counter table rocprofv3 writes."""
    import importlib.util
    import stat
    spec = importlib.util.spec_from_file_location('bench_for_test2', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fake = tmp_path / 'bin' / 'rocprofv3'
    fake.parent.mkdir()
    fake.write_text('''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
assert a[0] == '--pmc' and a[1] in ('FETCH_SIZE', 'WRITE_SIZE') and a[2] != '--kernel-trace', a
split = a.index('--')
prog = a[split + 1:]
assert os.path.basename(prog[0]).startswith('python') and prog[1].endswith('bench.py') and '--no-live-traffic' in prog, prog
assert not any(x in a[:split] for x in ('-s', '--sys-trace', '-r', '--runtime-trace', '--hip-trace', '--hsa-trace')), a
assert os.environ.get('SQ_BENCH_PMC_CHILD') == '1' and os.environ.get('SQ_BENCH_NO_REFERENCE_JOB') == '1'
d = os.path.join(a[a.index('-d') + 1], 'host', '123')
os.makedirs(d)
per_launch = {'FETCH_SIZE': (40.0e6, 42.0e6, 44.0e6), 'WRITE_SIZE': (83.0e6, 83.0e6, 83.0e6)}[a[1]]
with open(os.path.join(d, 'run_counter_collection.csv'), 'w') as fh:
    fh.write('"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\\n')
    fh.write(f'1,"synth_kernel<unsigned short>","{a[1]}",5.0\\n')
    for k, v in enumerate(per_launch):
        fh.write(f'{k + 2},"void (anonymous namespace)::fuse_overwrite_zg_kernel<float, true>(FuseParams, long)","{a[1]}",{v}\\n')
''')
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv('PATH', str(fake.parent) + os.pathsep + os.environ['PATH'])
    for k in [k for k in os.environ if k.startswith(('ROCPROF', 'ROCP_'))] + ['SQ_BENCH_PMC_CHILD']:
        monkeypatch.delenv(k, raising=False)
    traffic, source = bench.live_traffic(n_planes=40)
    assert traffic == (2 * 42.0e6 + 83.0e6) * 1024
    assert 'FETCH_SIZE' in source and '3 / 3 launches of 40 planes' in source
    # a process that is itself a PMC child (or being profiled) does not start passes of its own
    monkeypatch.setenv('SQ_BENCH_PMC_CHILD', '1')
    assert bench.live_traffic(n_planes=40) is None
    monkeypatch.delenv('SQ_BENCH_PMC_CHILD')
    # a failing pass leaves the caller with the committed measurement
    fake.write_text('#!/bin/sh\nexit 7\n')
    assert bench.live_traffic(n_planes=40) is None


def test_bench_live_traffic_ends_an_overrunning_pass_with_its_children(monkeypatch, tmp_path):
    """A PMC pass that overruns its time is ended as a process GROUP (profiler + the program under it), so no copy of
    the bench is left holding the GPU; the caller keeps the committed measurement."""
    import importlib.util
    import stat
    import time
    spec = importlib.util.spec_from_file_location('bench_for_test3', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pidfile = tmp_path / 'grandchild.pid'
    fake = tmp_path / 'bin' / 'rocprofv3'
    fake.parent.mkdir()
    fake.write_text(f'''#!/usr/bin/env python3
import subprocess, sys, time
child = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(60)'])
open({str(pidfile)!r}, 'w').write(str(child.pid))
time.sleep(60)
''')
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv('PATH', str(fake.parent) + os.pathsep + os.environ['PATH'])
    for k in [k for k in os.environ if k.startswith(('ROCPROF', 'ROCP_'))] + ['SQ_BENCH_PMC_CHILD']:
        monkeypatch.delenv(k, raising=False)
    t0 = time.perf_counter()
    assert bench.live_traffic(n_planes=40, timeout_s=2) is None
    assert time.perf_counter() - t0 < 20
    pid = int(pidfile.read_text())
    for _ in range(50):          # the group has been sent SIGKILL: the grandchild goes within moments
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        # a zombie still answers kill(0) until it is reaped by init: look at its state instead
        try:
            with open(f'/proc/{pid}/stat') as fh:
                if fh.read().split()[2] == 'Z':
                    break
        except FileNotFoundError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError('the program under the profiler survived the end of the pass')


def test_bench_committed_traffic_scales_with_the_planes_per_launch():
    """N > 1 lines carry the committed PMC measurement of a 10-plane launch of the headline geometry; a rank whose mean
    launch holds fewer planes (25 planes = 10 + 10 + 5) gets it scaled, and says so; unknown workloads get None."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_for_test4', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full, src = bench.committed_traffic('cfg4', 10)
    assert full and 'not measured in this run' in src and 'scaled' not in src
    part, src2 = bench.committed_traffic('cfg4', 25 / 3)
    assert abs(part - full * (25 / 3) / 10) < 1 and 'scaled to the 8.33333 planes' in src2
    assert bench.committed_traffic('cfg2', 1)[0] is None
    assert bench.committed_traffic('cfg3', 40)[0] > 1.5e11


def test_bench_line_keys_for_the_scaling_curve(monkeypatch):
    """every bench line names the strong-scaling point under ONE key pair (scale_workload / scale_value), the
    job's parallelism text follows the registration mode that ran, and a world larger than the node's GPUs stops with one line."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_for_test3', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # N = 1 region line: value is config 3, the curve's point comes from the job run afterwards on the same GPU
    region = {'value': 1577000.0, 'unit': 'Mvoxel/s', 'headline_job_on_this_gpu': {
        'workload': bench.WORKLOADS['cfg4']['desc'], 'value': 1552700.0, 'unit': 'Mvoxel/s', 'ms_per_step': 547.1, 'wall_ms_per_job': 910.0}}
    out = bench.with_scale_keys(dict(region), region['headline_job_on_this_gpu'])
    assert out['value'] == 1577000.0 and out['scale_value'] == 1552700.0 and out['scale_workload'] == bench.WORKLOADS['cfg4']['desc']
    assert out['scale_ms_per_job'] == 547.1 and out['scale_wall_ms_per_job'] == 910.0 and out['scale_unit'] == 'Mvoxel/s'
    # a job line (N ranks) is that point itself: same keys, same workload text as the N = 1 line's
    job = bench.with_scale_keys({'value': 9.9e6, 'unit': 'Mvoxel/s'}, {'workload': bench.WORKLOADS['cfg4']['desc'], 'value': 9.9e6,
                                                                       'unit': 'Mvoxel/s', 'ms_per_step': 85.0, 'wall_ms_per_job': 130.0})
    assert job['scale_value'] == job['value'] and job['scale_workload'] == out['scale_workload']
    # no job in the run: the keys are there and say null
    none = bench.with_scale_keys({'value': 1.0, 'unit': 'Mvoxel/s'}, None)
    assert none['scale_value'] is None and none['scale_workload'] is None and 'no headline job' in none['scale_note']
    # parallelism text per registration mode
    ap = bench.parallelism_text(8, 'nccl', True)
    assert 'pair table all-gathered over RCCL' in ap and 'rank 0 registers' not in ap and '8 GPUs' in ap
    cp = bench.parallelism_text(8, 'nccl', False)
    assert 'rank 0 registers' in cp and '8-int32 shift row' in cp
    assert 'gloo' in bench.parallelism_text(2, 'gloo', True) and '1 GPU ' in bench.parallelism_text(1, 'nccl', True)
    # one rank per GPU
    monkeypatch.delenv('SQ_DIST_BACKEND', raising=False)
    bench.check_device_count(8, 8)
    bench.check_device_count(1, 8)
    with pytest.raises(SystemExit) as e:
        bench.check_device_count(8, 1)
    assert '--gpus 8' in str(e.value) and '1 GPU' in str(e.value)
    monkeypatch.setenv('SQ_DIST_BACKEND', 'gloo')      # the rehearsal of N ranks on one card
    bench.check_device_count(5, 1)
