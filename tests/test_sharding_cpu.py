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
