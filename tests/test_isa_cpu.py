"""Static check of the compiled kernels (no GPU needed: hipcc cross-compiles gfx950 listings here)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SCAN = None


def _scan_module():
    global _SCAN
    if _SCAN is None:      # one module object: its listings are compiled once for both tests
        spec = importlib.util.spec_from_file_location('barrier_scan', os.path.join(ROOT, 'tools', 'barrier_scan.py'))
        _SCAN = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_SCAN)
    return _SCAN


def test_no_barrier_publishes_lds_writes_that_were_not_waited_for():
    """An s_barrier orders only LDS writes that have completed: the writing wave needs an s_waitcnt lgkmcnt(0) between its
    ds_write and the barrier.  ROCm 7.2 left that wait out in front of a loop-top barrier reached round the back edge
    (round 3: the device work queues lost work in 1-2 % of small launches, csrc/fuse.hip for_each_queued_item); the
    listing of every kernel is walked backwards from every barrier along all control-flow edges (tools/barrier_scan.py)."""
    mod = _scan_module()
    total, bad = mod.scan_all(verbose=False)
    assert total > 400, total            # fuse.hip alone has 60+, register.hip 300+
    assert not bad, bad[:10]


def test_no_spill_code_inside_the_kernels_loops():
    """VGPR spill / reload instructions may sit in a kernel's straight-line prologue (every lane enabled, run once), not inside
    its loops, where they run under the lane masks of divergent regions: an experiment build of the uint16 + gains feather
    kernel with such spills left voxels unwritten in queue mode (round 4, profiles/r04_exp_feather_edges.log -- a value every
    lane needs was reloaded inside a region only some lanes run).  The one exception is listed with its count: two reloads at
    the top of the queue walk of fuse_feather_zg_kernel<1, true, unsigned short>, used inside the region that reloads them.
    A change of code or toolchain that adds spill code to a loop has to be looked at (and parity re-run) before this list grows."""
    mod = _scan_module()
    files = sorted(f for f in os.listdir(os.path.join(ROOT, 'image-stitcher_amd', 'csrc')) if f.endswith('.hip'))
    found = mod.scan_spills_all(files)
    allowed = {'fuse_feather_zg_kernel<1, true, unsigned short>': 2}
    over = [(f, k, n) for f, k, n in found if n > max([v for name, v in allowed.items() if name in k] or [0])]
    assert not over, over


def test_spill_scan_tells_prologue_from_loop(tmp_path):
    """The scan itself, on a hand-written listing: spill code before the first loop and behind the last one is not counted, spill
    code between a label and a later branch back to it is -- nested loops once."""
    lst = tmp_path / 'k.s'
    lst.write_text('''
_Z1kv:
	scratch_store_dword off, v1, off ; 4-byte Folded Spill
.LBB0_1:
	v_add_u32_e32 v0, 1, v0
	scratch_load_dword v1, off, off ; 4-byte Folded Reload
.LBB0_2:
	scratch_store_dword off, v2, off offset:4 ; 4-byte Folded Spill
	s_cbranch_vccnz .LBB0_2
	s_cbranch_scc1 .LBB0_1
	scratch_load_dword v2, off, off offset:4 ; 4-byte Folded Reload
	s_endpgm
_Z2k2v:
	scratch_store_dword off, v1, off ; 4-byte Folded Spill
	s_cbranch_scc1 .LBB1_9
	scratch_load_dword v1, off, off ; 4-byte Folded Reload
.LBB1_9:
	s_endpgm
''')
    assert _scan_module().scan_spills(str(lst)) == {'_Z1kv': 2}
