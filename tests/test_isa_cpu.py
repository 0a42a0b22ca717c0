"""Static check of the compiled kernels (no GPU needed: hipcc cross-compiles gfx950 listings here)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_barrier_publishes_lds_writes_that_were_not_waited_for():
    """An s_barrier orders only LDS writes that have completed: the writing wave needs an s_waitcnt lgkmcnt(0) between its
    ds_write and the barrier.  ROCm 7.2 left that wait out in front of a loop-top barrier reached round the back edge
    (round 3: the device work queues lost work in 1-2 % of small launches, csrc/fuse.hip for_each_queued_item); the
    listing of every kernel is walked backwards from every barrier along all control-flow edges (tools/barrier_scan.py)."""
    spec = importlib.util.spec_from_file_location('barrier_scan', os.path.join(ROOT, 'tools', 'barrier_scan.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    total, bad = mod.scan_all(verbose=False)
    assert total > 400, total            # fuse.hip alone has 60+, register.hip 300+
    assert not bad, bad[:10]
