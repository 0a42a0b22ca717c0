"""Static check of the gfx950 listings: an s_barrier publishes LDS writes only if the writing wave has WAITED for them
(s_waitcnt lgkmcnt(0)) -- and ROCm 7.2 was seen to omit that wait in front of a loop-top barrier reached round the back
edge straight after a ds_write (csrc/fuse.hip, for_each_queued_item).  For every s_barrier of every kernel this walks the
control-flow graph backwards (fall-through and branch edges, up to DEPTH instructions per path) and reports paths that
reach an LDS write before an lgkmcnt(0) wait.

    python tools/barrier_scan.py            (compiles every csrc/*.hip to a listing under /tmp and scans it)
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEPTH = 80
WRITE = re.compile(r'^(ds_write|ds_store|ds_add|ds_sub|ds_min|ds_max|ds_or|ds_and|ds_xor|ds_inc|ds_dec|ds_cmpst|ds_wrxchg|ds_append)')


def scan(path):
    lines = open(path).read().splitlines()
    ins, labels, kernel_of = [], {}, []
    kern = None
    for raw in lines:
        s = raw.split(';')[0].strip()
        if not s or s.startswith('.') and not s.endswith(':'):
            continue
        if s.endswith(':'):
            name = s[:-1]
            labels[name] = len(ins)
            if name.startswith('_Z'):
                kern = name
            continue
        ins.append(s)
        kernel_of.append(kern)
    preds = [[] for _ in ins]
    for i, s in enumerate(ins):
        op = s.split()[0]
        if i + 1 < len(ins) and op not in ('s_branch', 's_endpgm', 's_setpc_b64'):
            preds[i + 1].append(i)
        if op == 's_branch' or op.startswith('s_cbranch'):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] < len(ins):
                preds[labels[tgt]].append(i)
    flagged = []
    n_barriers = 0
    for b, s in enumerate(ins):
        if s != 's_barrier':
            continue
        n_barriers += 1
        seen, stack, hit = set(), [(p, 0) for p in preds[b]], None
        while stack and hit is None:
            i, d = stack.pop()
            if (i in seen) or d > DEPTH:
                continue
            seen.add(i)
            t = ins[i]
            if t.startswith('s_waitcnt') and 'lgkmcnt(0)' in t:
                continue                      # this path has waited
            if t == 's_barrier':
                continue                      # an earlier barrier: its own check covers what lies before it
            if WRITE.match(t):
                hit = (i, t)
                break
            stack.extend((p, d + 1) for p in preds[i])
        if hit:
            flagged.append((kernel_of[b], b, hit[1]))
    return n_barriers, flagged


def scan_spills(path):
    """-> {kernel: number of VGPR spill / reload instructions (scratch_store / scratch_load, buffer_store / buffer_load ... offen
    marked 'Folded Spill' / 'Folded Reload') that sit INSIDE A LOOP} for every kernel of a listing.  Spill code in the straight-line
    prologue of a kernel runs once with every lane enabled; inside the item loops it runs under the lane masks of divergent
    regions -- round 4: the uint16 + gains feather kernel with such spills (an experiment that kept 12 more values alive) left row
    ends of one-tile items unwritten in queue mode; its `count` of the chunk loop, a value every lane needs, was reloaded inside a
    region only some lanes run (profiles/r04_exp_feather_edges.log).  A loop = the instructions between a label and a LATER instruction that branches back to it."""
    kern, idx, first_back_target, labels, spills = None, 0, {}, {}, {}
    rows = []
    for raw in open(path).read().splitlines():
        code = raw.split(';')[0].strip()
        if not code or (code.startswith('.') and not code.endswith(':')):
            continue
        if code.endswith(':'):
            name = code[:-1]
            if name.startswith('_Z'):
                kern = name
            labels[name] = (kern, len(rows))
            continue
        rows.append((kern, code, ('Folded Spill' in raw) or ('Folded Reload' in raw)))
    loops = []                                                # (target, source) of every back edge: the loop's instructions
    for i, (k, code, _) in enumerate(rows):
        op = code.split()[0]
        if op == 's_branch' or op.startswith('s_cbranch'):
            tgt = labels.get(code.split()[-1])
            if tgt and tgt[0] == k and tgt[1] <= i:
                loops.append((tgt[1], i))
    depth = [0] * (len(rows) + 1)
    for t, src in loops:
        depth[t] += 1
        depth[src + 1] -= 1
    inside = 0
    for i, (k, code, spill) in enumerate(rows):
        inside += depth[i]
        if spill and inside > 0:
            spills[k] = spills.get(k, 0) + 1
    return spills


def scan_spills_all(files=('fuse.hip',)):
    """-> [(file, demangled kernel, spill instructions inside loops)] over the given csrc files, compiled here for gfx950."""
    out = tempfile.mkdtemp(prefix='sq_isa_')
    found = []
    for f in files:
        for k, n in sorted(scan_spills(listing_of(f, out)).items()):
            name = subprocess.run(['c++filt', k or '?'], capture_output=True, text=True).stdout.strip()[:120]
            found.append((f, name, n))
    return found


_LISTINGS = {}      # csrc file -> its listing, compiled once per process


def listing_of(f, out):
    if f in _LISTINGS and os.path.exists(_LISTINGS[f]):
        return _LISTINGS[f]
    lst = os.path.join(out, f + '.s')
    _LISTINGS[f] = lst
    subprocess.run(['/opt/rocm/bin/hipcc', '-std=c++17', '-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-I' + os.path.join(ROOT, 'include'),
                    '-S', '--cuda-device-only', '-o', lst, os.path.join(ROOT, 'image-stitcher_amd', 'csrc', f)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return lst


def scan_all(verbose=True):
    """-> (barriers, [(file, kernel, barrier index, the LDS write)]) over every csrc/*.hip, compiled here for gfx950."""
    from concurrent.futures import ThreadPoolExecutor
    out = tempfile.mkdtemp(prefix='sq_isa_')
    files = sorted(f for f in os.listdir(os.path.join(ROOT, 'image-stitcher_amd', 'csrc')) if f.endswith('.hip'))
    with ThreadPoolExecutor(max_workers=4) as pool:
        listings = list(pool.map(lambda f: listing_of(f, out), files))
    total, bad = 0, []
    for f, lst in zip(files, listings):
        n, flagged = scan(lst)
        total += n
        bad += [(f,) + x for x in flagged]
        if verbose:
            print(f'{f}: {n} barriers, {len(flagged)} reached by an LDS write that has not been waited for')
            for k, b, w in flagged[:20]:
                name = subprocess.run(['c++filt', k or '?'], capture_output=True, text=True).stdout.strip()[:100]
                print(f'    {name}: barrier #{b} after `{w}`')
    return total, bad


def main():
    total, bad = scan_all()
    print(f'{total} barriers, {len(bad)} flagged')
    spills = scan_spills_all(sorted(f for f in os.listdir(os.path.join(ROOT, 'image-stitcher_amd', 'csrc')) if f.endswith('.hip')))
    for f, k, n in spills:
        print(f'{f}: {k}: {n} spill / reload instructions inside loops')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
