"""Lint for the gfx950 store-data hazard found in round 6 (tools/probe_store_war.hip, profiles/r06b_probe_store_war.log): a VALU
instruction that writes a data register of a multi-dword buffer / global store and is issued DIRECTLY behind that store can
reach memory in lanes 12..15 of every 16-lane row.  hipcc 7.2 inserts a wait state only for stores without a register soffset.
Compiles every kernel source to gfx950 assembly and reports each store whose next instruction (distance 1) or the one after
(distance 2, information only) is a VALU / LDS-return / VMEM-load write into its data registers.

    python tools/store_war_lint.py [source.hip ...]        (default: every eo_vae_amd/csrc/*.hip)"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, 'eo_vae_amd', 'csrc', '*.hip')))
STORE = re.compile(r'^\s*(buffer_store_dwordx[234]|global_store_dwordx[234]|flat_store_dwordx[234]|scratch_store_dwordx[234])\s+(.*)$')
REG = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)')


def regs_of(tok):
    m = REG.search(tok)
    if not m:
        return set()
    if m.group(1) is not None:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def store_data(op, rest):
    ops = [t.strip() for t in rest.split(',')]
    # buffer_store: vdata first; global/flat/scratch_store: vaddr, vdata
    return regs_of(ops[0] if op.startswith('buffer') else ops[1])


def scan(src):
    """[(distance, store line, next line)] for one source: data-register writes 1 or 2 instructions behind a multi-dword store."""
    asm = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', src, '-o', '-'],
                         capture_output=True, text=True).stdout
    lines = [l for l in asm.split('\n') if l.strip() and not l.strip().startswith((';', '.', '//')) and not l.strip().endswith(':')]
    hits = []
    for i, l in enumerate(lines):
        m = STORE.match(l)
        if not m:
            continue
        data = store_data(m.group(1), m.group(2))
        for dist in (1, 2):
            if i + dist >= len(lines):
                break
            nxt = lines[i + dist].strip()
            op = nxt.split()[0]
            if op.startswith(('v_', 'ds_read', 'buffer_load', 'global_load')) and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
                dst = regs_of(nxt[len(op):].split(',')[0])
                if dst & data:
                    hits.append((dist, l.strip(), nxt))
                    break
            if op.startswith(('s_nop', 's_waitcnt')) and dist == 1:
                break    # a wait state follows the store: safe at distance 1; distance 2 is then at least 2 states away
    return hits


if __name__ != '__main__':
    srcs = []
total = {1: 0, 2: 0}
for src in srcs:
    hits = scan(src)
    for h in hits:
        total[h[0]] += 1
    d1 = [h for h in hits if h[0] == 1]
    print(f'{os.path.basename(src):36s} stores with a data-register write directly behind: {len(d1):3d}   one instruction later: {len(hits) - len(d1):3d}')
    for h in d1[:6]:
        print('    ', h[1], ' || ', h[2])
if __name__ == '__main__':
    print('total distance-1 hazards:', total[1], ' distance-2:', total[2])
