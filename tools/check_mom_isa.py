"""ISA check of the LDS-DMA momentum kernel's level loop (CPU only: hipcc -S, no GPU).

The loop keeps one or two levels of DMA in flight and counts them itself (wait_vmcnt); a compiler-inserted
`s_waitcnt vmcnt(..)` inside the loop drains that pipeline (profiles/r02_notes.md).  For every instance of
k_momentum_dma this prints, inside the innermost loop: the vmcnt waits that are not the kernel's own switch
ladder, scratch accesses, and writes of M0 outside the DMA sequence.

usage: python tools/check_mom_isa.py [extra hipcc flags, e.g. -DDYCORE_FLOAT32]      exit code 1 on a finding in ANY
instance the build contains (the shipped library holds only instances it can launch; an experiment build,
-DDYCORE_AB_SWITCHES, also the older tile shapes and data flows)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'climate_model_amd', 'csrc')


def main():
    out = os.path.join(tempfile.mkdtemp(), 'dycore.s')
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-Wno-unused-function', '-Wno-inline-asm', '-Wno-pass-failed', '--offload-arch=gfx950',
           '--cuda-device-only', '-S', '-o', out, 'dycore_hip.hip'] + sys.argv[1:]
    r = subprocess.run(cmd, cwd=SRC, capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-3000:]); return 2
    L = open(out).read().split('\n')
    starts = [i for i, l in enumerate(L) if re.match(r'^_ZN3dyc14k_momentum_dmaI\w+:', l)]
    bad = 0
    for st in starts:
        name = L[st].split(':')[0]
        en = next(i for i in range(st, len(L)) if L[i].strip().startswith('.Lfunc_end'))
        body = L[st:en]
        m = re.search(r'k_momentum_dmaILb([01])ELi(\d+)ELb([01])ELb([01])ELi(\d)ELb([01])ELb([01])ELb([01])E', name)
        cur, tj, fw, turb, depth, pl, wt, mo = m.groups()
        # the level loop: the loop (lines grouped by their header label) that contains the DMA instructions
        groups = {}
        for i, l in enumerate(body):
            mm = re.search(r'Loop: Header=(BB\d+_\d+)', l) or re.search(r'^\.L(BB\d+_\d+):.*This Inner Loop Header', l)
            if mm: groups.setdefault(mm.group(1), []).append(i)
        spans = []
        for hdr, idx in groups.items():
            a = min(idx)
            # the loop's last block ends with its back edge (the code behind that branch, up to the next label, is the
            # epilogue of the sweep -- its loads of the level above a chunk are not in the loop)
            last = max(idx)
            b = next((i + 1 for i in range(last + 1, len(body)) if re.match(r'\s*s_c?branch', body[i]) or re.match(r'^\.LBB\d+_\d+:', body[i])), len(body))
            if any('global_load_lds' in l for l in body[a:b]): spans.append((a, b))
        # innermost DMA loops only (a WITH_TEMP instance has two: the momentum and the temperature wavefronts')
        spans = [sp for sp in spans if not any(o != sp and sp[0] <= o[0] and o[1] <= sp[1] for o in spans)]
        found = []
        ndma = 0
        for lo, hi in spans:
            for i in range(lo, hi):
                l = body[i].strip()
                if l.startswith('s_waitcnt') and 'vmcnt' in l:
                    # the kernel's own waits sit between ASMSTART / ASMEND markers
                    own = any('ASMSTART' in body[j] for j in range(max(lo, i - 3), i))
                    if not own: found.append('line %d: %s' % (i, l))
                if re.match(r'scratch_|buffer_(load|store)', l): found.append('line %d: %s (spill)' % (i, l))
            ndma += sum('global_load_lds' in l for l in body[lo:hi])
        lo, hi = (min(a for a, b in spans), max(b for a, b in spans)) if spans else (0, 0)
        tag = 'predictor' if cur == '1' else 'corrector'
        prod = True
        print('%-9s TJ=%s fw=%s turb=%s depth=%s phi_local=%s with_temp=%s moist=%s: %d level loop(s), %d DMA instructions, findings: %s' % (
            tag, tj, fw, turb, depth, pl, wt, mo, len(spans), ndma, found or 'none'))
        if found and prod: bad = 1
    return bad


if __name__ == '__main__':
    sys.exit(main())
