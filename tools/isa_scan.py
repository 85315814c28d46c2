"""Static screen of the kernels' ISA for what a profiler does not show directly: per kernel, the MFMA count, the register-to-register
moves (v_mov / v_accvgpr) inside the basic blocks that hold MFMAs, and scratch instructions.  Found round 3's measurement-hook bug in
csrc/conv2d_mfma.hip (61 moves per 24 MFMAs in the kernel-row loop).
    python3 tools/isa_scan.py [file.hip ...]        (default: every csrc/*.hip; needs hipcc, no GPU)
Kernels are listed when moves > MFMAs / 2 or any scratch instruction exists; the block-by-block histogram of one kernel:
    python3 tools/isa_scan.py --blocks <mangled-name-prefix> file.hip"""
import collections
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'percivaltts_amd', 'csrc')


def assemble(src):
    out = os.path.join(tempfile.gettempdir(), 'isa_' + os.path.basename(src) + '.s')
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off',
                    '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-S', '--cuda-device-only', '-o', out, src],
                   check=True, stderr=subprocess.DEVNULL)
    return out


def kernels(asm):
    cur, blk, stats = None, None, collections.OrderedDict()
    for l in open(asm):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur, blk = m.group(1), 'entry'
            stats[cur] = collections.OrderedDict()
            continue
        if cur is None:
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blk = m.group(1)
            continue
        if l.strip().startswith('s_endpgm'):
            cur = None
            continue
        m = re.match(r'^\s+((?:v_|s_|ds_|global_|buffer_|scratch_)\w+)', l)
        if m:
            stats[cur].setdefault(blk, collections.Counter())[m.group(1)] += 1
    return stats


def main():
    args = sys.argv[1:]
    blocks = None
    if args and args[0] == '--blocks':
        blocks, args = args[1], args[2:]
    files = args or sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    for f in files:
        st = kernels(assemble(f))
        for k, b in st.items():
            mf = sum(v for c in b.values() for n, v in c.items() if n.startswith('v_mfma'))
            if blocks is not None:
                if not k.startswith(blocks):
                    continue
                print(k)
                for name, c in b.items():
                    t = sum(c.values())
                    if t >= 40:
                        print('  %-12s total %4d  mfma %3d  valu %4d  salu %3d  lds %3d  vmem %3d  top %s' % (
                            name, t, sum(v for n, v in c.items() if 'mfma' in n),
                            sum(v for n, v in c.items() if n.startswith('v_') and 'mfma' not in n),
                            sum(v for n, v in c.items() if n.startswith('s_')), sum(v for n, v in c.items() if n.startswith('ds_')),
                            sum(v for n, v in c.items() if n.startswith(('global_', 'buffer_'))), c.most_common(4)))
                continue
            if mf == 0:
                continue
            mv = sum(c['v_mov_b32_e32'] + 2 * c['v_mov_b64_e32'] + c['v_accvgpr_write_b32'] + c['v_accvgpr_read_b32']
                     for c in b.values() if any(n.startswith('v_mfma') for n in c))
            scr = sum(v for c in b.values() for n, v in c.items() if n.startswith('scratch_'))
            if mv > 0.5 * mf or scr > 0:
                print('%-18s %-70s mfma %4d  moves in MFMA blocks %4d  scratch %3d' % (os.path.basename(f), k[:70], mf, mv, scr))


if __name__ == '__main__':
    main()
