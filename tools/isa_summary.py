#!/usr/bin/env python3
"""tools/isa_summary.py [ngw_kernels.s] [name filter ...] - per-kernel figures of the device ISA (`make -C gym_novel_gridworlds_amd/csrc asm`):
VGPRs / AGPRs / SGPRs allocated, SGPR spills, scratch bytes, LDS, static instruction count, and how many v_writelane / v_readlane /
scratch_ / s_swappc instructions the body holds (and, with --loop, how many sit inside its largest backward-branch loop).
The figures quoted in DESIGN.md / profiles/ come from this script."""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    path = args[0] if args and args[0].endswith('.s') else 'gym_novel_gridworlds_amd/csrc/ngw_kernels.s'
    filters = [a for a in args if not a.endswith('.s')]
    want_loop = '--loop' in sys.argv
    text = open(path).read().split('\n')
    # function bodies: "name:" ... ".Lfunc_endN:"
    bodies, cur, name = {}, None, None
    for ln in text:
        m = re.match(r'^(_Z[\w$.]+):\s*(;.*)?$', ln)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.startswith('.Lfunc_end'):
                bodies[name] = cur
                cur = None
            else:
                cur.append(ln)
    meta = {}
    for m in re.finditer(r'\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel', '\n'.join(text), re.S):
        d = dict(re.findall(r'\.amdhsa_(\w+) (\S+)', m.group(2)))
        meta[m.group(1)] = d
    # the YAML metadata holds the spill counts
    y = {}
    for m in re.finditer(r'- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?'
                         r'\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', '\n'.join(text), re.S):
        y[m.group(2)] = dict(agpr=int(m.group(1)), scratch=int(m.group(3)), sgpr=int(m.group(4)), sgpr_spill=int(m.group(5)), vgpr=int(m.group(6)),
                             vgpr_spill=int(m.group(7)))
    dm = demangle(list(y))
    rows = []
    for k, v in y.items():
        nm = dm.get(k, k)
        if filters and not any(f in nm for f in filters):
            continue
        body = bodies.get(k, [])
        ins = [b.strip() for b in body if b.startswith('\t') and not b.strip().startswith(('.', ';')) and b.strip()]
        cnt = lambda p: sum(1 for i in ins if i.startswith(p))
        row = dict(name=re.sub(r'\(anonymous namespace\)::', '', nm).split('(')[0], **v, insts=len(ins), writelane=cnt('v_writelane'), readlane=cnt('v_readlane'),
                   scratch_ops=cnt('scratch_'), calls=cnt('s_swappc'), lds=int(meta.get(k, {}).get('group_segment_fixed_size', 0)))
        if want_loop:
            # largest loop = the backward branch spanning the most instructions
            labels, pos = {}, 0
            for b in body:
                s = b.strip()
                lm = re.match(r'^(\.LBB\d+_\d+):', s)
                if lm:
                    labels[lm.group(1)] = pos
                elif b.startswith('\t') and s and not s.startswith(('.', ';')):
                    pos += 1
            best, pos = (0, 0, 0), 0
            for b in body:
                s = b.strip()
                if b.startswith('\t') and s and not s.startswith(('.', ';')):
                    bm = re.match(r'^s_cbranch_\w+ (\.LBB\d+_\d+)|^s_branch (\.LBB\d+_\d+)', s)
                    if bm:
                        t = labels.get(bm.group(1) or bm.group(2))
                        if t is not None and t <= pos and pos - t > best[0]:
                            best = (pos - t, t, pos)
                    pos += 1
            lo, hi = best[1], best[2]
            loop = ins[lo:hi + 1]
            row.update(loop_insts=len(loop), loop_readlane=sum(1 for i in loop if i.startswith('v_readlane')),
                       loop_writelane=sum(1 for i in loop if i.startswith('v_writelane')), loop_scratch=sum(1 for i in loop if i.startswith('scratch_')),
                       loop_calls=sum(1 for i in loop if i.startswith('s_swappc')))
        rows.append(row)
    rows.sort(key=lambda r: r['name'])
    keys = ['vgpr', 'agpr', 'sgpr', 'sgpr_spill', 'vgpr_spill', 'scratch', 'insts', 'writelane', 'readlane', 'scratch_ops', 'calls']
    if want_loop:
        keys += ['loop_insts', 'loop_readlane', 'loop_writelane', 'loop_scratch', 'loop_calls']
    print('| kernel | ' + ' | '.join(keys) + ' |')
    print('|---|' + '---|' * len(keys))
    for r in rows:
        print('| `%s` | ' % r['name'] + ' | '.join(str(r[k]) for k in keys) + ' |')


if __name__ == '__main__':
    main()
