#!/usr/bin/env python3
"""Instruction mix of a kernel's hottest loop from hipcc -S output.

usage: isa_mix.py file.s kernel_substring [--all]
Finds the kernel, splits it into basic blocks by label, prints for the largest
loop body (the block range between a label and a backward branch to it) the
count of instructions by class.  Diagnosis only."""
import re, sys, collections

def classify(op):
    if op.startswith('v_mov_b32') : return 'v_mov'
    if op.startswith(('v_cndmask',)): return 'v_cndmask'
    if op.startswith(('v_readlane','v_readfirstlane','v_writelane')): return 'v_lane'
    if op.startswith('v_cmp'): return 'v_cmp'
    if op.endswith('_f64') or '_f64_' in op: 
        if 'rcp' in op or 'div' in op or 'sqrt' in op or 'rsq' in op: return 'f64_special'
        return 'f64'
    if op.startswith('v_'): return 'v_other32'
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith('s_nop'): return 's_nop'
    if op.startswith('s_barrier'): return 's_barrier'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_','global_','flat_','scratch_')): return 'vmem'
    return 'other'

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = None
    for n, l in enumerate(lines):
        if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith(tuple('E0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ_')) and ':' in l:
            start = n; break
    if start is None: sys.exit('kernel not found')
    end = next(n for n in range(start, len(lines)) if lines[n].strip().startswith('.section') or lines[n].strip().startswith('s_endpgm'))
    # find end of function: .Lfunc_end
    end = next(n for n in range(start, len(lines)) if lines[n].startswith('.Lfunc_end'))
    body = lines[start:end]
    labels = {}
    insts = []
    for l in body:
        s = l.strip()
        if not s or s.startswith(';') or s.startswith('.') and not s.startswith('.LBB'): continue
        m = re.match(r'^(\.LBB[0-9_]+):', s)
        if m: labels[m.group(1)] = len(insts); continue
        op = s.split()[0]
        insts.append((op, s))
    # loops: backward branches
    loops = []
    for n, (op, s) in enumerate(insts):
        if op.startswith('s_cbranch') or op == 's_branch':
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= n:
                loops.append((labels[tgt], n))
    print(f'kernel has {len(insts)} instructions, {len(loops)} loops')
    for a, b in sorted(loops, key=lambda x: x[0]-x[1])[:4 if '--all' in sys.argv else 1]:
        c = collections.Counter(classify(op) for op, _ in insts[a:b+1])
        dpp = sum(1 for op, s in insts[a:b+1] if 'dpp' in s or 'row_' in s or 'wave_sh' in s)
        print(f'loop [{a},{b}] = {b-a+1} instructions; dpp-modified: {dpp}')
        for k, v in c.most_common(): print(f'   {k:12s} {v}')
        ops = collections.Counter(op for op, _ in insts[a:b+1])
        print('   top ops:', ', '.join(f'{k}:{v}' for k, v in ops.most_common(40)))
main()
