#!/usr/bin/env python3
"""Print the op-class sequence of the biggest basic block of a kernel in a hipcc -S listing.
usage: isa_seq.py file.s start_line end_line"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')[int(sys.argv[2]):int(sys.argv[3])]
blocks, cur = [], []
for l in lines:
    if re.match(r'^\.LBB', l):
        blocks.append(cur); cur = []
    cur.append(l)
blocks.append(cur)
big = max(blocks, key=len)
fast = {'v_add_u32', 'v_sub_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_mov_b32', 'v_lshrrev_b32', 'v_ashrrev_i32',
        'v_subrev_u32', 'v_cndmask_b32', 'v_not_b32'}
seq, cnt = '', {}
for l in big:
    t = l.strip().split()
    if not t or t[0].startswith(('.', ';')): continue
    op = t[0]; cnt[op] = cnt.get(op, 0) + 1
    if op.startswith('v_'): seq += 'a' if op.replace('_e32','').replace('_e64','') in fast else 'P'
    elif op.startswith('ds_'): seq += 'L'
    elif op.startswith('s_waitcnt'): seq += 'w'
    elif op.startswith('s_'): seq += '.'
    elif op.startswith(('global', 'buffer')): seq += 'G'
    else: seq += '?'
print(len(seq)); print(seq)
print(sorted(cnt.items(), key=lambda x: -x[1])[:40])
