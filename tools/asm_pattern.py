"""Print the instruction-class sequence of a kernel's .s (M mfma, d ds_read, G/S global load/store, T transcendental,
v other VALU, w s_waitcnt, X scratch, | branch, s scalar), run-length encoded.  Usage: asm_pattern.py file.s [start] [len]"""
import sys
lines = open(sys.argv[1]).read().split('\n')
seq = []
for l in lines:
    l = l.strip()
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
        continue
    op = l.split()[0]
    if op.startswith('v_mfma'): c = 'M'
    elif op.startswith('ds_read'): c = 'd'
    elif op.startswith('ds_'): c = 'D'
    elif op.startswith('global_load'): c = 'G'
    elif op.startswith('global_store'): c = 'S'
    elif op.startswith(('v_exp', 'v_rcp', 'v_sqrt', 'v_rsq')): c = 'T'
    elif op.startswith('s_waitcnt'): c = 'w'
    elif op.startswith('scratch'): c = 'X'
    elif op.startswith('v_'): c = 'v'
    elif op.startswith(('s_cbranch', 's_branch')): c = '|'
    else: c = 's'
    seq.append(c)
s = ''.join(seq)
out, i = [], 0
while i < len(s):
    j = i
    while j < len(s) and s[j] == s[i]:
        j += 1
    out.append(f"{s[i]}{j - i if j - i > 1 else ''}")
    i = j
txt = ' '.join(out)
k = max(txt.find('M') - 60, 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
print(txt[k:k + n])
print({c: s.count(c) for c in 'MdDGSTwXv'})
