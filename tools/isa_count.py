"""Static instruction-class counts of a physics_kernel ISA listing (hipcc -save-temps .s):  python tools/isa_count.py FILE.s [FILE2.s]"""
import sys, collections
def count(path):
    c = collections.Counter()
    for ln in open(path):
        t = ln.strip()
        if not t or t[0] in ';.' or t.endswith(':') or not (t.startswith('v_') or t.startswith('s_') or t.startswith('ds_') or t.startswith('global_') or t.startswith('scratch_') or t.startswith('buffer_')):
            continue
        m = t.split()[0]
        if m.startswith('v_accvgpr'): k = 'accvgpr'
        elif m.startswith('v_pk'): k = 'v_pk'
        elif 'dpp' in t: k = 'dpp'
        elif m.startswith('v_'): k = 'valu'
        elif m.startswith('s_cbranch') or m.startswith('s_branch'): k = 'branch'
        elif m.startswith('s_waitcnt'): k = 'waitcnt'
        elif m.startswith('s_nop'): k = 'nop'
        elif m.startswith('s_'): k = 'salu'
        elif m.startswith('ds_'): k = 'lds'
        else: k = 'vmem'
        c[k] += 1
    return c
for p in sys.argv[1:]:
    c = count(p); print(p, sum(c.values()), dict(sorted(c.items())))
