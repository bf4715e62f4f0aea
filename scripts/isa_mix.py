"""Static instruction mix of every kernel in csrc/rt_kernel.s (make asm): vector / scalar / branch / wait / LDS / memory (development aid)."""
import re, collections, sys, os
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tilecoderaytracer_amd", "csrc", "rt_kernel.s")
lines = open(path).read().splitlines()
name, c = None, None
def kind(op):
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_cbranch') or op in ('s_branch', 's_setpc_b64'): return 'branch'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith(('s_load', 's_buffer')): return 'smem'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'flat_', 'scratch_', 'buffer_')): return 'vmem'
    return 'other'
for line in lines:
    m = re.match(r'^(rt_render_kernel\w*):', line)
    if m:
        name, c = m.group(1), collections.Counter()
        continue
    if name and line.startswith('.Lfunc_end'):
        tot = sum(c.values())
        print(f"{name:34s} " + "  ".join(f"{k} {c[k]}" for k in ('valu', 'salu', 'branch', 'waitcnt', 'nop', 'smem', 'lds', 'vmem')) + f"  total {tot}")
        if len(sys.argv) > 1 and sys.argv[1] == name:
            ops = collections.Counter(o for o in ops_list if o.startswith('s_'))
            for o, n in ops.most_common(25): print(f"      {o:24s} {n}")
        name = None
        continue
    if name:
        s = line.strip()
        if not s or s.startswith(('.', ';', '//')) or s.endswith(':'): continue
        op = s.split()[0]
        c[kind(op)] += 1
        if not c.get('_init'):
            ops_list = []; c['_init'] = 0
        ops_list.append(op)
