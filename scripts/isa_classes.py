"""Static count of 'half-rate' vector instructions per kernel in csrc/rt_kernel.s (make asm): v_min/v_max/v_min3/v_max3/v_med3, v_cmp*, v_cndmask,
DPP forms, and any VALU op with a scalar-register operand issue at ~0.97 per CU and cycle where add/mul/fma issue at ~1.75
(scripts/ubench/issue_rate.hip).  Development aid."""
import re, collections, os
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tilecoderaytracer_amd", "csrc", "rt_kernel.s")
name, c = None, None
for line in open(path):
    m = re.match(r'^(rt_render_kernel\w*):', line)
    if m:
        name, c = m.group(1), collections.Counter(); continue
    if name and line.startswith('.Lfunc_end'):
        v = c['valu']
        print(f"{name:34s} valu {v:5d}: min/max {c['minmax']:4d}  cmp {c['cmp']:4d}  cndmask {c['cnd']:4d}  dpp {c['dpp']:4d}  readlane {c['rl']:4d}  "
              f"other with scalar operand {c['sop']:4d}  -> half-rate {c['slow']:5d} ({c['slow'] / max(v, 1):.2f})")
        name = None; continue
    if not name: continue
    s = line.strip()
    if not s.startswith('v_'): continue
    op = s.split()[0]
    c['valu'] += 1
    slow = True
    if re.match(r'v_(min|max|med)3?_', op): c['minmax'] += 1
    elif op.startswith('v_cmp'): c['cmp'] += 1
    elif op.startswith('v_cndmask'): c['cnd'] += 1
    elif '_dpp' in op or 'quad_perm' in s or 'row_' in s: c['dpp'] += 1
    elif op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane')): c['rl'] += 1
    elif re.search(r'(?<![a-z_])s\d+|s\[\d+:\d+\]|vcc|exec', s.split(None, 1)[1] if ' ' in s else ''): c['sop'] += 1
    else: slow = False
    if slow: c['slow'] += 1
