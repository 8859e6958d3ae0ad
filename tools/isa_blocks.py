"""Dev tool: per-basic-block instruction mix of one kernel from the -save-temps assembly (make -C software-rasterizer_amd asm).
usage: python tools/isa_blocks.py <mangled-substring> [--dump LABEL]"""
import re, sys
s = open(__file__.rsplit('/tools/', 1)[0] + '/software-rasterizer_amd/build/srz_kernels-hip-amdgcn-amd-amdhsa-gfx950.s').read()
key = sys.argv[1]
m = re.search(r'^(_Z\w*' + re.escape(key) + r'\w*):', s, re.M)
i = m.start(); j = s.index('.Lfunc_end', i)
blocks = []; cur = ('entry', [], '')
for l in s[i:j].split('\n'):
    t = l.strip()
    if not t or t.startswith(';'): continue
    mm = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', t)
    if mm:
        blocks.append(cur); cur = (mm.group(1), [], mm.group(2) or '')
    elif not t.startswith('.') and not t.endswith(':'):
        cur[1].append(t)
blocks.append(cur)
if '--dump' in sys.argv:
    lab = sys.argv[sys.argv.index('--dump') + 1]
    for n, ins, c in blocks:
        if n == lab: print('\n'.join(ins))
    sys.exit()
for n, ins, c in blocks:
    v = sum(1 for x in ins if x.startswith('v_')); sa = sum(1 for x in ins if x.startswith('s_'))
    ds = sum(1 for x in ins if x.startswith('ds_')); g = sum(1 for x in ins if x.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))
    br = [x.split()[0].replace('s_cbranch_', '').replace('s_branch', 'br') + '>' + x.split()[-1].split('_')[-1] for x in ins if 'branch' in x]
    print(f"{n.split('_')[-1]:>6s} n={len(ins):4d} valu={v:4d} salu={sa:4d} lds={ds:2d} vmem={g:2d} {' '.join(br):30s} {c[:60]}")
