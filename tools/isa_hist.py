"""Instruction histogram of one kernel from a -save-temps gfx950 .s file."""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
s = open(path).read()
for f in re.split(r'\n\s*\.globl\s+', s)[1:]:
    name = f.split('\n', 1)[0].strip()
    if pat not in name:
        continue
    body = f.split('.end_amdhsa_kernel')[0] if '.end_amdhsa_kernel' in f else f
    body = body.split('s_endpgm')[0] if False else body
    ins = [l.strip().split()[0] for l in body.split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    g = collections.Counter()
    for k, v in c.items():
        grp = ('valu_pk' if k.startswith('v_pk_') else 'valu_trans' if k.split('_e')[0] in ('v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_exp_f32', 'v_log_f32') else 'valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else 'vmem' if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'lds' if k.startswith('ds_') else 'other')
        g[grp] += v
    print(name[:90], 'total', len(ins), dict(g))
    for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
        print(f'  {k:28s}{v}')
