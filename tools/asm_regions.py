"""Static instruction counts of a kernel between the PTK_MARK comments of a -DPTK_MARKERS build.
Usage: asm_regions.py <file.s> <kernel symbol substring>.  Regions are taken in the order of the assembly text, so a
region's count includes whatever blocks the compiler laid out behind its marker (good enough to see where the code is)."""
import re, sys, collections
fast = {'v_fma_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_fmac_f32', 'v_mov_b32', 'v_and_b32', 'v_add_u32', 'v_fmaak_f32', 'v_fmamk_f32'}
slow8 = {'v_rcp_f32', 'v_sqrt_f32', 'v_rsq_f32', 'v_exp_f32', 'v_log_f32', 'v_sin_f32', 'v_cos_f32', 'v_rcp_iflag_f32'}
inside = False; region = 'entry'
stats = collections.OrderedDict()
for ln in open(sys.argv[1]):
    s = ln.strip()
    if not inside:
        if re.match(r'^_Z\S*' + re.escape(sys.argv[2]) + r'\S*:', s): inside = True
        continue
    if s.startswith('.Lfunc_end'): break
    m = re.search(r'PTK_MARK (\S+)', s)
    if m: region = m.group(1); continue
    if not s or s[0] in ';.' or s.endswith(':'): continue
    op = s.split()[0]
    d = stats.setdefault(region, collections.Counter())
    base = op.replace('_e32', '').replace('_e64', '')
    if op.startswith('v_'):
        d['valu'] += 1
        d['cycles'] += 2.5 if base in fast else (8.3 if base in slow8 else 4.3)
        if base == 'v_mov_b32': d['mov'] += 1
        if op.startswith('v_readlane') or op.startswith('v_writelane'): d['lane_spill'] += 1
    elif op.startswith('s_'): d['salu'] += 1
    elif op.startswith('ds_'): d['lds'] += 1
    elif op.startswith('scratch_'): d['scratch'] += 1
    elif op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_'): d['vmem'] += 1
print('%-22s %6s %7s %5s %6s %5s %4s %5s %7s' % ('region', 'valu', 'cycles', 'mov', 'lanesp', 'salu', 'lds', 'vmem', 'scratch'))
for r, d in stats.items():
    print('%-22s %6d %7.0f %5d %6d %5d %4d %5d %7d' % (r, d['valu'], d['cycles'], d['mov'], d['lane_spill'], d['salu'], d['lds'], d['vmem'], d['scratch']))
