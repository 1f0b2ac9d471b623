"""Per-kernel instruction census of the largest basic block (the main loop) of a hipcc -save-temps .s file — lab aid."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\w+):', s, re.M)]
for i, (pos, name) in enumerate(starts):
    if pat not in name:
        continue
    body = s[pos:starts[i + 1][0] if i + 1 < len(starts) else len(s)]
    body = body.split('.end_amdhsa_kernel')[0] if '.amdhsa_kernel' in body else body
    body = body.split('\n.Lfunc_end')[0]
    blocks = re.split(r'\n(?=\.LBB\d+_\d+:)', body)
    big = max(blocks, key=len)
    lines = [l.strip() for l in big.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    c = Counter(l.split()[0] for l in lines)
    print(name[:90], '| blocks', len(blocks), '| main block instructions', len(lines))
    keys = ('v_mfma_f32_16x16x32_bf16', 'ds_read_b128', 'ds_write_b128', 'global_load_dwordx4', 'global_load_lds_dwordx4', 'scratch_load_dwordx4',
            'scratch_store_dwordx4', 'scratch_load_dwordx2', 'scratch_store_dwordx2', 'scratch_load_dword', 'scratch_store_dword', 'v_accvgpr_read_b32', 'v_accvgpr_write_b32', 's_waitcnt', 's_barrier',
            'v_mov_b32', 's_nop', 'v_add_u32', 'v_lshl_add_u64', 'v_add_co_u32')
    print('   ', {k: c[k] for k in keys if c.get(k)})
    print('    waitcnts:', Counter(l for l in lines if l.startswith('s_waitcnt')).most_common(14))
