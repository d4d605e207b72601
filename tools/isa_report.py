"""Per-kernel resource / instruction report from a hipcc -save-temps device assembly file.
usage: python tools/isa_report.py build/gemm-hip-amdgcn-amd-amdhsa-gfx950.s [name filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
# function bodies: from "<name>:" label to its .end_amdhsa_kernel's preceding s_endpgm section
meta = {}
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    body = m.group(2)
    g = lambda k: re.search(r'\.amdhsa_' + k + r' (\S+)', body).group(1)
    meta[m.group(1)] = dict(vgpr=g('next_free_vgpr'), sgpr=g('next_free_sgpr'), scratch=g('private_segment_fixed_size'),
                            lds=g('group_segment_fixed_size'))
for name, md in meta.items():
    if flt not in name:
        continue
    i = s.find('\n' + name + ':')
    j = s.find('.end_amdhsa_kernel', i)
    code = s[i:j]
    cnt = lambda pat: len(re.findall(pat, code))
    vm0 = cnt(r's_waitcnt vmcnt\(0\)')
    print(f"{name[:110]}\n   vgpr {md['vgpr']} sgpr {md['sgpr']} scratch {md['scratch']}  mfma16 {cnt(r'v_mfma_f32_16x16x32')} mfma32 {cnt(r'v_mfma_f32_32x32x16')} "
          f"glds {cnt(r'global_load_lds')} ds_read_b128 {cnt(r'ds_read_b128')} ds_write_b32 {cnt(r'ds_write_b32')} flat {cnt(r'flat_')} "
          f"scratch_ops {cnt(r'scratch_')} waitcnt_vm0 {vm0} s_load {cnt(r's_load_')}")
