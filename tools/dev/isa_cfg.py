"""control-flow listing of a kernel's basic blocks in a label range: python tools/dev/isa_cfg.py file.s 'k_enc<4, true, false>' lo hi"""
import re, subprocess, sys
s = open(sys.argv[1]).read(); want = sys.argv[2]; lo, hi = int(sys.argv[3]), int(sys.argv[4])
for n in re.findall(r'^\s*\.amdhsa_kernel (\S+)', s, re.M):
    dem = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    if want not in dem: continue
    j = s.index('\n' + n + ':'); e = s.index('.Lfunc_end', j); body = s[j:e]
    label = 'entry'
    for b in re.split(r'\n(\.LBB\d+_\d+):', body):
        if re.fullmatch(r'\.LBB\d+_\d+', b): label = b; continue
        num = int(label.split('_')[1]) if label != 'entry' else -1
        if lo <= num <= hi:
            print(label, 'lines', b.count('\n'), 'mads', b.count('v_mad_u64_u32'), 'scr', len(re.findall(r'scratch_(?:load|store)', b)),
                  'glob', len(re.findall(r'global_(?:load|store)', b)), 'ds', len(re.findall(r'^\s+ds_', b, re.M)),
                  re.findall(r'^\s+(s_cbranch\S+|s_branch)\s+(\S+)', b, re.M))
