"""per basic block of a kernel in a -save-temps .s file: multiply-adds, VALU instructions, scratch accesses, LDS reads
usage: python tools/dev/isa_blocks.py file.s 'k_enc<4, true, false>'"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
want = sys.argv[2]
names = re.findall(r'^\s*\.amdhsa_kernel (\S+)', s, re.M)
for n in names:
    dem = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    if want not in dem: continue
    j = s.index('\n' + n + ':')
    e = s.index('.Lfunc_end', j)
    body = s[j:e]
    blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
    print(dem)
    label = 'entry'
    for i, b in enumerate(blocks):
        if re.fullmatch(r'\.LBB\d+_\d+', b): label = b; continue
        mads = b.count('v_mad_u64_u32')
        if mads < 200: continue
        valu = len(re.findall(r'^\s+v_', b, re.M))
        scr = len(re.findall(r'scratch_(?:load|store)', b))
        lds = len(re.findall(r'^\s+ds_', b, re.M))
        sw = len(re.findall(r's_waitcnt', b)); nop = len(re.findall(r's_nop', b))
        print(f'  {label}: mads {mads} valu {valu} scratch {scr} ds {lds} waitcnt {sw} nop {nop} lines {b.count(chr(10))}')
