"""opcode histogram of one basic block: python tools/dev/isa_hist.py file.s kernel-substring label"""
import re, subprocess, sys, collections
s = open(sys.argv[1]).read(); want = sys.argv[2]; lab = sys.argv[3]
for n in re.findall(r'^\s*\.amdhsa_kernel (\S+)', s, re.M):
    dem = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    if want not in dem: continue
    j = s.index('\n' + n + ':'); e = s.index('.Lfunc_end', j); body = s[j:e]
    i = body.index('\n' + lab + ':'); k = body.index('\n.LBB', i + 5)
    ops = collections.Counter(m.group(1) for m in re.finditer(r'^\s+([a-z_0-9]+)', body[i:k], re.M))
    print(dem, lab, dict(ops.most_common()))
