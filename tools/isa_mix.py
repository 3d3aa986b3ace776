"""Developer tool: instruction mix of one kernel in a device assembly file (hipcc -S --cuda-device-only)."""
import re
import sys
from collections import Counter
T = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^(_Z\w*%s\w*):' % name, T, re.M)
a = m.start()
b = T.index('.Lfunc_end', a)
ins = [l.strip() for l in T[a:b].splitlines() if re.match(r'\s+[a-z_0-9]+(\s|$)', l) and not l.strip().startswith(('.', ';'))]
print('instructions', len(ins))
for k, v in Counter(l.split()[0] for l in ins).most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 16):
    print(k, v)
sym = m.group(1)
print(re.findall(re.escape(sym) + r'\.(?:num_vgpr|num_agpr), \d+', T))
