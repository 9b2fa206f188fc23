"""Instruction statistics of a kernel's hop loop from the device assembly (CPU box, no GPU).
    python scripts/asm_loop.py hop_q64_kernel [out.s]
Compiles kmc_hip.hip to assembly, takes the innermost loop that contains a ds_bpermute and prints its blocks."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmc_amd import build
kern = sys.argv[1] if len(sys.argv) > 1 else 'hop_q64_kernel'
out = '/tmp/kmc_asm.s'
flags = [f for f in build.FLAGS if f not in ('-fPIC', '-shared')]
subprocess.check_call([build.hipcc()] + flags + ['--cuda-device-only', '-S', '-o', out, build.SRC] + sys.argv[3:],
                      stderr=subprocess.DEVNULL)
s = open(out).read()
m = re.search(r'^(_ZN3kmc\d+%s\w*):' % kern, s, re.M)
i = m.start()
body = s[i:s.index('s_endpgm', i)]
blocks, order, cur = {}, [], 'entry'
for l in body.split('\n'):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm: cur = mm.group(1)
    if cur not in blocks: blocks[cur] = []; order.append(cur)
    blocks[cur].append(l)
DEPTH = os.environ.get('DEPTH', '2')
depth2 = [b for b in order if any('Depth=' + DEPTH in l for l in blocks[b][:3])]
tot = {'valu': 0, 'salu': 0, 'lds': 0, 'vmem': 0, 'scratch': 0}
text = []
for b in depth2:
    c = {'valu': 0, 'salu': 0, 'lds': 0, 'vmem': 0, 'scratch': 0}
    for l in blocks[b]:
        t = l.strip()
        if t.startswith('v_'): c['valu'] += 1
        elif t.startswith('ds_'): c['lds'] += 1
        elif t.startswith('global_') or t.startswith('flat_'): c['vmem'] += 1
        elif t.startswith('scratch_'): c['scratch'] += 1
        elif t.startswith('s_') and not t.startswith(('s_waitcnt', 's_nop')): c['salu'] += 1
    for k in tot: tot[k] += c[k]
    print(b, c)
    text += blocks[b]
print('all depth-2 blocks (rare paths included):', tot)
if len(sys.argv) > 2 and sys.argv[2] != '-':
    open(sys.argv[2], 'w').write('\n'.join(text))
