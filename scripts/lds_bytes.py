#!/usr/bin/env python3
"""Static LDS traffic of the hop loops, from the device assembly (build container, no GPU):
    python scripts/lds_bytes.py            -> profiles/lds_bytes_static.json
For each hop kernel: every DS instruction in the blocks of the hop loop (loop depth 2), bytes per lane x 64 lanes.
An UPPER bound per hop (rarely-taken blocks are counted, every lane is taken as active); bench.py multiplies it by the
measured hops/s and sets it beside the LDS peak of MI355X_MICROARCH.md (256 B/clk/CU).  The measured companion is
SQ_LDS_IDX_ACTIVE (LDS-array cycles) from the counter passes."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmc_amd import build
BYTES = {'ds_read_b32': 4, 'ds_read_b64': 8, 'ds_read_b96': 12, 'ds_read_b128': 16, 'ds_read2_b32': 8, 'ds_read2_b64': 16,
         'ds_read2st64_b32': 8, 'ds_read2st64_b64': 16, 'ds_read_u8': 1, 'ds_read_u16': 2, 'ds_read_i8': 1, 'ds_read_i16': 2,
         'ds_write_b8': 1, 'ds_write_b16': 2, 'ds_write_b32': 4, 'ds_write_b64': 8, 'ds_write_b96': 12, 'ds_write_b128': 16,
         'ds_write2_b32': 8, 'ds_write2_b64': 16, 'ds_write2st64_b32': 8, 'ds_write2st64_b64': 16,
         'ds_and_b32': 8, 'ds_or_b32': 8, 'ds_mskor_b32': 8, 'ds_add_u32': 8, 'ds_xor_b32': 8,   # read-modify-write
         'ds_bpermute_b32': 4, 'ds_permute_b32': 4, 'ds_swizzle_b32': 4}                     # crossbar only, listed apart
KERNELS = {'hop_q64_kernel': '_ZN3kmc14hop_q64_kernelILb1ELb0EEEvNS_9HopParamsE',
           'hop_q64_kernel<any>': '_ZN3kmc14hop_q64_kernelILb1ELb1EEEvNS_9HopParamsE',
           # <ANY, classes, two per CU>: 64 x 64 and 30 x 30 both take the class-mapped re-sum (lat_cls_ok)
           'hop_lat_kernel': '_ZN3kmc14hop_lat_kernelILb0ELb1ELb0EEEvNS_9LatParamsE',
           'hop_lat_kernel<any>': '_ZN3kmc14hop_lat_kernelILb1ELb1ELb0EEEvNS_9LatParamsE',
           'hop_lat_kernel<2 per CU>': '_ZN3kmc14hop_lat_kernelILb0ELb1ELb1EEEvNS_9LatParamsE',
           'hop_big_kernel<site words>': '_ZN3kmc14hop_big_kernelILb0ELi2ELb1EEEvNS_12HopBigParamsE',
           'hop_big_kernel': '_ZN3kmc14hop_big_kernelILb1ELi2ELb0EEEvNS_12HopBigParamsE'}
asm = '/tmp/kmc_lds_bytes.s'
flags = [f for f in build.FLAGS if f not in ('-fPIC', '-shared')]
subprocess.check_call([build.hipcc()] + flags + ['--cuda-device-only', '-S', '-o', asm, build.SRC], stderr=subprocess.DEVNULL)
s = open(asm).read()
out = {'source_sha256': build.source_sha256(), 'note': __doc__.split('\n\n')[0].split('\n', 3)[-1] if False else
       'per hop, upper bound: every DS instruction in the hop loop blocks x 64 lanes; crossbar ops (bpermute) apart', 'kernels': {}}
for name, sym in KERNELS.items():
    m = re.search(r'^%s:' % re.escape(sym), s, re.M)
    if not m:
        print('missing', name); continue
    body = s[m.start():s.index('s_endpgm', m.start())].split('\n')
    blocks, cur = {}, 'entry'
    for l in body:
        mm = re.match(r'^(\.LBB\d+_\d+):', l) or re.match(r'^; (%bb\.\d+):', l)
        if mm: cur = mm.group(1)
        blocks.setdefault(cur, []).append(l)
    # the hop loop = the depth-2 loop that holds a ds_bpermute (q64 / lat) or, for the large-lattice kernel, all depth >= 2 blocks
    arr = xbar = n_ds = 0
    unknown = set()
    for b, ls in blocks.items():
        if not any(re.search(r'Depth=[23]', l) for l in ls[:3]): continue
        for l in ls:
            t = l.strip().split()
            if not t or not t[0].startswith('ds_'): continue
            op = t[0]
            if op not in BYTES: unknown.add(op); continue
            n_ds += 1
            if 'permute' in op or 'swizzle' in op: xbar += BYTES[op] * 64
            else: arr += BYTES[op] * 64
    out['kernels'][name] = {'symbol': sym, 'ds_instructions_in_loop_blocks': n_ds, 'lds_array_bytes_per_hop_upper': arr,
                            'crossbar_bytes_per_hop': xbar, 'unknown_ops': sorted(unknown)}
    print(name, out['kernels'][name])
json.dump(out, open(os.path.join(ROOT, 'profiles', 'lds_bytes_static.json'), 'w'), indent=1)
