#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (scripts/profile_bench.sh) into profiles/:
   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
   <tag>_pmc.json           per-launch means of every counter for the dominant hop kernel
   pmc_current.json         what bench.py's roofline reads: VALU-busy cycles, HBM bytes, instruction counts per launch,
                            with the hash of the sources the pass was taken on
usage: scripts/summarize_profile.py <tag> [kernel-substring]"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else 'hop_q64_kernel'
src = os.path.join(ROOT, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(ROOT, 'profiles')
stats = glob.glob(os.path.join(src, 'trace', '*', '*_kernel_stats.csv'))[0]
shutil.copy(stats, os.path.join(dst, tag + '_kernel_stats.csv'))
out = {'tag': tag, 'kernel': kern, 'command': 'kernel trace: python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline; '
       'counters: python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline', 'counters': {}}
for row in csv.DictReader(open(stats)):
    if kern in row['Name']:
        out['kernel_trace'] = {'calls': int(row['Calls']), 'avg_ms': float(row['AverageNs']) / 1e6,
                               'min_ms': float(row['MinNs']) / 1e6, 'max_ms': float(row['MaxNs']) / 1e6,
                               'share_percent': float(row['Percentage'])}
for f in sorted(glob.glob(os.path.join(src, 'pmc_*', '*', '*_counter_collection.csv'))):
    acc = {}
    for row in csv.DictReader(open(f)):
        if kern in row['Kernel_Name']:
            acc.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
            out['dispatch'] = {k: row[k] for k in ('Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'Scratch_Size',
                                                  'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count')}
    for k, v in acc.items():
        out['counters'][k] = {'mean_per_launch': sum(v) / len(v), 'launches': len(v)}
c = out['counters']
if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
    fetch_kb, write_kb = c['FETCH_SIZE']['mean_per_launch'], c['WRITE_SIZE']['mean_per_launch']
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-B
    # requests at 64 B, so it is doubled; WRITE_SIZE is exact.  (Narrow gathers are uncalibrated: the
    # doubled figure is the upper bound and the one reported.)
    # a run of 4096 replicas may be several concurrent launches (kmc_set_launch_groups): counters are per launch
    rpl = int(out['dispatch']['Grid_Size']) // int(out['dispatch']['Workgroup_Size'])
    hbm = {'kernel': kern, 'tag': tag, 'replicas': 4096, 'replicas_per_launch': rpl, 'hops_per_launch': 2048,
           'fetch_size_kb': fetch_kb, 'write_size_kb': write_kb,
           'fetch_correction': 2.0, 'hbm_bytes_per_launch': int((2.0 * fetch_kb + write_kb) * 1024),
           'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py; '
                     'profiles/%s_pmc.json' % tag}
    for k, name in (('SQ_INSTS_VALU', 'valu_insts_per_launch'), ('SQ_INSTS_SALU', 'salu_insts_per_launch')):
        if k in c:
            hbm[name] = c[k]['mean_per_launch']              # bench.py's valu_issue figure
    try:
        hbm['source_sha256'] = open(os.path.join(src, 'source_sha256.txt')).read().strip()
    except OSError:
        hbm['source_sha256'] = None
    out['source_sha256'] = hbm['source_sha256']
    if 'SQ_ACTIVE_INST_VALU' in c:
        hbm['active_inst_valu_per_launch'] = c['SQ_ACTIVE_INST_VALU']['mean_per_launch']       # quad-cycles, all waves
        if 'SQ_WAVE_CYCLES' in c:
            # fraction of a SIMD's cycles with the VALU busy while its 4 waves are resident
            hbm['valu_busy_while_resident'] = 4 * c['SQ_ACTIVE_INST_VALU']['mean_per_launch'] / c['SQ_WAVE_CYCLES']['mean_per_launch']
    if 'SQ_LDS_BANK_CONFLICT' in c and 'SQ_LDS_IDX_ACTIVE' in c:
        hbm['lds_bank_conflict_frac'] = c['SQ_LDS_BANK_CONFLICT']['mean_per_launch'] / c['SQ_LDS_IDX_ACTIVE']['mean_per_launch']
    if 'kernel_trace' in out:
        hbm['kernel_avg_ms'] = out['kernel_trace']['avg_ms']
    json.dump(hbm, open(os.path.join(dst, 'pmc_current.json'), 'w'), indent=1)
    out['hbm'] = hbm
json.dump(out, open(os.path.join(dst, tag + '_pmc.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
