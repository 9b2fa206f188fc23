#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (scripts/pmc_config.sh) into profiles/:
   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
   <tag>_pmc.json           per-launch means of every counter for the dominant hop kernel
   pmc_current.json         what bench.py's roofline reads, one entry per BASELINE config: per-launch counters (VALU-busy
                            cycles, instruction counts, LDS-array cycles, HBM bytes, the clock during the pass), with the
                            hash of the sources the pass was taken on
usage: scripts/summarize_profile.py <tag> <config> [kernel-substring]"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cfg = sys.argv[1], str(int(sys.argv[2]))
kern = sys.argv[3] if len(sys.argv) > 3 else 'hop_q64_kernel'
src = os.path.join(ROOT, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(ROOT, 'profiles')
stats = glob.glob(os.path.join(src, 'trace', '*', '*_kernel_stats.csv'))[0]
shutil.copy(stats, os.path.join(dst, tag + '_kernel_stats.csv'))
out = {'tag': tag, 'config': int(cfg), 'kernel': kern,
       'command': 'scripts/pmc_config.sh %s %s (kernel trace: bench.py --config %s, many steps; counters: --steps 6 --warmup 2, '
                  'one rocprofv3 --pmc run per counter group)' % (cfg, tag, cfg), 'counters': {}}
for row in csv.DictReader(open(stats)):
    if kern in row['Name']:
        out['kernel_trace'] = {'name': row['Name'], 'calls': int(row['Calls']), 'avg_ms': float(row['AverageNs']) / 1e6,
                               'min_ms': float(row['MinNs']) / 1e6, 'max_ms': float(row['MaxNs']) / 1e6,
                               'share_percent': float(row['Percentage'])}
        break
dur = {}
for f in sorted(glob.glob(os.path.join(src, 'pmc_*', '*', '*_counter_collection.csv'))):
    acc = {}
    for row in csv.DictReader(open(f)):
        if kern in row['Kernel_Name']:
            acc.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
            dur.setdefault(row['Counter_Name'], []).append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
            out['dispatch'] = {k: row[k] for k in ('Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'Scratch_Size',
                                                  'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count')}
            out['dispatch']['Kernel_Name'] = row['Kernel_Name']
    for k, v in acc.items():
        out['counters'][k] = {'mean_per_launch': sum(v) / len(v), 'launches': len(v),
                              'mean_launch_ns_during_pass': sum(dur[k]) / len(dur[k])}
c = out['counters']
try:
    sha = open(os.path.join(src, 'source_sha256.txt')).read().strip()
except OSError:
    sha = None
out['source_sha256'] = sha
if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
    fetch_kb, write_kb = c['FETCH_SIZE']['mean_per_launch'], c['WRITE_SIZE']['mean_per_launch']
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-B
    # requests at 64 B, so it is doubled; WRITE_SIZE is exact.  (Narrow gathers are uncalibrated: the
    # doubled figure is the upper bound and the one reported.)
    # a run may be several concurrent launches (kmc_set_launch_groups): counters are per launch
    rpl = int(out['dispatch']['Grid_Size']) // int(out['dispatch']['Workgroup_Size'])
    ent = {'kernel': out['dispatch']['Kernel_Name'], 'tag': tag, 'config': int(cfg), 'replicas_per_launch': rpl,
           'fetch_size_kb': fetch_kb, 'write_size_kb': write_kb, 'fetch_correction': 2.0,
           'hbm_bytes_per_launch': int((2.0 * fetch_kb + write_kb) * 1024),
           'source': 'rocprofv3 --pmc, one run per counter group, over bench.py --config %s; profiles/%s_pmc.json' % (cfg, tag)}
    for k, name in (('SQ_INSTS_VALU', 'valu_insts_per_launch'), ('SQ_INSTS_SALU', 'salu_insts_per_launch'),
                    ('SQ_INSTS_LDS', 'lds_insts_per_launch'), ('SQ_INSTS_VMEM_RD', 'vmem_rd_insts_per_launch'),
                    ('SQ_INSTS_VMEM_WR', 'vmem_wr_insts_per_launch'), ('SQ_INSTS_SMEM', 'smem_insts_per_launch'),
                    ('SQ_ACTIVE_INST_VALU', 'active_inst_valu_per_launch'), ('SQ_ACTIVE_INST_ANY', 'active_inst_any_per_launch'),
                    ('SQ_WAVE_CYCLES', 'wave_cycles_per_launch'), ('SQ_WAIT_ANY', 'wait_any_per_launch'),
                    ('SQ_WAVES', 'waves_per_launch'), ('SQ_LDS_IDX_ACTIVE', 'lds_idx_active_per_launch'),
                    ('SQ_LDS_BANK_CONFLICT', 'lds_bank_conflict_per_launch'), ('TCC_HIT_sum', 'l2_hits_per_launch'),
                    ('TCC_MISS_sum', 'l2_misses_per_launch')):
        if k in c:
            ent[name] = c[k]['mean_per_launch']
    if 'SQ_LDS_BANK_CONFLICT' in c and 'SQ_LDS_IDX_ACTIVE' in c:
        ent['lds_bank_conflict_frac'] = c['SQ_LDS_BANK_CONFLICT']['mean_per_launch'] / c['SQ_LDS_IDX_ACTIVE']['mean_per_launch']
    if 'GRBM_GUI_ACTIVE' in c:
        # busy cycles of the GPU over the dispatch's own duration in the SAME pass: the shader clock under --pmc
        g = c['GRBM_GUI_ACTIVE']
        # (the counter is summed over the chip's 8 XCDs)
        ent['pmc_clock_ghz'] = g['mean_per_launch'] / 8.0 / g['mean_launch_ns_during_pass']
        ent['pmc_launch_ms'] = g['mean_launch_ns_during_pass'] / 1e6
    if 'SQ_WAVE_CYCLES' in c:
        ent['sq_pass_launch_ms'] = c['SQ_WAVE_CYCLES']['mean_launch_ns_during_pass'] / 1e6
    if 'kernel_trace' in out:
        ent['kernel_avg_ms'] = out['kernel_trace']['avg_ms']
    ent['source_sha256'] = sha
    cur_path = os.path.join(dst, 'pmc_current.json')
    try:
        cur = json.load(open(cur_path))
        if 'configs' not in cur:
            cur = {'configs': {}}
    except (OSError, ValueError):
        cur = {'configs': {}}
    cur['configs'][cfg] = ent
    json.dump(cur, open(cur_path, 'w'), indent=1)
    out['entry'] = ent
json.dump(out, open(os.path.join(dst, tag + '_pmc.json'), 'w'), indent=1)
print(json.dumps(out.get('entry', out), indent=1))
