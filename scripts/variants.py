"""Build variant libraries (extra -D flags) and time them back to back in one GPU call.
    python scripts/variants.py --build name1:-DFOO name2:-DBAR,-DBAZ base:     (CPU box)
    python scripts/variants.py --run name1 name2 base                          (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == '--build':
    from kmc_amd import build
    for spec in sys.argv[2:]:
        name, _, fl = spec.partition(':')
        out = os.path.join(ROOT, 'kmc_amd', 'libkmc_hip_v_%s.so' % name)
        subprocess.check_call([build.hipcc()] + build.FLAGS + [f for f in fl.split(',') if f] + ['-o', out, build.SRC],
                              stderr=subprocess.DEVNULL)
        print(out)
else:
    # KMC_VAR_R: replica counts to time (default 4096: four waves per SIMD; 256 = one wave per CU, alone on its SIMD)
    for R in [int(x) for x in os.environ.get('KMC_VAR_R', '4096').split(',')]:
        for rep in range(2):
            for name in sys.argv[2:]:
                env = dict(os.environ, KMC_AMD_LIB=os.path.join(ROOT, 'kmc_amd', 'libkmc_hip_v_%s.so' % name))
                o = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'time_calls.py'), str(R)], env=env,
                                   capture_output=True, text=True, timeout=120).stdout.strip().split('\n')
                ks = [float(l.split('kernel')[1].split()[0]) for l in o if 'kernel' in l]
                print('R %5d %-12s presim %.3f  sim %.3f ms (min of 3)' % (R, name, min(ks[0::2]), min(ks[1::2])), flush=True)
