for v in ${VARIANTS:-base norot slow base norot slow}; do
  KMC_AMD_LIB=kmc_amd/libkmc_hip_v_$v.so timeout -k 10 200 python bench.py --steps 320 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '%.4g'%d['value'], d['ms_per_step'], (d.get('per_step_launch_rank0') or {}).get('hops_per_s_this_rank'))"
done
