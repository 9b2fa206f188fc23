for v in ${VARIANTS:-base prio base prio}; do
  KMC_AMD_LIB=kmc_amd/libkmc_hip_v_$v.so timeout -k 10 200 python bench.py --config 3 --steps 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '%.4g'%d['value'], d['ms_per_step'])"
done
