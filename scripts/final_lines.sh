set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout -k 10 200 python bench.py > $O/r3m_bench.json 2>$O/r3m_bench.err
timeout -k 10 200 python bench.py --steps 20 > $O/r3m_bench_s20.json 2>>$O/r3m_bench.err
timeout -k 10 200 python bench.py --config 1 > $O/r3m_bench_c1.json 2>>$O/r3m_bench.err
timeout -k 10 200 python bench.py --config 2 > $O/r3m_bench_c2.json 2>>$O/r3m_bench.err
timeout -k 10 200 python bench.py --config 3 > $O/r3m_bench_c3.json 2>>$O/r3m_bench.err
timeout -k 10 300 python bench.py --config 4 > $O/r3m_bench_c4.json 2>>$O/r3m_bench.err
timeout -k 10 200 python bench.py --replicas 512 > $O/r3m_bench_512.json 2>>$O/r3m_bench.err
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r3m_smoke.log 2>&1
timeout -k 10 100 python scripts/time_rates_full.py > $O/r3m_rates_full.txt 2>&1
echo all done
