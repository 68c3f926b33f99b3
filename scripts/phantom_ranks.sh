show() { tail -1 $1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); print(sys.argv[1], l['value'], l['ms_per_step'], l['roofline']['frac'], l['hbm_peak_gb'], l['config']['parallelism'][-90:])" $1; }
timeout 600 python bench.py --nodes 100000000 --steps 8 --warmup 4 --no-cpu-baseline --phantom-world 8 > gpurun_out/phantom3_100m_8.log 2>&1; show gpurun_out/phantom3_100m_8.log
timeout 600 python bench.py --nodes 100000000 --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/phantom3_100m_1.log 2>&1; show gpurun_out/phantom3_100m_1.log
timeout 600 python bench.py --nodes 100000000 --steps 8 --warmup 4 --no-cpu-baseline --round-walks 2097152 > gpurun_out/phantom3_100m_1_r21.log 2>&1; show gpurun_out/phantom3_100m_1_r21.log
timeout 600 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/phantom3_10m_1.log 2>&1; show gpurun_out/phantom3_10m_1.log
timeout 600 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --phantom-world 8 > gpurun_out/phantom3_10m_8.log 2>&1; show gpurun_out/phantom3_10m_8.log
