mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
VK_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 2>&1 | grep '^{"metric"' | cut -c1-150
timeout -k 10 1000 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/tests_full.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/tests_full.log
