for i in 1 2; do
echo SIDE; VK_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 2>&1 | grep '^{"metric"' | cut -c1-150
echo NOSIDE; VK_NO_SIDE_STREAM=1 VK_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 2>&1 | grep '^{"metric"' | cut -c1-150
done
