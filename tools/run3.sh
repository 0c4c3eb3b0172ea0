mkdir -p gpurun_out
timeout -k 10 300 python tools/debug_layers.py --n 4 --size 128 --dtype bf16 --train > gpurun_out/dbg_train_bf16.log 2>&1
echo "rc=$?"
