mkdir -p gpurun_out
timeout -k 10 300 python tools/debug_layers.py --n 2 --size 64 --dtype f32 --train > gpurun_out/dbg_train_f32.log 2>&1
echo "debug train rc=$?"; grep -c "<<<<<" gpurun_out/dbg_train_f32.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/model.log 2>&1
echo "model rc=$?"; tail -15 gpurun_out/model.log
