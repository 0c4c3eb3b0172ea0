# full GPU validation on a gpurun box:  gpurun --timeout 1200 -- 'bash tools/gpu_tests.sh'
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu.ids gpurun_out/smoke.log | tail -4
timeout -k 10 1000 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/tests_full.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/tests_full.log
