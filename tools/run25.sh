for s in 1234 1235 1236 1237 1238 1239; do
echo "NEW"; timeout -k 10 120 python tools/grad_err.py $s q 2>&1 | grep "seed"
echo "OLD"; VK_HALO_ROWSTAGED=1 timeout -k 10 120 python tools/grad_err.py $s q 2>&1 | grep "seed"
echo "TAP"; VK_NO_HALO=1 timeout -k 10 120 python tools/grad_err.py $s q 2>&1 | grep "seed"
done
