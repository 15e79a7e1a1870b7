mkdir -p gpurun_out/fz
fail=0
for seed in ${FUZZ_SEEDS:-31 32 33 34}; do
  timeout -k 10 300 python tools/fuzz_all.py $seed 300 > gpurun_out/fz/all_$seed.txt 2>&1 || { echo "fuzz_all $seed FAIL"; fail=1; }
  tail -1 gpurun_out/fz/all_$seed.txt | cut -c1-250
  timeout -k 10 300 python tools/fuzz_all.py $seed 300 wide > gpurun_out/fz/wide_$seed.txt 2>&1 || { echo "fuzz_all wide $seed FAIL"; fail=1; }
  tail -1 gpurun_out/fz/wide_$seed.txt | cut -c1-250
  timeout -k 10 300 python tools/fuzz_all.py $seed 300 k1024 > gpurun_out/fz/k1024_$seed.txt 2>&1 || { echo "fuzz_all k1024 $seed FAIL"; fail=1; }
  tail -1 gpurun_out/fz/k1024_$seed.txt | cut -c1-250
  timeout -k 10 300 python tools/fuzz_api.py $seed 100 > gpurun_out/fz/api_$seed.txt 2>&1 || { echo "fuzz_api $seed FAIL"; fail=1; }
  tail -1 gpurun_out/fz/api_$seed.txt
  timeout -k 10 300 python tools/fuzz_calls.py $seed 60 > gpurun_out/fz/calls_$seed.txt 2>&1 || { echo "fuzz_calls $seed FAIL"; fail=1; }
  tail -1 gpurun_out/fz/calls_$seed.txt
done
timeout -k 10 400 python tools/fuzz_all.py 35 40 big > gpurun_out/fz/big_35.txt 2>&1 || { echo "fuzz_all big FAIL"; fail=1; }
tail -1 gpurun_out/fz/big_35.txt | cut -c1-250
exit $fail
