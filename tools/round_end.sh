set -x
python -m pytest tests -q -m gpu 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 > gpurun_out/r06f_gpu_pytest.txt
bash tools/profile_round.sh r06f > /dev/null 2>&1
timeout 300 python tools/input_cost.py > gpurun_out/r06f_input_cost.txt 2>&1
timeout 1500 python tools/adversarial_parity.py 32 64 100 127 128 256 512 > gpurun_out/r06f_adversarial_parity.txt 2>&1; echo "rc $?" >> gpurun_out/r06f_adversarial_parity.txt
timeout 600 python tools/fuzz_parity.py 600 5 > gpurun_out/r06f_fuzz_parity.txt 2>&1; echo "rc $?" >> gpurun_out/r06f_fuzz_parity.txt
timeout 300 python tools/split_fold_census.py > gpurun_out/r06f_split_fold_census.txt 2>&1
timeout 400 python tools/configs_bench.py > gpurun_out/r06f_configs.json 2> gpurun_out/r06f_configs.err
timeout 300 python tools/batch_sweep.py > gpurun_out/r06f_batch_sweep.txt 2>&1
timeout 600 python tools/share_curve.py 1 3 8 16 32 2>&1 | grep -v amdgpu > gpurun_out/r06f_share_curve.txt
for k in pcg zeros; do timeout 120 python tools/clock_watch.py heart_sounds_segmentation_amd/libhssfsst.so $k 3 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/r06f_power.txt; done
cat gpurun_out/r06f_gpu_pytest.txt
tail -3 gpurun_out/r06f_adversarial_parity.txt; tail -3 gpurun_out/r06f_fuzz_parity.txt; tail -3 gpurun_out/r06f_split_fold_census.txt
