set -x
python -m pytest tests -q -m gpu 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 > gpurun_out/r06g_gpu_pytest.txt
bash tools/profile_round.sh r06g > /dev/null 2>&1
timeout 300 python tools/input_cost.py > gpurun_out/r06g_input_cost.txt 2>&1
timeout 1500 python tools/adversarial_parity.py 32 64 100 127 128 256 512 > gpurun_out/r06g_adversarial_parity.txt 2>&1; echo "rc $?" >> gpurun_out/r06g_adversarial_parity.txt
timeout 600 python tools/fuzz_parity.py 600 5 > gpurun_out/r06g_fuzz_parity.txt 2>&1; echo "rc $?" >> gpurun_out/r06g_fuzz_parity.txt
timeout 300 python tools/split_fold_census.py > gpurun_out/r06g_split_fold_census.txt 2>&1
timeout 400 python tools/configs_bench.py > gpurun_out/r06g_configs.json 2> gpurun_out/r06g_configs.err
timeout 300 python tools/batch_sweep.py > gpurun_out/r06g_batch_sweep.txt 2>&1
timeout 600 python tools/share_curve.py 1 3 8 16 32 2>&1 | grep -v amdgpu > gpurun_out/r06g_share_curve.txt
for k in pcg zeros; do timeout 120 python tools/clock_watch.py heart_sounds_segmentation_amd/libhssfsst.so $k 3 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/r06g_power.txt; done
timeout 300 python tools/flag_stress.py 30000 8 2>&1 | grep -v amdgpu | tail -3 > gpurun_out/r06g_flag_stress.txt
timeout 200 python tools/call_breakdown.py 2>&1 | grep -v amdgpu > gpurun_out/r06g_call_breakdown.txt
mkdir -p devlibs; [ -x devlibs/sync_latency ] || hipcc --offload-arch=gfx950 -O2 tools/sync_latency.hip -o devlibs/sync_latency > /dev/null 2>&1; ./devlibs/sync_latency > gpurun_out/r06g_sync_latency.txt 2>&1
cat gpurun_out/r06g_gpu_pytest.txt
tail -3 gpurun_out/r06g_adversarial_parity.txt; tail -3 gpurun_out/r06g_fuzz_parity.txt; tail -3 gpurun_out/r06g_split_fold_census.txt
