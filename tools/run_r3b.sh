mkdir -p gpurun_out/r3b
timeout -k 10 300 python -u -m pytest tests/test_gpu_pipeline.py -k random_configurations -x -q --timeout 180 --timeout-method thread > gpurun_out/r3b/pytest.log 2>&1
grep -E "mismatch|passed|failed" gpurun_out/r3b/pytest.log | cut -c1-600
timeout -k 10 300 python bench.py > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err || exit 1
cut -c1-2500 gpurun_out/r3b/bench.json
timeout -k 10 120 tools/micro/valu_rate > gpurun_out/r3b/valu_rate.txt 2>&1 || exit 1
timeout -k 10 120 tools/micro/mfma_beside_valu > gpurun_out/r3b/mfma_beside_valu.txt 2>&1 || exit 1
timeout -k 10 120 tools/micro/lds_rate > gpurun_out/r3b/lds_rate.txt 2>&1 || exit 1
timeout -k 10 120 tools/micro/vgpr_bank > gpurun_out/r3b/vgpr_bank.txt 2>&1 || exit 1
timeout -k 10 400 bash tools/sq_counters.sh round3a 2 ref > gpurun_out/r3b/sq.txt 2>&1 || exit 1
tail -5 gpurun_out/r3b/sq.txt
timeout -k 10 400 bash tools/pmc_traffic.sh r3b_traffic > gpurun_out/r3b/traffic.txt 2>&1 || exit 1
tail -12 gpurun_out/r3b/traffic.txt
echo done
