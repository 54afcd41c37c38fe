set -e
mkdir -p gpurun_out/r3a
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3a/pytest.log 2>&1
tail -3 gpurun_out/r3a/pytest.log
timeout -k 10 300 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
cat gpurun_out/r3a/bench.json | cut -c1-1500
timeout -k 10 400 bash tools/pmc_traffic.sh r3a_traffic > gpurun_out/r3a/traffic.txt 2>&1
tail -12 gpurun_out/r3a/traffic.txt
timeout -k 10 400 bash tools/sq_counters.sh round3a 2 ref > gpurun_out/r3a/sq.txt 2>&1
tail -5 gpurun_out/r3a/sq.txt
timeout -k 10 120 tools/micro/valu_rate > gpurun_out/r3a/valu_rate.txt 2>&1
timeout -k 10 120 tools/micro/mfma_beside_valu > gpurun_out/r3a/mfma_beside_valu.txt 2>&1
timeout -k 10 120 tools/micro/lds_rate > gpurun_out/r3a/lds_rate.txt 2>&1
timeout -k 10 120 tools/micro/vgpr_bank > gpurun_out/r3a/vgpr_bank.txt 2>&1
echo done
