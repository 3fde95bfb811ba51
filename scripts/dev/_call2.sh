mkdir -p gpurun_out/r5
FLOWGNN_LIB=scripts/dev/_pnamid.so timeout 600 python scripts/dev/pna_dma_race.py 12 > gpurun_out/r5/pna_dma_race_mid.log 2>&1
for i in 1 2 3 4 5 6; do timeout 300 python scripts/dev/with_lib.py scripts/dev/_pnamid.so tests/test_multi_device_gpu.py -q -k "bit_identical and PNA" 2>&1 | tail -2; done > gpurun_out/r5/pnamid_test.log 2>&1
timeout 300 python scripts/dev/with_lib.py scripts/dev/_pnamid.so tests/test_stress_concurrent_gpu.py -q -k "PNA" 2>&1 | tail -3 >> gpurun_out/r5/pnamid_test.log
timeout 300 python scripts/dev/threads_probe.py > gpurun_out/r5/threads_probe.log 2>&1
cat gpurun_out/r5/pna_dma_race_mid.log gpurun_out/r5/pnamid_test.log; tail -5 gpurun_out/r5/threads_probe.log
