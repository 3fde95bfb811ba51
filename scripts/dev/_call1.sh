set -x
mkdir -p gpurun_out/r5
timeout 900 python scripts/dev/pna_dma_race.py 6 > gpurun_out/r5/pna_dma_race.log 2>&1
timeout 600 python -m pytest tests/test_stress_concurrent_gpu.py -x -q > gpurun_out/r5/stress.log 2>&1
timeout 300 python bench.py > gpurun_out/r5/bench0.json 2> gpurun_out/r5/bench0.err
tail -5 gpurun_out/r5/stress.log; tail -40 gpurun_out/r5/pna_dma_race.log; cat gpurun_out/r5/bench0.json | cut -c1-600
