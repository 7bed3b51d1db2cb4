#!/bin/bash
# round-3: new band / SOC-band GPU tests
mkdir -p gpurun_out/r03m
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "band or soc or su2" > gpurun_out/r03m/tests.log 2>&1
tail -5 gpurun_out/r03m/tests.log
