#!/bin/bash
mkdir -p gpurun_out/r3
timeout -k 10 300 python tools/chunk_probe.py > gpurun_out/r3/exp_registration_chunks.log 2>&1; echo rc $?; cat gpurun_out/r3/exp_registration_chunks.log
