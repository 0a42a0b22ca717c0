#!/bin/bash
mkdir -p gpurun_out/r3
L=image-stitcher_amd/csrc/libsquidstitch_regexp.so
{
python tools/reg_time.py
for t in 128 256 512 1024; do SQ_LIB_PATH=$L SQ_REG_FWD_THREADS=$t timeout -k 10 120 python tools/reg_time.py || exit 1; done
for t in 128 256 512; do SQ_LIB_PATH=$L SQ_REG_INV_THREADS=$t timeout -k 10 120 python tools/reg_time.py || exit 1; done
for rl in 1 2 4; do for t in 256 512; do SQ_LIB_PATH=$L SQ_REG_RLF=$rl SQ_REG_FWD_THREADS=$t timeout -k 10 120 python tools/reg_time.py || exit 1; done; done
python tools/reg_time.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r3/exp_registration_threads.log
cat gpurun_out/r3/exp_registration_threads.log | cut -c1-200
