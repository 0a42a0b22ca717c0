#!/bin/bash
# r04 e: the structural alternatives of tools/membw_gains (A regs / B gains in LDS, a wave per plane / C LDS-DMA / P plain copy)
# once more, now with the canvas in a MIXED arena: round 3 found them neutral while the memory placement was the limiter
O=gpurun_out/r4; mkdir -p $O
MEMBW_MIXED=image-stitcher_amd/csrc/libsquidstitch.so timeout -k 10 400 tools/membw_gains 5 > $O/structures_mixed.log 2>&1 || { echo failed; tail -20 $O/structures_mixed.log; exit 1; }
timeout -k 10 400 tools/membw_gains 5 > $O/structures_plain.log 2>&1 || { echo failed; tail -20 $O/structures_plain.log; exit 1; }
echo "== mixed"; cat $O/structures_mixed.log; echo "== plain"; cat $O/structures_plain.log
