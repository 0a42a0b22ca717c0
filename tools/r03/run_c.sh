#!/bin/bash
# finer plane-stride sweep (membw_gains quick mode): which paddings of the tile / canvas plane strides avoid the collision?
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp_plane_stride_fine.log
: > $O
for pads in "0 0" "0 128" "0 384" "0 512" "0 768" "0 1024" "0 1280" "0 1536" "0 2048" "0 4096" "0 4352" "0 8448" \
            "128 0" "512 0" "1024 0" "2048 0" "4096 0" "4352 0" \
            "256 256" "256 512" "512 256" "256 768" "128 128" "1280 256" "256 1280" "768 256" "256 65792"; do
  echo "== src_pad dst_pad = $pads" >> $O
  timeout -k 10 120 tools/membw_gains 3 $pads 1 >> $O 2>&1 || { echo failed; tail -3 $O; exit 1; }
done
grep -E "^==|^A " $O | sed -E 's/ +[0-9.]+ ms +[0-9.]+ GB\/s//; s/== A bit for bit//; s/regs: gains \+ reciprocals in VGPRs, 5 planes per thread \(shipped structure\)//' | paste - - - 
