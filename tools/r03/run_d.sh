#!/bin/bash
# plane strides with "random-looking" low bits (multiples of 256): do they avoid the collision as a rule?
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp_plane_stride_hashed.log
: > $O
for s in 0 3635456 1782016 8023296; do for d in 0 2890496 6199040 1289984; do
  echo "== src_pad dst_pad = $s $d" >> $O
  timeout -k 10 120 tools/membw_gains 3 $s $d 1 >> $O 2>&1 || { echo failed; tail -3 $O; exit 1; }
done; done
grep -E "^==|^A " $O | sed -E 's/ +[0-9.]+ ms +[0-9.]+ GB\/s//; s/== A bit for bit//; s/regs: gains \+ reciprocals in VGPRs, 5 planes per thread \(shipped structure\)//' | paste - - - 
