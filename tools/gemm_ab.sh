#!/bin/bash
# A/B of the GEMM row-tile choice on the step's shapes (one process per arm: the choice is read once): tools/gemm_ab.sh
for tm in 256 224 0; do
  echo "== BSG_GEMM_TM=$tm (0 = automatic choice)"
  BSG_GEMM_TM=$tm python tools/gemm_shapes.py 100352,1024,1024 100352,1024,4096 100352,1024,3072 100352,3072,1024 100352,4096,1024 200704,1024,1024 200704,3072,1024 || exit 1
done
