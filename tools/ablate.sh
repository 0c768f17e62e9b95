#!/bin/bash
# ablation switches of the igemm kernel on one layer shape (results are WRONG by construction for dbg != 0)
# bits: 1 no refetch, 4 no barrier, 8 no buffer flip, 16 no epilogue, 32 no main loop
for d in ${DBG_LIST:-0 16 32 48 13 29}; do
  echo "dbg=$d"; LSP_HIP_DBG=$d python tools/tune_conv.py --only "$1" 2>/dev/null | grep -E " 64x64  split 1  g1| 128x64  split 1  g1| 64x64  split 4  g1" | head -2
done
