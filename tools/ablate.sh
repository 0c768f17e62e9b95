#!/bin/bash
# ablation of the igemm kernel on one layer shape (results are WRONG by construction when dbg != 0)
# bits: 1 no refetch, 2 no LDS restage, 4 no barrier, 8 no buffer flip, 16 no epilogue, 32 no main loop
for d in ${DBG_LIST:-0 16 32 48 15 31}; do
  echo "dbg=$d"; LSP_HIP_DBG=$d python tools/tune_conv.py --only "$1" 2>/dev/null | grep -E " 64x64  split 1  g1| 128x64  split 1  g1| 64x64  split 4  g1" | head -2
done
