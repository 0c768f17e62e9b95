#!/bin/bash
# Ablation switches of the igemm kernel on one layer shape (results are WRONG by construction for dbg != 0).
# The shipped library has the hooks compiled out: this script rebuilds liblspf2f.so with -DLSPF2F_ABLATE first and
# restores the product build afterwards.
# bits: 1 no refetch, 4 no barrier, 8 no buffer flip, 16 no epilogue, 32 no main loop
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
make -C "$here/livespeechportraits_amd/csrc" clean >/dev/null
make -C "$here/livespeechportraits_amd/csrc" -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_ABLATE" >/dev/null
for d in ${DBG_LIST:-0 16 32 48 13 29}; do
  echo "dbg=$d"; LSP_HIP_DBG=$d python "$here/tools/tune_conv.py" --only "$1" 2>/dev/null | grep -E " 64x64  split 1  g1| 128x64  split 1  g1| 64x64  split 4  g1" | head -2
done
make -C "$here/livespeechportraits_amd/csrc" clean >/dev/null
make -C "$here/livespeechportraits_amd/csrc" -j8 >/dev/null
