# debug job (GPU box; the box copy is scratch): where does a K-step of wino4_3x3 go?  Stamped builds with one ingredient removed each
set -e
O=gpurun_out/wino4_ablate; mkdir -p $O
for v in ${W4_VARIANTS:-"-DW4_ABL_NORAW+-DW4_ABL_NOU+-DW4_ABL_NOVALU" "-DW4_ABL_NORAW+-DW4_ABL_NOU+-DW4_ABL_NOVALU+-DW4_ABL_NOREAD" "-DW4_ABL_NOVALU"}; do
  f=$(echo $v | tr '+' ' ')
  make -C livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_WINO_STAMPS $f" > $O/build.log 2>&1
  echo "=== build flags: [$f]"
  for a in "64 256 1" "128 128 2"; do
    python tools/wino4_stamps.py $a 2>&1 | grep -v "amdgpu.ids\|XCD"
  done
done | tee $O/ablate2.txt
