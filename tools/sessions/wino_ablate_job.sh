# debug job (GPU box; the box copy is scratch): where does a K-step of the Winograd kernel go?  Stamp builds with one phase compiled out each
# (results are wrong in those builds; the step times are what is read)
mkdir -p gpurun_out/wino_ablate
for abl in "" "-DWINO_ABL_NOBARRIER" "-DWINO_ABL_NOHEAD" "-DWINO_ABL_NODMA" "-DWINO_ABL_NODMA -DWINO_ABL_NOHEAD" "-DWINO_ABL_NODMA -DWINO_ABL_NOHEAD -DWINO_ABL_NOBARRIER"; do
  rm -f livespeechportraits_amd/csrc/build/wino.o livespeechportraits_amd/csrc/build/api.o
  make -C livespeechportraits_amd/csrc -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_WINO_STAMPS $abl" > gpurun_out/wino_ablate/build.log 2>&1
  echo "=== build flags: [$abl]"
  for a in "128 128 1 1" "64 256 2 1" "512 32 1 4"; do
    python tools/wino_stamps.py $a 2>&1 | grep -E "workgroups|K loop:|no stamps"
  done
done | tee gpurun_out/wino_ablate/ablate.txt
rm -f livespeechportraits_amd/csrc/build/wino.o livespeechportraits_amd/csrc/build/api.o
make -C livespeechportraits_amd/csrc -j32 > gpurun_out/wino_ablate/rebuild.log 2>&1
