# debug job: rebuild the library with igemm phase stamps on the GPU box (the box copy is scratch) and print them for a few layer shapes
set -e
mkdir -p gpurun_out/stamps
make -C livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_IGEMM_STAMPS" > gpurun_out/stamps/build.log 2>&1
rm -f gpurun_out/stamps/igemm.txt
for a in "64 0 64 256 0 64 64" "128 0 128 128 0 64 64"; do
  python tools/time_conv.py $a >> gpurun_out/stamps/igemm.txt 2>&1
done
grep -v "^   -" gpurun_out/stamps/igemm.txt | grep -v XCD
