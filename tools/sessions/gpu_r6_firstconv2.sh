# GPU box job (round 6): first conv at one frame, kernel durations of the ablation arms from rocprofv3 --kernel-trace
# (-DLSPF2F_ABLATE build; bits: 1 no MFMAs, 2 no stores, 4 no window copies, 8 no weight copies)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/firstconv
make -C $R/livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_ABLATE" > $R/gpurun_out/firstconv/build.log 2>&1
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 3 12 13 14 15; do
  rm -rf /tmp/fc_$d
  LSP_HIP_DBG=$d rocprofv3 --kernel-trace --stats -d /tmp/fc_$d -o t -- python $R/tools/first_conv_time.py large 1 f32 > /tmp/fc_$d.log 2>&1
  db=$(find /tmp/fc_$d -name "*_results.db" | head -1)
  echo -n "dbg=$d  "; python $R/tools/rocprof_summary.py $db | grep first_conv
done | tee $R/gpurun_out/firstconv/ablate_rocprof.txt
