#!/bin/bash
# Winograd input transforms with v_pk_add_f32 (default build) against the scalar form (-DLSPF2F_NO_PK): two prebuilt libraries swapped in place, A-B-A-B
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4pk; mkdir -p $OUT
L=livespeechportraits_amd
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-8s b%s %8.1f fps %8.4f ms | %s' % ('$1', '$2', d['value'], d['ms_per_step'], ' '.join('%s x%d %.2f us f %.3f' % (k[:12], c['launches'], c['us_per_launch'], c['frac_mfma']) for k,c in pc.items() if k.startswith('wino'))))"; }
for b in 1 8; do
  for i in 1 2; do
    cp $L/_ab/liblspf2f_pk.so $L/liblspf2f.so; run pk $b
    cp $L/_ab/liblspf2f_nopk.so $L/liblspf2f.so; run scalar $b
  done
done 2>&1 | tee $OUT/ab.txt
cp $L/_ab/liblspf2f_pk.so $L/liblspf2f.so
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_network.py -m gpu -x -q -k "winograd or golden or batch8" > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/pytest.log
