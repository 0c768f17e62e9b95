#!/bin/bash
# A-B-A-B of two prebuilt libraries (livespeechportraits_amd/_ab/liblspf2f_{old,new}.so) swapped in place; the new one stays.  Usage: tools/gpu_ab_libs.sh <tag> [kernel-class prefix]
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/$1; mkdir -p $OUT; PFX=${2:-wino}
L=livespeechportraits_amd
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-5s b%s %8.1f fps %8.4f ms | %s' % ('$1', '$2', d['value'], d['ms_per_step'], ' '.join('%s x%d %.2f us' % (k[:14], c['launches'], c['us_per_launch']) for k,c in pc.items() if k.startswith('$PFX'))))"; }
for b in 1 8; do for i in 1 2; do
  cp $L/_ab/liblspf2f_new.so $L/liblspf2f.so; run new $b
  cp $L/_ab/liblspf2f_old.so $L/liblspf2f.so; run old $b
done; done 2>&1 | tee $OUT/ab.txt
cp $L/_ab/liblspf2f_new.so $L/liblspf2f.so
