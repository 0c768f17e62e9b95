# GPU box job (round 6): a side branch of the forward's graph walks the weights of the <= 16x16 levels towards the chip while the levels above compute
# (tune key tail_prefetch: 1 plain loads, 2 non-temporal; _at = layer index the branch forks in front of; _wgs = workgroups; _mb = only the first MB of the range)
mkdir -p gpurun_out/tail_prefetch
for t in "tail_prefetch=1" "tail_prefetch=2" "tail_prefetch=1,tail_prefetch_at=10" "tail_prefetch=1,tail_prefetch_at=15" "tail_prefetch=1,tail_prefetch_at=20" "tail_prefetch=1,tail_prefetch_at=10,tail_prefetch_wgs=128" "tail_prefetch=1,tail_prefetch_at=10,tail_prefetch_wgs=8" "tail_prefetch=1,tail_prefetch_at=10,tail_prefetch_mb=128" "tail_prefetch=2,tail_prefetch_at=15,tail_prefetch_mb=128,tail_prefetch_wgs=64"; do
  python tools/ab_tune.py $t large 1 f32 2 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/tail_prefetch/ab.txt
