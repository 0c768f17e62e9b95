"""A-B of any tune key of lspf2f_create_tuned on a whole forward (GPU): two engines on the same weights and inputs, outputs compared, time per forward of both arms
interleaved (A-B-A-B, graph replays between two events).
  python tools/ab_tune.py key=value[,key=value] [variant] [batch] [dtype] [reps] [norm: batch | instance]
e.g.  python tools/ab_tune.py out_wt=1 large 1 f32"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                        # noqa: E402
from livespeechportraits_amd import synth                            # noqa: E402
from livespeechportraits_amd.engine import Engine                    # noqa: E402
from livespeechportraits_amd.topology import build_topology          # noqa: E402


def main():
    tune = dict((k, int(v)) for k, v in (kv.split("=") for kv in sys.argv[1].split(",")))
    variant = sys.argv[2] if len(sys.argv) > 2 else "large"
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    norm = sys.argv[6] if len(sys.argv) > 6 else "batch"
    dev = torch.device("cuda:0")
    topo = build_topology(variant, norm=norm)
    sd = synth.make_state_dict(topo, 1234)
    feat, cand = synth.make_inputs(batch, 512, 99, 1)
    f, c = torch.from_numpy(feat).to(dev), torch.from_numpy(cand).to(dev)
    arms = {}
    for name, t in (("default", None), (sys.argv[1], tune)):
        e = Engine(variant, dtype=dtype, max_batch=batch, tune=t, norm=norm)
        e.load_state_dict(sd)                                       # returns the ignored num_batches_tracked keys
        e.bind(e.pack(), dev)
        arms[name] = (e, e.forward(f, c).clone())
    (na, (ea, oa)), (nb, (eb, ob)) = arms.items()
    print("%s %s batch %d: outputs bit-identical: %s (max-abs diff %.3e)" % (variant, dtype, batch, torch.equal(oa, ob), (oa.float() - ob.float()).abs().max().item()))
    changed = [(x["name"], x["kernel"], y["kernel"]) for x, y in zip(ea.layers(batch), eb.layers(batch)) if x["kernel"] != y["kernel"]]
    print("layers whose kernel name changes: %d%s" % (len(changed), "" if not changed else " (first: %s %s -> %s)" % changed[0]))
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300 if batch == 1 else 80
    for _ in range(reps):
        for name, (e, _) in arms.items():
            for _ in range(20):
                e.forward(f, c)
            torch.cuda.synchronize(); t0.record()
            for _ in range(n):
                e.forward(f, c)
            t1.record(); torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / n
            print("%-24s %.4f ms / forward  (%.1f frames/s)" % (name, ms, batch / ms * 1e3))


main()
