#!/usr/bin/env python3
"""bench.py -- 512x512 frames/sec of the Feature2FaceGenerator forward on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or started bare,
in which case this process becomes the launcher: it starts N rank processes of itself (one per GPU, backend nccl = RCCL), waits
for them and exits with their status.  Fewer than N devices is an error, not a silent 1-GPU run (LSP_DIST_BACKEND=gloo lets
ranks share devices: control-flow tests on a 1-GPU box only).

A step = one pass of the hot path (lspf2f_forward) over one batch of synthetic frames per GPU,
inputs already resident in HBM.  Default workload = BASELINE.json configs[1]: May ('large'),
batch 1, fp32.  Weak scaling: per-GPU work is fixed, value = frames all ranks rendered / time.
Rank 0 prints ONE JSON line (+ roofline and cpu_baseline objects).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402
import torch.distributed as dist   # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (not the 2:1-sparse marketing figure)
PEAK_HBM_GBS = 8000.0


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: become the launcher.  One child per rank with the torchrun environment
    contract (rendezvous on 127.0.0.1, a free port), children inherit stdout (rank 0 prints the JSON line); the first failing
    child ends the job.  Replaces nothing in the reference (its nn.DataParallel needs no launcher: models/networks.py:392-401)."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    gloo = os.environ.get("LSP_DIST_BACKEND") == "gloo"
    if ndev < 1:
        sys.exit("bench.py: no ROCm device visible (there is no CPU path)")
    if ndev < n and not gloo:
        sys.exit("bench.py: --gpus %d but only %d device(s) visible; RCCL needs one device per rank "
                 "(LSP_DIST_BACKEND=gloo shares devices, for control-flow tests only)" % (n, ndev))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts of this pool only support dmabuf IPC; without it RCCL's hipIpcGetMemHandle fails with "invalid
        # argument" as soon as two ranks exchange buffer handles.  The build / GPU images export it already; a child launched from a shell that
        # dropped it would only fail at the first collective, so it is pinned here (a platform setting, not a tuning switch).
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), LOCAL_RANK=str(r % ndev), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", LSP_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, alive = 0, list(procs)
    while alive and rc == 0:
        time.sleep(0.2)
        for p in list(alive):
            if p.poll() is not None:
                alive.remove(p)
                rc = rc or p.returncode
    for p in alive:          # a rank failed: stop the others (exact PIDs we started)
        p.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    sys.exit(rc)


RESULT_PRINTED = False      # rank 0 sets it once its JSON line is on stdout


def leave_group():
    """Tear the process group down without ever holding the job: the other ranks leave as soon as their part is done (rank 0 goes on alone with the roofline and CPU
    legs for a minute), and a backend teardown that waits for a peer that has already gone must not keep a finished measurement from returning -- a watchdog ends the
    process if destroy_process_group() has not returned after 30 s.  A forced exit is never silent: it is reported on stderr, and its status is 0 only when this rank's
    timed region is complete (every rank reaches leave_group() behind the closing barrier of the timed steps; rank 0 may still owe its JSON line, which it prints
    AFTER leaving the group) -- a rank that has to be forced out while a peer crashed is still told apart from a clean run by the message and by the peer's own status."""
    import threading

    def forced():
        sys.stderr.write("bench.py: rank %s: destroy_process_group() did not return within 30 s -- forced exit (timed region complete)\n" % os.environ.get("RANK", "0"))
        sys.stderr.flush()
        os._exit(0 if (RESULT_PRINTED or int(os.environ.get("RANK", "0")) != 0) else 3)
    t = threading.Timer(30.0, forced)
    t.daemon = True
    t.start()
    try:
        dist.destroy_process_group()
    finally:
        t.cancel()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--variant", default="large", choices=["large", "normal"])
    ap.add_argument("--batch", type=int, default=1, help="frames per GPU per step")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"],
                    help="f32 = the parity configuration (default); bf16 = BASELINE.json configs[2] storage path; f16 = the reference's opt.fp16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--layers", default=None, help="write the per-layer timing table to this file")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a.gpus)          # does not return

    import ctypes
    from livespeechportraits_amd import _native as N
    from livespeechportraits_amd import distributed as D
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    from livespeechportraits_amd.topology import build_topology

    if D.env_rank()[1] != a.gpus:          # checked before the rendezvous, which would wait for ranks that never come
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher and the argument disagree" % (a.gpus, D.env_rank()[1]))
    rank, world, local = D.init_process_group()
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path)"
    dev = torch.device("cuda:%d" % local)
    torch.cuda.set_device(dev)

    topo = build_topology(a.variant, size=a.size)
    B = a.batch
    eng = Engine(a.variant, size=a.size, max_batch=max(B, B if a.no_extra else 8), dtype=a.dtype)
    peak = PEAK_F32_MFMA_TFLOPS if a.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    sd = synth.make_state_dict(topo, 1234) if rank == 0 else None
    D.setup_engine(eng, sd, dev)          # pack on rank 0, ONE RCCL broadcast, bind everywhere

    # BASELINE.json configs[3]: distinct feature maps per rank and frame, ONE candidate stack shared by every rank -- rank 0 makes
    # it, one broadcast (RCCL) hands it to the others, like the weights
    feat_np, cand_np = synth.make_inputs(B, a.size, seed=99 + 1000 * rank, cand_batch=1)
    feat = torch.from_numpy(feat_np).to(dev)
    if rank != 0:
        cand_np = None
    cand = D.broadcast_tensor(None if cand_np is None else torch.from_numpy(cand_np), (1, 12, a.size, a.size), torch.float32, dev)
    out = torch.empty((B, 3, a.size, a.size), device=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        eng.forward(feat, cand, out)
    torch.cuda.synchronize()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        eng.forward(feat, cand, out)
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    # The shader clock the chip holds under this load, measured OUTSIDE the timed region (rank 0): the same loop once more while one wave on a side
    # stream counts shader cycles against the constant 100 MHz counter (lspf2f_clock_probe) for 80 % of the time the timed loop just took.
    clock_ghz = None
    if rank == 0:
        probe_stream = torch.cuda.Stream(device=dev)
        probe_buf = torch.zeros(2, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        N.check(eng.lib.lspf2f_clock_probe(ctypes.c_void_p(probe_buf.data_ptr()), int(min(2_000_000, max(50, 0.8e6 * elapsed))), ctypes.c_void_p(probe_stream.cuda_stream)))
        for _ in range(a.steps):
            eng.forward(feat, cand, out)
        torch.cuda.synchronize()
        pc, pt = (int(v) for v in probe_buf.cpu().tolist())
        clock_ghz = round(0.1 * pc / pt, 3) if pt > 0 else None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ev_ms = ev0.elapsed_time(ev1) / a.steps     # device-side, this rank's stream

    # BASELINE.json configs[3] shape beside the headline when N > 1: 8 frames per GPU per step, ONE candidate stack shared by
    # the batch, distinct feature maps per rank and frame (seed 99 + 1000*rank + i), same barrier / MAX-over-ranks protocol.
    # Efficiency = this aggregate / (N x the 1-GPU `extra.batch8_frames_per_s` of the N = 1 run) -- computed by the reader.
    config3 = None
    if world > 1 and not a.no_extra:
        f8 = torch.from_numpy(synth.make_inputs(8, a.size, seed=99 + 1000 * rank, cand_batch=1)[0]).to(dev)
        o8 = torch.empty((8, 3, a.size, a.size), device=dev)
        for _ in range(3):
            eng.forward(f8, cand, o8)
        torch.cuda.synchronize()
        barrier()
        n8 = max(5, a.steps // 5)
        t1 = time.perf_counter()
        for _ in range(n8):
            eng.forward(f8, cand, o8)
        torch.cuda.synchronize()
        mine = time.perf_counter() - t1
        barrier()
        t = torch.tensor([time.perf_counter() - t1, mine, -mine], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        config3 = {"frames_per_s": round(world * 8 * n8 / float(t[0].item()), 2), "global_batch": 8 * world, "frames_per_gpu": 8,
                   "steps": n8, "per_gpu_frames_per_s_slowest_fastest": [round(8 * n8 / float(t[1].item()), 2), round(8 * n8 / float(-t[2].item()), 2)],
                   "backend": dist.get_backend(), "ranks_in_group": dist.get_world_size(),
                   "note": "BASELINE.json configs[3]: batch 64 = 8 frames x 8 GPUs at N = 8; candidates shared; one RCCL weight broadcast at start-up, no per-frame collective"}

    if rank != 0:
        if world > 1:
            leave_group()
        return

    ms_per_step = 1e3 * elapsed / a.steps
    fps = world * B * a.steps / elapsed
    flops_step = topo.flops_per_frame() * B

    # ---- roofline: live, per kernel class.  ROCm 7.2 cannot time events recorded by graph nodes, so a kernel cannot be bracketed inside
    # the replay of the whole forward; instead the launches of ONE class (in network order, reading what the last forward left in the
    # workspace) are captured into their own graph and replayed between two hipEvents on the launch stream (lspf2f_subset_timed):
    # no host gaps, kernel boundaries included, the same launches the timed region replays.  The classes partition the forward, their
    # times add up to <= the timed step, and each class's average launch duration is what the committed rocprofv3 summary
    # (profiles/r03_kernel_stats_*.txt) shows for that kernel name.
    layers = eng.layers(B)
    eng.forward(feat, cand, out)

    def cls_of(l):
        k = l["kernel"]
        if k.startswith("igemm3x3"):
            return "igemm3x3<%dx%d,g%d>" % (l["tile_m"], l["tile_n"], l["k_group"])
        return k.split(" ")[0]
    classes = {}
    for i, l in enumerate(layers):
        classes.setdefault(cls_of(l), []).append(i)
    table = []
    elt = 4 if a.dtype == "f32" else 2

    def add_row(name, idxs, part, flops, exec_flops, nbytes, launches):
        sel = [0] * len(layers)
        for i in idxs:
            sel[i] = part
        ms = eng.subset_timed(feat, cand, sel, out, reps=10)
        table.append({"kernel": name, "launches": launches, "flops": int(flops), "exec_flops": int(exec_flops), "bytes": int(nbytes),
                      "ms": round(ms, 5), "us_per_launch": round(1e3 * ms / launches, 2),
                      "tflops": round(flops / (ms * 1e-3) / 1e12, 2), "exec_tflops": round(exec_flops / (ms * 1e-3) / 1e12, 2),
                      "gbs": round(nbytes / (ms * 1e-3) / 1e9, 1),
                      # frac_mfma = utilisation of the matrix pipe: FLOPs actually ISSUED / time / dense peak (never above 1).  algorithmic_mfma counts
                      # the FLOPs of the literal 3x3 convolution instead; flop_reduction = issued / algorithmic (Winograd 4/9 or 1/4, sub-pixel 4/9)
                      "frac_mfma": round(exec_flops / (ms * 1e-3) / 1e12 / peak, 4),
                      "algorithmic_mfma": round(flops / (ms * 1e-3) / 1e12 / peak, 4),
                      "flop_reduction": round(exec_flops / flops, 4) if flops else None,
                      "frac_hbm": round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)})
    for name, idxs in classes.items():
        fl = sum(layers[i]["flops_per_frame"] for i in idxs) * B
        ex = sum(layers[i]["exec_flops_per_frame"] for i in idxs) * B
        by = sum(layers[i]["act_bytes_per_frame"] for i in idxs) * B + sum(layers[i]["weight_bytes"] for i in idxs)
        add_row(name, idxs, 1 if name.startswith("igemm3x3") else 3, fl, ex, by, len(idxs))
    split = [i for i, l in enumerate(layers) if "+splitk_reduce" in l["kernel"]]
    if split:
        # the reduce launches: no arithmetic; bytes = the fp32 partial slabs they read + the tensor they write
        by = sum((layers[i]["split_k"] * 4 + elt) * layers[i]["cout"] * layers[i]["h_out"] ** 2 for i in split) * B
        add_row("splitk_reduce*", split, 2, 0, 0, by, len(split))
    table.sort(key=lambda r: -r["ms"])
    dom = table[0]
    class_ms = sum(r["ms"] for r in table)
    if a.layers:
        with open(a.layers, "w") as f:
            f.write("# per kernel class, %s batch %d %s: launches of one class replayed from their own graph between two hipEvents (bench.py)\n" % (a.variant, B, a.dtype))
            f.write("# frac_mfma = ISSUED FLOPs / time / dense peak (utilisation); algor. = the literal convolution's FLOPs / time / peak\n")
            f.write("%-28s %8s %10s %10s %9s %9s %9s %8s %9s %9s %9s\n" % ("kernel class", "launches", "GFLOP", "MB", "ms", "us/launch", "TFLOP/s", "GB/s", "frac_mfma", "algor.", "frac_hbm"))
            for r in table:
                f.write("%-28s %8d %10.3f %10.2f %9.4f %9.2f %9.2f %8.1f %9.4f %9.4f %9.4f\n" % (
                    r["kernel"], r["launches"], r["flops"] / 1e9, r["bytes"] / 1e6, r["ms"], r["us_per_launch"], r["tflops"], r["gbs"], r["frac_mfma"], r["algorithmic_mfma"], r["frac_hbm"]))
            f.write("# sum of classes %.4f ms; timed step (graph replay of the whole forward) %.4f ms\n" % (class_ms, ms_per_step))

    # HBM traffic of the dominant kernel from the committed PMC passes (offline: rocprofv3 --pmc cannot run inside this process);
    # only quoted for the workload it was measured on
    traffic, traffic_src = None, None
    pmc_path = next((q for q in (os.path.join(ROOT, "profiles", "%s_pmc_%s_b%d_%s.json" % (r, a.variant, B, a.dtype)) for r in ("r06", "r05", "r04", "r03")) if os.path.exists(q)),
                    os.path.join(ROOT, "profiles", "r04_pmc_%s_b%d_%s.json" % (a.variant, B, a.dtype)))
    family = "wino3x3" if dom["kernel"].startswith("wino3x3") else dom["kernel"].split("<")[0]
    if a.size == 512 and os.path.exists(pmc_path):
        pj = json.load(open(pmc_path))
        exact = pj["per_forward_bytes"].get(dom["kernel"])          # per template instance where the summary has it (wino3x3<1> / <2>)
        fam = pj["per_forward_bytes"].get(family)
        if exact and (exact["fetch_x2"] + exact["write"]) > 0:
            traffic = int(exact["fetch_x2"] + exact["write"])
            traffic_src = "profiles/%s (FETCH_SIZE x2 + WRITE_SIZE of the %s launches of one forward, rocprofv3 --pmc, separate passes)" % (os.path.basename(pmc_path), dom["kernel"])
        elif fam:
            # the PMC families are per kernel NAME (all template instances together); scaled to the dominant class by its share of the family's
            # algorithmic bytes when the family has more than one class in this plan
            fam_bytes = sum(r["bytes"] for r in table if r["kernel"].startswith(family))
            share = dom["bytes"] / fam_bytes if fam_bytes else 1.0
            traffic = int((fam["fetch_x2"] + fam["write"]) * share)
            traffic_src = ("profiles/%s (FETCH_SIZE x2 + WRITE_SIZE of the %s launches of one forward, rocprofv3 --pmc, separate passes; x %.2f = this class's share "
                           "of the family's algorithmic bytes)" % (os.path.basename(pmc_path), family, share))

    exec_step = sum(l["exec_flops_per_frame"] for l in layers) * B
    dom_exec_tf = dom["exec_flops"] / (dom["ms"] * 1e-3) / 1e12
    roofline = {
        "bound": "mfma",
        "kernel": "%s: the %d launches per forward of the dominant kernel class (all conv layers it executes; split-K reduce launches are their own row)" % (dom["kernel"], dom["launches"]),
        # `achieved` / `frac` = what the matrix pipe really did: MFMA FLOPs ISSUED by the launches / their time (/ dense peak) -- a utilisation, <= 1.
        # The contract's algorithmic figure (FLOPs of the literal 3x3 convolution) is carried beside it: Winograd F(2x2,3x3) and the
        # sub-pixel up-convs issue 4/9 of it, so it can exceed the peak; that is arithmetic saved, not utilisation.
        "achieved": round(dom_exec_tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(dom_exec_tf / peak, 4),
        "algorithmic_achieved": dom["tflops"], "algorithmic_frac": dom["algorithmic_mfma"], "flop_reduction": dom["flop_reduction"],
        "clock_ghz_observed": clock_ghz, "clock_ghz_peak": 2.4,
        "peak_at_observed_clock": round(peak * clock_ghz / 2.4, 1) if clock_ghz else None,
        "frac_at_observed_clock": round(dom_exec_tf / (peak * clock_ghz / 2.4), 4) if clock_ghz else None,
        "clock_note": "`peak` assumes the 2.4 GHz maximum; clock_ghz_observed = shader cycles / 100-MHz ticks counted by one wave on a side stream while "
                      "the timed region ran (lspf2f_clock_probe): what the chip held under this load",
        "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": dom["bytes"],
        "flops_per_launch_set": dom["exec_flops"], "algorithmic_flops_per_launch_set": dom["flops"], "ms_per_launch_set": dom["ms"], "us_per_launch": dom["us_per_launch"],
        "method": "launches of one kernel class replayed from their own hipGraph between two hipEvents on the launch stream (lspf2f_subset_timed); "
                  "agrees with the rocprofv3 --kernel-trace --stats averages committed under profiles/",
        "per_class": table, "sum_of_classes_ms": round(class_ms, 4),
        "whole_forward": {"achieved": round(exec_step / (ev_ms * 1e-3) / 1e12, 2),
                          "frac": round(exec_step / (ev_ms * 1e-3) / 1e12 / peak, 4),
                          "algorithmic_achieved": round(flops_step / (ev_ms * 1e-3) / 1e12, 2),
                          "algorithmic_frac": round(flops_step / (ev_ms * 1e-3) / 1e12 / peak, 4),
                          "flop_reduction": round(exec_step / flops_step, 4),
                          "ms_device": round(ev_ms, 4)},
    }

    cpu_baseline = None
    if not a.no_cpu_baseline:
        from oracle import torch_oracle
        sd_t = torch_oracle.to_torch(sd)
        x = torch.cat([torch.from_numpy(feat_np[:1]), torch.from_numpy(cand_np)], 1)
        n_timed = 4
        tmin, tmed, threads = torch_oracle.time_cpu(sd_t, x, topo.nres, topo.num_downs, repeats=n_timed)
        cpu_baseline = {"value": round(1.0 / tmin, 3), "unit": "frames/s", "cores": threads, "kind": "port",
                        "median_value": round(1.0 / tmed, 3),
                        "kind_note": "port = oracle/torch_oracle.py: the reference's own torch op sequence, asserted bit-identical to the reference "
                                     "modules by oracle/make_golden.py (the Python reference cannot travel to the GPU box)",
                        "sample": "%s generator, batch 1, %dx%d fp32, 1 warm-up + %d timed frames of "
                                  "oracle/torch_oracle.py (torch %s CPU/oneDNN, %d threads)"
                                  % (a.variant, a.size, a.size, n_timed, torch.__version__, threads)}

    extra = None
    if not a.no_extra and world == 1 and B == 1:
        extra = {}

        def guard(key, fn, *args):
            """an informational probe must never cost the headline line that is already measured: its failure is recorded in its own slot"""
            try:
                extra[key] = fn(*args)
            except Exception as ex:                      # noqa: BLE001
                extra[key] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
                try:
                    torch.cuda.synchronize()
                except Exception:                        # noqa: BLE001
                    pass
    if extra is not None:
      try:
        # throughput configuration (BASELINE.json configs[3] shape: 8 frames per GPU, shared candidates)
        f8 = torch.from_numpy(synth.make_inputs(8, a.size, seed=99, cand_batch=1)[0]).to(dev)
        o8 = torch.empty((8, 3, a.size, a.size), device=dev)
        for _ in range(3):
            eng.forward(f8, cand, o8)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n8 = max(5, a.steps // 5)
        for _ in range(n8):
            eng.forward(f8, cand, o8)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        extra.update({"batch8_frames_per_s": round(8 * n8 / dt, 2),
                      "batch8_tflops": round(topo.flops_per_frame() * 8 * n8 / dt / 1e12, 2)})
        # PCIe-inclusive rate of a demo.py-style loop (never `value`): per frame, H2D of a host feature map
        # (1 MiB, pinned), forward with fused tensor2im, D2H of the uint8 frame (0.75 MiB), synchronised
        hfeat = torch.from_numpy(feat_np).pin_memory()
        hout = torch.empty((1, a.size, a.size, 3), dtype=torch.uint8).pin_memory()
        dfeat = torch.empty_like(feat)
        du8 = torch.empty((1, a.size, a.size, 3), dtype=torch.uint8, device=dev)
        for _ in range(3):
            dfeat.copy_(hfeat, non_blocking=True); eng.forward_image(dfeat, cand, du8); hout.copy_(du8, non_blocking=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        nl = 100
        for _ in range(nl):
            dfeat.copy_(hfeat, non_blocking=True)
            eng.forward_image(dfeat, cand, du8)
            hout.copy_(du8, non_blocking=True)
            torch.cuda.synchronize()
        extra["pcie_inclusive_frames_per_s_batch1_uint8"] = round(nl / (time.perf_counter() - t2), 2)
        # the same loop with the edge map drawn on the device: per frame ~1.5 KB of landmarks + shoulder points go
        # H2D instead of a host-rasterised 1 MiB feature map (datasets/face_dataset.py:276-323 + demo.py:262-265)
        from livespeechportraits_amd.feature_map import FeatureMapRasteriser
        rast = FeatureMapRasteriser(a.size, 18, dev)
        rng = np.random.default_rng(0)
        lm = (a.size * 0.5 + rng.normal(0, a.size * 0.12, (1, 73, 2))).astype(np.float32)
        sh = np.stack([np.linspace(0, a.size, 18), np.full(18, a.size * 0.9)], 1)[None].astype(np.float32)
        hpts = torch.from_numpy(np.concatenate([lm, sh], 1)).pin_memory()
        dpts = torch.empty((1, 91, 2), device=dev)
        for _ in range(3):
            dpts.copy_(hpts, non_blocking=True)
            rast.rasterise_points(dpts, out=dfeat); eng.forward_image(dfeat, cand, du8); hout.copy_(du8, non_blocking=True)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(nl):
            dpts.copy_(hpts, non_blocking=True)                 # 728 bytes H2D instead of the 1 MiB feature map
            rast.rasterise_points(dpts, out=dfeat)
            eng.forward_image(dfeat, cand, du8)
            hout.copy_(du8, non_blocking=True)
            torch.cuda.synchronize()
        extra["pcie_inclusive_frames_per_s_batch1_uint8_landmarks_in"] = round(nl / (time.perf_counter() - t3), 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        l8 = torch.from_numpy(np.repeat(lm, 8, 0)).to(dev); s8 = torch.from_numpy(np.repeat(sh, 8, 0)).to(dev)
        m8 = torch.empty((8, 1, a.size, a.size), device=dev)
        rast.rasterise(l8, s8, out=m8); torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            rast.rasterise(l8, s8, out=m8)
        e1.record(); torch.cuda.synchronize()
        extra["edge_map_rasteriser"] = {"us_per_frame_batch8": round(e0.elapsed_time(e1) * 1e3 / 160, 2),
                                        "hbm_frac_of_8TBs": round(8 * a.size * a.size * 4 / (e0.elapsed_time(e1) * 1e-3 / 20) / 8e12, 4),
                                        "note": "lspraster_edge_maps: 88 thick edges per frame -> fp32 [8,1,512,512]; parity unpinned vs cv2 (bit-exact to oracle/raster_oracle.c)"}
      except Exception as ex:                            # noqa: BLE001
        extra["inline_probes_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:300])
      cores = None if a.no_cpu_baseline else (cpu_baseline or {}).get("cores", 8)
      guard("torch_rocm_baseline", torch_rocm_extra, dev, sd, topo, feat_np, cand_np, a)
      guard("small_unet_native_plan", small_unet_extra, dev, a)
      guard("concurrent_batch1_forwards", concurrent_extra, dev, eng, a)
      guard("render_loop_end_to_end", render_loop_extra, dev, sd, a)
      guard("headpose", headpose_extra, dev, cores)
      guard("manifold_projection", manifold_extra, dev, cores)
      guard("audio_recurrent", recurrent_extra, dev, cores)
      guard("pipeline_gpu_stages_plumbing_only", pipeline_extra, dev, eng, cand)
      # BASELINE.json configs[2] LAST and compact: the driver keeps only the tail of this line; the per-frame detail sits in its own key ahead of it
      guard("config2_detail", config2_extra, dev, a)
      c2 = extra.get("config2_detail") or {}
      if "error" in c2:
          extra["config2_normal_b8_bf16"] = c2
      else:
          vo = c2.get("vs_fp32_oracle") or {}
          extra["config2_normal_b8_bf16"] = {
              "frames_per_s": c2.get("frames_per_s"), "ms_per_step": c2.get("ms_per_step"), "steps": c2.get("steps"), "dtype": "bf16", "batch": 8,
              "executed_frac_of_dense_bf16_peak": c2.get("whole_forward_frac_of_dense_bf16_peak"),
              "algorithmic_frac_of_dense_bf16_peak": c2.get("whole_forward_algorithmic_frac_of_dense_bf16_peak"),
              "dominant_class": c2.get("dominant_class"),
              "max_abs_vs_fp32_oracle": max(vo.get("max_abs_per_frame") or [float("nan")]), "mean_abs_vs_fp32_oracle": max(vo.get("mean_abs_per_frame") or [float("nan")]),
              "declared_tolerance": "1e-2 max / 2.5e-3 mean per frame (parity-unpinned: the reference has no bf16 path)"}

    line = {
        "metric": "512x512 frames/sec (Feature2FaceGenerator fwd)", "value": round(fps, 3), "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "%s generator (%s), batch %d per GPU, %dx%d, %s, synthetic weights+inputs"
                               % (a.variant, "May" if a.variant == "large" else "Obama1", B, a.size, a.size,
                                  {"f32": "fp32", "bf16": "bf16 storage / fp32 accumulate (parity-unpinned: tolerance declared in tests)",
                                   "f16": "fp16 storage / fp32 accumulate (the reference's opt.fp16; pinned on the autocast oracle in tests)"}[a.dtype]),
                   "global_batch": world * B, "parallelism": "dp%d (frames sharded, one RCCL weight broadcast)" % world,
                   "gflop_per_frame": round(topo.flops_per_frame() / 1e9, 2)},
        "roofline": roofline, "cpu_baseline": cpu_baseline,
    }
    if config3:
        extra = dict(extra or {}, config3_batch8_per_gpu=config3)
    if world > 1:
        line["config"]["collective_backend"] = dist.get_backend()
        line["config"]["ranks_in_group"] = dist.get_world_size()
    if extra:
        line["extra"] = extra
    print(json.dumps(line))
    sys.stdout.flush()
    if world > 1:
        leave_group()


def torch_rocm_extra(dev, sd, topo, feat_np, cand_np, a):
    """The as-shipped-on-this-hardware comparator: the same generator through PyTorch-ROCm / MIOpen
    on the same MI355X, same weights and inputs, fp32, batch 1 and batch 8 (candidates expanded like the reference's cat would
    need).  Baseline leg: oracle/torch_oracle.py is the reference's module sequence restated op for op (asserted bit-identical
    to the reference classes on CPU by oracle/make_golden.py), here simply moved to cuda:0.  cudnn.benchmark = True is what the
    reference sets (models/base_model.py:46-47)."""
    from oracle import torch_oracle
    prev = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = True
    out = {"kind": "port", "device": torch.cuda.get_device_name(dev), "torch": torch.__version__,
           "note": "oracle/torch_oracle.py (the reference's op sequence, bit-identical to its modules on CPU) on cuda:0: ATen -> MIOpen, fp32, eager, cudnn.benchmark=True"}
    try:
        sd_d = {k: v.to(dev) for k, v in torch_oracle.to_torch(sd).items()}
        for b in (1, 8):
            f = torch.from_numpy(synth_inputs(b, a.size)[0]).to(dev)
            x = torch.cat([f, torch.from_numpy(cand_np).to(dev).expand(b, -1, -1, -1)], 1).contiguous()
            for _ in range(3):
                y = torch_oracle.generator_forward(sd_d, x, topo.nres, topo.num_downs)
            torch.cuda.synchronize()
            n = 10 if b == 1 else 4
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                y = torch_oracle.generator_forward(sd_d, x, topo.nres, topo.num_downs)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            out["batch%d" % b] = {"frames_per_s": round(b / (ms * 1e-3), 2), "ms_per_step": round(ms, 3),
                                  "tflops": round(topo.flops_per_frame() * b / (ms * 1e-3) / 1e12, 2)}
            del y
    except Exception as ex:      # a MIOpen failure must not take the headline line with it
        out["error"] = "%s: %s" % (type(ex).__name__, str(ex)[:300])
    torch.backends.cudnn.benchmark = prev
    torch.cuda.empty_cache()
    return out


def synth_inputs(b, size):
    from livespeechportraits_amd import synth
    return synth.make_inputs(b, size, seed=99, cand_batch=1)


def render_loop_extra(dev, sd, a):
    """The product's frame loop (render_loop.render_frames = demo.py:260-272 batched): pinned host feature maps in, uint8 HWC frames out on the host -- gather, H2D, generator +
    tensor2im, D2H, hand-out -- through the drop-in model object, 8 frames per batch, two lanes (stream + handle each).  PCIe-inclusive: never `value`."""
    import argparse as _ap
    import livespeechportraits_amd as L
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.render_loop import render_frames
    opt = _ap.Namespace(model="feature2face", gpu_ids=[0], isTrain=False, size=a.variant, ngf=64, n_downsample_G=8, fp16=0, checkpoints_dir="/tmp", name="bench",
                        load_epoch="none", verbose=False)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):                  # (the reference's constructor prints)
        model = L.create_model(opt)
    model._g().load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    model.eval()
    feats, cand = synth.make_inputs(8, a.size, seed=5, cand_batch=1)
    c = torch.from_numpy(cand).to(dev)
    maps = [torch.from_numpy(feats[i % 8]).pin_memory() for i in range(256)]
    r = {}
    for lanes in (1, 2):
        render_frames(model, iter(maps[:32]), c, batch=8, streams=lanes)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = render_frames(model, iter(maps), c, batch=8, streams=lanes)
        r["lanes_%d_frames_per_s" % lanes] = round(len(out) / (time.perf_counter() - t0), 1)
    r["note"] = "256 frames, batch 8, host feature maps (1 MiB each, pinned) -> uint8 frames on the host; round 4's loop (one stream, event waits, torch CPU copies): 160-290 on the same boxes"
    return r


def concurrent_extra(dev, eng, a):
    """NOT the headline: N independent batch-1 forwards in flight at once (N handles on ONE packed blob, one stream each) against the single stream `value` is
    measured on.  A batch-1 forward is a chain of dependent launches, each a single round of workgroups, so prologues, tails and kernel boundaries overlap nothing;
    independent frames on other streams fill those holes.  In a FRESH process four streams add 26 % (667 -> 838 frames/s, tools/multistream_probe.py,
    profiles/r05_concurrent_streams.txt): the size of what a cross-layer overlap inside one forward could win.  In THIS process, late in its life, the figure reads lower (why is not established; streams
    created earlier are ruled out), so the probe is also run in a process of its own (`fresh_process`).  A caller with frames in hand batches them instead (the batch-8 rows
    are faster still); this is the number for frames that arrive one by one."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    blob = eng._blob_dev
    engines, streams, ins, outs = [eng], [torch.cuda.Stream(dev)], [], []
    for i in range(1, 4):
        e = Engine(a.variant, size=a.size, max_batch=eng.max_batch, dtype=a.dtype)
        e.bind(blob, dev)
        engines.append(e); streams.append(torch.cuda.Stream(dev))
    for i in range(4):
        f, c = synth.make_inputs(1, a.size, 99 + i, 1)
        ins.append((torch.from_numpy(f).to(dev), torch.from_numpy(c).to(dev)))
        outs.append(torch.empty((1, 3, a.size, a.size), device=dev))
    torch.cuda.synchronize()
    r = {"note": "N handles on one packed blob, one stream each, batch 1 per forward; outputs bit-identical to the single-stream run; `value` above is the 1-stream number"}
    for n in (1, 2, 4):
        def run(reps):
            for _ in range(reps):
                for i in range(n):
                    with torch.cuda.stream(streams[i]):
                        engines[i].forward(ins[i][0], ins[i][1], outs[i])
        run(5); torch.cuda.synchronize()
        reps = max(20, a.steps)
        t0 = time.perf_counter(); run(reps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        r["streams_%d" % n] = {"frames_per_s": round(n * reps / dt, 1), "ms_per_frame": round(1e3 * dt / (n * reps), 4)}
    for e in engines[1:]:
        e.close()
    # the same measurement in a process of its own (tools/multistream_probe.py): the figure that is not distorted by this process's stream history
    try:
        import re, subprocess
        root = os.path.dirname(os.path.abspath(__file__))
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "multistream_probe.py"), a.variant, a.dtype, "4", "1"], capture_output=True, text=True, timeout=180)
        fresh = {}
        for m in re.finditer(r"(\d+) concurrent stream\(s\): ([0-9.]+) frames/s .*bit-identical to the single-stream run: (True|False)", p.stdout):
            fresh["streams_%s" % m.group(1)] = {"frames_per_s": float(m.group(2)), "bit_identical": m.group(3) == "True"}
        if fresh:
            r["fresh_process"] = fresh
    except Exception as ex:      # the probe is an extra: never fail the bench line over it
        r["fresh_process_error"] = str(ex)[:200]
    return r


def small_unet_extra(dev, a):
    """opt.size == 'small' (Feature2FaceGenerator_Unet, models/networks.py:680-769): the native plan behind include/lspunet.h, fp32 at one and
    eight frames and the fp16 storage plan of opt.fp16, forwards enqueued back to back between two events; no shipped configuration selects the variant."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.unet_small import SmallUnetEngine
    sd = synth.make_unet_small_state_dict()
    r = {"workload": "pix2pix U-Net, 23-channel input, ngf 64, 8 levels, %dx%d, synthetic weights+inputs" % (a.size, a.size)}
    for dt in ("f32", "f16"):
        e = SmallUnetEngine(dtype=dt)
        e.load_state_dict(sd, "model", dev)
        for B in (1, 8):
            x = torch.from_numpy(synth.symmetric(B * 23 * a.size * a.size, 0.6, 3).reshape(B, 23, a.size, a.size)).to(dev)
            for _ in range(3):
                e.forward(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            n = 40 if B == 1 else 12
            for _ in range(n):
                e.forward(x)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            r["%s_batch%d" % (dt, B)] = {"ms_per_forward": round(ms, 4), "frames_per_s": round(B / ms * 1e3, 1)}
        e.close()
    r["note"] = "parity: tests/test_unet_small.py (fp32 vs the reference module's frozen output <= 1e-4, measured 1.4e-6; fp16 pinned on the autocast oracle); host-sequenced form of round 4: 0.90 / 4.61 ms"
    return r


def config2_extra(dev, a):
    """BASELINE.json configs[2] inside the default run: Obama1 ('normal'), batch 8, 512x512, bf16 storage / fp32 accumulate.  Frames/s
    with the same timing protocol as the headline, the whole-forward fraction of the dense bf16 MFMA peak, and every frame's
    distance from the fp32 oracle (the reference has no bf16 path: a declared tolerance, NOT a parity result)."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    from livespeechportraits_amd.topology import build_topology
    topo = build_topology("normal", size=a.size)
    sd = synth.make_state_dict(topo, 1234)
    eng = Engine("normal", size=a.size, max_batch=8, dtype="bf16")
    eng.load_state_dict(sd)
    eng.bind(eng.pack(), dev)
    feat_np, cand_np = synth.make_inputs(8, a.size, seed=99, cand_batch=1)
    feat, cand = torch.from_numpy(feat_np).to(dev), torch.from_numpy(cand_np).to(dev)
    out = torch.empty((8, 3, a.size, a.size), device=dev)
    for _ in range(5):
        eng.forward(feat, cand, out)
    torch.cuda.synchronize()
    n = max(10, a.steps // 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        eng.forward(feat, cand, out)
    e1.record(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    r = {"frames_per_s": round(8 * n / dt, 1), "ms_per_step": round(1e3 * dt / n, 4), "steps": n, "dtype": "bf16",
         # executed = MFMA FLOPs the plan issues (sub-pixel up-convs: 4/9 of the literal convolution's), the utilisation; algorithmic = SURVEY.md 8d's count
         "whole_forward_frac_of_dense_bf16_peak": round(sum(l["exec_flops_per_frame"] for l in eng.layers(8)) * 8 / (e0.elapsed_time(e1) * 1e-3 / n) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
         "whole_forward_algorithmic_frac_of_dense_bf16_peak": round(topo.flops_per_frame() * 8 / (e0.elapsed_time(e1) * 1e-3 / n) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
         "workload": "normal generator (Obama1), batch 8, %dx%d, bf16 storage / fp32 accumulate, candidates shared" % (a.size, a.size)}
    if not a.no_cpu_baseline:
        from oracle import torch_oracle
        sd_t = torch_oracle.to_torch(sd)
        x = torch.cat([torch.from_numpy(feat_np), torch.from_numpy(cand_np).expand(8, -1, -1, -1)], 1)
        ref = torch_oracle.generator_forward(sd_t, x, topo.nres, topo.num_downs)
        d = (out.cpu() - ref).abs().flatten(1)
        r["vs_fp32_oracle"] = {"max_abs_per_frame": [round(float(v), 6) for v in d.max(1).values],
                               "mean_abs_per_frame": [round(float(v), 6) for v in d.mean(1)],
                               "note": "parity-unpinned by construction (the reference has only fp16 autocast); tests/test_gpu_plans.py declares 1e-2 max / 2.5e-3 mean per frame for this variant"}
    # the plan's dominant kernel class, replayed from its own graph like the headline's roofline rows (lspf2f_subset_timed)
    layers = eng.layers(8)
    groups = {}
    for i, l in enumerate(layers):
        k = l["kernel"]
        groups.setdefault("igemm3x3<%dx%d>" % (l["tile_m"], l["tile_n"]) if k.startswith("igemm3x3") else k.split(" ")[0], []).append(i)
    best = None
    for name, idxs in groups.items():
        sel = [0] * len(layers)
        for i in idxs:
            sel[i] = 1 if name.startswith("igemm3x3") else 3
        ms = eng.subset_timed(feat, cand, sel, out, reps=10)
        if best is None or ms > best[1]:
            best = (name, ms, idxs)
    name, ms, idxs = best
    ex = sum(layers[i]["exec_flops_per_frame"] for i in idxs) * 8
    r["dominant_class"] = {"kernel": name, "launches": len(idxs), "ms": round(ms, 4), "us_per_launch": round(1e3 * ms / len(idxs), 2),
                           "executed_frac_of_dense_bf16_peak": round(ex / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}
    eng.close()
    return r


def headpose_extra(dev, cpu_threads):
    """SURVEY.md 8f rank 3 beside the headline: Audio2HeadposeModel.generate_sequences on the demo clip's length
    (687 frames + frame_future 15, default network, synthetic weights), device tensors in -> device tensor out.
    loop_ms = fill_ms + us_per_frame * nframe from two clip lengths; the CPU leg times the reference's
    sliding-window algorithm (oracle/a2h_oracle.py, kind "port") on a bounded sample."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.a2h_engine import HeadposeEngine
    cfg, ff = dict(synth.A2H_DEFAULTS), 15
    sd = synth.make_a2h_state_dict(cfg)
    eng = HeadposeEngine(max_audio_frames=687 + ff)
    eng.load_state_dict(sd)
    eng.bind(dev)
    med = {}
    for nframe in (87, 687):
        audio, pre = synth.make_a2h_inputs(nframe + ff, cfg)
        au, pr = torch.from_numpy(audio).to(dev), torch.from_numpy(pre).to(dev)
        noise = torch.from_numpy(synth.symmetric(nframe * 12, 1.0, 5).reshape(nframe, 12)).to(dev)
        eng.generate_timed(au, pr, noise, None, 0.3, ff)
        t = sorted(eng.generate_timed(au, pr, noise, None, 0.3, ff)[1:] for _ in range(5))
        if eng.status() != 0:
            raise RuntimeError("head-pose kernel hand-off timed out")
        med[nframe] = t[2]
    slope = (med[687][1] - med[87][1]) / 600.0
    out = {"metric": "head poses/s (Audio2Headpose.generate_sequences, 687-frame clip, fp32, synthetic weights)",
           "value": round(687 / ((med[687][0] + med[687][1]) * 1e-3), 1), "precompute_ms": round(med[687][0], 3),
           "loop_ms": round(med[687][1], 3), "us_per_frame": round(1e3 * slope, 2),
           "fill_ms": round(med[87][1] - slope * 87, 3)}
    if cpu_threads:
        from oracle import a2h_oracle
        n = 32
        audio, pre = synth.make_a2h_inputs(n + ff, cfg)
        noise = synth.symmetric(n * 12, 1.0, 5).reshape(n, 12)
        torch.set_num_threads(int(cpu_threads))
        a2h_oracle.generate_sequences(sd, cfg, audio[:4 + ff], pre, noise, np.ones((n, 1), np.float32), 0.3, ff)   # warm-up
        t0 = time.perf_counter()
        a2h_oracle.generate_sequences(sd, cfg, audio, pre, noise, np.ones((n, 1), np.float32), 0.3, ff)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(n / dt, 2), "unit": "head poses/s", "cores": int(cpu_threads), "kind": "port",
                               "sample": "%d frames of oracle/a2h_oracle.generate_sequences: the reference's per-frame 255-wide "
                                         "re-evaluation (torch %s CPU)" % (n, torch.__version__)}
    return out


def pipeline_extra(dev, eng, cand):
    """BASELINE.json configs[4] shape, PLUMBING ONLY: the device stages of demo.py chained on one stream for a 687-frame clip
    (data/Input/00083.wav is 687 frames at 60 fps = 11.45 s) with synthetic weights and stand-in data -- waveform -> mel -> APC
    encoder -> manifold projection -> Audio2Feature and Audio2Headpose -> (stand-in landmarks) -> edge-map rasteriser -> renderer
    (batch 8, fused tensor2im, uint8 frames left on the device).  NOT in it, because the reference does them on the CPU with
    per-person assets that cannot be obtained: smoothing / head-pose post-processing, the 3-D projection that turns mouth
    features and poses into the 73 landmarks (random stand-in landmarks are rasterised instead), JPEG / video writing.  One
    number: clip frames per second of wall-clock over those device stages."""
    from livespeechportraits_amd import manifold, synth
    from livespeechportraits_amd.a2h_engine import HeadposeEngine
    from livespeechportraits_amd.apc import APC_encoder
    from livespeechportraits_amd.audio2feature import Audio2Feature
    import argparse as _ap
    nframe, ff_head, ff_feat = 687, 15, 18
    apc = APC_encoder(80, 512, 3, False)
    apc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_apc_state_dict().items()})
    apc = apc.to(dev).eval()
    a2f = Audio2Feature(_ap.Namespace(feature_decoder="LSTM", loss="L2", A2L_GMM_ndim=75, A2L_GMM_ncenter=1, predict_length=1, APC_hidden_size=512))
    a2f.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_a2f_state_dict().items()}, strict=False)
    a2f = a2f.to(dev).eval()
    cfg = dict(synth.A2H_DEFAULTS)
    a2h = HeadposeEngine(max_audio_frames=nframe + ff_head)
    a2h.load_state_dict(synth.make_a2h_state_dict(cfg)); a2h.bind(dev)
    db = torch.from_numpy(synth.make_feature_database(30000, 8, 512, 24)[0]).to(dev)
    from livespeechportraits_amd import mel as mel_mod
    from livespeechportraits_amd.feature_map import FeatureMapRasteriser
    wave = torch.from_numpy((0.1 * np.random.default_rng(1).standard_normal(int(nframe / 60 * 16000) + 8)).astype(np.float32)).to(dev)   # 11.45 s at 16 kHz
    rast = FeatureMapRasteriser(eng.size, 18, dev)
    rng = np.random.default_rng(2)
    lms = torch.from_numpy((eng.size * 0.5 + rng.normal(0, eng.size * 0.12, (nframe, 73, 2))).astype(np.float32)).to(dev)      # stand-in landmarks
    shs = torch.from_numpy(np.tile(np.stack([np.linspace(0, eng.size, 18), np.full(18, eng.size * 0.9)], 1)[None], (nframe, 1, 1)).astype(np.float32)).to(dev)
    pre = torch.zeros(12, device=dev)
    noise = torch.from_numpy(synth.symmetric(nframe * 12, 1.0, 5).reshape(nframe, 12)).to(dev)
    maps = torch.empty((8, 1, eng.size, eng.size), device=dev)
    frames = torch.empty((8, eng.size, eng.size, 3), dtype=torch.uint8, device=dev)

    def clip():
        mel = mel_mod.compute_mel(wave).unsqueeze(0)                                            # [1, 1374, 80] from the waveform, on the device
        feats = apc.forward(mel, torch.Tensor([2 * nframe]))[0]                                 # [1374, 512]
        feats = manifold.project(feats.contiguous(), db, 10, 1.0)
        tail = feats[-1:].expand(2 * ff_feat, -1)
        mouth = a2f.forward(torch.cat([feats, tail]).unsqueeze(0))[0, ff_feat:]                 # [687, 75]
        a2h_in = torch.cat([feats.reshape(nframe, 1024), feats.reshape(nframe, 1024)[-1:].expand(ff_head, -1)]).contiguous()
        poses = a2h.generate(a2h_in, pre, noise, None, 0.3, ff_head)                            # [687, 12]
        for i in range(0, nframe, 8):
            b = min(8, nframe - i)
            rast.rasterise(lms[i:i + b], shs[i:i + b], out=maps[:b])                            # edge maps drawn on the device
            eng.forward_image(maps[:b], cand, frames[:b])
        return mouth, poses

    clip(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); clip(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    if a2h.status() != 0 or a2f.status() != 0 or apc._engine.status() != 0:
        raise RuntimeError("a hand-off kernel timed out")
    ts.sort()
    return {"label": "plumbing only: synthetic weights, stand-in feature maps, device stages only (see bench.py pipeline_extra)",
            "clip_frames": nframe, "seconds_median": round(ts[1], 4), "clip_frames_per_s": round(nframe / ts[1], 1),
            "x_realtime_at_60fps": round((nframe / 60.0) / ts[1], 2)}


def recurrent_extra(dev, cpu_threads):
    """the two recurrent audio stages at the 687-frame clip's lengths -- APC_encoder
    (3 x GRU-512 over 1374 mel frames, demo.py:186-191) and Audio2Feature (MLP + 3 x LSTM-256 + MLP over
    687 + frame_future 18 = 705 steps, demo.py:205).  Device tensors in/out; two lengths give the per-step slope
    (all three layers advance together in the wavefront kernel, so a step is one time step of the whole stack)."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.apc import APC_encoder
    from livespeechportraits_amd.audio2feature import Audio2Feature
    import argparse as _ap

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    out = {}
    apc = APC_encoder(80, 512, 3, False)
    apc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_apc_state_dict().items()})
    apc = apc.to(dev).eval()
    ms = {}
    for T in (274, 1374):
        mel = torch.from_numpy(synth.make_mel(T)).to(dev).unsqueeze(0)
        ms[T] = timed(lambda: apc.forward(mel, torch.Tensor([T])))
    if apc._engine.status() != 0:
        raise RuntimeError("GRU kernel hand-off timed out")
    out["apc_gru"] = {"ms_1374_steps": round(ms[1374], 3), "us_per_step": round(1e3 * (ms[1374] - ms[274]) / 1100.0, 3),
                      "mel_frames_per_s": round(1374 / (ms[1374] * 1e-3), 1)}
    opt = _ap.Namespace(feature_decoder="LSTM", loss="L2", A2L_GMM_ndim=75, A2L_GMM_ncenter=1, predict_length=1, APC_hidden_size=512)
    a2f = Audio2Feature(opt)
    a2f.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_a2f_state_dict().items()}, strict=False)
    a2f = a2f.to(dev).eval()
    ms2 = {}
    for rows in (410, 1410):
        x = torch.from_numpy(synth.symmetric(rows * 512, 0.5, 77).reshape(1, rows, 512)).to(dev)
        ms2[rows] = timed(lambda: a2f.forward(x))
    if a2f.status() != 0:
        raise RuntimeError("LSTM kernel hand-off timed out")
    out["audio2feature_lstm"] = {"ms_705_steps": round(ms2[1410], 3), "us_per_step": round(1e3 * (ms2[1410] - ms2[410]) / 500.0, 3),
                                 "frames_per_s": round(705 / (ms2[1410] * 1e-3), 1)}
    if cpu_threads:
        from oracle import rnn_oracle
        torch.set_num_threads(int(cpu_threads))
        sd, mel = synth.make_apc_state_dict(), synth.make_mel(1374)
        rnn_oracle.apc_forward(sd, mel[:64])
        t0 = time.perf_counter(); rnn_oracle.apc_forward(sd, mel); t_apc = time.perf_counter() - t0
        sd2, feats = synth.make_a2f_state_dict(), synth.symmetric(1410 * 512, 0.5, 77).reshape(1410, 512)
        rnn_oracle.a2f_forward(sd2, feats[:64])
        t0 = time.perf_counter(); rnn_oracle.a2f_forward(sd2, feats); t_a2f = time.perf_counter() - t0
        out["cpu_baseline"] = {"apc_gru_ms": round(1e3 * t_apc, 1), "audio2feature_ms": round(1e3 * t_a2f, 1), "cores": int(cpu_threads), "kind": "port",
                               "sample": "oracle/rnn_oracle.py (torch nn.GRU / nn.LSTM on the host, torch %s), the same full sequences, one run each" % torch.__version__}
    return out


def manifold_extra(dev, cpu_threads):
    """KNN + LLE projection of one clip's APC features (demo.py:196-200): 1374 feature rows
    (2 per video frame of the 687-frame clip), a 30 000-row synthetic database, d 512, K 10.  Device tensors in/out."""
    from livespeechportraits_amd import manifold, synth
    n, m, d, K = 1374, 30000, 512, 10
    db_np, q_np = synth.make_feature_database(m, n, d, 24)
    db, q = torch.from_numpy(db_np).to(dev), torch.from_numpy(q_np).to(dev)
    for _ in range(2):
        manifold.project(q, db, K, 1.0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); manifold.project(q, db, K, 1.0); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    out = {"metric": "feature rows/s (KNN_with_torch + compute_LLE_projection_all_frame + blend, n 1374, m 30000, d 512, K 10)",
           "value": round(n / (ts[2] * 1e-3), 1), "ms_median": round(ts[2], 3), "ms_min_max": [round(ts[0], 3), round(ts[-1], 3)],
           "note": "per-kernel split in profiles/r01_kernel_stats_side_paths.txt (distance GEMM ~0.51 ms = 83 TFLOP/s, top-K 0.17, LLE 0.14)"}
    if cpu_threads:
        from oracle import lle_oracle
        torch.set_num_threads(int(cpu_threads))
        t0 = time.perf_counter()
        ind = lle_oracle.knn(q_np, db_np, K)
        t1 = time.perf_counter()
        lle_oracle.lle_all(q_np[:300], db_np, ind[:300])
        t2 = time.perf_counter()
        total = (t1 - t0) + (t2 - t1) * n / 300.0
        out["cpu_baseline"] = {"value": round(n / total, 1), "unit": "feature rows/s", "cores": int(cpu_threads), "kind": "port",
                               "sample": "oracle/lle_oracle.py: full KNN (%.3f s) + the per-frame LLE loop on 300 of %d rows (%.3f s, scaled)"
                                         % (t1 - t0, n, t2 - t1)}
    return out


if __name__ == "__main__":
    main()
