#!/usr/bin/env python3
"""Headline benchmark: images/sec of one mFormerV1_sm training step (forward + loss + backward
[+ gradient all-reduce] + AdamW) on synthetic 3x224x224 batches, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` is the whole-job images/sec with inputs resident in HBM.
Beside it: `roofline` for the dominant kernel class (the MFMA forward/data-gradient GEMM),
measured live with HIP events on the launch stream in extra, untimed steps; and
`cpu_baseline`, the CPU oracle (oracle/, a port of the reference's arithmetic) timed on this
host's cores on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

TASKS = (("taxa_L10", 1000), ("taxa_L20", 300), ("taxa_L30", 80), ("taxa_L40", 20))
FLOP_PER_IMG = 25.79e9          # fwd+bwd FLOPs per image, mFormerV1_sm @224
PEAK_BF16_TFLOPS = 2500.0       # dense MFMA peak (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0        # dense fp8 MFMA peak: what an `--dtype fp8` line's roofline is priced against
PEAK_HBM_GBS = 8000.0
MEASURED_HBM_COPY_GBS = 5000.0   # what a plain streaming copy gets from this HBM (read + written bytes; tools/ubench/hbm_rw.hip: 4.8-5.3 TB/s)


def log(msg):
    print(f"[bench +{time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


T0 = time.perf_counter()


def host_cores():
    """CPU threads this process may actually use: min(affinity, cgroup quota), capped by LNX_CPU_CORES
    (default 16, the GPU box's per-GPU CPU share) -- os.cpu_count() reports the whole host."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("LNX_CPU_CORES", "16"))))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None,
                    help="per-GPU batch.  Default: 256 at every --gpus N (BASELINE config 2's batch on every rank: weak scaling, the "
                         "per-GPU work does not change with N).  --batch 128 at --gpus 8 is BASELINE config 3's shape (global batch 1024)")
    ap.add_argument("--arch", default="sm")
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8"], help="fp8: bf16 plus MXFP8 forward products in the RoPE blocks (config 5)")
    ap.add_argument("--no-optim", action="store_true", help="time forward+backward only")
    ap.add_argument("--force-dp", action="store_true", help="single GPU rehearsal of the data-parallel path: RCCL world size 1, collectives issued")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="with --gpus N > 1 on a box with ONE GPU: all N ranks use cuda:0 and the collectives run over gloo (RCCL refuses two ranks on "
                         "one device).  Exercises every line of the N > 1 path with real ranks; the line it prints is marked as a rehearsal and is "
                         "NOT a throughput or scaling measurement")
    ap.add_argument("--torch-optim", action="store_true", help="torch.optim.AdamW(fused=True) instead of linnaeus_amd.optim.FusedAdamW")
    ap.add_argument("--drop-in", action="store_true",
                    help="time the model the way the reference's train.py:147-176,279-316 drives it: torch cross_entropy per task, "
                         "loss.backward(), clip_grad_norm_, torch.optim.AdamW.step, zero_grad (nothing from linnaeus_amd but the model)")
    ap.add_argument("--recompute", action="store_true",
                    help="gradient checkpointing on (TRAIN.GRADIENT_CHECKPOINTING: block inputs kept, activations recomputed in backward); "
                         "not the headline configuration")
    ap.add_argument("--host-input", action="store_true",
                    help="every step's batch starts in pinned HOST memory and is moved by linnaeus_amd.prefetch.DevicePrefetcher "
                         "(copy stream, two batches in flight): the PCIe-inclusive rate -- reported in DESIGN.md, never the headline")
    ap.add_argument("--flat-file", action="store_true",
                    help="with the PCIe-inclusive rate of --host-input, but from the synthetic flat-file reader (linnaeus_amd.flatdata: config 2's "
                         "'synthetic HDF5' without h5py): memory-mapped uint8 images collated by a background thread, uint8 over PCIe, converted to "
                         "float NCHW on the device -- reported in DESIGN.md, never the headline")
    ap.add_argument("--eval", action="store_true", help="inference throughput in the reference's throughput_test protocol instead of the training step")
    ap.add_argument("--eval-batches", default="64,128,256,512")
    ap.add_argument("--eval-iters", type=int, default=100)
    ap.add_argument("--eval-warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sched-calibration", action="store_true",
                    help="skip the warm-up comparison of the two backward schedules (weight-gradient stream on / off) and keep the library default (on)")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--profile-steps", type=int, default=3)
    return ap.parse_args()


def make_model(args):
    from linnaeus_amd import arch_config, build_model

    cfg = arch_config(args.arch, args.img)
    cfg.DATA.TASK_KEYS_H5 = [t for t, _ in TASKS]
    cfg.MODEL.CLASSIFICATION.HEADS = {t: {"TYPE": "Linear"} for t, _ in TASKS}
    cfg.TRAIN.GRADIENT_CHECKPOINTING.ENABLED_NORMAL_STEPS = False
    model = build_model(cfg, num_classes={t: c for t, c in TASKS})
    return cfg, model


def committed_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (tools/profile_round.sh -> profiles/rNN_<tag>_gemm_nt_traffic.json): the file of the HIGHEST round, its "final" pass if
    there is one, else the newest by modification time.  Returns (bytes_per_launch, file name)."""
    import re

    d = os.path.join(REPO, "profiles")
    best = None
    for f in os.listdir(d) if os.path.isdir(d) else []:
        m = re.match(r"r(\d+)_(.*)_gemm_nt_traffic\.json$", f)
        if m:
            key = (int(m.group(1)), m.group(2) == "final", os.path.getmtime(os.path.join(d, f)))
            if best is None or key > best[0]:
                best = (key, f)
    if best is None:
        return None, None
    with open(os.path.join(d, best[1])) as fh:
        return json.load(fh)["bytes_per_launch"], "profiles/" + best[1]


def eval_throughput(args):
    """`--eval`: inference images/sec measured the way the reference's own tool measures it (evaluation/throughput_tester.py:56-90,
    inputs as evaluation/synthetic_data.py:6-22): model in eval mode under no_grad, uniform-random images and metadata resident on
    the device, `--eval-warmup` untimed calls then `--eval-iters` timed calls between two device synchronisations, once per batch
    size; images/sec = batch * iterations / seconds.  (The reference's function itself runs unchanged on this model wherever the
    reference is importable; this is the same protocol for the GPU box, where it is not.)  One JSON line; `value` = best batch."""
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    _, model = make_model(args)
    model = model.to(dev).eval()
    model.set_compute_dtype(args.dtype)
    meta_width = sum(model.meta_dims)
    rows = []
    with torch.no_grad():
        for bs in (int(b) for b in args.eval_batches.split(",")):
            x = torch.rand(bs, 3, args.img, args.img, device=dev)
            meta = torch.rand(bs, meta_width, device=dev) if meta_width else None
            for _ in range(args.eval_warmup):
                model(x, meta)
            torch.cuda.reset_peak_memory_stats(dev)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.eval_iters):
                model(x, meta)
            torch.cuda.synchronize(dev)
            sec = time.perf_counter() - t0
            rows.append({"batch_size": bs, "imgs_per_sec": round(bs * args.eval_iters / sec, 3), "ms_per_call": round(sec / args.eval_iters * 1e3, 3),
                         "torch_peak_alloc_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 3)})
            log(f"eval batch {bs}: {rows[-1]['imgs_per_sec']:.0f} img/s")
    best = max(rows, key=lambda r: r["imgs_per_sec"])
    print(json.dumps({
        "metric": f"images/sec (eval forward, no_grad) mFormerV1_{args.arch} 3x{args.img}x{args.img}", "value": best["imgs_per_sec"],
        "unit": "images/sec", "n_gpus": 1, "higher_is_better": True, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"throughput_test protocol of the reference (evaluation/throughput_tester.py:56-90): eval mode, no_grad, "
                               f"{args.eval_warmup} warm-up + {args.eval_iters} timed iterations per batch size, inference plans",
                   "best_batch": best["batch_size"]},
        "results": rows}), flush=True)


# |GPU loss - oracle loss| allowed by `loss_check` (sum of four batch-mean cross-entropies, ~20): 3x what was measured on the MI355X
LOSS_CHECK_TOL = {"fp32": 2e-3, "bf16": 0.05, "fp8": 0.25}


def cpu_baseline(args, cfg, gpu_loss=None):
    """Time the CPU oracle (fp32, all host cores) on a bounded sample of the same workload.  `gpu_loss(sd, x, meta, tg)`: the HIP model's
    loss at the oracle's weights and inputs (DropPath off) -- compared with the oracle's own loss on them (`loss_check`), so that the run
    that prints a throughput has also shown, in the same process, that the timed path computes the reference's numbers."""
    from oracle import mformer_oracle as O

    cores = host_cores()
    torch.set_num_threads(cores)
    spec = O.Spec(heads=TASKS, drop_path_rate=0.0)
    sd = {k: v.requires_grad_(True) for k, v in O.seeded_state_dict(O.param_shapes(spec), 1).items()}
    g = torch.Generator().manual_seed(42)
    B = args.cpu_batch
    x = torch.rand(B, 3, args.img, args.img, generator=g)
    meta = torch.rand(B, 5, generator=g)
    tg = {t: torch.randint(1, c, (B,), generator=g) for t, c in TASKS}

    seen = {}

    def step():
        for v in sd.values():
            v.grad = None
        out = O.forward(sd, spec, x, meta)
        loss = sum(F.cross_entropy(out[t], tg[t]) for t, _ in TASKS)
        loss.backward()
        seen["loss"] = float(loss.detach())

    def fwd():
        with torch.no_grad():
            O.forward(sd, spec, x, meta)

    # 3 warm-up + 10 timed steps, forward-only and forward+backward
    warm, n = 3, 10
    res = {}
    for name, fn in (("fwd", fwd), ("fwd_bwd", step)):
        for _ in range(warm):
            fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        res[name] = B * n / (time.perf_counter() - t0)
    ref = None
    rp = os.path.join(REPO, "profiles", "r03_reference_cpu_timing.json")  # the imported reference timed in the build container
    if os.path.exists(rp):
        with open(rp) as fh:
            ref = json.load(fh)
    check = None
    if gpu_loss is not None:
        got = gpu_loss({k: v.detach() for k, v in sd.items()}, x, meta, tg)
        tol = LOSS_CHECK_TOL[args.dtype]
        check = {"gpu": round(got, 5), "oracle": round(seen["loss"], 5), "abs_diff": round(abs(got - seen["loss"]), 5), "tolerance": tol, "batch": B,
                 "what": f"sum over the 4 tasks of the batch-mean cross-entropy at the oracle's seeded weights and inputs, DropPath off: the HIP plan ({args.dtype}) "
                         "on cuda:0 against the fp32 CPU oracle, same process, after the timed legs"}
        if not abs(got - seen["loss"]) <= tol:
            raise RuntimeError(f"loss_check failed: {check}")
    return {"value": round(res["fwd_bwd"], 3), "unit": "images/sec", "cores": cores, "kind": "port", "fwd_only_images_per_sec": round(res["fwd"], 3),
            "sample": f"oracle fp32, batch {B}, {warm} warm-up + {n} timed steps each of forward-only and forward+loss+backward (value = the latter), {args.img}x{args.img}",
            "reference_in_build_container": ref, "loss_check": check}


def live_profile(args, model, state, loss_fn, ips_per_gpu):
    """Untimed extra steps on rank 0 with HIP events on the launch stream.  Pass 1: an event pair around every launch of the timed
    kernel classes (time, FLOPs, algorithmic HBM bytes per class) -- with the backward's weight-gradient stream switched OFF
    (lnx_plan_set_wgrad_stream), so that a kernel's duration is its own and not its share of two kernels running side by side.
    Pass 1b: the same with the stream on, as the timed steps run (gemm_nt / gemm_tn only: `overlapped`).  Pass 2: an event pair around
    every whole RoPE block and nothing inside it, stream on (what the attention + MLP blocks cost in the step -> north-star's own fraction)."""
    from linnaeus_amd import _lib as L

    lib = L.lib()
    NC = 10
    x, meta, tg = state["x"], state["meta"], state["tg"]
    hook, model._segment_hook = model._segment_hook, None  # no collectives in the profiled steps
    model.zero_grad(set_to_none=True)
    loss_fn(model(x, meta), tg).backward()  # makes this leg's plan the active one (another leg may have run since)
    st = model._active

    def run_steps(begin):
        L.check(begin(st["handle"]), "profile_begin")
        for _ in range(args.profile_steps):
            model.zero_grad(set_to_none=True)
            loss_fn(model(x, meta), tg).backward()
        ms, work, byts, cnt = (C.c_double * NC)(), (C.c_double * NC)(), (C.c_double * NC)(), (C.c_int * NC)()
        L.check(lib.lnx_plan_profile_end_ex(st["handle"], ms, work, byts, cnt), "profile_end_ex")
        return ms, work, byts, cnt

    was = lib.lnx_plan_set_wgrad_stream(st["handle"], 0)
    ms, work, byts, cnt = run_steps(lib.lnx_plan_profile_begin)
    if was < 0 or lib.lnx_plan_set_wgrad_stream(st["handle"], was) < 0:
        L.check(1, "lnx_plan_set_wgrad_stream")
    oms, _, _, ocnt = run_steps(lib.lnx_plan_profile_begin) if was == 1 else (None, None, None, None)
    sms, swork, _, scnt = run_steps(lib.lnx_plan_profile_begin_spans)
    model._segment_hook = hook
    n = args.profile_steps
    if cnt[0] == 0 or ms[0] <= 0:
        return None, {}, None
    kernels = {}
    names = ["gemm_nt", "gemm_tn", "attn_fwd", "attn_bwd", "dwconv7", "dwconv7_wgrad", "convmlp_fwd", "convmlp_bwd"]
    for i, nm in enumerate(names):
        if cnt[i] == 0:
            continue
        per = {"ms_per_step": round(ms[i] / n, 4), "launches_per_step": cnt[i] // n, "avg_launch_us": round(ms[i] * 1e3 / cnt[i], 2)}
        if i < 4 or i >= 6:
            per["tflops"] = round(work[i] / (ms[i] * 1e-3) / 1e12, 2)
        else:
            per["gbs"] = round(work[i] / (ms[i] * 1e-3) / 1e9, 1)
        if byts[i] > 0:
            per["algorithmic_gb_per_step"] = round(byts[i] / n / 1e9, 3)
            per["algorithmic_gbs"] = round(byts[i] / (ms[i] * 1e-3) / 1e9, 1)
        kernels[nm] = per
    if oms is not None:
        # the same classes as the timed steps run them: weight-gradient products beside the data-gradient chain (each kernel's span then
        # covers time it shares with the other stream's kernel -- the sum over classes exceeds the wall time)
        kernels["overlapped"] = {"what": "per-launch HIP-event time with the weight-gradient stream ON (the timed steps' mode); the classes above are "
                                         "timed with it OFF (lnx_plan_set_wgrad_stream(plan, 0)): a kernel alone on the chip",
                                 "gemm_nt_ms_per_step": round(oms[0] / n, 4), "gemm_tn_ms_per_step": round(oms[1] / n, 4)}
    for i, nm in ((8, "rope_block_spans"), (9, "conv_block_spans")):
        if scnt[i]:
            kernels[nm] = {"ms_per_step": round(sms[i] / n, 4), "spans_per_step": scnt[i] // n, "tflops": round(swork[i] / (sms[i] * 1e-3) / 1e12, 2)}
    # ---- the dominant class: forward + data-gradient GEMMs.  Two roofs, both stated; `bound` is the one its own FLOP/byte puts it under.
    a_tf = work[0] / (ms[0] * 1e-3) / 1e12
    alg_b = byts[0] / cnt[0]                      # algorithmic bytes per launch
    intensity = work[0] / byts[0]                 # FLOP per algorithmic byte
    # an fp8 line is priced against the fp8 peak: its RoPE-block forward products and their data gradients run on MXFP8 operands
    peak_tf = PEAK_FP8_TFLOPS if args.dtype == "fp8" else PEAK_BF16_TFLOPS
    ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
    # HBM bytes per launch from the PMC counters: they cannot be read from inside this process, so the figure comes from the
    # committed rocprofv3 --pmc passes of this same command (profiles/*_gemm_nt_traffic.json); null for other workloads
    traffic = traffic_src = None
    if args.arch == "sm" and state["x"].shape[0] == 256 and args.dtype == "bf16" and args.img == 224:
        traffic, traffic_src = committed_traffic()
    launch_s = ms[0] * 1e-3 / cnt[0]
    hbm = {"algorithmic_bytes_per_launch": round(alg_b), "counter_bytes_per_launch": traffic, "flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
           "achieved_tbs_algorithmic": round(alg_b / launch_s / 1e12, 3),
           "achieved_tbs_counter": round(traffic / launch_s / 1e12, 3) if traffic else None,
           "frac_of_spec": round((traffic or alg_b) / launch_s / (PEAK_HBM_GBS * 1e9), 4),
           "frac_of_measured": round((traffic or alg_b) / launch_s / (MEASURED_HBM_COPY_GBS * 1e9), 4),
           "spec_gbs": PEAK_HBM_GBS, "measured_copy_gbs": MEASURED_HBM_COPY_GBS, "measured_copy_source": "tools/ubench/hbm_rw.hip, profiles/r03_hbm_rw.log",
           "mfma_frac_ceiling_at_spec_hbm": round(min(1.0, intensity / ridge), 3),
           "mfma_frac_ceiling_at_measured_hbm": round(min(1.0, intensity * MEASURED_HBM_COPY_GBS * 1e9 / (peak_tf * 1e12)), 3)}
    roofline = {"bound": "hbm" if intensity < ridge else "mfma",
                "kernel": f"gemm_nt class <{args.dtype}>: forward + data-gradient GEMMs with M >= 1024 (gemm_nt_v2 / v4 one-shot and gemm_nt_v7 / v9 persistent LDS-DMA kernels, fused epilogues)",
                "achieved": round(a_tf, 2), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(a_tf / peak_tf, 4),
                "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": kernels["gemm_nt"]["avg_launch_us"],
                "launches_per_step": kernels["gemm_nt"]["launches_per_step"], "flops_per_step": work[0] / n, "hbm": hbm,
                "note": f"frac = achieved / dense {'fp8' if args.dtype == 'fp8' else 'bf16'} MFMA peak.  The class's FLOP per algorithmic byte is below the ridge, so by its own bytes it is HBM-bound: "
                        "its MFMA fraction cannot exceed hbm.mfma_frac_ceiling_* at the stated bandwidths",
                "measured_with": "HIP events around every launch of the class, extra untimed steps, weight-gradient stream OFF (each kernel alone on the chip: "
                                 "rocprofv3 --kernel-trace of `LNX_WGRAD_STREAM=0 python3 bench.py ...` agrees, profiles/*_kernel_stats.csv).  In the timed steps the "
                                 "weight-gradient products run beside these kernels on a second stream: kernels.overlapped has the class's time in that mode"}
    if oms is not None and ocnt[0]:
        roofline["overlapped_avg_launch_us"] = round(oms[0] * 1e3 / ocnt[0], 2)
    rope = None
    if scnt[8]:
        # north-star: ">= 60 % of MFMA peak on the attention + MLP blocks".  FLOPs = the plan's own count for the RoPE blocks
        # (forward + 2x backward; BASELINE.md section 2: 15.94 GFLOP/img for sm @224), time = the block spans of pass 2
        tf = swork[8] / (sms[8] * 1e-3) / 1e12
        rope = {"frac": round(tf / peak_tf, 4), "peak_tflops": peak_tf, "achieved_tflops": round(tf, 2), "ms_per_step": round(sms[8] / n, 3),
                "gflop_per_image": round(swork[8] / n / state["x"].shape[0] / 1e9, 3), "blocks_timed_per_step": scnt[8] // n,
                "what": "HIP-event spans around every whole RoPE2DMHSABlock forward and backward on the launch stream (LayerNorms, qkv, attention, proj, Mlp, "
                        "weight gradients, reduces; no events between the kernels), untimed extra steps"}
    return roofline, kernels, rope


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks ourselves, one process per GPU, as
    the driver's own command line does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py ...`).  This parent never touches the GPU (no HIP call, no torch.cuda call) and never execs: it
    runs the launcher as a CHILD, relays rank 0's JSON line on stdout and exits with the child's return code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {n} without WORLD_SIZE: launching {' '.join(cmd[1:9])} ...")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's cross-process buffer sharing needs it on this stack
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:  # ranks print nothing else on stdout; keep the last JSON object in case a library does
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            line = ln
        elif ln:
            print(ln, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    sys.exit(rc if rc != 0 or line is not None else 1)


def legs_for(n_gpus, batch):
    """What `bench.py --gpus N` times.  [(name, per-GPU batch)], the first leg is the line's `value`.
    N = 1: BASELINE config 2 (256 images) is `value`; a second leg, `config3_n1`, times config 3's per-GPU shape (128 images) on this one
    GPU in the same run, so that the N > 1 lines -- quoted at 128 images per GPU -- have an N = 1 denominator at THEIR batch (VERDICT r4
    item 3).  N > 1: BASELINE config 3 / BASELINE.md section 3's shape -- 128 images per GPU, global batch 128 N (1024 at 8 GPUs) -- is
    `value`; the 256-per-GPU weak leg is timed in the same run and reported beside it (`weak256`).  --batch B: that one batch at any N."""
    if batch is not None:
        return [("batch", batch)]
    return [("config2", 256), ("config3_n1", 128)] if n_gpus == 1 else [("config3", 128), ("weak256", 256)]


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU (python bench.py --gpus N starts them itself)")
    if args.eval:
        return eval_throughput(args)
    legs = legs_for(2 if (args.force_dp and world == 1) else world, args.batch)  # --force-dp: the N > 1 legs rehearsed on one GPU
    if world == 1 and not args.force_dp and len(legs) > 1 and (args.arch != "sm" or args.img != 224 or args.host_input or args.flat_file or args.drop_in or args.recompute):
        legs = legs[:1]  # config3_n1 belongs to the headline workload only
    rehearsal = bool(args.rehearse_one_gpu and world > 1)
    if rehearsal:
        local = 0  # every rank on the one card
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # stdout carries the one JSON line and nothing else: anything a library prints on the way (RCCL's version banner goes to
    # stdout through printf) is sent to stderr by pointing fd 1 at fd 2 until the line is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    dist = None
    if world > 1 or args.force_dp:
        import torch.distributed as dist

        os.environ.setdefault("NCCL_DEBUG", "VERSION")  # one line on rank 0: the RCCL build that carries the collectives
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29544")
            dist.init_process_group("nccl", rank=0, world_size=1)
        elif rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl")
        # No device_id= here on purpose: binding the process group to the device at init (eager communicator creation) makes
        # EVERY step ~1 ms slower on this stack (13.6 -> 14.6 ms at batch 128, with no collective ever issued;
        # tools/bench_segments.py --pg [--lazy]); the lazily created communicator does not.  torch.cuda.set_device() above
        # tells RCCL which GPU this rank owns; barriers name it explicitly.

    def barrier():
        if rehearsal:
            dist.barrier()  # (gloo takes no device_ids)
        else:
            dist.barrier(device_ids=[local])

    torch.manual_seed(42 + rank)
    cfg, model = make_model(args)
    model = model.to(dev)
    model.set_compute_dtype(args.dtype)
    model.train()
    model.use_checkpoint = bool(args.recompute)
    net = model
    if world > 1 or args.force_dp:
        from linnaeus_amd.ddp import DataParallel

        net = DataParallel(model, single_rank_collectives=args.force_dp)
    elif not (args.drop_in or args.torch_optim):
        # gradients written straight into the flat arena the fused optimizer reads.  The reference's own glue (--drop-in) and a
        # torch optimizer get the model's default grad_mode ("autograd": one arena copy per backward + AccumulateGrad), which
        # is what a drop-in user pays
        model.grad_mode = "direct"
    if args.no_optim:
        opt = None
    elif args.drop_in:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05)  # what optimizers/build.py builds for OPTIMIZER.NAME adamw
    elif args.torch_optim:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05, fused=True)
    else:
        from linnaeus_amd.optim import FusedAdamW

        opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05)  # one multi-tensor HIP launch per step

    from linnaeus_amd.loss import multitask_cross_entropy

    params = [p_ for p_ in model.parameters() if p_.requires_grad]
    cleanup = []

    def make_step(B):
        """Inputs of one leg (resident in HBM unless --host-input / --flat-file) and the step function over them."""
        g = torch.Generator(device=dev).manual_seed(42 + rank)
        state = {"x": torch.rand(B, 3, args.img, args.img, device=dev, generator=g), "meta": torch.rand(B, 5, device=dev, generator=g),
                 "tg": {t: torch.randint(1, c, (B,), device=dev, generator=g) for t, c in TASKS}}
        feed = None
        if args.flat_file:
            import queue
            import tempfile
            import threading

            from linnaeus_amd.aug import u8hwc_to_f32chw
            from linnaeus_amd.flatdata import FlatBatchLoader, FlatSyntheticDataset, write_synthetic_flat
            from linnaeus_amd.prefetch import DevicePrefetcher

            fd, path = tempfile.mkstemp(prefix=f"lnx_bench_{rank}_", suffix=".flat")
            os.close(fd)
            write_synthetic_flat(path, 4 * B, args.img, dict(TASKS), meta=(("TEMPORAL", 2), ("SPATIAL", 3)), seed=42 + rank, null_fraction=0.0)
            ds = FlatSyntheticDataset(path)
            os.unlink(path)  # the memory map keeps the pages; nothing is left in the temp dir whatever happens next
            q = queue.Queue(maxsize=4)
            stop = threading.Event()

            def produce():  # the reference's loader workers: read + collate off the training thread, straight into pinned memory
                for b in FlatBatchLoader(ds, B, shuffle=True, seed=rank, raw_uint8=True, epochs=None, workers=4, pin=True):
                    while not stop.is_set():
                        try:
                            q.put(b, timeout=0.2)
                            break
                        except queue.Full:
                            pass
                    if stop.is_set():
                        return

            th = threading.Thread(target=produce, daemon=True)
            th.start()

            def stop_producer():  # joined before the interpreter (and the HIP runtime its pinned allocations use) shuts down
                stop.set()
                while th.is_alive():
                    try:
                        q.get_nowait()
                    except queue.Empty:
                        pass
                    th.join(timeout=0.2)

            cleanup.append(stop_producer)

            def drain():
                while True:
                    yield q.get()

            def to_step(it):
                for raw, onehot, aux, _masks, _gids in it:
                    yield u8hwc_to_f32chw(raw), aux, {t: onehot[t].argmax(-1) for t, _ in TASKS}

            feed = to_step(iter(DevicePrefetcher(drain(), dev)))
        elif args.host_input:
            from linnaeus_amd.prefetch import DevicePrefetcher

            host = [(torch.rand(B, 3, args.img, args.img).pin_memory(), torch.rand(B, 5).pin_memory(),
                     {t: torch.randint(1, c, (B,)).pin_memory() for t, c in TASKS}) for _ in range(3)]

            def cycle():
                while True:
                    yield from host

            feed = iter(DevicePrefetcher(cycle(), dev))

        def step():
            if feed is not None:
                state["x"], state["meta"], state["tg"] = next(feed)
            x, meta, tg = state["x"], state["meta"], state["tg"]
            if args.drop_in:
                # the reference's step glue (train.py:147-176,279-316) around the drop-in model, torch ops only
                out = net(x, meta)
                loss = sum(F.cross_entropy(out[t].float(), tg[t]) for t, _ in TASKS)
                loss.backward()
                torch.nn.utils.clip_grad_norm_(params, 1.0)
                if opt is not None:
                    opt.step()
                    opt.zero_grad(set_to_none=True)
                return loss
            model.zero_grad(set_to_none=True)
            out = net(x, meta)
            loss = multitask_cross_entropy(out, tg)  # sum over the 4 tasks of the batch-mean CE: one HIP launch for all tasks
            loss.backward()
            if opt is not None:
                opt.step()
            return loss

        return step, state

    def run_leg(name, B):
        """W untimed + EXACTLY K timed steps between barrier + device synchronisation on both sides, MAX over ranks; then (data
        parallel only) the same steps without the gradient collectives, outside the timed region."""
        step, state = make_step(B)
        log(f"leg {name}: batch {B}/GPU on {dev}; warm-up {args.warmup} steps")
        for _ in range(args.warmup):
            step()
        # Backward schedule, chosen during warm-up (untimed): the library's default runs the weight-gradient products on a second stream
        # beside the data-gradient chain (+1...3 % on one GPU); whether that still pays beside a real collective's streams cannot be
        # rehearsed on one GPU, so both schedules are timed here over a few steps -- all ranks, the slowest rank counts -- and the
        # default is kept unless the plain schedule is more than 1 % faster.  Same gradients either way (include/lnx.h).
        sched = None
        if not args.no_sched_calibration and os.environ.get("LNX_WGRAD_STREAM") != "0":
            def few(n=4):
                step()
                torch.cuda.synchronize()
                if dist:
                    barrier()
                t = time.perf_counter()
                for _ in range(n):
                    step()
                torch.cuda.synchronize()
                t = (time.perf_counter() - t) / n
                if dist:
                    tt = torch.tensor([t], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    t = tt.item()
                return t
            model.set_wgrad_stream(True)
            t_on = few()
            model.set_wgrad_stream(False)
            t_off = few()
            keep_on = t_on <= 1.01 * t_off
            model.set_wgrad_stream(keep_on)
            sched = {"weight_gradient_stream": "on" if keep_on else "off", "calibration_ms_on": round(t_on * 1e3, 3), "calibration_ms_off": round(t_off * 1e3, 3),
                     "what": "4 untimed steps each during warm-up, max over ranks; 'on' (the library default) unless 'off' is > 1 % faster"}
            # ... and the stream the metadata-head chains run on (round 5): their own side stream (single-GPU default), the weight-gradient stream
            # (DataParallel's default: one stream fewer beside the collective's) or the launch stream; the default is kept unless another
            # placement is more than 1 % faster
            if os.environ.get("LNX_NO_SIDE_STREAM") is None and os.environ.get("LNX_META_STREAM") is None:
                default_mode = 2 if dist else 1
                t_meta = {}
                for mode in ([2, 0] if dist else [1, 2, 0]):
                    model.set_meta_stream(mode)
                    t_meta[mode] = few(3)
                best = min(t_meta, key=t_meta.get)
                keep = default_mode if t_meta[default_mode] <= 1.01 * t_meta[best] else best
                model.set_meta_stream(keep)
                sched["metadata_stream"] = {"kept": {0: "launch stream", 1: "own side stream", 2: "weight-gradient stream"}[keep],
                                            "calibration_ms": {{0: "launch", 1: "side", 2: "weight_gradient"}[k_]: round(v_ * 1e3, 3) for k_, v_ in t_meta.items()}}
            step()
        torch.cuda.synchronize()
        if dist:
            barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        host_dt = time.perf_counter() - t0  # the host's share: Python + launch calls, returned before the GPU has finished
        torch.cuda.synchronize()
        if dist:
            barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = tt.item()
        res = {"name": name, "per_gpu_batch": B, "global_batch": B * world, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "images_per_sec": round(world * B * args.steps / dt, 2), "host_enqueue_ms_per_step": round(host_dt / args.steps * 1e3, 3),
               "loss": round(float(loss.item()), 4), "step": step, "state": state, "backward_schedule": sched}
        log(f"leg {name}: timed {args.steps} steps: {res['ms_per_step']:.2f} ms/step, {res['images_per_sec']:.0f} img/s")
        # ---- data parallel only: n1_equiv = what ONE GPU does alone at this per-GPU batch (the same steps under no_sync: no
        # gradient collective is issued), exposed_allreduce_ms = step time with collectives - without, and
        # scaling_efficiency = images/sec / (n_gpus * n1_equiv): all three from this one run, at this leg's batch
        if dist and hasattr(net, "no_sync"):
            k = max(3, min(args.steps, 10))
            with net.no_sync():
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(k):
                    step()
                torch.cuda.synchronize()
                dt_ns = time.perf_counter() - t1
            tt = torch.tensor([dt_ns], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ms_ns = tt.item() / k * 1e3
            n1 = B / (ms_ns * 1e-3)
            res.update({"ms_per_step_no_sync": round(ms_ns, 3), "exposed_allreduce_ms": round(res["ms_per_step"] - ms_ns, 3),
                        "n1_equiv_images_per_sec": round(n1, 2), "scaling_efficiency": round(res["images_per_sec"] / (world * n1), 4)})
        return res

    results = [run_leg(name, B) for name, B in legs]
    head = results[0]
    B = head["per_gpu_batch"]
    step, state = head["step"], head["state"]
    ips = head["images_per_sec"]

    rccl = None
    if dist:
        # the world size the communicator itself reports, and proof that a collective saw that many contributions
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        rccl = {"world_size": dist.get_world_size(), "allreduce_of_ones": int(one.item()), "backend": dist.get_backend(),
                "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
                "nccl_env": {k_: v_ for k_, v_ in os.environ.items() if k_.startswith(("NCCL_", "RCCL_"))}}

    # ---- data parallel: stream budget and per-bucket issue -> done times (3 untimed steps with the reducer's telemetry on, all ranks) ----
    dp_diag = None
    if dist and hasattr(net, "bucket_report"):
        net.telemetry = True
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        barrier()
        net.telemetry = False
        dp_diag = {"buckets": net.bucket_report(), "stream_budget": net.stream_budget(),
                   "what": "buckets: bytes all-reduced per backward segment and the time from issue (the segment's kernels enqueued) to done (the collective's end "
                           "event), mean over 3 untimed steps, this rank; stream_budget: every HIP stream a step issues work on"}
        step()
        torch.cuda.synchronize()

    # ---- live per-kernel-class timing (untimed extra steps, rank 0 only) ----
    roofline, kernels, rope = None, {}, None
    if head["backward_schedule"]:  # (a later leg's calibration may have left the other schedule set)
        model.set_wgrad_stream(head["backward_schedule"]["weight_gradient_stream"] == "on")
    if rank == 0 and args.profile_steps > 0:
        roofline, kernels, rope = live_profile(args, model, state, multitask_cross_entropy, ips / world)

    if dist:
        # every rank gets here before any communicator is torn down (rank 0 has just run its profiled steps alone)
        torch.cuda.synchronize()
        barrier()
    for fn in cleanup:
        fn()
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        log("kernel profile done; timing the CPU oracle")

        def gpu_loss(sd, x, meta, tg):
            # the timed model itself, at the oracle's weights (same names and shapes: Linear heads), DropPath multipliers switched off
            model.load_state_dict(sd, strict=True)
            model.train()
            n_drop = sum(len(st_) for st_ in model.stages[:2]) + 2 * sum(len(st_) for st_ in model.stages[2:])
            model._inject_drop = [None] * n_drop
            try:
                model.zero_grad(set_to_none=True)
                out = model(x.to(dev), meta.to(dev))  # the training plan the legs timed (train mode, gradients enabled)
                return float(multitask_cross_entropy(out, {t: v.to(dev) for t, v in tg.items()}).detach())
            finally:
                model._inject_drop = None

        cpu = cpu_baseline(args, cfg, gpu_loss if (args.arch == "sm" and not args.drop_in) else None)
        log("cpu baseline done")
    line = {
        "metric": f"images/sec (train fwd+bwd) mFormerV1_{args.arch} 3x{args.img}x{args.img}",
        "value": ips, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "host_enqueue_ms_per_step": head["host_enqueue_ms_per_step"],
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"mFormerV1_{args.arch} train step (forward + 4-task CE loss + backward"
                               f"{' + RCCL gradient all-reduce' if world > 1 else ''}{'' if args.no_optim else ' + AdamW (' + ('torch fused' if args.torch_optim else 'one HIP launch') + ')'}), "
                               f"{'bf16 operands with MXFP8 forward products in the RoPE blocks' if args.dtype == 'fp8' else args.dtype + ' operands'} / fp32 accumulate+residual, batch {B}/GPU, 3x{args.img}x{args.img} synthetic, "
                               "DropPath 0.2, gradient checkpointing off",
                   "baseline_config": {"config2": "BASELINE.json configs[1]: train fwd+bwd bf16, batch 256, 1 GPU",
                                       "config3": "BASELINE.json configs[2] / BASELINE.md section 3: 128 images per GPU, global batch 128 N (1024 at 8 GPUs)",
                                       "weak256": "256 images per GPU", "batch": "--batch given on the command line"}[head["name"]],
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}", "tasks": dict(TASKS), "grad_mode": model.grad_mode},
        # FLOP_PER_IMG is the sm @224 figure: other architectures / sizes report no whole-step fraction
        "step_mfma_roofline_frac": round(ips / world * FLOP_PER_IMG / (PEAK_BF16_TFLOPS * 1e12), 4) if (args.arch == "sm" and args.img == 224) else None,
        "rope_blocks_mfma_frac": rope["frac"] if rope else None, "rope_blocks": rope,
        "loss": head["loss"], "backward_schedule": head["backward_schedule"],
        "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
    }
    dp_keys = ("ms_per_step_no_sync", "exposed_allreduce_ms", "n1_equiv_images_per_sec", "scaling_efficiency")
    if rccl:
        line["rccl_ranks"] = rccl["world_size"]
        if dp_diag:
            rccl.update(dp_diag)
        line["rccl"] = rccl
    if "scaling_efficiency" in head:
        # value / (n_gpus * n1_equiv), both measured in THIS run at THIS leg's per-GPU batch: not value / (N * the N = 1 line's
        # value), which at N = 1 is quoted on 256 images
        line["scaling_efficiency"] = head["scaling_efficiency"]
        line["data_parallel"] = {k_: head[k_] for k_ in dp_keys}
    for extra in results[1:]:
        line[extra["name"]] = {k_: v_ for k_, v_ in extra.items() if k_ not in ("step", "state", "name")}
        if extra["name"] == "config3_n1":
            line["config3_n1"]["what"] = ("BASELINE config 3's per-GPU shape (128 images) on this one GPU, same run, same protocol (warm-up, calibration, K timed steps): the "
                                          "N = 1 denominator for the N > 1 lines' `value`, which is quoted at 128 images per GPU")
            line["config3_n1"]["per_image_rate_vs_config2"] = round(extra["images_per_sec"] / ips, 4)
    if args.recompute:
        line["config"]["workload"] = line["config"]["workload"].replace("gradient checkpointing off", "gradient checkpointing ON (recompute plan)")
        line["workspace_gb"] = round(model._active["ws"].numel() / 1e9, 2)
    if args.host_input:
        line["config"]["workload"] += "; inputs start in pinned host memory every step (DevicePrefetcher: PCIe-inclusive, NOT the headline)"
    if args.flat_file:
        line["config"]["workload"] += ("; inputs read every step from a memory-mapped synthetic flat file by a loader thread, uint8 over PCIe through "
                                       "DevicePrefetcher, uint8 -> float NCHW on the device (PCIe- and reader-inclusive, NOT the headline)")
    if rehearsal:
        line["rehearsal"] = (f"{world} ranks sharing ONE GPU, collectives over gloo (--rehearse-one-gpu): exercises the N > 1 code path with real ranks; "
                             "value / ms_per_step are NOT a throughput or scaling measurement")
        line["config"]["workload"] = line["config"]["workload"].replace("RCCL gradient all-reduce", "gloo gradient all-reduce (REHEARSAL on one GPU)")
    if args.drop_in:
        line["config"]["workload"] = line["config"]["workload"].replace("forward + 4-task CE loss + backward", "DROP-IN: torch CE + loss.backward() + clip_grad_norm_ + torch.optim.AdamW (reference train.py glue)")
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
