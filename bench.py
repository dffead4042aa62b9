#!/usr/bin/env python
"""Benchmark of the iPOKE second-stage train step on MI355X (BASELINE.json metric: video-frames/sec).

    python bench.py --gpus N --steps K --warmup W            # N = 1: in process; N > 1: re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
                                                             # the same job launched by hand: one rank per GPU over RCCL

One *step* = the reference's second-stage optimisation step on one synthetic batch that is
already resident in HBM: frozen poke/image/motion encoders (no grad) -> flow forward -> FlowLoss -> flow backward ->
gradient all-reduce over RCCL (N > 1) -> fused Adam-amsgrad + weight-operand refresh.  Workload at N = 1 is
BASELINE.json configs[1]: plants_128 (z = 64, 16 x 3 x 128 x 128 clips, per-GPU batch 20, bf16 matrix-core inputs).
Rank 0 prints ONE JSON line; `value` is the whole-job frames/s (weak scaling: per-GPU batch fixed).

Extra objects in the line:
  roofline     -- the flow's dominant contraction (NICE 1x1 conv, a [B*64, 2048] x [2048, 2048] implicit GEMM) timed with
                  HIP events on its own launches; algorithmic FLOPs / launch over the dense bf16 MFMA peak.
  cpu_baseline -- the CPU oracle (oracle/, plain PyTorch fp32 restatement pinned to the reference's goldens) doing
                  the same step on a bounded sample of the same workload, on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

# The step keeps four streams busy at once (chain, weight gradients, optimizer, encoder prefetch) and the HIP runtime deals streams onto
# its hardware queues in creation order: with the default of 4 queues ONE more stream in the process (measured: an idle one) makes two of
# the busy four share a queue -- 53.8 instead of 51.3 ms per step.  RCCL brings its own streams at N > 1.  Eight queues measure the same
# as four on one GPU (52.1 / 52.1 vs 52.1 / 51.9 ms) and leave room; must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ipoke_amd import _lib, configs, dist as D, ops                      # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3       # f32-input matrix cores (v_mfma_f32_16x16x4_f32): 1/16 of the bf16 rate, same guide
FLOW_GFLOP = {32: 134.95304192, 64: 158.35070464}      # per sample forward
ENC_GFLOP = {128: 82.96, 64: 20.52}


def synthetic_batch(B, T, size, seed, device):
    """SURVEY.md §8d synthetic inputs, generated on the host generator then moved to HBM before the timed region."""
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(B, T, 3, size, size, generator=g) * 2 - 1
    flow = torch.randn(B, 2, size, size, generator=g)
    mask = (torch.rand(B, 1, size, size, generator=g) < 0.05).float()
    poke = [torch.randn(B, 2, size, size, generator=g) * mask, torch.zeros(B, 5, 2, dtype=torch.int64)]
    batch = {"images": images, "flow": flow, "poke": poke, "sample_ids": torch.zeros(B, T, dtype=torch.int64)}
    return {k: ([p.to(device) for p in v] if isinstance(v, list) else v.to(device)) for k, v in batch.items()}


def build_model(cfg, dtype, device, seed=0):
    from ipoke_amd.second_stage import PokeMotionModel
    torch.manual_seed(seed)
    conf = configs.second_stage_config(cfg["spatial_size"], cfg["z_dim"], cfg["n_frames"], cfg["batch_size"])
    model = PokeMotionModel(conf, dirs={}, dtype=dtype, device=device, max_batch=cfg["batch_size"])
    return model


def randomise_couplings(model, seed=0):
    """After the data-dependent init every coupling is the identity (zero-init weight norm).  Give the weight-norm
    gains small non-zero values so that gradients, optimizer state and clocks are those of a model in training."""
    g = torch.Generator(device=model.flow.flat_params.device).manual_seed(seed)
    with torch.no_grad():
        for name, p in model.flow.named_parameters():
            if name.endswith("weight_g"):
                p.copy_(0.02 + 0.01 * torch.rand(p.shape, device=p.device, generator=g))
    model.flow.mark_weights_updated()


PMC_ROUND = "r06"          # the committed PMC passes roofline.traffic is read from (NOT measured by this run: see traffic_source)


def gemm_source_hash():
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, "ipoke_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()[:16]


def pmc_traffic(kernel_substr, grid_size):
    """HBM-side bytes per launch from the committed in-situ PMC passes (separate rocprofv3 --pmc runs of this bench, summarised
    by scripts/pmc_summary.py): 2 x FETCH_SIZE (the gfx950 correction for 16-byte-per-lane streaming reads) + WRITE_SIZE, KB."""
    try:
        tot = 0.0
        for counter, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            rows = json.load(open(os.path.join(ROOT, "profiles", f"{PMC_ROUND}_bench_pmc_{counter}.json")))
            hit = [r for r in rows if kernel_substr in r["kernel"] and r["grid_size"] == grid_size and r["counter"] == counter]
            if not hit or hit[0].get("gemm_hip_sha16") != gemm_source_hash():      # counters of another build of the kernel: no figure
                return None
            tot += mul * hit[0]["mean"] * 1024.0
        return round(tot)
    except Exception:
        return None


def kernel_roofline(B, dtype, iters=50):
    """Time the dominant GEMM ([B*64,2048] x [2048,2048]^T, 1x1 conv + ELU) on its own launches with HIP events."""
    dev = "cuda"
    hid, M = 2048, B * 64
    td = ops.torch_dtype(dtype)
    a = torch.randn(M, hid, device=dev).to(td)
    w = (torch.randn(hid, hid, device=dev) / hid ** 0.5).to(td)
    c = torch.empty(M, hid, device=dev, dtype=td)
    d = ops.conv_desc(B, (1, 8, 8), (1, 8, 8), (1, 1, 1), (1, 1, 1), (0, 0, 0))
    d.A = a.data_ptr(); d.a_sn = 64 * hid; d.a_sh = 8 * hid; d.a_sw = hid; d.a_sc = 1; d.Kc_real = hid; d.Kc = hid
    d.W = w.data_ptr(); d.ldw = hid; d.Nout = hid; d.act = _lib.ACT_ELU; d.C = c.data_ptr(); d.ldc = hid
    from ctypes import byref
    dt, stream = ops._dt(dtype), _lib.current_stream()
    _lib.check(_lib.lib().ipoke_conv_forward_repeat(byref(d), dt, 5, stream))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()                       # recorded on the current stream == the stream the kernels are launched on
    _lib.check(_lib.lib().ipoke_conv_forward_repeat(byref(d), dt, iters, stream))     # native back-to-back launches
    e1.record()
    torch.cuda.synchronize()
    avg_s = e0.elapsed_time(e1) * 1e-3 / iters
    flops = 2.0 * M * hid * hid
    achieved = flops / avg_s / 1e12
    # HBM-side bytes per launch of this kernel at this shape from the PMC passes committed under profiles/
    # (FETCH_SIZE x 2 -- the gfx950 correction for 16-byte-per-lane streaming reads -- plus WRITE_SIZE); null for other shapes
    traffic = None
    if M == 1280 and dtype == "bf16":
        traffic = pmc_traffic("igemm_nt_glds_kernel<bool _Accum, int, E, 4, 5, 2, 3, 1, true, 2>", str(256 * 512))
    peak = MFMA_BF16_PEAK_TFLOPS if dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "bytes/launch: 2 x FETCH_SIZE + WRITE_SIZE of this kernel's dispatches INSIDE the train step",
            "traffic_source": f"profiles/{PMC_ROUND}_bench_pmc_*.json: separate rocprofv3 --pmc passes of this command (not this run; PMC collection and timing cannot share a run)",
            "algorithmic_bytes_per_launch": int(2 * (M * hid + hid * hid + M * hid)),
            "kernel": "igemm_nt (NICE conv2 1x1, M=%d N=K=2048, %s)" % (M, dtype), "avg_launch_us": round(avg_s * 1e6, 2),
            "algorithmic_gflop_per_launch": round(flops / 1e9, 3)}


MCF_GFLOP = {32: 2.53, 64: 8.95}          # the 800 masked-conv flows of one sample, forward


def insitu_rooflines(run_step, B, z, dtype):
    """One extra (untimed) step with HIP events around every launch of the dominant kernel families, on the streams they
    run on, while the rest of the step -- side-stream weight gradients, optimizer slices -- runs as usual."""
    from ctypes import c_double, c_int
    L = _lib.lib()
    tags = (c_int * 5)(1, 2, 3, 4, 6)        # 6 = IPOKE_TAG_NN_SQUARE: the conv2 data gradient read from the K-major weight (igemm_nn_glds)
    counts, mean = (c_int * 5)(), (c_double * 5)()
    torch.cuda.synchronize()
    _lib.check(L.ipoke_timing_start())
    run_step()
    _lib.check(L.ipoke_timing_stop(tags, 5, counts, mean))
    hid, M = 2048, B * 64
    gemm_gf = 2.0 * M * hid * hid / 1e9
    unit_gf = B * MCF_GFLOP[z] / 200.0             # 200 MaCowUnits (4 flows each): mean over the 15 channel widths

    def entry(k, kernel, gflop, what):
        us = mean[k]
        ach = gflop / us * 1e3 if us > 0 else 0.0           # GFLOP / us = 1000 TFLOP/s
        return {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None, "kernel": kernel, "launches_in_step": int(counts[k]),
                "avg_launch_us": round(us, 2), "algorithmic_gflop_per_launch": round(gflop, 3), "measured": what}
    how = "HIP events around each launch on its own stream inside one full train step (in situ, side streams active)"
    split = counts[4] > 0        # bf16: the data gradient runs in its own kernel (K-major weight) and is reported on its own
    head = entry(0, f"igemm_nt_glds (NICE conv2 1x1 forward{'' if split else ' + data gradient'}, M={M} N=K=2048, {dtype})", gemm_gf, how)
    if split:
        both = (counts[0] * mean[0] + counts[4] * mean[4]) / (counts[0] + counts[4])
        head["forward_and_data_gradient_avg_launch_us"] = round(both, 2)          # the blended figure rounds 1-5 reported as `roofline`
        head["forward_and_data_gradient_frac"] = round(gemm_gf / both * 1e3 / MFMA_BF16_PEAK_TFLOPS, 4)
    return [head] + ([entry(4, f"igemm_nn_glds (NICE conv2 data gradient, weight read K-major, M={M} N=K=2048, {dtype})", gemm_gf, how)] if split else []) + [
            entry(1, f"igemm_tn_glds (NICE conv2 weight gradient, 2048x2048 over M={M}, {dtype}; side stream)", gemm_gf, how),
            entry(2, "macow_unit_fwd (4 masked-conv flows + 2 ActNorms per launch; mean over channel widths 8..64)", unit_gf, how),
            entry(3, "macow_unit_bwd (data path of the same unit; FLOPs counted as the forward's)", unit_gf, how)]


KERNEL_FAMILIES = {      # in-situ timing tags (include/ipoke_hip.h) -> what the family is
    1: "igemm_nt_glds, 1x1 square GEMM (NICE conv2 forward / data gradient)",
    2: "igemm_tn_glds, square weight gradient (NICE conv2)",
    5: "macow_unit_inv (4 masked-conv flows + 2 ActNorms inverted per launch, two samples per workgroup)",
    6: "igemm_nn_glds, 1x1 square GEMM with the weight read K-major (NICE conv2 data gradient)",
    17: "igemm_nt / igemm_nt_glds implicit-GEMM convolutions (all shapes that no stationary-input kernel takes)",
    18: "conv3x3_s8 (3x3 on the 8x8 latent, stationary input)",
    19: "conv3x3_halo (3x3, halo-staged 8x16 patches)",
    20: "conv3x3_halo16 (3x3 / 3x3x3 wide layers, halo-staged 16x16 patches x 128 channels)",
    21: "conv3x3_c64 (<= 64 output channels, filter resident in LDS, persistent workgroups)",
    22: "conv3x3_k8 (one 16-byte chunk of input channels: the data gradient of the decoder's last convolution; nothing staged)",
    32: "igemm_tn / igemm_tn_glds weight gradients (split-M slabs)",
}


def config_rooflines(run_step, dtype):
    """Per-configuration rooflines: ONE extra, untimed step with HIP events around every convolution, weight
    gradient and flow-unit launch on the stream it runs on (ipoke_timing_start_all); per kernel family the launches, the summed
    durations and the summed ALGORITHMIC work (FLOPs; input + weights + output bytes, each once).  Each family is priced against the
    roof that bounds it (the larger of flops / MFMA peak and bytes / 8 TB/s); the list is ordered by time -- its head is the
    configuration's dominant kernel."""
    from ctypes import c_double, c_int
    L = _lib.lib()
    ids = sorted(KERNEL_FAMILIES)
    n = len(ids)
    tags = (c_int * n)(*ids)
    counts, us, fl, by = (c_int * n)(), (c_double * n)(), (c_double * n)(), (c_double * n)()
    torch.cuda.synchronize()
    _lib.check(L.ipoke_timing_start_all())
    run_step()
    _lib.check(L.ipoke_timing_stop_ex(tags, n, counts, us, fl, by))
    peak_tf = MFMA_BF16_PEAK_TFLOPS if dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
    out = []
    for k, tag in enumerate(ids):
        if not counts[k] or us[k] <= 0.0:
            continue
        t = us[k] * 1e-6
        tf, gbs = fl[k] / t / 1e12, by[k] / t / 1e9
        mfma_bound = fl[k] / (peak_tf * 1e12) >= by[k] / 8e12
        e = {"bound": "mfma" if mfma_bound else "hbm", "achieved": round(tf if mfma_bound else gbs, 2), "peak": peak_tf if mfma_bound else 8000.0,
             "unit": "TFLOP/s" if mfma_bound else "GB/s", "frac": round((tf / peak_tf) if mfma_bound else (gbs / 8000.0), 4), "traffic": None,
             "kernel": KERNEL_FAMILIES[tag], "launches_in_step": int(counts[k]), "total_us_in_step": round(us[k], 1),
             "avg_launch_us": round(us[k] / counts[k], 2), "algorithmic_gflop_in_step": round(fl[k] / 1e9, 1),
             "algorithmic_mbytes_in_step": round(by[k] / 1e6, 1), "achieved_tflops": round(tf, 2), "achieved_gbs": round(gbs, 1),
             "measured": "HIP events around each launch on its own stream inside one full step of THIS configuration (in situ)"}
        out.append(e)
    out.sort(key=lambda e: -e["total_us_in_step"])
    return out


def usable_cores():
    """Host cores this process may actually use: CPU affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except Exception:
            pass
    return n


def cpu_baseline(cfg, clips=2, seed=1, timed_steps=3):
    """The CPU oracle doing the same optimisation step on `clips` clips of the same workload (bounded sample)."""
    from oracle import flow_ref, vae_ref
    from ipoke_amd.utils.detfill import deterministic_fill_
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    # identity-initialised couplings produce exact zeros / subnormals in the backward pass; without flush-to-zero the
    # x86 microcode path makes the same arithmetic ~25x slower (measured: 725 s vs 29 s per step on 8 cores)
    torch.set_flush_denormal(True)
    size, z, T = cfg["spatial_size"], cfg["z_dim"], cfg["n_frames"]
    t_build = time.time()
    fs = vae_ref.SpadeCondMotionModel(configs.first_stage_config(size, z, T)).eval()
    pe = vae_ref.FirstStageWrapper(configs.encoder2d_config(size, 2)).eval()
    ce = vae_ref.FirstStageWrapper(configs.encoder2d_config(size, 3)).eval()
    flow = flow_ref.SupervisedMacowTransformer(configs.flow_arch(z))
    for m, pfx in ((fs, "first_stage."), (pe, "poke_embedder."), (ce, "conditioner.")):
        deterministic_fill_(m, prefix=pfx)
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for k, v in flow.state_dict().items():      # the GPU leg's state: initialised flags, randomise_couplings() gains
            if k.endswith("initialized"):
                v.fill_(1)
            elif k.endswith("weight_g"):
                v.copy_(0.02 + 0.01 * torch.rand(v.shape, generator=gen))
            elif k.endswith("bias"):
                v.zero_()
            elif k.endswith("log_scale"):
                v.zero_()
    opt = torch.optim.Adam(flow.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-5, amsgrad=True)
    loss_fn = flow_ref.FlowLoss()
    batch = synthetic_batch(clips, T, size, seed, "cpu")
    t_build = time.time() - t_build

    def step():
        with torch.no_grad():
            poke_emb, *_ = pe.encoder(batch["flow"])
            cond, *_ = ce.encoder(batch["images"][:, 0])
            motion, mu, _ = fs.enc_motion(batch["images"].transpose(1, 2))
        opt.zero_grad()
        out, logdet = flow(motion.detach(), torch.cat([cond, poke_emb], 1))
        loss, _ = loss_fn(out, logdet)
        loss.backward()
        opt.step()

    t0 = time.time(); step(); t_warm = time.time() - t0            # warm-up (allocator, thread pools, first-touch of 15 GB)
    times = []
    for _ in range(timed_steps):
        t0 = time.time(); step(); times.append(time.time() - t0)
    times.sort()
    dt = times[len(times) // 2]
    return {"value": round(clips * T / dt, 4), "unit": "video-frames/sec", "cores": ncores, "kind": "port",
            "sample": f"{clips} clip(s) per step of the same workload (16x3x{size}x{size}, z={z}), same coupling initialisation as the GPU leg: "
                      f"encoders + flow fwd + FlowLoss + bwd + Adam-amsgrad, oracle/ PyTorch fp32 CPU; 1 warm-up step ({t_warm:.1f}s), "
                      f"median of {timed_steps} timed steps = {dt:.1f}s (min {times[0]:.1f}s, max {times[-1]:.1f}s; model build {t_build:.0f}s untimed)"}


def cpu_baseline_subprocess(config, clips, timeout_s):
    """Run the CPU leg in a child process (fresh thread pools, bounded wall time)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--config", config, "--cpu-clips", str(clips)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"value": None, "unit": "video-frames/sec", "cores": None, "kind": "port",
                "sample": "cpu leg failed: " + (out.stderr.strip().splitlines() or ["?"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "video-frames/sec", "cores": usable_cores(), "kind": "port",
                "sample": f"cpu leg exceeded {timeout_s}s for {clips} clip(s); lower bound {clips * 16 / timeout_s:.4f} frames/s not reached"}


def secondary_subprocess(config, steps, warmup, timeout_s=600, extra=()):
    """Time a secondary workload (c4: first-stage train step, c5: sampling, c2 in the reference's own fp32 arithmetic) with this same
    script in a child process -- same timing contract, fresh HIP context -- and return the fields of its JSON line that matter beside
    the headline (incl. the configuration's OWN roofline objects)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", config, "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline",
           "--no-secondary"] + list(extra)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith("{"):
                d = json.loads(ln)
                keep = {k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "algorithmic_tflop_per_step_per_gpu",
                                          "step_mfma_frac", "step_hbm_frac_12P", "hipgraph", "loss", "roofline", "roofline_other_kernels",
                                          "executed_tflop_per_step_per_gpu", "step_mfma_frac_executed", "kernel_time_share", "roofline_note",
                                          "handoff_timeouts") if k in d}
                keep["workload"] = d["config"]["workload"]
                return keep
        return {"error": (out.stderr.strip().splitlines() or ["no output"])[-1][:300]}
    except subprocess.TimeoutExpired:
        return {"error": f"exceeded {timeout_s}s"}


def secondary(args, cfg, rank, world, device):
    """c4: first-stage VAE train step; c5: sampling.  Same timing contract as the headline run."""
    B, T, size, z = cfg["batch_size"], cfg["n_frames"], cfg["spatial_size"], cfg["z_dim"]
    batch = synthetic_batch(B, T, size, seed=1 + rank, device=device)
    if args.config == "fvd":
        # the FVD evaluation's device work per I3D batch (utils/metrics.py:679-731, 787-800): 224x224 bilinear resize + de-normalisation
        # + the I3D trunk to 400 logits; fp32 arithmetic as in the reference unless --dtype bf16 is forced with --fvd-dtype
        from ipoke_amd import fvd
        from ipoke_amd.utils.detfill import deterministic_fill_
        net = fvd.I3D(400, "rgb", dtype=args.fvd_dtype, device="cpu")
        deterministic_fill_(net, prefix="i3d.")
        net.to(device)
        vids = batch["images"]
        minval = fvd._resized_min(vids)
        net.logits_of_videos(vids, (224, 224), minval)
        net.gflop = 0.0
        net.logits_of_videos(vids, (224, 224), minval)
        gflop = net.gflop
        step = lambda i: net.logits_of_videos(vids, (224, 224), minval)
        metric, frames = "video-frames/sec (FVD evaluation: 224x224 resize + I3D logits)", world * B * T
        workload = (f"I3D (Kinetics RGB, 12.7 M parameters, 58 convolutions) on {T}x3x{size}x{size} clips resized to 224x224, batch {B} "
                    f"(bs_i3d), {args.fvd_dtype} arithmetic, {gflop / B:.1f} GFLOP per clip")
        args.dtype = args.fvd_dtype
    elif args.config == "c4gan":
        # the reference's real first-stage step (first_stage_motion_model.py:160-277, config/first_stage.yaml d_t / d_s, w_vgg = 10):
        # L1 + KL + VGG perceptual loss + temporal discriminator (hinge + gradient penalty) + spatial discriminator + generator terms
        import numpy as np
        from ipoke_amd.discriminator import PatchDiscriminator, TemporalDiscriminator
        from ipoke_amd.first_stage import SpadeCondMotionModel
        from ipoke_amd.first_stage_gan import FirstStageGANTrainer
        if world > 1:
            raise SystemExit("c4gan is a single-GPU measurement")
        torch.manual_seed(0)
        model = SpadeCondMotionModel(configs.first_stage_config(size, z, T), dirs={}, dtype=args.dtype).to(device)
        d_t = {"bce_loss": False, "gp_weight": 1.0, "num_classes": 1, "patch_temp_disc": False, "fmap_weight": 1.0, "gen_weight": 1.0,
               "max_frames": 12}
        d_s = {"bce_loss": False, "gp_weight": 0.0, "fmap_weight": 1.0, "gen_weight": 1.0, "n_examples": 16}
        disc_t = TemporalDiscriminator(size, d_t, dtype=args.dtype).to(device)
        disc_s = PatchDiscriminator(d_s, dtype=args.dtype).to(device)
        from ipoke_amd.utils.detfill import deterministic_fill_
        from ipoke_amd.vgg import VGGLoss
        vgg = VGGLoss(dtype=args.dtype)
        deterministic_fill_(vgg.vgg, prefix="vgg19.")          # random-init weights of the VGG-19 architecture (no pretrained weights offline)
        vgg.to(device)
        gan = FirstStageGANTrainer(model, disc_t, disc_s, {"training": {"lr": 2e-4, "weight_decay": 1e-5, "w_l1": 10.0, "w_kl": 1e-7, "w_vgg": 10.0},
                                                            "d_t": d_t, "d_s": d_s, "data": {"max_frames": T - 1}}, vgg_loss=vgg)
        eps = torch.randn(B, z, 8, 8, generator=torch.Generator().manual_seed(7 + rank)).to(device)
        rng = np.random.RandomState(3)
        step = lambda i: gan.step(batch["images"], eps, *gan.draw(batch["images"], rng))["loss"]
        metric, frames = "video-frames/sec (first-stage adversarial train step: L1 + KL + VGG + d_t hinge/GP + d_s + generator terms)", world * B * T
        workload = (f"first_stage {T}x3x{size}x{size} clips, z={z}, generator fwd+bwd, 3-D ResNet-18 discriminator on {d_t['max_frames']} frames "
                    f"(4 forward + tangent pass + backward), PatchGAN on {d_s['n_examples']} frames, VGG-19 perceptual loss on {B * (T - 1)} frame pairs, three Adam steps; per-GPU batch {B}")
    elif args.config == "c4":
        from ipoke_amd.first_stage import SpadeCondMotionModel
        from ipoke_amd.first_stage_train import FirstStageTrainer
        torch.manual_seed(0)
        model = SpadeCondMotionModel(configs.first_stage_config(size, z, T), dirs={}, dtype=args.dtype).to(device)
        for p in model.parameters():
            D.broadcast_(p.data, src=0)
        trainer = FirstStageTrainer(model)
        eps = torch.randn(B, z, 8, 8, generator=torch.Generator().manual_seed(7 + rank)).to(device)
        trainer.enable_data_parallel()             # N > 1: flat-buffer all-reduce of the gradients, mean folded into the Adam update
        step = lambda i: trainer.step(batch["images"], eps)[0]
        metric, frames = "video-frames/sec (first-stage VAE train step, L1 + KL)", world * B * T
        workload = f"first_stage {T}x3x{size}x{size} clips, z={z}, encoder + ConvGRU + SPADE decoder fwd+bwd, per-GPU batch {B}"
    else:
        model = build_model(cfg, args.dtype, device)
        with torch.no_grad():
            model.forward_density(batch)                  # data-dependent init
        randomise_couplings(model)
        step = lambda i: model.forward_sample(batch, n_samples=1, n_logged_vids=1)
        metric, frames = "video-frames/sec (sampling: reverse flow + decode)", world * B * (T - 1)
        workload = f"{cfg['name']} forward_sample, z={z}, reverse flow + {T - 1}-frame decode at {size}x{size}, per-GPU batch {B}"
    for i in range(args.warmup):
        out = step(i)
    import gc
    gc.collect(); gc.freeze()                     # see main(): no generation-2 collection inside the timed loop
    D.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    torch.cuda.synchronize(); D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, device)
    handoff_timeouts = None
    if args.config == "c5":
        handoff_timeouts = list(model.flow.engine.handoff_timeouts())
        if any(handoff_timeouts):
            raise SystemExit(f"in-launch hand-off time-outs {handoff_timeouts}: the timed steps are invalid")
    # one extra step on EVERY rank (the data-parallel hook of c4 holds a collective) with the per-family event timing on -- taken right
    # behind the timed eager loop, BEFORE the hipGraph / two-batches-in-flight variants of c5 (round 5 took it after them and its
    # single-stream families summed to more than the step: the variants leave other streams and graph state behind)
    fams = config_rooflines(lambda: step(args.warmup + args.steps), args.dtype) if args.config != "fvd" else []
    graph = None
    if args.config == "c5" and not args.quick:
        def timed_again():
            for i in range(3):
                step(i)
            D.barrier(); torch.cuda.synchronize()
            tg = time.perf_counter()
            for i in range(args.steps):
                step(args.warmup + i)
            torch.cuda.synchronize(); D.barrier()
            return D.max_over_ranks(time.perf_counter() - tg, device)
        # configs[4] "hipGraph-captured": (a) the reverse flow replayed from the engine's own captured graph, decoder eager;
        # (b) the WHOLE device side of forward_sample (encoders + reverse flow + ConvGRU + batched decode) as one captured graph
        model.flow.set_graph_mode(True)
        eg = timed_again()
        model.flow.set_graph_mode(False)
        model.set_sample_graph(True)
        ef = timed_again()
        model.set_sample_graph(False)
        # two batches in flight (PokeMotionModel.sample_stream): reverse flow of batch k+1 beside the decode of batch k
        # (IPOKE_BENCH_NO_PIPELINE=1 skips it: profiler runs -- a kernel-trace run of this loop did not finish within 15 minutes)
        ep = None
        if os.environ.get("IPOKE_BENCH_NO_PIPELINE") != "1":
            for _ in model.sample_stream([batch] * 3):
                pass
            D.barrier(); torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in model.sample_stream([batch] * args.steps):
                pass
            torch.cuda.synchronize(); D.barrier()
            ep = D.max_over_ranks(time.perf_counter() - tp, device)
        graph = {"pipelined_ms_per_step": None if ep is None else round(ep / args.steps * 1e3, 3),
                 "pipelined_value": None if ep is None else round(frames / (ep / args.steps), 2),
                 "pipelined_what": "the same K batches through sample_stream: every batch's kernels unchanged, the decode of batch k on a second "
                                   "stream beside the reverse flow of batch k+1 (throughput of a validation / test loop; per-batch latency is the headline value)",
                 "flow_graph_ms_per_step": round(eg / args.steps * 1e3, 3), "flow_graph_value": round(frames / (eg / args.steps), 2),
                 "full_graph_ms_per_step": round(ef / args.steps * 1e3, 3), "full_graph_value": round(frames / (ef / args.steps), 2),
                 "what": "same K steps; flow_graph: reverse flow replayed as a captured hipGraph, decoder eager; full_graph: conditioning "
                         "encoders + reverse flow + ConvGRU + frame-batched decode replayed as ONE captured hipGraph (PokeMotionModel.set_sample_graph); "
                         "the headline value of this line is the eager path"}
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {"metric": metric, "value": round(frames / (elapsed / args.steps), 2), "unit": "video-frames/sec", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": workload, "global_batch": world * B, "clip_frames": T, "parallelism": f"dp{world}",
                           "weights": "random init of the named architecture (no checkpoints offline)"},
                "roofline": None}
        # a kernel family runs on ONE stream: launches that sum to more than the step were not timed inside a representative step
        bad = [f["kernel"] for f in fams if f["total_us_in_step"] * 1e-3 > ms]
        if bad:
            line["roofline_note"] = ("in-situ family timing rejected: " + "; ".join(bad) + f" sum to more than the {ms:.2f} ms step "
                                     "(single-stream families): perturbed timing step, no roofline reported for this configuration")
            fams = []
        if fams:                           # the dominant kernel family OF THIS configuration; the rest beside it
            line["roofline"] = fams[0]
            line["roofline_other_kernels"] = fams[1:5]
            done = sum(f["algorithmic_gflop_in_step"] for f in fams) / 1e3
            peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
            line["executed_tflop_per_step_per_gpu"] = round(done, 2)          # convolutions + weight gradients + flow units actually launched
            line["step_mfma_frac_executed"] = round(done / (ms * 1e-3) / peak, 4)
            line["kernel_time_share"] = {"timed_families_ms": round(sum(f["total_us_in_step"] for f in fams) / 1e3, 2), "step_ms": round(ms, 2)}
        if args.config in ("c4", "c4gan"):
            line["loss"] = round(float(out.item()), 4)
        if args.config == "fvd":
            peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
            line["algorithmic_tflop_per_step_per_gpu"] = round(gflop / 1e3, 3)
            line["step_mfma_frac"] = round(gflop / 1e3 / (ms * 1e-3) / peak, 4)
            line["roofline"] = {"bound": "mfma", "achieved": round(gflop / 1e3 / (ms * 1e-3), 2), "peak": peak, "unit": "TFLOP/s",
                                "frac": line["step_mfma_frac"], "traffic": None,
                                "kernel": "whole I3D batch (58 implicit-GEMM launches + 13 pools + resize), wall clock of the step"}
        if graph:
            line["hipgraph"] = graph
        if handoff_timeouts is not None:
            line["handoff_timeouts"] = handoff_timeouts
        if args.config == "c5":
            line["algorithmic_tflop_per_step_per_gpu"] = round(B * (FLOW_GFLOP[z] + 244.3) / 1e3, 2)     # un-hoisted
            line["step_mfma_frac"] = round(line["algorithmic_tflop_per_step_per_gpu"] / (ms * 1e-3) /
                                           (MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS), 4)
        print(json.dumps(line), flush=True)
    D.barrier()


def self_launch_command(argv, gpus, environ, script=None, port=None):
    """``python bench.py --gpus N`` (N > 1) without a torchrun environment: the command that re-runs this script as N ranks, one per GPU
    -- exactly the launch line of the driver's contract -- or None when this process is already a rank (WORLD_SIZE set), N = 1, or this
    is the CPU-baseline child.  Rank 0 of the relaunched job prints the one JSON line."""
    if gpus <= 1 or "WORLD_SIZE" in environ or "--cpu-baseline-only" in argv:
        return None
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4", "c4gan", "c5", "fvd"],
                    help="c2 (default) is the configuration BASELINE.json's metric is quoted on; c4 = first-stage VAE train step "
                         "(L1 + KL), c5 = sampling (reverse flow + 15-frame decode): secondary workloads of SURVEY.md §8d")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--fvd-dtype", default="f32", choices=["bf16", "f32"], help="arithmetic of --config fvd (the reference's I3D runs fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=2, help="clips per CPU-baseline step (the per-step Adam / weight stream is amortised over them)")
    ap.add_argument("--cpu-timeout", type=int, default=600)
    ap.add_argument("--no-secondary", action="store_true", help="skip the c4 / c5 lines attached to the default c2 line")
    ap.add_argument("--secondary-steps", type=int, default=10)
    ap.add_argument("--quick", action="store_true", help="c5: the eager latency only (no hipGraph / two-batches-in-flight variants)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(dict(configs.BENCH_CONFIGS[args.config]), clips=args.cpu_clips)), flush=True)
        return

    relaunch = self_launch_command(sys.argv[1:], args.gpus, os.environ)
    if relaunch is not None:                 # plain `python bench.py --gpus N`: become the N-rank job (same stdout, same exit code)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this driver
        sys.stdout.flush()
        os.execv(sys.executable, relaunch)

    _lib.require_gpu()
    rank, world, local = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if os.environ.get("IPOKE_DIST_SINGLE_GPU") != "1" and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but {torch.cuda.device_count()} GPU(s) visible (one rank per GPU over RCCL; the single-GPU test "
                         "mode is IPOKE_DIST_BACKEND=gloo IPOKE_DIST_SINGLE_GPU=1)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    cfg = dict(configs.BENCH_CONFIGS[args.config])
    if args.batch:
        cfg["batch_size"] = args.batch
    B, T, size, z = cfg["batch_size"], cfg["n_frames"], cfg["spatial_size"], cfg["z_dim"]

    if args.config in ("c4", "c4gan", "c5", "fvd"):
        secondary(args, cfg, rank, world, device)
        return

    from ipoke_amd.trainer import SecondStageTrainer
    model = build_model(cfg, args.dtype, device)
    trainer = SecondStageTrainer(model)
    batch = synthetic_batch(B, T, size, seed=1 + rank, device=device)
    trainer.sync_initial_state(batch)                 # data-dependent init on rank 0's statistics, then broadcast
    randomise_couplings(model)
    for i in range(args.warmup):
        trainer.train_step(batch, i, next_batch=batch)
    # the model is ~7 000 parameter objects plus their modules: a generation-2 collection of Python's cycle GC inside the timed loop
    # stalls the host for > 100 ms (seen as one 185 ms step in 30): collect now and move the survivors out of the collector's sight
    import gc
    gc.collect(); gc.freeze()
    D.barrier(); torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()                             # per-step boundaries on the stream (no host synchronisation)
        loss = trainer.train_step(batch, args.warmup + i, next_batch=batch)
    marks[args.steps].record()
    torch.cuda.synchronize(); D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, device)
    handoff_timeouts = list(model.flow.engine.handoff_timeouts())      # (row-split unit, fused conv3 + coupling) launches that gave up: must be 0, 0
    if any(handoff_timeouts):
        raise SystemExit(f"in-launch hand-off time-outs {handoff_timeouts}: the timed steps are invalid")
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    loss_val = float(loss.item())
    ms = elapsed / args.steps * 1e3
    frames = world * B * T
    value = frames / (elapsed / args.steps)

    insitu = insitu_rooflines(lambda: trainer.train_step(batch, args.warmup + args.steps, next_batch=batch), B, z, args.dtype) \
        if args.dtype == "bf16" else None
    if rank == 0:
        roof = kernel_roofline(B, args.dtype)
        if insitu:                       # the dominant kernel as it runs INSIDE the step; the isolated figure is kept beside it
            iso = roof
            roof = dict(insitu[0]); roof["traffic"] = iso["traffic"]; roof["traffic_unit"] = iso["traffic_unit"]; roof["traffic_source"] = iso["traffic_source"]
            roof["algorithmic_bytes_per_launch"] = iso["algorithmic_bytes_per_launch"]
            roof["isolated_avg_launch_us"] = iso["avg_launch_us"]; roof["isolated_frac"] = iso["frac"]
        P_bytes = model.flow.engine.n_params * 4
        step_tflop = B * (3 * FLOW_GFLOP[z] + ENC_GFLOP[size] + 0.47) / 1e3
        line = {
            "metric": "video-frames/sec (second-stage train step)", "value": round(value, 2), "unit": "video-frames/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{cfg['name']} second_stage train, {T}x3x{size}x{size} clips, z={z}, flow 2048 hidden "
                                   f"({model.flow.engine.n_params / 1e9:.3f} B params), per-GPU batch {B}",
                       "global_batch": world * B, "clip_frames": T, "parallelism": f"dp{world}",
                       "weights": "random init of the named architecture (no checkpoints offline)"},
            "loss": round(loss_val, 3), "handoff_timeouts": handoff_timeouts,
            "ms_per_step_median": round(median_ms, 3), "ms_per_step_min": round(per_step[0], 3), "ms_per_step_max": round(per_step[-1], 3),
            "algorithmic_tflop_per_step_per_gpu": round(step_tflop, 2),
            "step_mfma_frac": round(step_tflop / (ms * 1e-3) / (MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS), 4),
            "step_hbm_frac_12P": round(12 * P_bytes / (ms * 1e-3) / 8e12, 4),
            "roofline": roof,
        }
        if insitu:
            line["roofline_other_kernels"] = insitu[1:]
        if not args.no_secondary and world == 1 and args.config == "c2":
            # BASELINE configs[3] / configs[4], timed by the same driver invocation (their own lines: --config c4 / c5)
            del model, trainer, batch
            torch.cuda.empty_cache()
            line["secondary"] = {c: secondary_subprocess(c, args.secondary_steps, 3) for c in ("c4", "c5")}
            # BASELINE configs[2]: the per-GPU workload of the 8-GPU iper_128 job (z = 32, per-GPU batch 40) on this one GPU
            line["secondary"]["c3"] = secondary_subprocess("c3", 5, 3)
            if args.dtype == "bf16":
                # every BASELINE configuration also in the reference's own arithmetic (exact-f32 matrix cores: 157 TFLOP/s peak -- the
                # parity-tight mode; configs[3] / configs[4] name no dtype, the reference computes fp32)
                line["secondary"]["c2_f32"] = secondary_subprocess("c2", args.secondary_steps, 3, extra=("--dtype", "f32"))
                line["secondary"]["c4_f32"] = secondary_subprocess("c4", 5, 2, extra=("--dtype", "f32"))
                line["secondary"]["c5_f32"] = secondary_subprocess("c5", 5, 2, extra=("--dtype", "f32", "--quick"))
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (the other ranks would idle at the barrier)
            line["cpu_baseline"] = cpu_baseline_subprocess(args.config, args.cpu_clips, args.cpu_timeout)
        # LAST key: every configuration's ms per step in one short object (a 2 KB tail of the line still shows all of them)
        summ = {args.config if args.dtype == "bf16" else args.config + "_f32": round(ms, 3)}
        for name, sec in line.get("secondary", {}).items():
            summ[name] = sec.get("ms_per_step") if isinstance(sec, dict) else None
        line["summary"] = {"ms_per_step": summ, "roofline_frac": line["roofline"]["frac"] if line.get("roofline") else None,
                           "cpu_frames_per_s": (line.get("cpu_baseline") or {}).get("value")}
        print(json.dumps(line), flush=True)
    D.barrier()


if __name__ == "__main__":
    main()
