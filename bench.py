#!/usr/bin/env python3
"""Headline benchmark: 4 s @ 16 kHz utterances/sec, forward + SI-SNR, DCCRN-CL, on N MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One step = one pass of the hot path over one batch of synthetic utterances resident in HBM:
  DCCRN_(causal, skips 012345, mask).forward(noisy, train=False)  ->  ete_train_se_loss([0,0,1]).final_ete_loss
  (STFT -> 6 complex-conv encoder blocks -> complex LSTM -> dense -> 6 complex transposed-conv decoder blocks
   -> mask -> ISTFT, + STFT(clean) + STFT losses + SI-SNR), i.e. supervised_dccrn/train.py:233-237 of the reference
  without the backward pass.  Utterances are independent: each rank runs its own batch (replicas, no collective
  on the data path); value = all ranks' utterances / max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` (dominant cgemm instantiation, HIP-event timed on the launching
stream inside the timed region) and `cpu_baseline` (the torch-CPU oracle on the host cores, N = 1 only).
Other workloads: --workload cvae_elbo | nsvae_kl | twophase (BASELINE.json configs 2, 3, 5, forward + loss).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NFFT, HOP, WIN, LEN = 512, 100, 400, 64000
SKIP = [0, 1, 2, 3, 4, 5]
PEAK_F32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense bf16 MFMA peak (2:1-sparsity figures never used)
DEFAULT_BATCH = 64       # utterances per GPU per step
METRIC = "4s@16kHz utterances/sec fwd+SI-SNR, DCCRN-CL, 1/2/4/8 MI355X vs host CPU"


def synth_state(module, seed):
    synth = importlib.import_module("i-dccrn-vae_amd.utils.synth")
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(synth.synth_state_dict(shapes, seed), strict=True)
    return module


def make_inputs(B, seed, device):
    import torch
    g = torch.Generator().manual_seed(seed)
    clean = torch.randn(B, LEN, generator=g) * 0.1
    g2 = torch.Generator().manual_seed(seed + 1)
    noise = torch.randn(B, LEN, generator=g2) * 0.1
    return (clean + noise).to(device), clean.to(device), noise.to(device)


def build_workload(name, B, device, rank):
    """-> (step callable, utterances per step, description dict)"""
    import torch
    pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
    nl = importlib.import_module("i-dccrn-vae_amd.model.nsvae_loss")
    pl = importlib.import_module("i-dccrn-vae_amd.model.pretrain_pvaes_loss")
    cn = importlib.import_module("i-dccrn-vae_amd.model.causal_netconfig")
    np_ = cn.get_net_params()
    noisy, clean, noise = make_inputs(B, 123 + 1000 * rank, device)
    if name == "dccrn_cl":
        model = synth_state(pm.DCCRN_(NFFT, HOP, np_, True, device, WIN, SKIP, "mask", False, None, None), 51).to(device)
        loss = nl.ete_train_se_loss([0.0, 0.0, 1.0])

        def step():
            est, est_stft = model(noisy, train=False)
            return loss.final_ete_loss(est_stft, model.stft(clean), clean, est)[0]
        return step, B, {"workload": "supervised DCCRN-CL forward(train=False) + final_ete_loss (weights 0/0/1)",
                         "batch_per_gpu": B}
    if name == "cvae_elbo":
        ns, zdim = 5, 128
        enc = synth_state(pm.pvae_dccrn_encoder_skip_prepare(np_, True, device, zdim, NFFT, HOP, WIN, ns), 52).to(device)
        dec = synth_state(pm.pvae_dccrn_decoder_skip_prepare(np_, True, device, ns, zdim, NFFT, HOP, WIN, "real_imag", SKIP), 53).to(device)
        loss = pl.complex_standard_vae_loss(torch.ones(1), 1.0, 0.0, 'multiple', 'real_imag', [1.0, 1.0, 0.0], ns)

        def step():
            z, miu, ls, dl, skiper, C, F, stft_x = enc(clean, train=False)
            recon, pred = dec(stft_x, z, skiper, C, F, train=False)
            return loss.cal_loss(clean, recon, stft_x, pred, miu, ls, dl, z, 100)[0]
        return step, B, {"workload": "pretrained CVAE forward (eval BN) + ELBO, num_samples=5", "batch_per_gpu": B}
    if name == "nsvae_kl":
        ns, zdim = 2, 128
        ce = synth_state(pm.pvae_dccrn_encoder_skip_prepare(np_, True, device, zdim, NFFT, HOP, WIN, ns), 54).to(device)
        ne = synth_state(pm.pvae_dccrn_encoder_skip_prepare(np_, True, device, zdim, NFFT, HOP, WIN, ns), 55).to(device)
        se = synth_state(pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, device, zdim, NFFT, HOP, WIN, ns, 2), 56).to(device)
        loss = nl.standard_nsvae_loss_true_kl(1.0, 0, 1.0, 0.0, zdim, ns, 2, 'original', 'False', SKIP, 'both')

        ops = importlib.import_module("i-dccrn-vae_amd").ops

        def step():
            # three independent encoders: one HIP stream each (their H = 384 / 768 recurrences are latency-bound)
            c, n, s = ops.concurrent([lambda: ce(clean, train=False), lambda: ne(noise, train=False),
                                      lambda: se(noisy, train=False)], device)
            return loss.final_nsvae_loss(c[1], n[1], s[1], s[5], c[2], n[2], s[2], s[6], c[3], n[3], s[3], s[7],
                                         s[0], s[4], c[4], n[4], s[8])[0]
        return step, B, {"workload": "NSVAE: 2 frozen CVAE/NVAE encoders + noisy encoder (latent_num=2) + nsvae KL loss",
                         "batch_per_gpu": B}
    if name == "twophase":
        ns, zdim = 2, 128
        se = synth_state(pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, device, zdim, NFFT, HOP, WIN, ns, 2), 56).to(device)
        dec = synth_state(pm.nsvae_pvae_dccrn_decoder_twophase(np_, True, device, ns, zdim, NFFT, HOP, WIN, "mask", True, SKIP, False), 57).to(device)
        loss = nl.two_phase_loss([0, 0, 1], 1.0, zdim, 1)

        def step():
            s = se(noisy, train=False)
            recon, pred = dec(s[11], s[0], s[8], s[9], s[10], train=False, pad='sig')
            return loss.phase_2_loss(pred, s[11], clean, recon, None, None, None, None)[0]
        return step, B, {"workload": "two-phase decoder fine-tune forward: frozen NSVAE encoder + decoder(mask, pad='sig') + SI-SNR",
                         "batch_per_gpu": B}
    raise SystemExit(f"unknown workload {name}")


def host_cores() -> int:
    """CPU share of this process (the GPU box exposes more logical CPUs than the job may use)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("IDV_CPU_THREADS", "16"))))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(seconds_budget=20.0):
    """The CPU oracle (torch-CPU restatement of the reference's op sequence, MKLDNN convolutions, all host
    cores) on a bounded sample of the SAME workload: B = 2 utterances per pass."""
    import torch
    from oracle import idccrn_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    np_ = O.net_params(True, 32)
    pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
    m = pm.DCCRN_(NFFT, HOP, np_, True, "cpu", WIN, SKIP, "mask", False, None, None)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = O.synth_state_dict(shapes, 51)
    B = 2
    noisy, clean, _ = make_inputs(B, 123, "cpu")

    def one():
        with torch.no_grad():
            est, pred, _ = O.dccrn_forward(noisy, sd, np_, True, NFFT, HOP, WIN, SKIP, "mask", False)
            return O.multiple_recon_loss(pred, O.stft(clean, NFFT, HOP, WIN), clean, est, [0.0, 0.0, 1.0])[0]
    one()
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 24:
            break
    return {"value": round(B * n / el, 4), "unit": "utterances/sec", "cores": cores, "kind": "port",
            "sample": f"{n} passes of B={B} 4 s utterances, DCCRN-CL forward + final_ete_loss, torch-CPU oracle "
                      f"(oracle/idccrn_oracle.py), {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH, help="utterances per GPU per step")
    ap.add_argument("--workload", default="dccrn_cl")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default=os.environ.get("IDV_PRECISION", "bf16x3"), choices=["fp32", "bf16x3"],
                    help="conv contraction arithmetic: exact fp32 MFMA, or split-bf16 (3 bf16 MFMAs, fp32 accumulate)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)                 # rehearsal of N ranks on fewer GPUs (IDV_BENCH_BACKEND=gloo) shares devices
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("IDV_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    ops = importlib.import_module("i-dccrn-vae_amd").ops
    ops.set_precision(args.precision)
    torch.set_grad_enabled(False)
    step, utt_per_step, cfg = build_workload(args.workload, args.batch, device, rank)

    log(f"workload {args.workload} B={args.batch} built; warmup {args.warmup}")
    for w in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {w} done")

    def barrier():
        if world > 1:
            dist.barrier()

    ops.LAUNCH_LOG = []
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    launches, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
    loss_val = float(last)
    log(f"timed {args.steps} steps in {elapsed:.3f} s")

    # Under sub-batch stream overlap a kernel's event interval includes the time its blocks wait behind the other
    # stream's kernel, so kernel quality (the roofline) is measured in a second, single-stream pass of the same steps;
    # the overlapped intervals of the timed region are reported beside it.
    overlapped = None
    n_streams = ops.stream_split(args.batch) if args.workload == "dccrn_cl" else 1
    if n_streams > 1:
        overlapped = launches
        keep, ops.STREAM_SPLIT = ops.STREAM_SPLIT, 1
        step()
        torch.cuda.synchronize()
        ops.LAUNCH_LOG = []
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        serial_elapsed = time.perf_counter() - t1
        launches, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
        ops.STREAM_SPLIT = keep
        log(f"single-stream roofline pass: {args.steps} steps in {serial_elapsed:.3f} s")
    else:
        serial_elapsed = elapsed

    dt = importlib.import_module("i-dccrn-vae_amd.utils.dist_timing")
    red_dev = device if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")
    value, elapsed = dt.job_throughput(elapsed, float(utt_per_step * args.steps), red_dev)

    # dominant kernel: group conv launches by kernel instantiation, HIP-event durations on the launch stream
    def kernel_name(cfg_id):
        if cfg_id in (-99, -98):
            return f"void (anonymous namespace)::ctconv_c1_bf16_kernel<{'true' if cfg_id == -98 else 'false'}>(CgemmArgs)"
        if cfg_id > 0:
            d = str(cfg_id)
            mode, t = (1 if len(d) == 7 else 0), d[-6:]
            return f"void cgemm_kernel<{mode}, {t[0]}, {t[1]}, {t[2]}, {t[3]}, {t[4]}, {t[5]}, false, false, true>(CgemmArgs)"
        if cfg_id <= -100000000:       # idv_cconv_img_config digits <MODE, WM, WN, FO_T, JC_W, MT_W, IMGIN, AD>
            d = str(-cfg_id - 100000000).rjust(8, "0")
            return (f"void (anonymous namespace)::cgemm_bf16_kernel<{d[0]}, {d[1]}, {d[2]}, {d[3]}, {d[4]}, false, {d[5]}, "
                    f"{'true' if d[6] == '1' else 'false'}, {d[7]}>(CgemmArgs)")
        d = str(-cfg_id - 1000000).rjust(6, "0")
        return f"void (anonymous namespace)::cgemm_bf16_kernel<{d[0]}, {d[1]}, {d[2]}, {d[3]}, {d[4]}, false, {d[5]}, false, 2>(CgemmArgs)"

    def group(entries):
        out = {}
        for cfg_id, macs, e0, e1 in entries:
            g = out.setdefault(cfg_id, [0, 0.0, 0])
            g[0] += macs
            g[1] += e0.elapsed_time(e1) * 1e-3
            g[2] += 1
        return out

    groups = group(launches)
    roofline = None
    if groups:
        dom = max(groups, key=lambda k: groups[k][1])
        macs, secs, n = groups[dom]
        tot_macs = sum(g[0] for g in groups.values())
        tot_secs = sum(g[1] for g in groups.values())
        split = dom < 0
        peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        ach = 2 * macs / secs / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath) and args.batch == DEFAULT_BATCH and args.workload == "dccrn_cl":     # single-stream launches
            tk = json.load(open(tpath))["kernels"].get(kernel_name(dom))
            if tk:
                traffic = tk["hbm_bytes_per_launch"]
        roofline = {
            "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": traffic,
            "traffic_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB from the committed rocprofv3 --pmc passes "
                            "(profiles/r01_traffic.json, same command and batch); null when the run differs from that command",
            "peak_note": ("dense bf16 MFMA peak; achieved counts ALGORITHMIC fp32 flops (4 real convs), the kernel executes 3 bf16 "
                          "MFMA products per algorithmic product" if split else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
            "kernel": kernel_name(dom), "launches": n, "avg_launch_ms": round(secs / n * 1e3, 4),
            "algorithmic_gflop_per_launch": round(2 * macs / n / 1e9, 3),
            "all_conv_launches": {"achieved": round(2 * tot_macs / tot_secs / 1e12, 3),
                                  "frac": round(2 * tot_macs / tot_secs / 1e12 / peak, 4),
                                  "share_of_step_time": round(tot_secs / serial_elapsed, 4)},
            "per_kernel": {kernel_name(k): {"tflops": round(2 * v[0] / v[1] / 1e12, 2), "ms_per_step": round(v[1] / args.steps * 1e3, 3)}
                           for k, v in sorted(groups.items())},
        }
        if overlapped is not None:
            og = group(overlapped)
            om, osec, on = og.get(dom, (0, 1.0, 1))
            roofline["measured_in"] = (f"single-stream pass (IDV_STREAM_SPLIT=1) of the same {args.steps} steps, "
                                       f"{round(serial_elapsed / args.steps * 1e3, 3)} ms/step; `value` is the {n_streams}-stream timed region")
            roofline["timed_region_overlapped"] = {
                "streams": n_streams, "launches": on, "avg_launch_ms": round(osec / on * 1e3, 4),
                "achieved": round(2 * om / osec / 1e12, 3), "frac": round(2 * om / osec / 1e12 / peak, 4),
                "note": "event intervals on each sub-batch stream; they include CU time-slicing with the other stream"}
        if split:
            roofline["clock_note"] = ("committed counters (profiles/r01h: GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES): the conv kernels run "
                                      "power-limited at 1.65-1.80 GHz and the MFMA pipe is busy 65-78 % of those cycles; `peak` is the "
                                      "dense bf16 rate at the nominal 2.4 GHz")
            roofline["executed_bf16_tflops"] = round(3 * ach, 1)
            roofline["frac_executed"] = round(3 * ach / peak, 4)
            roofline["vs_fp32_mfma_peak"] = round(ach / PEAK_F32_MFMA_TFLOPS, 3)

    if rank == 0:
        out = {
            "metric": METRIC, "value": round(value, 3), "unit": "utterances/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16x3 (split-fp32 operands on bf16 MFMA, fp32 accumulate)", "data": "synthetic (0.1*N(0,1) clean + noise, seeded; random-init weights)",
            "config": dict(cfg, utterance="4 s @ 16 kHz (64000 samples, 641 frames)", parallelism=f"replicas x{world}",
                           loss=loss_val),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "dccrn_cl":
            log("cpu baseline ...")
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
