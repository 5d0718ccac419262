#!/usr/bin/env python3
"""Headline benchmark: 4 s @ 16 kHz utterances/sec, forward + SI-SNR, DCCRN-CL, on N MI355X.

  python bench.py --gpus N --steps K --warmup W
      N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (WORLD_SIZE set:
      this process is one rank), or plain `python bench.py --gpus N`: the process then starts that launcher as a child and
      relays rank 0's JSON line (it touches no GPU itself).

One step = one pass of the hot path over one batch of synthetic utterances resident in HBM:
  DCCRN_(causal, skips 012345, mask).forward(noisy, train=False)  ->  ete_train_se_loss([0,0,1]).final_ete_loss
  (STFT -> 6 complex-conv encoder blocks -> complex LSTM -> dense -> 6 complex transposed-conv decoder blocks
   -> mask -> ISTFT, + STFT(clean) + STFT losses + SI-SNR), i.e. supervised_dccrn/train.py:233-237 of the reference
  without the backward pass.  Utterances are independent: each rank runs its own batch (replicas, no collective
  on the data path); value = all ranks' utterances / max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` (dominant contraction kernel, HIP-event timed on the launching stream
inside the timed region) and `cpu_baseline` (the reference's op sequence on stock torch-CPU operators, oracle/stock_cpu.py,
N = 1 only).  The headline is measured at the reference's precision (`dtype` "f32": exact fp32 MFMA, one stream); the
split-bf16 mode is attached as the clearly labelled secondary record `alt` (same workload, `dtype` "bf16x3"), together with
a two-stream run of it whose output is compared bit for bit with the one-stream output (`two_stream_bit_exact`).

Other workloads (--workload): enhance | enhance_complex_mask (the evaluation script's inference, 10 latent samples per
utterance), cvae_elbo | nsvae_kl | twophase (BASELINE.json configs 2, 3, 5: forward + loss) and the
TRAIN steps forward + loss + backward + Adam, as the reference's trainers run them: dccrn_cl_train
(supervised_dccrn/train.py:233-243), cvae_train (pretrained_vaes/train.py:281-301), nsvae_train (train_nsvae.py:487-574,
config 4) and twophase_train (train_second_phase_decoder.py:376-433, config 5).  With N > 1 the train steps shard the
global batch, all-reduce the batch-norm moment sums (Sync-CBN) and average the gradients over RCCL (parallel.py).
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


STREAM = False      # --stream: every step takes a fresh batch from the host-side synthetic stream through DevicePrefetcher


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
    if name in ("enhance", "enhance_complex_mask"):
        # enhancement inference as the evaluation script runs it (test_se_cvaefinetune.py:251-311, num_samples = 10 in
        # test_se_cvaefinetune.sh:8), batched: noisy encoder (eval) -> decoder(s) on 10 latent draws per utterance -> mean over
        # the samples (latent_to_use 1) or the two-decoder complex-mask estimator (latent_to_use 2)
        inf = importlib.import_module("i-dccrn-vae_amd.inference")
        ns, zdim = 10, 128
        se = synth_state(pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, device, zdim, NFFT, HOP, WIN, ns, 2), 56).to(device)
        dec = synth_state(pm.nsvae_pvae_dccrn_decoder_twophase(np_, True, device, ns, zdim, NFFT, HOP, WIN, "mask", True, SKIP, False), 57).to(device)
        if name == "enhance":
            def step():
                return inf.compute_sisdr(inf.enhance_vae(se, dec, noisy, check=False), clean).mean()
            what = "enhancement inference: noisy encoder + fine-tuned decoder on 10 latent samples, mean over samples, SI-SDR"
        else:
            dn = synth_state(pm.nsvae_pvae_dccrn_decoder_twophase(np_, True, device, ns, zdim, NFFT, HOP, WIN, "mask", True, SKIP, False), 58).to(device)

            def step():
                return inf.compute_sisdr(inf.enhance_vae_two_latents(se, dec, dn, noisy, "complex_mask", 2, check=False), clean).mean()
            what = ("enhancement inference, latent_to_use 2: noisy encoder + speech and noise decoders on 10 latent samples each, "
                    "complex-mask estimator, ISTFT, SI-SDR")
        return step, B, {"workload": what, "batch_per_gpu": B, "num_samples": ns}
    par = importlib.import_module("i-dccrn-vae_amd.parallel")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = world > 1 or (FORCE_DIST and "WORLD_SIZE" in os.environ)     # rehearsal: the N-rank code path with one rank

    def train_step(models, fwd_loss):
        """zero_grad -> forward + loss -> backward -> [gradient all-reduce] -> Adam.step (lr 1e-3, weight_decay 1e-3:
        supervised_dccrn/train.py:109)."""
        params = [p_ for m in models for p_ in m.parameters() if p_.requires_grad]
        # torch.optim.Adam semantics, one multi-tensor HIP launch per step (optim.py / csrc/bucket.hip); IDV_TORCH_ADAM=1: stock
        hip_adam = os.environ.get("IDV_TORCH_ADAM", "0") != "1"
        optim = importlib.import_module("i-dccrn-vae_amd.optim")
        opt = (optim.Adam if hip_adam else torch.optim.Adam)(params, lr=1e-3, weight_decay=1e-3)
        red = par.GradAllReduce(params) if dist_on else None
        if dist_on:
            par.enable_sync_bn()

        def step():
            with torch.enable_grad():
                opt.zero_grad(set_to_none=True)
                loss = fwd_loss()
                loss.backward()
            if red is not None and hip_adam and len(red.buckets) == 1:
                red.reduce(into_grads=False)              # gather -> RCCL all-reduce; Adam reads the averaged bucket
                opt.step(grad_bucket=red.bucket())
            else:
                if red is not None:
                    red.reduce()
                opt.step()
            return loss.detach()
        return step

    if name == "dccrn_cl_train":
        model = synth_state(pm.DCCRN_(NFFT, HOP, np_, True, device, WIN, SKIP, "mask", False, None, None), 51).to(device)
        model.train()
        loss = nl.ete_train_se_loss([0.0, 0.0, 1.0])

        def fl():
            est, est_stft = model(noisy, train=True)
            return loss.final_ete_loss(est_stft, model.stft(clean), clean, est)[0]
        return train_step([model], fl), B, {"workload": "supervised DCCRN-CL TRAIN step: forward(train=True) + final_ete_loss "
                                            "(weights 0/0/1) + backward + Adam", "batch_per_gpu": B}
    if name == "cvae_train":
        ns, zdim = 5, 128
        enc = synth_state(pm.pvae_dccrn_encoder_skip_prepare(np_, True, device, zdim, NFFT, HOP, WIN, ns), 52).to(device)
        dec = synth_state(pm.pvae_dccrn_decoder_skip_prepare(np_, True, device, ns, zdim, NFFT, HOP, WIN, "real_imag", SKIP), 53).to(device)
        loss = pl.complex_standard_vae_loss(torch.ones(1), 1.0, 0.0, 'multiple', 'real_imag', [1.0, 1.0, 0.0], ns)

        def fl():
            z, miu, ls, dl, skiper, C, F, stft_x = enc(clean, train=True)
            recon, pred = dec(stft_x, z, skiper, C, F, train=True)
            return loss.cal_loss(clean, recon, stft_x.detach(), pred, miu, ls, dl, z, 100)[0]
        return train_step([enc, dec], fl), B, {"workload": "CVAE pre-training TRAIN step: encoder + decoder (num_samples=5) + ELBO "
                                               "+ backward + Adam", "batch_per_gpu": B}
    if name == "nsvae_train":
        ns, zdim = 2, 128
        ce = synth_state(pm.pvae_dccrn_encoder_skip_prepare(np_, True, device, zdim, NFFT, HOP, WIN, ns), 54).to(device)
        ne = synth_state(pm.pvae_dccrn_encoder_skip_prepare(np_, True, device, zdim, NFFT, HOP, WIN, ns), 55).to(device)
        se = synth_state(pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, device, zdim, NFFT, HOP, WIN, ns, 2), 56).to(device)
        for m_ in (ce, ne):
            for p_ in m_.parameters():
                p_.requires_grad = False
        loss = nl.standard_nsvae_loss_true_kl(1.0, 0, 1.0, 0.0, zdim, ns, 2, 'original', 'False', SKIP, 'both')

        def fl():
            with torch.no_grad():
                c = ce(clean, train=False)
                n = ne(noise, train=False)
            s = se(noisy, train=True)
            return loss.final_nsvae_loss(c[1], n[1], s[1], s[5], c[2], n[2], s[2], s[6], c[3], n[3], s[3], s[7],
                                         s[0], s[4], c[4], n[4], s[8])[0]
        return train_step([se], fl), B, {"workload": "NSVAE TRAIN step (config 4): 2 frozen encoders (eval) + noisy encoder "
                                         "(train, latent_num=2) + nsvae KL loss + backward + Adam", "batch_per_gpu": B}
    if name == "twophase_train":
        ns, zdim = 2, 128
        se = synth_state(pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, device, zdim, NFFT, HOP, WIN, ns, 2), 56).to(device)
        dec = synth_state(pm.nsvae_pvae_dccrn_decoder_twophase(np_, True, device, ns, zdim, NFFT, HOP, WIN, "mask", True, SKIP, False), 57).to(device)
        for p_ in se.parameters():
            p_.requires_grad = False
        loss = nl.two_phase_loss([0, 0, 1], 1.0, zdim, 1)

        feed = None
        if STREAM:
            dl = importlib.import_module("i-dccrn-vae_amd.dataset.dataload")
            feed = iter(dl.DevicePrefetcher(dl.SyntheticMixtures(B, LEN, seed=123 + 1000 * rank), device))

        def fl():
            x_noisy, x_clean = (noisy, clean) if feed is None else next(feed)[:2]
            s = se(x_noisy, train=False)
            recon, pred = dec(s[11], s[0], s[8], s[9], s[10], train=True, pad='sig')
            return loss.phase_2_loss(pred, s[11], x_clean, recon, None, None, None, None)[0]
        return train_step([dec], fl), B, {"workload": "decoder fine-tune TRAIN step (config 5): frozen NSVAE encoder (eval) + "
                                          "decoder(mask, pad='sig', train) + SI-SNR + backward + Adam", "batch_per_gpu": B}
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


def cpu_baseline(seconds_budget=20.0, train=False):
    """The reference's op sequence on STOCK torch-CPU operators (nn.Conv2d / nn.ConvTranspose2d on MKLDNN, the fused CPU
    nn.LSTM, torch.stft / istft; oracle/stock_cpu.py, pinned to the oracle in tests/test_host_cpu.py) on a bounded sample
    of the SAME workload: B = 2 utterances per pass, all host cores of this job."""
    import torch
    from oracle import idccrn_oracle as O
    from oracle import stock_cpu
    cores = host_cores()
    torch.set_num_threads(cores)
    np_ = O.net_params(True, 32)
    pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
    m = pm.DCCRN_(NFFT, HOP, np_, True, "cpu", WIN, SKIP, "mask", False, None, None)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = O.synth_state_dict(shapes, 51)
    net = stock_cpu.StockDCCRN(np_, NFFT, HOP, WIN).load_reference_state(sd)
    B = 2
    noisy, clean, _ = make_inputs(B, 123, "cpu")

    def one():
        with torch.set_grad_enabled(train):
            est, pred = net(noisy, train=train)
            loss = O.multiple_recon_loss(pred, net.stft(clean), clean, est, [0.0, 0.0, 1.0])[0]
            if train:
                net.zero_grad(set_to_none=True)
                loss.backward()
            return loss
    one()
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 24:
            break
    what = "forward(train=True) + final_ete_loss + backward" if train else "forward + final_ete_loss"
    return {"value": round(B * n / el, 4), "unit": "utterances/sec", "cores": cores, "kind": "port",
            "sample": f"{n} passes of B={B} 4 s utterances, DCCRN-CL {what}, stock torch-CPU modules as the reference "
                      f"assembles them (oracle/stock_cpu.py), {cores} threads",
            "vs_reference_note": "build-container cross-check against the imported reference: BASELINE.md section 2"}


def self_launch(n_gpus: int) -> int:
    """Parent of an N-rank run started as `python bench.py --gpus N`: start `python -m torch.distributed.run` as a CHILD
    process (one rank per GPU, rendezvous on 127.0.0.1), pass its stderr through, relay the ranks' stdout (rank 0 prints the
    one JSON line) and return its exit code.  No GPU call and no exec in this process."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"self-launch: {' '.join(cmd)}")
    import signal
    import threading
    # own session: the launcher and its N ranks form one process group that can be ended as a whole -- on SIGINT / SIGTERM of
    # this parent, and at an overall deadline (a hung rank must not block the relay loop for ever): IDV_BENCH_DEADLINE seconds
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    state = {"why": None}

    def end_group(why):
        if proc.poll() is None:
            state["why"] = why
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(proc.pid, sig)
                except (ProcessLookupError, PermissionError):
                    break
                try:
                    proc.wait(timeout=10)
                    break
                except subprocess.TimeoutExpired:
                    continue

    deadline = float(os.environ.get("IDV_BENCH_DEADLINE", "1500"))
    timer = threading.Timer(deadline, end_group, args=(f"deadline of {deadline:.0f} s",))
    timer.daemon = True
    timer.start()
    old = {sig: signal.signal(sig, lambda n, f: (end_group(f"signal {n}"), sys.exit(128 + n))) for sig in (signal.SIGINT, signal.SIGTERM)}
    json_line = None
    try:
        for line in proc.stdout:
            line = line.rstrip("\n")
            if line.startswith("{") and '"metric"' in line:
                json_line = line
            elif line:
                print(line, file=sys.stderr, flush=True)
        rc = proc.wait()
    finally:
        timer.cancel()
        end_group("launcher exit")
        for sig, h in old.items():
            signal.signal(sig, h)
    if state["why"] and state["why"] != "launcher exit":
        log(f"self-launch: ranks ended by the launcher ({state['why']})")
        return rc if rc else 1
    if rc == 0 and json_line is None:
        log("self-launch: the ranks exited 0 without a JSON line")
        return 1
    if json_line is not None:
        print(json_line, flush=True)
    return rc


TRAIN_WORKLOADS = ("dccrn_cl_train", "cvae_train", "nsvae_train", "twophase_train")
# Rehearsal of the N-rank code path on ONE GPU: under `python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1` a
# process group of one rank is created over RCCL, the barriers, the timing all-reduce and (train workloads, with IDV_DP_FORCE=1)
# the Sync-CBN moment and gradient-bucket all-reduces are issued; the numbers are those of one GPU.
FORCE_DIST = os.environ.get("IDV_BENCH_FORCE_DIST", "0") == "1"


def kernel_name(cfg_id):
    if cfg_id == -95:
        return ("void (anonymous namespace)::wgrad_kernel<5, 2, 1, 1, 4, 1, 16, 2>(WgradArgs) x 3 Gauss products + wgrad_combine_kernel + "
                "wgrad_unpack_gauss_kernel")
    if cfg_id == -97:
        return "void (anonymous namespace)::wgrad_kernel<5, 2, 1, 1, 4, 1, 16, 2>(WgradArgs) + wgrad_unpack_conv_kernel"
    if cfg_id == -96:
        return "void (anonymous namespace)::wgrad_bf16_kernel(WgradArgs) + wgrad_unpack_conv_kernel"
    if cfg_id in (-99, -98):
        return f"void (anonymous namespace)::ctconv_c1_bf16_kernel<{'true' if cfg_id == -98 else 'false'}>(CgemmArgs)"
    if cfg_id == 1000001:
        return "void (anonymous namespace)::ctconv_c1_f32_kernel<false>(CgemmArgs, int)"
    if cfg_id in (5000002, 5000003):      # the conv form (cgemm_tw2.hip)
        return f"void (anonymous namespace)::cconv_tw2_kernel<{'true' if cfg_id == 5000003 else 'false'}, false, 0>((anonymous namespace)::Tw2Args)"
    if cfg_id in (5000000, 5000001):      # ops.TW_CFG (+ 1: taps to the left): the two phase kernels of a transposed-conv layer
        a, b = kernel_parts(cfg_id)
        return a.replace("((anonymous namespace)::TwArgs)", "") + " + " + b.replace("void (anonymous namespace)::cconv_tw_kernel", "")
    if 4000000 <= cfg_id < 5000000:       # ops.WINO_CFG + 1000 * transposed + idv_cconv_wino_config digits WM WN CIK
        tr, d = (cfg_id - 4000000) // 1000, str((cfg_id - 4000000) % 1000)
        if tr:                                # the two phase kernels of a transposed-conv layer (+ their half-tile variants)
            a, b = kernel_parts(cfg_id)
            return a.replace("((anonymous namespace)::WinoArgs)", "") + " + " + b.replace("void (anonymous namespace)::cconv_wino_kernel", "")
        tail = "2, 3, false, 2, 3" if d[2] == "2" else f"{d[2]}, 3, false, 1, {3 * int(d[2]) // 2}"      # two channels per chunk: 2 workgroups / CU
        return f"void (anonymous namespace)::cconv_wino_kernel<2, {d[0]}, {d[1]}, {tail}>((anonymous namespace)::WinoArgs)"
    if 3000000 <= cfg_id < 4000000:       # idv_cconv_gauss_config digits 3 MODE WM WN FO_T JC_W OCC
        d = str(cfg_id)
        occ = int(d[6])
        # cgemm_gauss.hip launch_cfg: 2 channels per K chunk for the one-workgroup 5- and 1-row conv tiles, 4 elsewhere
        cik = 2 if (d[1] == "0" and d[4] in "15" and (occ == 1 or d[2:4] == "14")) else 4
        jt = 32 * int(d[5]) * int(d[3])
        fr = 2 * int(d[4]) + 3 if d[1] == "0" else int(d[4]) + 2
        nbuf = 3 if 3 * cik * 3 * fr * (jt + 8) * 4 * occ <= 156 * 1024 else 2
        return (f"void (anonymous namespace)::cgemm_gauss_kernel<{d[1]}, {d[2]}, {d[3]}, {d[4]}, {d[5]}, {cik}, false, true, {nbuf}, {occ}>"
                "((anonymous namespace)::GaussArgs)")
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


def kernel_parts(cfg_id):
    """The kernel name(s), as rocprofv3 prints them, behind one timed launch of configuration cfg_id: one name, except for a
    transposed-conv layer on the Winograd form (even-row phase kernel + odd-row phase kernel)."""
    if cfg_id in (5000000, 5000001):
        left = "true" if cfg_id == 5000001 else "false"
        return [f"void (anonymous namespace)::cconv_tw_kernel<0, 8, {left}, 0, 2, true, false>((anonymous namespace)::TwArgs)",
                f"void (anonymous namespace)::cconv_tw_kernel<1, 8, {left}, 0, 2, false, false>((anonymous namespace)::TwArgs)"]
    if 4001000 <= cfg_id < 5000000:
        d = str((cfg_id - 4000000) % 1000)
        # even-row phase at two workgroups per CU: four channels per chunk (4 x 1 waves) / two (2 x 2)
        even = "4, 3, false, 2, 4" if d[:2] == "41" else ("2, 3, false, 2, 3" if d[:2] == "22" else f"{d[2]}, 3, false, 1, 0")
        return [f"void (anonymous namespace)::cconv_wino_kernel<0, {d[0]}, {d[1]}, {even}>((anonymous namespace)::WinoArgs)",
                f"void (anonymous namespace)::cconv_wino_kernel<1, {d[0]}, {d[1]}, {d[2]}, 3, false, 2, 8>((anonymous namespace)::WinoArgs)"]
    return [kernel_name(cfg_id)]


def group_launches(entries):
    out = {}
    for cfg_id, macs, e0, e1 in entries:
        g = out.setdefault(cfg_id, [0, 0.0, 0])
        g[0] += macs
        g[1] += e0.elapsed_time(e1) * 1e-3
        g[2] += 1
    return out


def roofline_of(launches, steps, step_seconds, precision, batch, workload):
    """Roofline block for the dominant contraction kernel of a timed region (HIP-event intervals on the launch stream)."""
    groups = group_launches(launches)
    if not groups:
        return None
    dom = max(groups, key=lambda k: groups[k][1])
    macs, secs, n = groups[dom]
    tot_macs = sum(g[0] for g in groups.values())
    tot_secs = sum(g[1] for g in groups.values())
    split = dom < 0 and dom not in (-97, -95)
    peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
    ach = 2 * macs / secs / 1e12
    traffic, tsrc = None, None
    tname = "r04_traffic_f32.json" if precision == "fp32" else "r04_traffic_bf16x3.json"
    tpath = os.path.join(ROOT, "profiles", tname)
    if os.path.exists(tpath) and batch == DEFAULT_BATCH and workload == "dccrn_cl":
        tks = [json.load(open(tpath))["kernels"].get(nm) for nm in kernel_parts(dom)]
        if all(tks):                           # a transposed-conv layer on the Winograd form is TWO kernels: their traffic adds up
            traffic, tsrc = sum(tk["hbm_bytes_per_launch"] for tk in tks), "profiles/" + tname
    r = {
        "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
        # `frac` = ALGORITHMIC flops / peak as the bench contract defines it (it exceeds 1 where the kernels execute fewer products than
        # the reference's four real convolutions x ten taps); the matrix pipe's own utilisation is `frac_executed` below
        "frac_algorithmic": round(ach / peak, 4),
        "traffic": traffic,
        "traffic_note": f"memory-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB from the committed rocprofv3 --pmc passes "
                        f"({tsrc}, same command and batch; the L2's fabric requests, Infinity-Cache hits included: DESIGN.md 5); "
                        "null when the run differs from that command",
        "peak_note": ("dense bf16 MFMA peak; achieved counts ALGORITHMIC fp32 flops (4 real convs), the kernel executes 3 bf16 "
                      "MFMA products per algorithmic product" if split else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
        "kernel": kernel_name(dom), "kernel_parts": kernel_parts(dom), "launches": n, "avg_launch_ms": round(secs / n * 1e3, 4),
        "algorithmic_gflop_per_launch": round(2 * macs / n / 1e9, 3),
        "all_conv_launches": {"achieved": round(2 * tot_macs / tot_secs / 1e12, 3),
                              "frac": round(2 * tot_macs / tot_secs / 1e12 / peak, 4),
                              "share_of_step_time": round(tot_secs / step_seconds / steps, 4)},
        "per_kernel": {kernel_name(k): {"tflops": round(2 * v[0] / v[1] / 1e12, 2), "ms_per_step": round(v[1] / steps * 1e3, 3)}
                       for k, v in sorted(groups.items())},
    }
    if 4000000 <= dom < 5000000:
        # Winograd-transformed frequency taps on top of the three complex products (cgemm_wino.hip): 7 of 10 real products per pair
        # of input rows x 3 of 4 per complex product; an odd row count pads the last pair (counted as executed work it is not)
        r["executed"] = round(0.525 * ach, 3)
        r["frac_executed"] = round(0.525 * ach / peak, 4)
        r["peak_note"] = ("dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32); achieved counts ALGORITHMIC flops (the reference's 4 real "
                          "convolutions x 10 taps per complex transposed convolution, SURVEY 8(d)); the kernels execute 3 real products "
                          "(Gauss) x 7 of 10 frequency-tap products (Winograd F(2,3) + F(2,2)): `executed` = 0.525 x achieved, "
                          "excluding the padding row of an odd row count.  One `launch` here = one decoder layer = the even-row and the odd-row "
                          "phase kernel back to back (kernel_parts): avg_launch_ms = the sum of their average durations in the rocprofv3 summary")
    if dom in (5000000, 5000001, 5000002, 5000003):
        # time-Winograd on top (cgemm_tw.hip): 3 of 4 products per pair of output columns as well
        r["executed"] = round(0.525 * 0.75 * ach, 3)
        r["frac_executed"] = round(0.525 * 0.75 * ach / peak, 4)
        r["peak_note"] = ("dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32); achieved counts ALGORITHMIC flops (the reference's 4 real "
                          "convolutions x 10 taps per complex transposed convolution, SURVEY 8(d)); the kernels execute 3 real products "
                          "(Gauss) x 7 of 10 frequency-tap products (Winograd F(2,3) + F(2,2)) x 3 of 4 time-tap products (F(2,2) over "
                          "pairs of output columns): `executed` = 0.394 x achieved, excluding padding (the last row pair of an odd row "
                          "count, the 28th tile slot of the odd-row phase).  One `launch` here = one decoder layer = the even-row and "
                          "the odd-row phase kernel back to back (kernel_parts)")
    if 3000000 <= dom < 4000000 or dom == -95:
        # three real products per complex product (Gauss, cgemm_gauss.hip): `achieved` counts the reference's 4 real convolutions
        r["executed"] = round(0.75 * ach, 3)
        r["frac_executed"] = round(0.75 * ach / peak, 4)
        r["peak_note"] = ("dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32); achieved counts ALGORITHMIC flops (the reference's 4 real "
                          "convolutions per complex convolution, SURVEY 8(d)), the kernel executes 3 (Gauss): `executed` = 0.75 x achieved")
    if split:
        r["executed_bf16_tflops"] = round(3 * ach, 1)
        r["frac_executed"] = round(3 * ach / peak, 4)
        r["vs_fp32_mfma_peak"] = round(ach / PEAK_F32_MFMA_TFLOPS, 3)
    return r


def two_stream_probe(step, steps, ops, utt_per_step, batch):
    """Two HIP streams (sub-batch pipelining, opt-in via IDV_STREAM_SPLIT=2): throughput + ONE bit-exactness check of its
    output against the one-stream output of the same input.  Leaves ops.STREAM_SPLIT as it found it."""
    import torch
    pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
    probe = {}
    orig_forward = pm.DCCRN_.forward
    keep = ops.STREAM_SPLIT

    def spy(self, signal, train=True):
        r = orig_forward(self, signal, train)
        probe["est"] = r[0]
        return r
    pm.DCCRN_.forward = spy
    try:
        ops.STREAM_SPLIT = 1
        step()
        one = probe["est"].clone()
        ops.STREAM_SPLIT = 2
        step()
        b_el, _, _ = timed(step, steps, ops)
        torch.cuda.synchronize()
        return {"two_stream": {"value": round(utt_per_step * steps / b_el, 3), "ms_per_step": round(b_el / steps * 1e3, 3),
                               "streams": ops.stream_split(batch)},
                "bit_exact": bool(torch.equal(one, probe["est"]))}
    finally:
        pm.DCCRN_.forward = orig_forward
        ops.STREAM_SPLIT = keep


def timed(step, steps, ops, barrier=None):
    import torch
    ops.LAUNCH_LOG = []
    if barrier:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    if barrier:
        barrier()
    el = time.perf_counter() - t0
    launches, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
    ops.coop_check(sync=False)              # a timed-out cooperative recurrence (NaN outputs) voids the measurement: raise, no line
    return el, launches, last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU per step (default 64; train workloads 32)")
    ap.add_argument("--workload", default="dccrn_cl")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the secondary bf16x3 record")
    ap.add_argument("--stream", action="store_true", help="twophase_train: stream fresh synthetic DNS3-shaped mixtures from the "
                    "host every step (pinned memory + copy stream) instead of re-using one resident batch")
    ap.add_argument("--precision", default=os.environ.get("IDV_PRECISION", "fp32"), choices=["fp32", "bf16x3"],
                    help="conv contraction arithmetic of the HEADLINE: exact fp32 MFMA (the reference's precision, default), "
                         "or split-bf16 (3 bf16 MFMAs, fp32 accumulate)")
    args = ap.parse_args()
    train = args.workload in TRAIN_WORKLOADS
    global STREAM
    STREAM = args.stream
    if args.batch is None:
        args.batch = 32 if train else (16 if args.workload.startswith("enhance") else DEFAULT_BATCH)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It never touches the GPU (no torch import,
        # no HIP call) and never exec()s: the N ranks are children of torch.distributed.run, rank 0's JSON line is relayed.
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ... "
                         "(or unset WORLD_SIZE and let bench.py start the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)                 # rehearsal of N ranks on fewer GPUs (IDV_BENCH_BACKEND=gloo) shares devices
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist_on = world > 1 or (FORCE_DIST and "WORLD_SIZE" in os.environ)
    if dist_on:
        # RCCL before any other GPU work of this process
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("IDV_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    ops = importlib.import_module("i-dccrn-vae_amd").ops
    ops.set_precision(args.precision)
    torch.set_grad_enabled(False)                # train workloads enable grad inside their step
    step, utt_per_step, cfg = build_workload(args.workload, args.batch, device, rank)

    log(f"workload {args.workload} B={args.batch} precision={args.precision} built; warmup {args.warmup}")
    for w in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {w} done")

    def barrier():
        if dist_on:
            dist.barrier()

    elapsed, launches, last = timed(step, args.steps, ops, barrier)
    loss_val = float(last)
    log(f"timed {args.steps} steps in {elapsed:.3f} s")
    n_streams = ops.stream_split(args.batch) if args.workload == "dccrn_cl" else 1

    dt = importlib.import_module("i-dccrn-vae_amd.utils.dist_timing")
    red_dev = device if (not dist_on or dist.get_backend() == "nccl") else torch.device("cpu")
    value, elapsed_max = dt.job_throughput(elapsed, float(utt_per_step * args.steps), red_dev, force=dist_on)
    roofline = roofline_of(launches, args.steps, elapsed / args.steps, args.precision, args.batch, args.workload)
    if roofline is not None and n_streams > 1:
        roofline["note"] = (f"timed region ran {n_streams} sub-batch streams (IDV_STREAM_SPLIT): event intervals include CU "
                            "time-slicing with the other stream")

    # ---- informational: the same fp32 workload with the opt-in two-stream sub-batch split (never `value`), checked bit for bit
    two_f32 = None
    if (world == 1 and not args.no_alt and args.workload == "dccrn_cl" and args.precision == "fp32" and n_streams == 1):
        two_f32 = two_stream_probe(step, args.steps, ops, utt_per_step, args.batch)
        log(f"fp32 two-stream (opt-in): {two_f32}")

    # ---- secondary record: the split-bf16 mode on the same workload (narrower arithmetic than the reference: never `value`)
    alt = None
    if (world == 1 and not args.no_alt and args.workload == "dccrn_cl" and args.precision == "fp32"):
        keep_split = ops.STREAM_SPLIT
        try:
            ops.set_precision("bf16x3")
            ops.STREAM_SPLIT = 1
            for _ in range(2):
                step()
            a_el, a_launch, a_last = timed(step, args.steps, ops)
            alt = {"dtype": "bf16x3 (split-fp32 operands on the bf16 MFMA: w*x ~ w_hi*x_hi + w_hi*x_lo + w_lo*x_hi, fp32 "
                            "accumulate; ~16-17 significant bits per operand, waveform 5e-6 from the fp32 reference)",
                   "value": round(utt_per_step * args.steps / a_el, 3), "unit": "utterances/sec",
                   "ms_per_step": round(a_el / args.steps * 1e3, 3), "streams": 1, "loss": float(a_last),
                   "roofline": roofline_of(a_launch, args.steps, a_el / args.steps, "bf16x3", args.batch, args.workload)}
            ts = two_stream_probe(step, args.steps, ops, utt_per_step, args.batch)
            alt["two_stream"], alt["two_stream_bit_exact"] = ts["two_stream"], ts["bit_exact"]
        finally:
            ops.STREAM_SPLIT = keep_split
            ops.set_precision(args.precision)
        log(f"alt bf16x3: {alt['value']} utt/s, two-stream {alt.get('two_stream')}, bit exact {alt.get('two_stream_bit_exact')}")

    if rank == 0:
        out = {
            "metric": METRIC, "value": round(value, 3), "unit": "utterances/sec",
            "n_gpus": world, "rccl_ranks": (dist.get_world_size() if dist_on else 1),
            "backend": ((dist.get_backend() + (" (RCCL over xGMI)" if dist.get_backend() == "nccl" else " (rehearsal, no RCCL)"))
                        if dist_on else "none (single process)"),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16x3 (split-fp32 operands on bf16 MFMA, fp32 accumulate)",
            "data": "synthetic (0.1*N(0,1) clean + noise, seeded; random-init weights)",
            "config": dict(cfg, utterance="4 s @ 16 kHz (64000 samples, 641 frames)",
                           parallelism=(f"data-parallel x{world}: sharded batch, Sync-CBN moment all-reduce, gradient all-reduce (RCCL)"
                                        if train else f"replicas x{world}"),
                           streams=n_streams, loss=loss_val, input=("streamed from the host per step" if args.stream else "resident in HBM")),
            "roofline": roofline,
        }
        if two_f32 is not None:
            out["two_stream"] = dict(two_f32["two_stream"], bit_exact=two_f32["bit_exact"],
                                     note="opt-in IDV_STREAM_SPLIT=2 (DESIGN.md 5.1); `value` is the one-stream default")
        if alt is not None:
            out["alt"] = alt
        if world == 1 and not args.no_cpu_baseline and args.workload in ("dccrn_cl", "dccrn_cl_train"):
            log("cpu baseline ...")
            out["cpu_baseline"] = cpu_baseline(train=train)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
