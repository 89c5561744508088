"""bench.py - frames/s of TDC-Video's video-encoding hot path (encode + compress) on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W                      (N=1; N>1: starts its own N ranks, one per GPU)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W   (N>1, one rank/GPU)

A "step" = one pass of the hot path (S0-S10 of SURVEY.md 3.2: DINOv2-g + SigLIP-so400m towers, adjacent-frame
segmentation, aux projectors, SVA 576->144, mm_projector, unpad/newline, batched Q-Former TDC compressor,
vision_proj + L2, token emission) over ONE synthetic video of T frames whose pixels are already resident in HBM.
For N>1 the T frames of the one video are sharded by contiguous ranges over the ranks (strong scaling), with the RCCL
exchanges of tdc-video_amd/dist.py (boundary-frame DINOv2 features, similarities all-gather, key-frame query hand-off,
emitted-token all-gather).
Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` (dominant kernel = tdc_gemm MFMA
kernel, timed live with events on the launch stream) and `cpu_baseline` (the oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
# what back-to-back bf16 16x16x32 MFMAs on dense random REGISTER operands sustain at the chip's power limit (no memory traffic;
# all-zero operands: 2448): tools/mfma_power.cpp, profiles/r02_mfma_power.log.  Reported beside `frac`, never instead of it.
MFMA_DENSE_SUSTAINED_TFLOPS = 1899.0
PMC_SUMMARY = "r05_gemm_pmc_summary.json"   # profiles/: counters of the current code (tools/run_gemm_pmc.sh), see `traffic_source`


def model_cfg(H, K, T):
    return dict(hidden_size=H, vision_hidden_size=1024, num_query_group=1, query_num_list=[144], image_token_len=144,
                mm_vision_tower_aux_token_len_list=[576, 576], connector_depth=3, context_token_num=K,
                tokenizer_model_max_length=10 ** 9, inference_max_length=16, max_num_segments=24, model_type="qwen2",
                text_input=True, add_static=True)


def random_state_dict(H, K, device, gen, siglip_px=384, dino_pos_grid=37, std=0.02):
    """Random-init weights of the reference architecture, reference state-dict names, fp32."""
    sd = {}

    def w(name, *shape, s=std):
        sd[name] = torch.randn(*shape, device=device, generator=gen) * s

    def ln(name, n):
        sd[name + ".weight"] = 1.0 + 0.05 * torch.randn(n, device=device, generator=gen)
        sd[name + ".bias"] = 0.02 * torch.randn(n, device=device, generator=gen)

    def lin(name, n, k, bias=True):
        w(name + ".weight", n, k)
        if bias:
            w(name + ".bias", n)
    # SigLIP-so400m/14: 1152, 27 layers, 16 heads, MLP 4304
    p = "vision_tower_aux_list.0.vision_tower."
    D = 1152
    w(p + "embeddings.patch_embedding.weight", D, 3, 14, 14); w(p + "embeddings.patch_embedding.bias", D)
    w(p + "embeddings.position_embedding.weight", (siglip_px // 14) ** 2, D)
    for i in range(27):
        q = p + "encoder.layers.%d." % i
        ln(q + "layer_norm1", D); ln(q + "layer_norm2", D)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(q + "self_attn." + n, D, D)
        lin(q + "mlp.fc1", 4304, D); lin(q + "mlp.fc2", D, 4304)
    # DINOv2-giant/14: 1536, 40 layers, 24 heads, SwiGLU 4096, LayerScale
    p = "vision_tower_aux_list.1.vision_tower."
    D = 1536
    w(p + "embeddings.patch_embeddings.projection.weight", D, 3, 14, 14)
    w(p + "embeddings.patch_embeddings.projection.bias", D)
    w(p + "embeddings.cls_token", 1, 1, D); w(p + "embeddings.position_embeddings", 1, 1 + dino_pos_grid ** 2, D)
    for i in range(40):
        q = p + "encoder.layer.%d." % i
        ln(q + "norm1", D); ln(q + "norm2", D)
        for n in ("query", "key", "value"):
            lin(q + "attention.attention." + n, D, D)
        lin(q + "attention.output.dense", D, D)
        sd[q + "layer_scale1.lambda1"] = torch.full((D,), 1.0, device=device)
        sd[q + "layer_scale2.lambda1"] = torch.full((D,), 1.0, device=device)
        lin(q + "mlp.weights_in", 8192, D); lin(q + "mlp.weights_out", D, 4096)
    ln(p + "layernorm", D)
    # connector
    C = 1024
    for i, dv in enumerate((1152, 1536)):
        q = "mm_projector_aux_%d." % i
        lin(q + "0", C, dv); lin(q + "2", C, C); ln(q + "3", C)
    w("vision_query", 1, C, s=1.0)
    for i in range(3):
        q = "vision_sampler_0.layers.%d." % i
        w(q + "pos_embed_0", 4, C, s=1.0); w(q + "pos_embed_1", 4, C, s=1.0)
        lin(q + "proj_context", C, C, False); lin(q + "proj_in", C, 2 * C, False)
        lin(q + "proj_out.linear_1", C, C, False); lin(q + "proj_out.linear_2", C, C, False)
        ln(q + "norm", C)
        ln(q + "cross_attn.q_proj.0", C); lin(q + "cross_attn.q_proj.1", C, C, False)
        for tw in range(2):
            for kv in "kv":
                ln(q + "cross_attn.%s_proj_%d.0" % (kv, tw), C); lin(q + "cross_attn.%s_proj_%d.1" % (kv, tw), C, C, False)
        lin(q + "cross_attn.o_proj", C, C, False)
    lin("mm_projector.0", H, C); lin("mm_projector.2", H, H)
    w("image_newline", H, s=1.0); w("frame_seg", H, s=1.0)
    # Q-Former (bert-base: 12 x 768, 12 heads, FFN 3072, cross-attention in even layers)
    Dq = 768
    q = "Qformer.bert."
    w(q + "embeddings.word_embeddings.weight", 30522, Dq); w(q + "embeddings.position_embeddings.weight", 512, Dq)
    ln(q + "embeddings.LayerNorm", Dq)
    for i in range(12):
        l = q + "encoder.layer.%d." % i
        for n in ("query", "key", "value"):
            lin(l + "attention.self." + n, Dq, Dq)
        lin(l + "attention.output.dense", Dq, Dq); ln(l + "attention.output.LayerNorm", Dq)
        if i % 2 == 0:
            lin(l + "crossattention.self.query", Dq, Dq)
            lin(l + "crossattention.self.key", Dq, H); lin(l + "crossattention.self.value", Dq, H)
            lin(l + "crossattention.output.dense", Dq, Dq); ln(l + "crossattention.output.LayerNorm", Dq)
        for a, b in (("intermediate", "output"), ("intermediate_query", "output_query")):
            lin(l + a + ".dense", 3072, Dq); lin(l + b + ".dense", Dq, 3072); ln(l + b + ".LayerNorm", Dq)
    lin("query_proj", Dq, H); lin("vision_proj", H, Dq)
    return sd


def build_mixin_lm(cfg, sd, dev, dtype, extra=None, vocab=2048):
    """The drop-in boundary around the same weights (`--via-mixin`, tests): a stub language model - `config`, `embed_tokens`,
    `dtype`, nothing else; the LLM is stubbed in every BASELINE config - hosting tdc-video_amd/model.py's CambrianMetaModel /
    CambrianMetaForCausalLM exactly as CambrianQwenForCausalLM hosts the reference's mixins
    (/root/reference/tdc/language_model/cambrian_qwen.py:205-232), its parameters loaded from `sd` (reference names).
    cfg: model_cfg(...) dict; extra: the non-reference keys (tdc_frame_cap, tdc_tower_dtype, ...)."""
    import types
    import torch.nn as nn
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.model import CambrianMetaForCausalLM, CambrianMetaModel
    ns = types.SimpleNamespace(mm_vision_tower_aux_list=["siglip/CLIP-ViT-SO400M-14-384", "facebook/dinov2-giant-res378"],
                               mm_projector_type="sva", connector_only=True, tokenizer_padding_side="right", **cfg)
    for k, v in (extra or {}).items():
        setattr(ns, k, v)
    # SigLIP's learned position table follows the input size (336 px: 24 x 24 positions instead of 27 x 27)
    from tdc_video_amd.model import SIGLIP_SO400M
    n_pos = sd["vision_tower_aux_list.0.vision_tower.embeddings.position_embedding.weight"].shape[0]
    if n_pos != SIGLIP_SO400M["n_pos"]:
        ns.tdc_tower_archs = {"siglip": dict(SIGLIP_SO400M, n_pos=n_pos)}

    class StubBase(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config = config
            self.embed_tokens = nn.Embedding(vocab, config.hidden_size)

        @property
        def dtype(self):
            return dtype

    class StubModel(CambrianMetaModel, StubBase):
        pass

    class StubLM(nn.Module, CambrianMetaForCausalLM):
        def __init__(self, config):
            nn.Module.__init__(self)
            self.config = config
            self.model = StubModel(config)

        def get_model(self):
            return self.model
    with torch.device(dev):
        lm = StubLM(ns)
        for t in lm.model.vision_tower_aux_list:
            t.load_model()
    m = lm.model
    missing, unexpected = m.load_state_dict({k: v for k, v in sd.items() if not k.startswith("vision_tower_aux_list")},
                                            strict=False)
    assert not unexpected, unexpected[:5]
    for i, t in enumerate(m.vision_tower_aux_list):
        pre = "vision_tower_aux_list.%d.vision_tower." % i
        tsd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        if i == 1 and "embeddings.position_embeddings" in tsd:      # the stub allocates DINOv2's 37 x 37 table (518 px)
            assert tsd["embeddings.position_embeddings"].shape == t.vision_tower.state_dict()["embeddings.position_embeddings"].shape
        miss, unexp = t.vision_tower.load_state_dict(tsd, strict=False)
        assert not unexp and not miss, (miss[:3], unexp[:3])
    m.embed_tokens.to(device=dev, dtype=dtype)
    return lm


def synth_video(lo, hi, px, device, dtype, seed=1234, scene_len=21):
    """frames [lo, hi) of a deterministic synthetic video (scene = constant base image + per-frame noise)."""
    g = torch.Generator(device=device)
    out = torch.empty(hi - lo, 3, px, px, device=device, dtype=dtype)
    for t in range(lo, hi):
        sc = t // scene_len
        g.manual_seed(seed * 7919 + sc)
        base = torch.rand(3, px, px, device=device, generator=g) * 2 - 1
        g.manual_seed(seed * 104729 + 1000003 + t)
        out[t - lo] = (base + 0.1 * torch.randn(3, px, px, device=device, generator=g)).to(dtype)
    return out


def cpu_baseline(sd_cpu, cfg, H, K, px_s, px_d):
    """The oracle (CPU restatement, torch fp32 eager) timed on this host's cores on a bounded sample of the same workload
    (BASELINE.md section 3): both towers at full depth on a 4-frame batch - 1 warm-up (two layers per tower: same shapes, warms
    the thread pool and the primitive caches) + 3 timed runs, median and spread reported -, connector + compressor on a
    32-frame clip; per-frame costs are added."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tdc_oracle as orc
    cores = torch.get_num_threads()
    W = dict(sd_cpu)
    Ws = {k[len("vision_tower_aux_list.0.vision_tower."):]: v for k, v in W.items() if k.startswith("vision_tower_aux_list.0.")}
    Wd = {k[len("vision_tower_aux_list.1.vision_tower."):]: v for k, v in W.items() if k.startswith("vision_tower_aux_list.1.")}

    def cut(sd, pre, n):
        return {k: v for k, v in sd.items() if not k.startswith(pre) or int(k[len(pre):].split(".")[0]) < n}
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        n_v = 4
        x = torch.rand(n_v, 3, px_s, px_s, generator=g) * 2 - 1
        y = torch.rand(n_v, 3, px_d, px_d, generator=g) * 2 - 1
        orc.siglip_tower(x, cut(Ws, "encoder.layers.", 2), 16)         # warm-up, outside the timed region
        orc.dino_tower(y, cut(Wd, "encoder.layer.", 2), 24)
        runs = []
        for _ in range(3):
            t0 = time.time()
            orc.siglip_tower(x, Ws, 16)
            orc.dino_tower(y, Wd, 24)
            runs.append((time.time() - t0) / n_v)
        runs.sort()
        t_v = runs[1]
        # compressor stage on a 32-frame clip of post-tower features
        Tc = 32
        sig = torch.randn(Tc, 576, 1152, generator=g)
        din = torch.randn(Tc, 576, 1536, generator=g)
        t1 = time.time()
        aux = [orc.mm_projector_aux(sig, W, 0), orc.mm_projector_aux(din, W, 1)]
        q, _ = orc.sva(aux, W["vision_query"][0], [(384, 384)] * Tc, W, 12)
        feat = orc.mm_projector(q, W)
        frames, _ = orc.unpad_newline(feat, [(384, 384)] * Tc, W["image_newline"])
        sims = orc.adjacent_cosine(din)
        seg = orc.select_segments(sims, 24)
        orc.tdc_compress(torch.stack(frames), seg, torch.arange(12) + 1000, W, K, 12, 10 ** 9)
        t_c = (time.time() - t1) / Tc
    fps = 1.0 / (t_v + t_c)
    return dict(value=round(fps, 4), unit="frames/s", cores=cores, kind="port",
                spread=[round(1.0 / (runs[2] + t_c), 4), round(1.0 / (runs[0] + t_c), 4)],
                sample="oracle/tdc_oracle.py fp32 eager, %d torch threads: both towers at full depth on a %d-frame batch, 1 warm-up + 3 "
                       "timed runs (median %.2f s/frame, min %.2f, max %.2f) + connector & TDC compressor on a 32-frame clip "
                       "(%.3f s/frame), per-frame costs added; `spread` = frames/s at the slowest / fastest tower run"
                       % (cores, n_v, t_v, runs[0], runs[2], t_c))


def pmc_leg(gemms, budget_s, H, live_gemm_ms):
    """The in-run counter leg (see the call site): -> (traffic bytes per launch, MFMA-busy fraction, MFMA-busy of the Q-Former's
    stacked K/V GEMM, provenance string, full per-shape summary) or None when it could not run."""
    import collections
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    binary = os.path.join(ROOT, "tools", "bin", "gemm_pmc")
    t0 = time.monotonic()
    if not os.path.exists(binary):          # normally built by __graft_entry__.build(); a 10-second hipcc job otherwise
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            return None
        os.makedirs(os.path.dirname(binary), exist_ok=True)
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-o", binary, os.path.join(ROOT, "tools", "gemm_pmc.cpp"),
                            "-L" + os.path.join(ROOT, "tdc-video_amd"), "-ltdc_hip", "-Wl,-rpath,$ORIGIN/../../tdc-video_amd"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write("bench.py: building tools/bin/gemm_pmc failed: %s\n" % r.stderr[-300:])
            return None
    cnt = collections.Counter((r["M"], r["N"], r["K"], r["act"], r["res"], r["out_f32"]) for r in gemms)
    shapes = [k + (c,) for k, c in sorted(cnt.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1])]
    work = tempfile.mkdtemp(prefix="tdc_bench_pmc_")
    try:
        shapes_file = os.path.join(work, "shapes.txt")
        with open(shapes_file, "w") as fh:
            for sh in shapes:
                fh.write("%d %d %d %d %d %d %d\n" % sh)
        log = []
        done = pmc_summary.collect(binary, shapes_file, work, reps=2, timeout_s=budget_s - (time.monotonic() - t0), log=log)
        if "FETCH_SIZE" not in done or "WRITE_SIZE" not in done:
            sys.stderr.write("bench.py: in-run counter leg incomplete (%s): %s\n" % (done, log[-1:] if log else "not started"))
            return None
        summ = pmc_summary.summarise(work, shapes, reps=2)
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("bench.py: in-run counter leg failed: %r\n" % (e,))
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)
    kv = max((r.get("mfma_busy_frac", 0) for r in summ["shapes"] if r["N"] >= 9216 and r["K"] == H), default=None)
    src = "measured in this run: %d rocprofv3 --pmc passes (one counter each: %s) over tools/bin/gemm_pmc, a torch-free replay of this " \
          "run's %d GEMM launches (%d shapes, the second of two launches per shape) through the same libtdc_hip.so, started after " \
          "the timed region; %.0f s" % (len(done), ", ".join(done), summ["launches_per_step"], len(shapes), time.monotonic() - t0)
    if summ.get("gemm_ms_per_step_at_collection"):
        src += "; live / replayed GEMM time per step = %.3f" % (live_gemm_ms / summ["gemm_ms_per_step_at_collection"])
    summ["collected_by"] = "bench.py --pmc (in-run leg)"
    return round(summ["per_launch_hbm_bytes"]), summ.get("mfma_busy_frac"), kv, src, summ


def self_launch(n, guard_s=480.0, cmd=None):
    """`python bench.py --gpus N` without a launcher (the reference's eval drivers start one worker per GPU the same way,
    /root/reference/eval/eval_mlvu.py:129-157): N fresh child processes of this script, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set; rank 0's stdout (the JSON line) is this process's stdout,
    the other ranks' stdout goes to stderr.  Returns the worst child return code; when one child fails the others are
    terminated (a failed rank is never restarted in place - a child that has touched the GPU is never re-exec'd).
    guard_s: wall-clock guard (`--launch-timeout`).  A rank stuck in a collective must not hold the caller's GPU lease for
    hours (the reference's own 8-hour NCCL timeout, tdc/train.py:892, is the anti-pattern): when the guard expires every live
    child is terminated (then killed) and the launcher returns 124, printing no JSON line of its own.
    cmd: the child command line (default: this script with this process's arguments; tests pass a stand-in)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    if cmd is None:
        cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("TDC_BENCH_CU_SPLIT") and n == 2:
            # experiment hook (never set by the driver; tools/half_chip_gate.sh): both ranks on ONE GPU, each confined to half
            # of the CUs of every XCD (mask bit i = XCD i % 8, CU slot i / 8: every XCD keeps 16 CUs per rank - a mask that
            # empties an XCD would leave the workgroups dispatched to it without a CU).  Set before the child touches the GPU.
            env["ROC_GLOBAL_CU_MASK"] = "0x" + ("f" * 32 if r == 0 else "f" * 32 + "0" * 32)
            env["TDC_BENCH_PERSIST_GRID"] = "128"      # the child passes it to tdc_gemm_set_persistent_grid (the library reads no environment)
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else sys.stderr))
    worst = 0
    live = set(range(n))
    t_end = time.monotonic() + guard_s if guard_s and guard_s > 0 else None
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and worst == 0:
                worst = rc
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, rc))
                for o in live:
                    procs[o].terminate()
        if live and t_end is not None and time.monotonic() > t_end:
            sys.stderr.write("bench.py: launch guard of %.0f s expired with rank(s) %s still running; stopping them\n"
                             % (guard_s, sorted(live)))
            for o in live:
                procs[o].terminate()
            worst = worst or 124
            break
        time.sleep(0.05)
    for pr in procs:
        try:
            pr.wait(timeout=30)
        except subprocess.TimeoutExpired:
            pr.kill()
            pr.wait()
    return worst


def rank_guard(seconds):
    """In-rank wall-clock guard (`--rank-timeout`): under a foreign launcher (torch.distributed.run) nothing else bounds a
    rank that waits for a peer which never arrives.  A daemon timer; on expiry the rank says so on stderr and leaves with
    code 124 (plain exit - never an exec - so the launcher sees a failed rank and stops the others)."""
    import threading
    if not seconds or seconds <= 0:
        return None

    def fire():
        sys.stderr.write("bench.py: rank %s exceeded --rank-timeout %.0f s; exiting 124\n" % (os.environ.get("RANK", "0"), seconds))
        sys.stderr.flush()
        os._exit(124)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--K", type=int, default=144, help="context_token_num (BASELINE configs: 144; reference default 16)")
    ap.add_argument("--hidden", type=int, default=3584, help="LLM embed dim (Qwen2-7B)")
    ap.add_argument("--dtype", default="mixed", choices=["mixed", "bf16", "fp16", "fp8"],
                    help="mixed (default): the two ViT towers in bf16, connector + Q-Former in fp16 - the type whose compressed "
                         "tokens pass the north_star's 1e-3 against the fp32 oracle at full depth (tests/test_hip_configs.py); "
                         "fp8 (BASELINE config 5): e4m3 operands for the towers' LayerNorm-fed GEMMs (qkv, fc1) through "
                         "v_mfma_f32_16x16x128_f8f6f4, bf16 everywhere else")
    ap.add_argument("--res", default="fp16", choices=["fp16", "fp32"],
                    help="the towers' residual stream in HBM: fp16 (default; the reference's own arithmetic - its HF towers run "
                         "under torch_dtype=float16 - 4 B per element and residual add, sums formed in fp32 and rounded once) or fp32 "
                         "(8 B per element: rounds 1-3)")
    ap.add_argument("--dino-dtype", default="", choices=["", "bf16", "fp16"], help="operand type of the DINOv2 tower alone "
                    "(config.tdc_dino_dtype; default: the towers' type): fp16 keeps the a5 similarities at the reference's precision")
    ap.add_argument("--selection-refine", type=int, default=-1, choices=[-1, 0, 1], help="a5 at the reference's precision under bf16 "
                    "DINOv2 operands (VideoEncoder.selection_refine): -1 = automatic (on for bf16 DINOv2 operands), 0 / 1.  The synthetic "
                    "video's decisive similarity ranks are 0.43 apart, so nothing is re-encoded in the timed region either way")
    ap.add_argument("--ln-fuse", action="store_true", help="the towers' pre-LayerNorms folded into the neighbouring GEMMs "
                    "(VideoEncoder(ln_fuse=True)); over the fp16 residual stream this needs --dtype fp16: the consumer GEMMs read the "
                    "stream itself, no LayerNorm kernel runs inside the layer loop")
    ap.add_argument("--px", type=int, default=384, help="SigLIP input size (DINO uses px-6: 378); 336 -> 336/336")
    ap.add_argument("--fp8-level", type=int, default=1, choices=[1, 2, 3],
                    help="--dtype fp8: 1 = qkv / fc1 (quantised by the LayerNorm kernel), 2 = also out-proj / fc2, "
                         "3 = as 2 with the MLP hidden written as e4m3 by fc1 itself")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tower-batch", type=int, default=512)
    ap.add_argument("--two-streams", type=int, default=-1, help="run the two towers on two HIP streams: 0 / 1, -1 = on at <= 128 "
                    "frames per rank (small batches leave partly filled last tile rounds, which the other tower's workgroups fill)")
    ap.add_argument("--gemm-shape-times", default="", help="write per-shape GEMM times of the profiled step to this file")
    ap.add_argument("--xattn-mode", type=int, default=1, choices=[0, 1, 2], help="Q-Former cross-attention block: 0 = per-kernel "
                    "sequence (stacked K/V GEMM, q GEMM, tdc_attention, dense GEMM, LayerNorm), 1 = output projection + residual + "
                    "LayerNorm in one kernel behind q GEMM + tdc_attention, 2 = the whole block in one kernel per layer "
                    "(tdc_qformer_xattn)")
    ap.add_argument("--recompute-halo", action="store_true", help="N > 1: every rank re-encodes its right neighbour's first frame "
                    "through DINOv2 instead of receiving its features point to point (fallback form of the boundary exchange)")
    ap.add_argument("--audio", action="store_true", help="BASELINE config 4: + T seconds of 16 kHz audio through BEATs "
                    "on the device, 50 audio tokens per frame in the Q-Former KV (1 GPU only)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "on", "off"], help="N = 1: after the timed region collect FETCH_SIZE / "
                    "WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE in fresh child processes (rocprofv3 --pmc over a torch-free "
                    "replay of this run's GEMM launches) and report them as roofline.traffic / mfma_busy_pmc; auto = when rocprofv3 exists")
    ap.add_argument("--pmc-budget", type=float, default=90.0, help="wall-clock budget of that leg in seconds")
    ap.add_argument("--pmc-out", default="", help="write the leg's per-shape summary (JSON) to this file")
    ap.add_argument("--launch-timeout", type=float, default=-1.0, help="N > 1 started as plain `python bench.py --gpus N`: wall-clock "
                    "guard of the launcher in seconds; on expiry every rank is stopped and the exit code is 124 (0 = no guard; default "
                    "600 + 4 x (steps + warmup))")
    ap.add_argument("--rank-timeout", type=float, default=-1.0, help="wall-clock guard inside every rank in seconds (0 = none; "
                    "default 600 + 4 x (steps + warmup): a step takes 1.2 s at T = 512 on one GPU)")
    ap.add_argument("--collective-timeout", type=float, default=180.0, help="N > 1: timeout of the process group (rendezvous and "
                    "every collective) in seconds")
    ap.add_argument("--via-mixin", action="store_true", help="drive the step through the drop-in boundary instead of VideoEncoder: "
                    "a stub language model hosting model.CambrianMetaForCausalLM, one call of prepare_inputs_labels_for_multimodal("
                    "input_ids, ..., images=[siglip[1,T,...], dino[1,T,...]], image_sizes, video_indices=[None], prompts=[ids]) per "
                    "step with config.tdc_frame_cap = T (N > 1: + config.tdc_shard_frames, every rank holds the whole video as the "
                    "reference's eval workers do); the JSON line says config.entry = \"mixin\"")
    ap.add_argument("--dump-gemm-shapes", default=None, help="write the GEMM launches of one step (for tools/gemm_pmc)")
    args = ap.parse_args()
    for k in ("launch_timeout", "rank_timeout"):
        if getattr(args, k) < 0:
            setattr(args, k, 600.0 + 4.0 * (args.steps + args.warmup))

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above has touched the GPU (no torch.cuda call), and
        # nothing below this branch runs in this process.
        sys.exit(self_launch(args.gpus, args.launch_timeout))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus N` by itself, or under "
                 "torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    # test hooks (never set by the driver): TDC_BENCH_ONE_GPU=1 maps every rank onto GPU 0 and TDC_DIST_BACKEND=gloo
    # swaps the transport, so the N>1 code path can be exercised on a 1-GPU box
    if os.environ.get("TDC_BENCH_ONE_GPU"):
        local = 0
    guard = rank_guard(args.rank_timeout)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import datetime
        import torch.distributed as dist
        backend = os.environ.get("TDC_DIST_BACKEND", "nccl")
        # a bounded timeout on the rendezvous and on every collective: a rank that never arrives fails the job within minutes
        tmo = datetime.timedelta(seconds=args.collective_timeout)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import ops
    from tdc_video_amd.pipeline import VideoEncoder
    from tdc_video_amd import segment as seg
    if os.environ.get("TDC_BENCH_PERSIST_GRID"):     # experiment hook of self_launch's CU split (tools/half_chip_gate.sh)
        from tdc_video_amd import lib as tlib
        tlib.load().tdc_gemm_set_persistent_grid(int(os.environ["TDC_BENCH_PERSIST_GRID"]))

    dtype = torch.float16 if args.dtype in ("fp16", "mixed") else torch.bfloat16          # connector, Q-Former, outputs
    tower_dtype = torch.bfloat16 if args.dtype == "mixed" else dtype                       # the two ViT towers
    T, K, H = args.frames, args.K, args.hidden
    px_s = args.px
    px_d = args.px - 6 if args.px == 384 else args.px
    cfg = model_cfg(H, K, T)
    gen = torch.Generator(device=dev).manual_seed(0)
    sd = random_state_dict(H, K, dev, gen, siglip_px=px_s)
    two_streams = args.two_streams if args.two_streams >= 0 else int((T + world - 1) // world <= 128)
    res16 = args.res == "fp16"
    product_setting = {"dtype": str(dtype).replace("torch.", ""), "tdc_tower_dtype": str(tower_dtype).replace("torch.", ""),
                       "tdc_tower_res_dtype": "float16" if res16 else "float32", "tdc_tower_batch": args.tower_batch,
                       "tdc_fp8_towers": args.fp8_level if args.dtype == "fp8" else 0, "tdc_frame_cap": T,
                       "tdc_two_streams": bool(two_streams)}
    if args.ln_fuse:
        product_setting["tdc_ln_fuse"] = True
    if args.selection_refine >= 0:
        product_setting["tdc_selection_refine"] = bool(args.selection_refine)
    if args.dino_dtype:
        product_setting["tdc_dino_dtype"] = {"bf16": "bfloat16", "fp16": "float16"}[args.dino_dtype]
    lm = None
    if args.via_mixin:
        if args.audio or args.recompute_halo:
            sys.exit("bench.py: --via-mixin covers the video path (no --audio / --recompute-halo)")
        extra = {k: v for k, v in product_setting.items() if k != "dtype"}
        if world > 1:
            extra["tdc_shard_frames"] = True
        lm = build_mixin_lm(cfg, sd, dev, dtype, extra)
        enc = lm.get_model().tdc_engine(device=dev, dtype=dtype)       # built by the mixin from its own parameters
    else:
        enc = VideoEncoder(sd, cfg, dtype=dtype, device=dev, siglip_heads=16, dino_heads=24, qformer_heads=12,
                           tower_batch=args.tower_batch, fp8_towers=args.fp8_level if args.dtype == "fp8" else 0,
                           tower_dtype=tower_dtype, tower_res_dtype=torch.float16 if res16 else None, ln_fuse=args.ln_fuse,
                           dino_dtype={"bf16": torch.bfloat16, "fp16": torch.float16, "": None}[args.dino_dtype],
                           selection_refine=None if args.selection_refine < 0 else bool(args.selection_refine))
        enc.two_streams = bool(two_streams)
    enc.xattn_mode = args.xattn_mode
    wav = None
    if args.audio:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from bench_beats import random_beats_state
        from tdc_video_amd.beats import BEATS_ITER3_CFG, BeatsEncoder
        from tdc_video_amd import weights as Wt
        enc.cfg["audio_input"] = True
        enc.c.audio_proj = Wt.make_lin(torch.randn(H, 768, device=dev, generator=gen) * 0.02,
                                       torch.zeros(H, device=dev), dtype, dev)
        enc.beats = BeatsEncoder(random_beats_state(BEATS_ITER3_CFG), BEATS_ITER3_CFG, dtype=dtype, device=dev)
        # 1 frame per second; every rank holds the waveform (16 MB at T = 512) and encodes only its own 10-s windows
        gw = torch.Generator(device=dev).manual_seed(77)
        wav = (0.1 * torch.randn(1, 16000 * T, device=dev, generator=gw)).half()
    sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sd_cpu = {k: v.cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()

    prompt_ids = [101] + list(range(2000, 2010)) + [102]          # 12 BERT ids
    lo, hi = seg.shard_ranges(T, world)[rank]
    if lm is not None:
        # the reference's call (tdc/language_model/cambrian_qwen.py:415-438): one sample, <image> after the system prompt,
        # 64 text tokens; every rank is handed the whole video, as the reference's per-GPU eval workers are
        vs = synth_video(0, T, px_s, dev, tower_dtype)
        vd = synth_video(0, T, px_d, dev, tower_dtype, seed=4321) if px_d != px_s else vs
        ids = torch.arange(100, 165, device=dev)
        ids[14] = -200                                                  # IMAGE_TOKEN_INDEX (tdc/constants.py)
        ids = ids[None]
        images = [vs[None], vd[None]]

        def step():
            o = lm.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, images, image_sizes=[(384, 384)],
                                                        video_indices=[None], prompts=[prompt_ids], audios=[None])
            return o[4][0]
    elif world == 1:
        vs = synth_video(0, T, px_s, dev, tower_dtype)
        vd = synth_video(0, T, px_d, dev, tower_dtype, seed=4321) if px_d != px_s else vs

        def step():
            return enc.encode_video(vs, vd, (384, 384), budget_text_len=64, n_text_tokens=64, prompt_ids=prompt_ids,
                                    frame_cap=T, audio={"audio_wav": wav} if wav is not None else None)
    else:
        from tdc_video_amd import dist as tdist
        vs = synth_video(lo, hi, px_s, dev, tower_dtype)       # a rank only ever holds its own frames
        halo = 1 if (args.recompute_halo and lo < hi < T) else 0        # + the right neighbour's first frame, re-encoded here
        vd = synth_video(lo, hi + halo, px_d, dev, tower_dtype, seed=4321 if px_d != px_s else 1234)
        sharded = tdist.ShardedVideoEncoder(enc, rank, world)

        def step():
            return sharded.encode_video(vs, vd, T, (384, 384), n_text_tokens=64, prompt_ids=prompt_ids,
                                        audio={"audio_wav": wav} if wav is not None else None,
                                        recompute_halo=bool(args.recompute_halo))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    rank_ms = None
    ranks_seen = [0]
    if world > 1:
        # who took part, as seen over the process group itself (RCCL at N > 1): every rank's id and device index
        me = torch.tensor([rank, local], device=dev, dtype=torch.int64)
        seen = [torch.zeros_like(me) for _ in range(world)]
        torch.distributed.all_gather(seen, me)
        ranks_seen = sorted(int(x[0].item()) for x in seen)
        rank_devices = [int(x[1].item()) for x in sorted(seen, key=lambda x: int(x[0].item()))]
        # per-rank step times (a straggler shows as max >> min), then the MAX over the ranks as the job's time
        mine = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        rank_ms = [round(float(x.item()) / args.steps * 1e3, 2) for x in every]
        dt = max(float(x.item()) for x in every)
    ms_per_step = dt / args.steps * 1e3
    fps = T * args.steps / dt

    # ---- roofline of the dominant kernel (tdc_gemm MFMA kernel): one extra pass under the library's launch profiler
    # (tdc_profile_*: hipEvents recorded inside libtdc_hip.so around every GEMM / attention / LayerNorm / fused cross-attention
    # launch, on the stream the kernel is launched on) - the SAME host path as the timed steps: the C++ composites
    # tdc_vit_fwd / tdc_connector_fwd / tdc_qformer_fwd, not a per-kernel Python replay of them
    ops.profile_start()
    step()
    torch.cuda.synchronize()
    recs = ops.profile_stop()
    this_args = dict(frames=T, K=K, hidden=H, gpus=world, tower_batch=args.tower_batch, dtype=args.dtype, px=args.px,
                     audio=bool(args.audio), fp8_level=args.fp8_level if args.dtype == "fp8" else 0,
                     two_streams=int(two_streams), res=args.res)
    if args.ln_fuse:
        this_args["ln_fuse"] = True
    gemms = [r for r in recs if r["kind"] == "gemm"]
    shape_of = lambda r: (r["M"], r["N"], r["K"], r["act"], int(r["res"] != 0), r["out_f32"])      # noqa: E731
    if args.dump_gemm_shapes and rank == 0:
        with open(args.dump_gemm_shapes + ".args.json", "w") as fh:
            json.dump(this_args, fh)
        import collections
        cnt = collections.Counter((r["M"], r["N"], r["K"], r["act"], r["res"], r["out_f32"]) for r in gemms)
        with open(args.dump_gemm_shapes, "w") as fh:
            # res: 0 = none, 1 = fp32, 2 = 16-bit (the replay tool allocates the residual / output accordingly)
            for (M_, N_, K_, act_, res_, of_), c in sorted(cnt.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1]):
                fh.write("%d %d %d %d %d %d %d\n" % (M_, N_, K_, act_, res_, of_, c))
    if args.gemm_shape_times and rank == 0:
        import collections
        agg = collections.OrderedDict()
        for r in gemms:
            a = agg.setdefault((r["M"], r["N"], r["K"], r["act"], r["res"], r["out_f32"]), [0, 0.0, 0.0])
            a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]
        with open(args.gemm_shape_times, "w") as fh:
            fh.write("# M N K act res(0 none, 1 fp32, 2 16-bit) out_f32 | launches, total ms of the profiled step, TFLOP/s (real dims)\n")
            for shp, (c, ms_, w) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                fh.write("%7d %5d %5d %d %d %d | %4d %9.3f ms %8.1f TF/s\n" % (shp + (c, ms_, w / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0.0)))
    g_ms = sum(r["ms"] for r in gemms)
    g_fl = sum(r["flops"] for r in gemms)
    attns = [r for r in recs if r["kind"] == "attn"]
    a_ms = sum(r["ms"] for r in attns)
    a_fl = sum(r["flops"] for r in attns)
    ln_ms = sum(r["ms"] for r in recs if r["kind"] == "ln")
    achieved = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
    # the north_star's kernel scope: the Q-Former cross-attention BLOCK = K/V projections of the encoder tokens
    # + q projection + QK^T / softmax / PV + output projection + residual + LayerNorm, all six cross layers; tdc_qformer_fwd tags
    # every launch of it (TDC_PROF_TAG_XATTN_BLOCK); algorithmic FLOPs = the GEMM / attention counts
    xb = [r for r in recs if r["tag"] == 1]
    xb_ms = sum(r["ms"] for r in xb)
    xb_fl = sum(r["flops"] for r in xb)
    xb_kinds = {}
    for r in xb:
        xb_kinds[r["kind"]] = round(xb_kinds.get(r["kind"], 0.0) + r["ms"], 3)
    xb_tf = xb_fl / (xb_ms * 1e-3) / 1e12 if xb_ms > 0 else None
    # HBM-side traffic per launch and MFMA-pipe occupancy from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, gfx950 correction;
    # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)).  rocprofv3's counter collection cannot run inside this (torch)
    # process on this image, so rank 0 at N = 1 starts FRESH child processes after the timed region: one rocprofv3 --pmc pass per
    # counter over tools/bin/gemm_pmc, a torch-free program that replays THIS run's GEMM launch list through the same
    # libtdc_hip.so (pmc_leg).  When that leg is unavailable (no rocprofv3, budget spent, N > 1, fp8) the numbers fall back to
    # the committed summary of the same measurement - only when this run's arguments match it - and `traffic_source` says which.
    traffic = mfma_busy = kv_busy = traffic_source = None
    pmc_shapes = None
    if rank == 0 and world == 1 and args.pmc != "off" and args.dtype != "fp8":
        got = pmc_leg(gemms, args.pmc_budget, H, g_ms)
        if got is not None:
            traffic, mfma_busy, kv_busy, traffic_source, pmc_shapes = got
    pmc = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    if traffic is None and os.path.exists(pmc):
        summ = json.load(open(pmc))
        if summ.get("bench_args") == this_args:
            traffic = round(summ["per_launch_hbm_bytes"])
            mfma_busy = summ.get("mfma_busy_frac")
            kv_busy = max((r.get("mfma_busy_frac", 0) for r in summ["shapes"] if r["N"] >= 9216 and r["K"] == H), default=None)
            traffic_source = "replayed, not measured in this run: profiles/%s (tools/gemm_pmc.cpp on this launch list, " \
                             "collected at commit %s)" % (PMC_SUMMARY, summ.get("collected_at_commit", "?"))
            if summ.get("gemm_ms_per_step_at_collection"):   # staleness check: this run's GEMM time / the replayed run's
                traffic_source += "; live / replayed GEMM time per step = %.3f" % (g_ms / summ["gemm_ms_per_step_at_collection"])
    if args.pmc_out and pmc_shapes is not None and rank == 0:
        json.dump(pmc_shapes, open(args.pmc_out, "w"), indent=1)
    # fp8 runs: the dense f8f6f4 MFMA peak (5 PFLOP/s) when every tower GEMM runs on e4m3 operands (level 2); at level 1
    # a third of the GEMM FLOPs stay in bf16, the bf16 peak is kept as the (conservative) yardstick
    peak = 2.0 * MFMA_PEAK_TFLOPS if (args.dtype == "fp8" and args.fp8_level >= 2) else MFMA_PEAK_TFLOPS
    a_tf = a_fl / (a_ms * 1e-3) / 1e12 if a_ms > 0 else None
    # whole step: every algorithmic FLOP of the pass (GEMMs + attention; the rest is byte work) over the TIMED step time
    x_fl = sum(r["flops"] for r in recs if r["kind"] == "xattn")       # the fused cross-attention kernel's own GEMM / attention FLOPs
    step_tf = (g_fl + a_fl + x_fl) / (ms_per_step * 1e-3) / 1e12 * (world if world > 1 else 1)
    roofline = dict(bound="mfma", achieved=round(achieved, 1), peak=peak, unit="TFLOP/s",
                    frac=round(achieved / peak, 4), traffic=traffic, traffic_source=traffic_source,
                    kernel="gemm256p_kernel / gemm_kernel (tdc_gemm)",
                    launches=len(gemms), avg_launch_us=round(g_ms * 1e3 / max(1, len(gemms)), 2),
                    host_path="C++ composites (tdc_vit_fwd / tdc_connector_fwd / tdc_qformer_fwd): the path of the timed steps; "
                              "events recorded inside libtdc_hip.so (tdc_profile_*)",
                    layernorm_ms_per_step=round(ln_ms, 2),
                    gemm_ms_per_step=round(g_ms, 2), gemm_tflop_per_step=round(g_fl / 1e12, 2),
                    attention=dict(ms_per_step=round(a_ms, 2), tflops=round(a_tf, 1) if a_tf else None,
                                   frac=round(a_tf / MFMA_PEAK_TFLOPS, 4) if a_tf else None,
                                   launches=len(attns)),
                    whole_step=dict(tflops=round(step_tf, 1), frac=round(step_tf / (peak * world), 4),
                                    note="(GEMM + attention + fused cross-attention FLOPs of this rank x ranks) / timed step; peak x ranks"),
                    xattn_block=dict(ms_per_step=round(xb_ms, 3), tflop_per_step=round(xb_fl / 1e12, 3),
                                     tflops=round(xb_tf, 1) if xb_tf else None,
                                     frac=round(xb_tf / MFMA_PEAK_TFLOPS, 4) if xb_tf else None, launches=len(xb),
                                     ms_by_kernel=xb_kinds, xattn_mode=int(enc.xattn_mode),
                                     note="Q-Former cross-attention block of all 6 cross layers (SURVEY D7: K/V projections + q-proj + "
                                          "QK^T/softmax/PV + out-proj + residual + LayerNorm), live events; the north_star's >= 40 % target"),
                    mfma_busy_pmc=mfma_busy, xattn_kv_mfma_busy_pmc=kv_busy)
    if args.dtype != "fp8":
        roofline["power_limited_mfma"] = dict(
            tflops=MFMA_DENSE_SUSTAINED_TFLOPS, frac=round(achieved / MFMA_DENSE_SUSTAINED_TFLOPS, 4),
            note="replayed constant, not measured in this run: bf16 16x16x32 MFMAs on dense random register operands, no memory "
                 "traffic, every CU busy (tools/mfma_power.cpp, profiles/r02_mfma_power.log); all-zero operands reach 2448")
    if world > 1:
        # every rank is done with the process group: leave it in step (a rank that exits while a peer is still inside a
        # collective shows up as a watchdog abort in that peer's log)
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    res = {
        "metric": "frames/sec encoded+compressed (576->K tokens) at T=%d" % T,
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": ("fp8 (e4m3 operands: %s) + bf16, %s tower residual stream" % ("all four tower GEMMs" if args.fp8_level >= 2 else "qkv / fc1", args.res)) if args.dtype == "fp8"
                 else ("bf16 (ViT tower operands, %s residual stream) + fp16 (connector, Q-Former)" % args.res) if args.dtype == "mixed"
                 else "%s (%s tower residual stream)" % (args.dtype, args.res),
        "data": "synthetic",
        # fp8 is a throughput mode: its outputs are held to a measured contract against the bf16 path (tests/test_hip_configs.py:
        # test_config5_fp8_contract_vs_bf16), not to the oracle's tolerances
        "parity": "throughput mode (e4m3 operands: contract vs the bf16 path, DESIGN.md section 2)" if args.dtype == "fp8"
                  else "parity mode (oracle tolerances, DESIGN.md section 2)",
        "config": {"workload": "one %d-frame video, SigLIP-so400m/14@%d + DINOv2-giant/14@%d towers (729 patches -> 576 "
                               "tokens), SVA 576->144, mm_projector -> H=%d, Q-Former TDC K=%d (N=156, 12 prompt ids), "
                               "random-init weights, frame cap lifted to T, LLM stubbed%s"
                               % (T, px_s, px_d, H, K, ", + %d s of 16 kHz audio through BEATs (50 audio tokens / frame in "
                                  "the Q-Former KV)" % T if args.audio else ""),
                   "frames": T, "K": K, "hidden": H, "px": px_s, "tower_residual": args.res,
                   # the model-level settings (tdc-video_amd/model.py: CambrianMetaModel.tdc_engine) that build this encoder
                   "product_setting": product_setting,
                   "entry": "mixin: CambrianMetaForCausalLM.prepare_inputs_labels_for_multimodal on a stub LM (value counts its text "
                            "splice and padding too)" if lm is not None else "engine: VideoEncoder.encode_video",
                   "parallelism": "frames sharded over %d GPU(s)" % world, "two_streams": bool(two_streams),
                   "emitted_tokens": int(out.shape[0]) - (64 if lm is not None else 0),
                   "selection_refine": {"on": enc.selection_eps is not None, "eps": enc.selection_eps,
                                        "note": "a5 under bf16 DINOv2 operands: pairs whose similarities decide the selection and lie within "
                                                "2 eps of the decisive ranks are re-encoded by an fp16-operand DINOv2 tower (bands of more than "
                                                "max(8, frames / 8) frames - plateaus - are left alone); none at T = 512 on this video"}},
        "rank_ms_per_step": None if rank_ms is None else {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms},
        "ranks_seen": ranks_seen, "rank_devices": rank_devices if world > 1 else [local],
        "dist_backend": (os.environ.get("TDC_DIST_BACKEND", "nccl") + " (RCCL)" * (os.environ.get("TDC_DIST_BACKEND", "nccl") == "nccl")) if world > 1 else None,
        "roofline": roofline,
    }
    if sd_cpu is not None:
        res["cpu_baseline"] = cpu_baseline(sd_cpu, cfg, H, K, px_s, px_d)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
