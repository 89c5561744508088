"""VideoEncoder: the hot path (stages S0-S10) as a sequence of libtdc_hip.so launches.

Python only sequences kernels and owns the buffers (torch tensors as device memory); every FLOP runs in the HIP
library.  Stage methods mirror the reference's stages so the parity tests can compare stage by stage:
  tower()        a2-a4   tdc/multimodal_encoder/{siglip,dino}_encoder.py + HF ViTs
  frame_sims()   a5      tdc/cambrian_arch.py:832-842
  connector()    a6-a10  tdc/cambrian_arch.py:1002-1053 (aux projectors, SVA), :1149-1150 (mm_projector), :1176-1293
  compress()     a11-a19 tdc/cambrian_arch.py:1507-1709 (all chunks of a video batched into ONE Q-Former pass)
"""
import math

import torch

from . import lib as L
from . import ops
from . import segment as seg
from . import weights as Wt
from .weights import pad64


class _LRU(dict):
    """Small bounded cache for the host-built index / mask / prompt tables (keys: geometry or prompt-id tuples): a server that
    sees many prompts and resolutions does not grow host and device memory without bound; the least recently used entry
    leaves when the 33rd arrives."""

    def __init__(self, maxsize=32):
        super().__init__()
        self.maxsize = maxsize

    def _touch(self, key):
        v = dict.pop(self, key)
        dict.__setitem__(self, key, v)          # dicts keep insertion order: the last entry is the most recent
        return v

    def get(self, key, default=None):
        return self._touch(key) if key in self else default

    def __getitem__(self, key):
        return self._touch(key)

    def __setitem__(self, key, value):
        if key in self:
            dict.pop(self, key)
        dict.__setitem__(self, key, value)
        while len(self) > self.maxsize:
            dict.pop(self, next(iter(self)))


class VideoEncoder:
    def __init__(self, sd, cfg, dtype=torch.float16, device="cuda", siglip_heads=16, dino_heads=24,
                 qformer_heads=12, tower_batch=None, fp8_towers=False, tower_dtype=None, ln_fuse=False,
                 tower_res_dtype=None, dino_dtype=None, selection_refine=None, selection_eps=1e-3, selection_max_fraction=0.125):
        """sd: reference-named state dict without the leading 'model.'; cfg: dict of reference config keys.
        dtype: 16-bit type of the connector, the Q-Former and every tensor handed to the caller; tower_dtype (default:
        dtype): 16-bit type of the two ViT towers - their last kernel (the token-grid resample) writes `dtype` rows.  bf16
        towers under an fp16 connector / compressor keep the compressed tokens within 1e-3 of the fp32 reference arithmetic (measured
        1.0e-4 at full depth; all-bf16: 6e-4 ... 2e-3): the 16-bit error of the context tokens is made behind the towers.
        dino_dtype (default: tower_dtype): operand type of the DINOv2 tower alone.  Its features alone drive the a5 segment
        selection (tdc/cambrian_arch.py:832-849): fp16 there keeps the adjacent-frame similarities at the reference's own
        precision (3.4e-5 instead of 4.2e-4 with bf16 operands) while SigLIP stays in tower_dtype.
        selection_refine (None = automatic: on when the DINOv2 operands are bf16 and not e4m3) / selection_eps: the a5 segment
        selection at the reference's precision under bf16 DINOv2 operands.  bf16 similarities are within selection_eps of the fp16
        tower's (measured 1.7e-4 ... 5.0e-4; default bound 1e-3); when the ranks that decide the selection are closer than
        4 selection_eps, the pairs inside the band [v_n+1 - 2 eps, v_n + 2 eps] - and only they - are re-encoded by a second, fp16-operand copy of the DINOv2
        tower (+2.2 GB of weights) and re-ranked (segment.selection_band / select_refined): exactly what ranking the fp16 tower's
        similarities selects.  A video whose decisive ranks are further apart (the bench's: 0.43) pays nothing; a band whose
        frames exceed max(8, selection_max_fraction x frames) - a plateau of near-identical similarities at the decisive rank, where
        the reference's own fp16 similarity values tie - is left to the fast tower's ranking (segment.band_allowed).
        fp8_towers (BASELINE config 5): the towers' LayerNorms emit e4m3 rows with per-row scales and the qkv / fc1 GEMMs
        run on fp8 operands (v_mfma_f32_16x16x128_f8f6f4); everything else stays in `tower_dtype`.
        ln_fuse: the towers' pre-LayerNorms folded into the neighbouring GEMMs (weights.ln_fusion_enabled; off by default).
        tower_batch: frames per tower launch sequence; None (default) = all frames of the call up to TOWER_BATCH_MAX, halved
        until the tower workspace fits the free HBM (`auto_tower_batch`) - the result does not depend on it, bit for bit
        (tests: test_batch_invariance_and_determinism), the rate does (512 / 256 / 128 / 64 frames: 428.8 / 425.7 / 420.1 /
        407.6 frames/s, profiles/r04_tower_batch.log).  The reference's own chunk is 64 (tdc/cambrian_arch.py:698-745).
        tower_res_dtype: type of the towers' residual stream in HBM - None / torch.float32: fp32 (the out-projection / fc2
        epilogues read-modify-write 8 B per element); torch.float16 (or bfloat16): that 16-bit type - 4 B per element, half
        the LayerNorm input bytes, sums formed in fp32 and rounded once per residual add.  fp16 is the reference's own
        arithmetic (its HF towers run under torch_dtype=float16, tdc/builder.py:69).  Composes with fp8_towers (round 6) and, when the
        operands have the stream's type, with ln_fuse."""
        self.cfg = dict(cfg)
        self.tower_res_dtype = None if tower_res_dtype in (None, torch.float32) else tower_res_dtype
        assert self.tower_res_dtype in (None, torch.float16, torch.bfloat16)
        # the LayerNorm fold over a 16-bit stream: the consumer GEMMs read the stream itself as their A operand, so the
        # operands of both towers have the stream's type
        assert not (ln_fuse and self.tower_res_dtype is not None) or \
            (tower_dtype or dtype) == self.tower_res_dtype == (dino_dtype or tower_dtype or dtype), \
            "ln_fuse over a 16-bit residual stream needs tower operands of the stream's type"
        self.dtype, self.dev = dtype, torch.device(device)
        self._tower_dtype = tower_dtype = dtype if tower_dtype is None else tower_dtype
        self.tower_batch = tower_batch
        self.qheads = qformer_heads
        s_sd = Wt._strip(sd, "vision_tower_aux_list.0.vision_tower.")
        d_sd = Wt._strip(sd, "vision_tower_aux_list.1.vision_tower.")
        self.towers = {}
        if s_sd:
            self.towers["siglip"] = Wt.prep_siglip(s_sd, siglip_heads, tower_dtype, self.dev, fp8=fp8_towers, ln_fuse=ln_fuse)
        if d_sd:
            self.towers["dino"] = Wt.prep_dino(d_sd, dino_heads, dino_dtype or tower_dtype, self.dev, fp8=fp8_towers,
                                               ln_fuse=ln_fuse)
        for t in self.towers.values():
            t["dtype"] = dino_dtype or tower_dtype if t.kind == "dino" else tower_dtype
        # a5 at the reference's precision under bf16 DINOv2 operands: a second, fp16-operand DINOv2 tower for the pairs whose
        # similarities decide the selection (encode_video_with / dist.ShardedVideoEncoder: selection_band -> refine_pairs)
        if selection_refine is None:
            selection_refine = bool(d_sd) and (dino_dtype or tower_dtype) == torch.bfloat16 and not fp8_towers
        self.selection_eps = None
        if selection_refine and d_sd and (dino_dtype or tower_dtype) != torch.float16:
            tp = Wt.prep_dino(d_sd, dino_heads, torch.float16, self.dev)
            tp["dtype"] = torch.float16
            # its residual stream: fp16 (the reference's arithmetic) when the engine's is 16-bit, else the engine's fp32
            tp["res_dtype"] = torch.float16 if self.tower_res_dtype is not None else None
            self.towers["dino_precise"] = tp
            self.selection_eps = float(selection_eps)
        self.selection_max_fraction = float(selection_max_fraction)
        self.c = Wt.prep_connector(sd, cfg, dtype, self.dev)
        tok = cfg.get("mm_vision_tower_aux_token_len_list", [576, 576])
        self.out_grid = [int(round(t ** 0.5)) for t in tok]
        self.side = int(round(cfg.get("query_num_list", [144])[0] ** 0.5))
        self._tables = _LRU(32)
        self.K = cfg.get("context_token_num", 16)
        self.H = cfg["hidden_size"]
        self.two_streams = False
        self.native_qformer = True    # Q-Former through the C++ composite tdc_qformer_fwd
        # cross-attention block of the Q-Former: 2 = one kernel per layer (tdc_qformer_xattn), 1 = q GEMM + tdc_attention,
        # then output projection + residual + LayerNorm in one kernel, 0 = per-kernel sequence; applies when the shape allows it
        # (bert-base width, K % 16 == 0; mode 2 also Nenc % 4 == 0, Nenc <= 224)
        self.xattn_mode = 1
        self.native_towers = True     # towers through the C++ composite tdc_vit_fwd (per-kernel Python path when False)
        self.beats = None             # beats.BeatsEncoder for raw-waveform audio input

    @property
    def tower_dtype(self):
        return self.__dict__.get("_tower_dtype") or self.dtype

    def _tdt(self, t):
        """operand type of tower `t` (its weights' type): tower_dtype unless dino_dtype set the DINOv2 tower apart"""
        return t.get("dtype") or self.tower_dtype

    # ------------------------------------------------------------------------------------------------ towers
    def _bil(self, n_in, n_out):
        key = (n_in, n_out)
        if key not in self._tables:
            self._tables[key] = ops.bilinear_tables(n_in, n_out, self.dev)
        return self._tables[key]

    def tower(self, name, px):
        """px [B,3,H,W] (fp32, fp16 or bf16) -> features [B*g*g, pad64(D)] 16-bit `dtype` (g = 24)."""
        t = self.towers[name]
        out_grid = self.out_grid[0 if name == "siglip" else 1]      # "dino" / "dino_precise": the second entry
        outs = []
        tb = int(self.tower_batch) if self.tower_batch else self.auto_tower_batch(t, px)
        for s in range(0, px.shape[0], tb):
            outs.append(self._tower_batch(t, px[s:s + tb].contiguous(), out_grid))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    TOWER_BATCH_MAX = 512      # the largest batch measured (profiles/r04_tower_batch.log); one 512-frame batch needs ~15 GB

    def auto_tower_batch(self, t, px, free_bytes=None):
        """Frames per tower batch when the caller gave none: min(frames, TOWER_BATCH_MAX), halved until tdc_vit_workspace_bytes
        fits 60 % of the HBM this process can still get (free on the device + what torch's allocator holds unused + the
        workspace it would replace).  free_bytes: override of that figure (tests)."""
        import ctypes as C
        B = max(1, min(int(px.shape[0]), self.TOWER_BATCH_MAX))
        if len(t.layers) == 0 or not px.is_cuda:
            return B
        Hpx, Wpx = int(px.shape[2]), int(px.shape[3])
        m = self._vit_struct(t, Hpx // t.patch, Wpx // t.patch)[0]
        lib = L.load()
        if free_bytes is None:
            free, _total = torch.cuda.mem_get_info(self.dev)
            unused = torch.cuda.memory_reserved(self.dev) - torch.cuda.memory_allocated(self.dev)
            wkey = "_vit_ws_" + t.kind if getattr(self, "two_streams", False) else "_vit_ws"
            ws = getattr(self, wkey, None)
            free_bytes = free + max(0, unused) + (ws.numel() if ws is not None else 0)
        budget = 0.6 * free_bytes
        while B > 1 and lib.tdc_vit_workspace_bytes(C.byref(m), B, Hpx, Wpx) > budget:
            B = (B + 1) // 2
        return B

    # ---- native composite (csrc/api.cpp: tdc_vit_fwd): the whole tower batch is one C call ------------------------------
    def _vit_struct(self, t, gh, gw):
        """ctypes mirror of tdc_vit_model for tower `t` on a gh x gw patch grid (cached; keeps the arrays alive)."""
        import ctypes as C
        cache = t.setdefault("_c_structs", {})
        if (gh, gw) in cache:
            return cache[(gh, gw)]
        pos, cls_row = Wt.tower_pos(t, gh, gw, self.dev)

        def lin(l):
            return L.Lin(l.w.data_ptr(), l.b.data_ptr() if l.b is not None else None, l.w.shape[0], l.w.shape[1])
        layers = (L.VitLayer * len(t.layers))()
        zeros = None
        if t.get("fp8"):    # one zero ln_c1 vector long enough for every lin of the tower (tdc_vit_layer.zeros)
            zn = max(l.w.shape[0] for Lr in t.layers for l in (Lr.qkv, Lr.out, Lr.fc1, Lr.fc2))
            zeros = t.setdefault("_zeros", torch.zeros(zn, dtype=torch.float32, device=self.dev))
        for i, Lr in enumerate(t.layers):
            layers[i] = L.VitLayer(Lr.ln1_g.data_ptr(), Lr.ln1_b.data_ptr(), Lr.ln2_g.data_ptr(), Lr.ln2_b.data_ptr(),
                                   lin(Lr.qkv), lin(Lr.out), lin(Lr.fc1), lin(Lr.fc2),
                                   Lr.qkv_c1.data_ptr() if Lr.qkv_c1 is not None else None,
                                   Lr.fc1_c1.data_ptr() if Lr.fc1_c1 is not None else None,
                                   Lr.qkv.wscale or 0.0, Lr.fc1.wscale or 0.0,
                                   zeros.data_ptr() if zeros is not None else None,
                                   Lr.out.wscale or 0.0, Lr.fc2.wscale or 0.0, Lr.fc1.w2max, Lr.fc1.bmax)
        m = L.VitModel()
        m.dtype, m.out_dtype_p1 = ops._dtcode(self._tdt(t)), ops._dtcode(self.dtype) + 1
        rd = t["res_dtype"] if "res_dtype" in t else getattr(self, "tower_res_dtype", None)
        m.res_dtype_p1 = 0 if rd is None else ops._dtcode(rd) + 1
        m.dim, m.heads, m.head_dim, m.n_layers, m.patch, m.has_cls = t.dim, t.heads, t.head_dim, len(t.layers), t.patch, \
            t.has_cls
        m.act = {"gelu_tanh": L.ACT_GELU_TANH, "gelu_erf": L.ACT_GELU_ERF, "swiglu": L.ACT_SWIGLU}[t.act]
        m.eps = t.eps
        m.patch_lin = lin(t.patch_lin)
        m.pos, m.ldpos = pos.data_ptr(), pos.stride(0)
        m.cls_row = cls_row.data_ptr() if cls_row is not None else None
        fl = t.get("final_ln")
        m.lnf_g, m.lnf_b = (fl[0].data_ptr(), fl[1].data_ptr()) if fl else (None, None)
        m.layers_host = layers
        m.fused = int(bool(t.fused))
        m.fp8 = int(t.get("fp8") or 0)
        cache[(gh, gw)] = (m, layers, pos, cls_row)
        return cache[(gh, gw)]

    def _tower_batch_native(self, t, px, out_grid):
        import ctypes as C
        B, _, H, W = px.shape
        assert H == W and px.is_contiguous() and len(t.layers) > 0   # trailing pixels (384 = 27*14 + 6) are dropped: "valid" conv
        g = H // t.patch
        m = self._vit_struct(t, g, g)[0]
        lib = L.load()
        need = lib.tdc_vit_workspace_bytes(C.byref(m), B, H, W)
        # one workspace per tower when the towers run concurrently on two streams, otherwise one shared buffer
        wkey = "_vit_ws_" + t.kind if getattr(self, "two_streams", False) else "_vit_ws"
        ws = getattr(self, wkey, None)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
            setattr(self, wkey, ws)
        D = t.dim
        out = torch.empty(B * out_grid * out_grid, pad64(D), device=self.dev, dtype=self.dtype)
        i0, i1, fr = self._bil(g, out_grid)
        f32 = ops.px_kind(px, self._tdt(t))
        L.check(lib.tdc_vit_fwd(C.byref(m), ops._ptr(px), f32, B, H, W, out_grid, ops._ptr(i0), ops._ptr(i1),
                                ops._ptr(fr), ops._ptr(out), out.stride(0), ops._ptr(ws), ws.numel(), ops._stream()),
                "tdc_vit_fwd")
        return out

    def _tower_batch(self, t, px, out_grid):
        """One tower batch.  Default: the C++ composite (tdc_vit_fwd).  With `native_towers = False`: the SAME launches issued one
        by one from Python - the form the tests compare the composite with, bit for bit - over either residual stream
        (`tower_res_dtype`: None = fp32, read-modify-written as 8 B per element; a 16-bit type = tdc_vit_model.res_dtype_p1)."""
        if getattr(self, "native_towers", True) and len(t.layers) > 0:
            return self._tower_batch_native(t, px, out_grid)
        dt, dev = self._tdt(t), self.dev
        rd = t["res_dtype"] if "res_dtype" in t else getattr(self, "tower_res_dtype", None)
        s32 = rd is None                         # fp32 residual stream
        B = px.shape[0]
        D, Dp = t.dim, pad64(t.dim)
        patches, gh, gw = ops.im2col(px, t.patch, dt)
        assert gh == gw, "square inputs only (reference pads to square, mm_datautils.py:286-314)"
        P = gh * gw
        S = P + t.has_cls
        pos, cls_row = Wt.tower_pos(t, gh, gw, dev)
        x = torch.empty(B * S, Dp, device=dev, dtype=torch.float32 if s32 else rd)        # the residual stream
        ops.gemm(patches, t.patch_lin.w, t.patch_lin.b, res=pos, r_map=(P, 0, t.has_cls, 1), out=x, out_f32=s32,
                 c_map=(P, S, t.has_cls, 1))
        if t.has_cls:
            (ops.set_rows if s32 else ops.set_rows16)(x, B, S, 0, cls_row)

        def ln(g, b, **kw):                      # LayerNorm of the stream -> 16-bit operand rows (or e4m3 rows + scales)
            return ops.layernorm(x, g, b, t.eps, D, dt, x16_kernel=not s32, **kw)

        def update(a, lin, **kw):                # x <- x + a lin^T + b, in place (fp32: 8 B per element; 16-bit: one rounding)
            if not s32 and "out_dtype" in kw:    # e4m3 operands over the 16-bit stream: C / res are of the stream's type
                kw["out_dtype"] = rd
            return ops.gemm(a, lin.w, lin.b, res=x, out=x, out_f32=s32, **kw)
        h16 = torch.empty(B * S, Dp, device=dev, dtype=dt)
        qkv = torch.empty(B * S, t.layers[0].qkv.w.shape[0], device=dev, dtype=dt) if t.layers else None
        attn = torch.zeros(B * S, Dp, device=dev, dtype=dt)
        act = {"gelu_tanh": L.ACT_GELU_TANH, "gelu_erf": L.ACT_GELU_ERF, "swiglu": L.ACT_SWIGLU}[t.act]
        mlp_w = t.layers[0].fc2.w.shape[1] if t.layers else 0
        mlp = torch.empty(B * S, mlp_w, device=dev, dtype=dt) if t.layers else None
        scale = t.head_dim ** -0.5
        # t.fused: the block's LayerNorms are folded into the GEMMs around them - the residual-stream GEMM emits the
        # 16-bit row copy (into h16) and per-slot statistics, the next GEMM folds (mean, rstd) into its epilogue
        fused = bool(t.fused)
        fp8 = int(t.get("fp8") or 0)
        assert s32 or not fused or rd == dt, "the fold over a 16-bit stream reads the stream as the GEMM operand"
        slots = D // 64
        part = torch.empty(slots, B * S, 2, device=dev, dtype=torch.float32) if fused else None
        stats = torch.empty(B * S, 2, device=dev, dtype=torch.float32) if fused else None
        # fp32 stream: the producer also writes the 16-bit row copy the consumer reads; 16-bit stream: the consumer reads the
        # stream itself - only the partials are emitted and no LayerNorm kernel runs inside the layer loop
        emit = dict(x16=h16, ln_part=part) if (fused and s32) else dict(ln_part=part) if fused else {}
        xa = h16 if s32 else x                   # A operand of the folded consumers
        if fp8:     # e4m3 LayerNorm rows (per-row scales in `stats`) for the fp8-operand qkv / fc1 GEMMs
            h8 = torch.empty(B * S, t.layers[0].qkv.w.shape[1], device=dev, dtype=torch.uint8)
            stats = torch.empty(B * S, 2, device=dev, dtype=torch.float32)
        if fp8 >= 2:    # ... and e4m3 copies of the attention output / MLP hidden for out-proj / fc2
            a8 = torch.empty(B * S, t.layers[0].out.w.shape[1], device=dev, dtype=torch.uint8)
            m8 = torch.empty(B * S, t.layers[0].fc2.w.shape[1], device=dev, dtype=torch.uint8)
            mlp_n = t.layers[0].fc1.w.shape[0] // (2 if t.act == "swiglu" else 1)
            mlp_w = max(mlp_w, m8.shape[1])
            mlp = torch.empty(B * S, mlp_w, device=dev, dtype=dt)
        if fp8 >= 3:    # ... the hidden written as e4m3 by fc1 itself, with its row scales in stats2
            stats2 = torch.empty(B * S, 2, device=dev, dtype=torch.float32)
        for li, Lr in enumerate(t.layers):
            if fp8:
                ln(Lr.ln1_g, Lr.ln1_b, y8=h8, y8_stats=stats, y8_wscale=Lr.qkv.wscale)
                ops.gemm(h8, Lr.qkv.w, Lr.qkv.b, out=qkv, ln_stats=stats, ln_c1=Lr.qkv.zeros, out_dtype=dt)
            elif Lr.qkv_c1 is None:
                ln(Lr.ln1_g, Lr.ln1_b, y16=h16)
                ops.gemm(h16, Lr.qkv.w, Lr.qkv.b, out=qkv)
            else:
                ops.gemm(xa, Lr.qkv.w, Lr.qkv.b, out=qkv, ln_stats=stats, ln_c1=Lr.qkv_c1)
            ld = qkv.stride(0)
            ops.attention(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], attn, B, t.heads, t.head_dim, S, S, scale,
                          S * ld, S * ld, S * ld, S * attn.stride(0))
            if fp8 >= 2:
                ops.quantize_rows_fp8(attn, Dp, Lr.out.wscale, y8=a8, stats=stats)
                update(a8, Lr.out, ln_stats=stats, ln_c1=Lr.out.zeros, out_dtype=dt)
            else:
                update(attn, Lr.out, **emit)
            if fused:
                ops.ln_finalize(part, slots, B * S, t.eps, stats)
                ops.gemm(xa, Lr.fc1.w, Lr.fc1.b, act=act, out=mlp, ln_stats=stats, ln_c1=Lr.fc1_c1)
            elif fp8:
                ln(Lr.ln2_g, Lr.ln2_b, y8=h8, y8_stats=stats, y8_wscale=Lr.fc1.wscale)
                if fp8 >= 3:
                    ops.gemm(h8, Lr.fc1.w, Lr.fc1.b, act=act, out=m8, ln_stats=stats, ln_c1=Lr.fc1.zeros, out_dtype=dt,
                             out_stats=stats2, out_w2max=Lr.fc1.w2max, out_bmax=Lr.fc1.bmax, out_wscale=Lr.fc2.wscale)
                else:
                    ops.gemm(h8, Lr.fc1.w, Lr.fc1.b, act=act, out=mlp, ln_stats=stats, ln_c1=Lr.fc1.zeros, out_dtype=dt)
            else:
                ln(Lr.ln2_g, Lr.ln2_b, y16=h16)
                ops.gemm(h16, Lr.fc1.w, Lr.fc1.b, act=act, out=mlp)
            if fp8 >= 3:
                update(m8, Lr.fc2, ln_stats=stats2, ln_c1=Lr.fc2.zeros, out_dtype=dt)
            elif fp8 == 2:
                ops.quantize_rows_fp8(mlp, mlp_n, Lr.fc2.wscale, y8=m8, stats=stats)
                update(m8, Lr.fc2, ln_stats=stats, ln_c1=Lr.fc2.zeros, out_dtype=dt)
            elif fused and li + 1 < len(t.layers):
                update(mlp, Lr.fc2, **emit)
                ops.ln_finalize(part, slots, B * S, t.eps, stats)
            else:
                update(mlp, Lr.fc2)
        src, sdt = x, (dt if s32 else rd)
        if t.get("final_ln"):
            ln(t.final_ln[0], t.final_ln[1], y16=h16)
            src, sdt = h16, dt
        return ops.resample_tokens(src, B, t.has_cls, gh, out_grid, D, sdt, self._bil(gh, out_grid), out_dtype=self.dtype)

    # ------------------------------------------------------------------------------------------------ a5
    def sims_tensor(self, dino_feat, T):
        """adjacent-frame cosine similarity of the DINO features -> fp32 device tensor [T-1]."""
        n = dino_feat.shape[0] // T * dino_feat.shape[1]
        return ops.frame_cossim(dino_feat, T, n)

    # The segmentation is host logic on T - 1 similarities.  Reading them with .tolist() on the compute stream would also wait
    # for whatever was enqueued behind them (the SigLIP tower: a quarter of the step) and leave the device idle while the host
    # plans and enqueues the connector.  mark() / after(): an event behind the similarity kernel and a side stream that waits
    # for just that event - the host gets the numbers while the tower behind them is still running.
    def mark(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        return ev

    def tower_stream(self):
        """the ONE side stream the SigLIP tower runs on under `two_streams` (kept for the engine's life: its per-tower workspace
        `_vit_ws_siglip` and the cached tables are allocated under it once and never migrate between pool streams)"""
        st = self.__dict__.get("_tower_stream")
        if st is None:
            st = self._tower_stream = torch.cuda.Stream(device=self.dev)
        return st

    def after(self, ev):
        side = self.__dict__.get("_side_stream")
        if side is None:
            side = self._side_stream = torch.cuda.Stream(device=self.dev)
        side.wait_event(ev)
        return torch.cuda.stream(side)

    def fetch(self, t, ev):
        """device tensor t (ready at event ev) -> python list, without waiting for later work on the compute stream"""
        with self.after(ev):
            t.record_stream(torch.cuda.current_stream(self.dev))
            return t.cpu().tolist()

    def frame_sims(self, dino_feat, T):
        """same as a python list (one D2H copy: the segmentation is host logic)."""
        return self.sims_tensor(dino_feat, T).tolist()

    def precise_dino(self, px):
        """px [n, 3, H, W] -> DINOv2 features [n * P, pad64(D)] from the fp16-operand copy of the tower (selection refinement)"""
        return self.tower("dino_precise", px)

    def pair_sims(self, feats, pairs):
        """feats: {frame: [P, Dp] feature rows}; pairs: [(a, b)] -> device fp32 tensor of cos-sim(feats[a], feats[b]) (the a5
        kernel on consecutive (a, b) blocks, every second value)"""
        rows = torch.cat([feats[f] for ab in pairs for f in ab], 0)
        return self.sims_tensor(rows, 2 * len(pairs))[0::2].contiguous()

    # ------------------------------------------------------------------------------------------------ a6-a10
    def aux_project(self, feat, i):
        a = self.c.aux[i]
        h = ops.gemm(feat, a.fc1.w, a.fc1.b, act=L.ACT_GELU_ERF)
        y = ops.gemm(h, a.fc2.w, a.fc2.b, out_f32=True)
        y16, _ = ops.layernorm(y, a.ln_g, a.ln_b, 1e-5, self.c.C, self.dtype)
        return y16

    def sva(self, aux, T, image_sizes):
        """aux: two [T*P, pad64(C)] 16-bit tensors -> queries [T*nq, pad64(C)] (vision_sampler_0, 3 layers)."""
        c, dt, dev = self.c, self.dtype, self.dev
        C, Cp = c.C, pad64(c.C)
        side = self.side
        nq = side * side
        P = aux[0].shape[0] // T
        n = int(round(P ** 0.5))
        r = n // side
        mask, _ = self._window_mask(T, P, image_sizes)      # host geometry, cached per (T, image size)
        ctx = ops.token_mean(aux[0], T, P)                                    # [T, Cp]
        q16 = torch.zeros(T * nq, Cp, device=dev, dtype=dt)
        q16[:, :C] = c.vision_query.to(dt).to(dev)[None, :]
        for Lr in c.sva:
            cproj = ops.gemm(ctx, Lr.proj_context.w)                          # [T, Cp]
            cin = ops.gemm(cproj, Lr.proj_in_c.w, out_f32=True)               # [T, Cp] fp32: per-frame bias
            qin = ops.gemm(q16, Lr.proj_in_q.w, res=cin, r_map=(nq, 1, 0, 0), out_f32=True)   # proj_in(cat[q, ctx])
            kvs = []
            for tw in range(2):
                xn, _ = ops.layernorm(aux[tw], c.ones_C, c.zeros_C, 1e-5, C, dt, add=Lr.pos[tw], add_period=P,
                                      add_mode=1)
                kvs.append(ops.gemm(xn, Lr.kv[tw].w, Lr.kv[tw].b))            # [T*P, 2C]: K | V
            qn, _ = ops.layernorm(qin, Lr.q_ln[0], Lr.q_ln[1], 1e-5, C, dt)
            qs = ops.gemm(qn, Lr.q_proj.w)
            att = ops.sva_attention(qs, kvs, mask, T, side, r, C, 16)
            q2 = ops.gemm(att, Lr.o_proj.w, res=qin, out_f32=True)            # queries + attention_output
            qn2, _ = ops.layernorm(q2, Lr.norm[0], Lr.norm[1], 1e-5, C, dt)
            h = ops.gemm(qn2, Lr.out1.w, act=L.ACT_GELU_ERF)
            q16 = ops.gemm(h, Lr.out2.w, res=q16)                             # + residual (layer input)
        return q16

    def mm_project(self, q16):
        h = ops.gemm(q16, self.c.mm1.w, self.c.mm1.b, act=L.ACT_GELU_ERF)
        return ops.gemm(h, self.c.mm2.w, self.c.mm2.b)

    def unpad_newline(self, feat, T, image_sizes):
        """feat [T*nq, Hp] -> X [T*N, Hp] with the newline column appended (all frames of a video share image_size).
        The gather map depends only on (T, image sizes): built once on the host, kept on the device."""
        side = self.side
        key = ("unpad", T, tuple(image_sizes[0]) if len(set(map(tuple, image_sizes))) == 1 else tuple(map(tuple, image_sizes)))
        hit = self._tables.get(key)
        if hit is None:
            import numpy as np
            cache, blocks, sizes = {}, [], []
            for t in range(T):
                k = tuple(image_sizes[t])
                if k not in cache:
                    m, sz = seg.unpad_newline_map(side, k, 0)
                    cache[k] = (np.asarray(m, dtype=np.int32).reshape(-1, 2), sz)
                m, sz = cache[k]
                blk = m.copy()
                blk[:, 1] += np.where(m[:, 0] == 0, t * side * side, 0).astype(np.int32)
                blocks.append(blk)
                sizes.append(sz)
            idx_np = np.concatenate(blocks, 0)
            ops.check_pairs_host(idx_np, [T * side * side, 1])
            idx = torch.from_numpy(idx_np).to(self.dev).contiguous()
            hit = self._tables[key] = (idx, sizes)
        idx, sizes = hit
        Hp = feat.shape[1]
        assert feat.shape[0] >= T * side * side
        X = ops.gather_rows([feat, self.c.image_newline], idx, idx.shape[0], Hp, validated=True)
        return X, list(sizes)

    # ---- native composite (csrc/api.cpp: tdc_connector_fwd): a6-a9 in one C call ------------------------------------------
    def _connector_struct(self):
        """ctypes mirror of tdc_connector_model (cached; keeps the arrays alive)."""
        if getattr(self, "_conn_struct", None) is not None:
            return self._conn_struct
        c = self.c

        def lin(l):
            return L.Lin(l.w.data_ptr(), l.b.data_ptr() if l.b is not None else None, l.w.shape[0], l.w.shape[1])
        layers = (L.SvaLayer * len(c.sva))()
        for i, Lr in enumerate(c.sva):
            x = L.SvaLayer()
            x.proj_context, x.proj_in_c, x.proj_in_q = lin(Lr.proj_context), lin(Lr.proj_in_c), lin(Lr.proj_in_q)
            x.pos[0], x.pos[1], x.ldpos = Lr.pos[0].data_ptr(), Lr.pos[1].data_ptr(), Lr.pos[0].stride(0)
            x.kv[0], x.kv[1] = lin(Lr.kv[0]), lin(Lr.kv[1])
            x.q_ln_g, x.q_ln_b = Lr.q_ln[0].data_ptr(), Lr.q_ln[1].data_ptr()
            x.q_proj, x.o_proj = lin(Lr.q_proj), lin(Lr.o_proj)
            x.norm_g, x.norm_b = Lr.norm[0].data_ptr(), Lr.norm[1].data_ptr()
            x.out1, x.out2 = lin(Lr.out1), lin(Lr.out2)
            layers[i] = x
        vq = torch.zeros(pad64(c.C), device=self.dev, dtype=self.dtype)
        vq[: c.C] = c.vision_query.to(self.dtype).to(self.dev)
        m = L.ConnectorModel()
        m.dtype, m.C, m.side, m.heads, m.n_layers = ops._dtcode(self.dtype), c.C, self.side, 16, len(c.sva)
        for i in range(2):
            a = c.aux[i]
            m.aux[i] = L.AuxProj(lin(a.fc1), lin(a.fc2), a.ln_g.data_ptr(), a.ln_b.data_ptr())
        m.vision_query, m.ones_C, m.zeros_C = vq.data_ptr(), c.ones_C.data_ptr(), c.zeros_C.data_ptr()
        m.layers_host = layers
        m.mm1, m.mm2 = lin(c.mm1), lin(c.mm2)
        self._conn_struct = (m, layers, vq)
        return self._conn_struct

    def _window_mask(self, T, P, image_sizes):
        side = self.side
        r = int(round(P ** 0.5)) // side
        key = ("mask", T, P, tuple(image_sizes[0]) if len(set(map(tuple, image_sizes))) == 1 else tuple(map(tuple, image_sizes)))
        hit = self._tables.get(key)
        if hit is None:
            import numpy as np
            cache, rows = {}, []
            for t in range(T):
                k = tuple(image_sizes[t])
                if k not in cache:
                    m0 = seg.window_mask_bytes(side, r, k)
                    # both towers share the geometry (same grid)
                    cache[k] = np.asarray([a + b for a, b in zip(m0, m0)], dtype=np.uint8)
                rows.append(cache[k])
            hit = self._tables[key] = torch.from_numpy(np.concatenate(rows, 0)).to(self.dev).contiguous()
        return hit, r

    def _connector_native(self, sig_feat, dino_feat, T, image_sizes):
        import ctypes as C
        m = self._connector_struct()[0]
        P = sig_feat.shape[0] // T
        mask, r = self._window_mask(T, P, image_sizes)
        m.r = r
        assert dino_feat.shape[0] == T * P and P == (self.side * r) ** 2
        assert sig_feat.shape[1] >= m.aux[0].fc1.k and dino_feat.shape[1] >= m.aux[1].fc1.k
        lib = L.load()
        need = lib.tdc_connector_workspace_bytes(C.byref(m), T)
        ws = getattr(self, "_conn_ws", None)
        if ws is None or ws.numel() < need:
            ws = self._conn_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        out = torch.empty(T * self.side * self.side, pad64(self.c.H), device=self.dev, dtype=self.dtype)
        L.check(lib.tdc_connector_fwd(C.byref(m), ops._ptr(sig_feat), sig_feat.stride(0), ops._ptr(dino_feat),
                                      dino_feat.stride(0), T, ops._ptr(mask), ops._ptr(out), out.stride(0), ops._ptr(ws),
                                      ws.numel(), ops._stream()), "tdc_connector_fwd")
        return out

    def connector(self, sig_feat, dino_feat, T, image_sizes, keep=None):
        if keep is None and getattr(self, "native_connector", True):
            feat = self._connector_native(sig_feat, dino_feat, T, image_sizes)
            return self.unpad_newline(feat, T, image_sizes)
        aux = [self.aux_project(sig_feat, 0), self.aux_project(dino_feat, 1)]
        q = self.sva(aux, T, image_sizes)
        feat = self.mm_project(q)
        X, sizes = self.unpad_newline(feat, T, image_sizes)
        if keep is not None:
            keep.update(aux0=aux[0], aux1=aux[1], sva=q, mm_proj=feat)
        return X, sizes

    # ------------------------------------------------------------------------------------------------ a11-a19
    def qformer(self, enc, F, Nenc, query, qsrc, prompt_ids):
        """Batched Q-Former (tdc/Qformer.py:804-965) over F compressed frames.
        enc [F*Nenc, Hp] 16-bit; query [nC*K, Dq_pad] (query_proj output per chunk); qsrc int32 [F] chunk of frame.
        Returns last hidden state rows of the K query tokens as the combined buffer h16 [F*S, Dq_pad] and S."""
        qf, dt = self.c.qformer, self.dtype
        Dq = qf.dim
        K = self.K
        heads = self.qheads
        hd = Dq // heads
        ids = self._prompt_tensor(prompt_ids)
        Lt = 0 if ids is None else ids.numel()
        S = K + Lt
        h32, h16 = ops.qformer_embed(query, qsrc, qf.word, qf.pos, ids, qf.emb_ln[0], qf.emb_ln[1], 1e-12, F, K, Dq, dt)
        # the cross-attention block: fused form = key GEMM + transposed value GEMM + one kernel per cross layer
        mode = int(getattr(self, "xattn_mode", 1)) if qf.cross_k is not None else 0
        fused = mode >= 2 and ops.qformer_xattn_supported(Dq, heads, K, Nenc)
        out_fused = not fused and mode >= 1 and ops.qformer_xattn_supported(Dq, heads, K, 8)
        if fused or out_fused:
            self._tile_cross_weights()
        ops.profile_tag(L.PROF_TAG_XATTN_BLOCK)
        if fused:
            k_all = ops.gemm(enc, qf.cross_k.w, qf.cross_k.b, M=F * Nenc)     # [F*Nenc, n_cross*Dq]
            vt_all = torch.empty(qf.cross_v.w.shape[0], pad64(F * Nenc + 4), device=self.dev, dtype=dt)   # ldvt >= round_up(F*Nenc, 8)
            ops.gemm(qf.cross_v.w, enc[:F * Nenc], out=vt_all, c_pad8=True)   # V^T = Wv enc^T: [n_cross*Dq, F*Nenc]
        else:
            kv_all = ops.gemm(enc, qf.cross_kv.w, qf.cross_kv.b)              # [F*Nenc, n_cross*2*Dq]
        ops.profile_tag(0)
        Dp = h16.shape[1]
        ctx = torch.zeros(F * S, Dp, device=self.dev, dtype=dt)
        ctx_q = torch.zeros(F * K, Dp, device=self.dev, dtype=dt)
        t32 = torch.empty(F * S, Dp, device=self.dev, dtype=torch.float32)
        t16 = torch.empty(F * K, Dp, device=self.dev, dtype=dt) if out_fused else None
        scale = 1.0 / math.sqrt(hd)
        qmap = (K, S, 0, 1)
        tmap = (Lt, S, K, 1) if Lt else None
        for Lr in qf.layers:
            q16 = False          # this layer's query rows between the cross-attention output and the FFN LayerNorm: 16-bit only
            qkv = ops.gemm(h16, Lr.qkv.w, Lr.qkv.b)
            ld = qkv.stride(0)
            ops.attention(qkv[:, 0:Dq], qkv[:, Dq:2 * Dq], qkv[:, 2 * Dq:3 * Dq], ctx, F, heads, hd, S, S, scale,
                          S * ld, S * ld, S * ld, S * ctx.stride(0))
            ops.gemm(ctx, Lr.attn_out.w, Lr.attn_out.b, res=h32, out=t32, out_f32=True)
            ops.layernorm(t32, Lr.attn_ln[0], Lr.attn_ln[1], 1e-12, Dq, dt, y16=h16, y32=h32)
            ops.profile_tag(L.PROF_TAG_XATTN_BLOCK)
            if Lr.cross is not None and fused:
                j = Lr.cross.idx
                ops.qformer_xattn(h16, h32, F, K, S, Lr.cross.q_tiled, Lr.cross.q.b, Lr.cross.out_tiled, Lr.cross.out.b,
                                  k_all[:, j * Dq:(j + 1) * Dq], vt_all[j * Dq:(j + 1) * Dq], qf.cross_bv[j * Dq:(j + 1) * Dq],
                                  Nenc, Lr.cross.ln[0], Lr.cross.ln[1], 1e-12, Dq, heads, scale)
            elif Lr.cross is not None:
                cq = ops.gemm(h16, Lr.cross.q.w, Lr.cross.q.b, M=F * K, a_map=qmap)          # [F*K, Dq]
                j = Lr.cross.idx
                kk = kv_all[:, j * 2 * Dq: j * 2 * Dq + Dq]
                vv = kv_all[:, j * 2 * Dq + Dq: (j + 1) * 2 * Dq]
                ldk = kv_all.stride(0)
                ops.attention(cq[:, :Dq], kk, vv, ctx_q, F, heads, hd, K, Nenc, scale, K * cq.stride(0), Nenc * ldk,
                              Nenc * ldk, K * ctx_q.stride(0))
                if out_fused:
                    # the residual of this kernel and of the query FFN behind it is the 16-bit hidden state (the reference's
                    # half-precision arithmetic, tdc/Qformer.py:285-289,335-341): the fp32 copy of the query rows is neither
                    # read nor written until the FFN's LayerNorm rewrites both copies - half the bytes of the kernel
                    ops.qformer_xattn_out(h16, None, F, K, S, ctx_q, Lr.cross.out_tiled, Lr.cross.out.b, Lr.cross.ln[0],
                                          Lr.cross.ln[1], 1e-12, Dq, heads, res16=True)
                    q16 = True
                else:
                    ops.gemm(ctx_q, Lr.cross.out.w, Lr.cross.out.b, res=h32, r_map=qmap, out=t32, out_f32=True, M=F * K)
                    ops.layernorm(t32, Lr.cross.ln[0], Lr.cross.ln[1], 1e-12, Dq, dt, y16=h16, y32=h32, rows=F * K,
                                  y_map=qmap)
            ops.profile_tag(0)
            m = ops.gemm(h16, Lr.ffn_q.fc1.w, Lr.ffn_q.fc1.b, act=L.ACT_GELU_ERF, M=F * K, a_map=qmap)
            if q16:
                ops.gemm(m, Lr.ffn_q.fc2.w, Lr.ffn_q.fc2.b, res=h16, r_map=qmap, out=t16, M=F * K)
            else:
                ops.gemm(m, Lr.ffn_q.fc2.w, Lr.ffn_q.fc2.b, res=h32, r_map=qmap, out=t32, out_f32=True, M=F * K)
            if Lt:
                m2 = ops.gemm(h16, Lr.ffn_t.fc1.w, Lr.ffn_t.fc1.b, act=L.ACT_GELU_ERF, M=F * Lt, a_map=tmap)
                t32b = ops.gemm(m2, Lr.ffn_t.fc2.w, Lr.ffn_t.fc2.b, res=h32, r_map=tmap, out_f32=True, M=F * Lt)
            ops.layernorm(t16 if q16 else t32, Lr.ffn_q.ln[0], Lr.ffn_q.ln[1], 1e-12, Dq, dt, y16=h16, y32=h32, rows=F * K,
                          y_map=qmap)
            if Lt:
                ops.layernorm(t32b, Lr.ffn_t.ln[0], Lr.ffn_t.ln[1], 1e-12, Dq, dt, y16=h16, y32=h32, rows=F * Lt,
                              y_map=tmap)
        return h16, S

    def audio_tokens(self, beats_windows, sample_indices, T, lo=0, hi=None, window_sizes=None):
        """a20 (tdc/cambrian_arch.py:1552-1598): BEATs window features -> audio tokens of frames [lo, hi) of the T frames,
        [hi - lo, 50, 768] 16-bit (pooling of short / dropped seconds with tdc_adaptive_pool_tokens, zero padded tail).
        beats_windows: list of [1, n_w, 768] for ALL windows, or a dict {window: features} holding at least the windows the
        frames [lo, hi) draw from, together with window_sizes = token count of every window."""
        dt, dev = self.dtype, self.dev
        hi = T if hi is None else hi
        if isinstance(beats_windows, dict):
            assert window_sizes is not None
            src = beats_windows
        else:
            src = dict(enumerate(beats_windows))
            window_sizes = [int(w.reshape(-1, w.shape[-1]).shape[0]) for w in beats_windows]
        Da = next(iter(src.values())).shape[-1]
        plan = seg.audio_plan(list(window_sizes), [int(v) for v in sample_indices])
        wins = {}

        def win(w):
            if w not in wins:
                w2 = src[w].reshape(-1, Da).to(dev)
                assert w2.shape[0] == window_sizes[w]
                buf = torch.zeros(w2.shape[0], pad64(Da), device=dev, dtype=dt)
                buf[:, :Da] = w2.to(dt)
                wins[w] = buf
            return wins[w]
        out = torch.zeros(hi - lo, 50, Da, device=dev, dtype=dt)

        def pooled50(x):
            return x if x.shape[0] == 50 else ops.adaptive_pool_tokens(x.contiguous(), x.shape[0], 50, 1)
        for i in range(lo, min(hi, T, len(plan))):
            parts, direct = plan[i]
            toks = [pooled50(win(w)[s:e]) for (w, s, e) in parts]
            x = toks[0] if len(toks) == 1 else pooled50(torch.cat(toks, 0))
            out[i - lo] = x[:, :Da]
        return out

    def beats_windows(self, wav, mask=None, only=None):
        """tdc/cambrian_arch.py:1552-1560: BEATs features of the consecutive 10-s windows of wav [1, n] (16 kHz).
        `self.beats` is a beats.BeatsEncoder (set by the owner of the weights, e.g. model.initialize_audio).
        only = set of window indices -> dict {window: features} of just those windows."""
        if getattr(self, "beats", None) is None:
            raise RuntimeError("raw audio given but no BEATs encoder is attached (VideoEncoder.beats)")
        return self.beats.window_features(wav, only=only, mask=mask)

    def local_audio(self, audio, sample_indices, T, lo=0, hi=None):
        """a20 for the frames [lo, hi) of the T kept frames: the caller's `audio` (None, a [T, 50, 768] token tensor, or
        the dict forms of prepare_inputs_labels_for_multimodal) -> [hi - lo, 50, 768] audio tokens (None without audio).
        With a raw waveform only the BEATs windows those frames draw from are encoded (frame sharding: dist.py)."""
        hi = T if hi is None else hi
        if audio is None:
            return None
        if not isinstance(audio, dict):
            return audio[lo:hi]
        if audio.get("audio_tokens") is not None:
            return audio["audio_tokens"][lo:hi]
        wins = audio.get("beats_windows")
        if wins is not None:
            return self.audio_tokens(wins, sample_indices, T, lo, hi)
        wav = audio["audio_wav"]                               # raw 16 kHz waveform: BEATs on the device (8(f)-1)
        if getattr(self, "beats", None) is None:
            raise RuntimeError("raw audio given but no BEATs encoder is attached (VideoEncoder.beats)")
        sizes = self.beats.window_token_counts(wav.shape[1])
        plan = seg.audio_plan(sizes, [int(v) for v in sample_indices])
        need = sorted({w for i in range(lo, min(hi, T, len(plan))) for (w, _, _) in plan[i][0]})
        if lo == 0 and hi >= T:
            need = None                                        # the whole video: every window, one batched call
        feats = self.beats_windows(wav, audio.get("audio_wav_mask"), only=need)
        if need is None:
            return self.audio_tokens(feats, sample_indices, T, lo, hi)
        return self.audio_tokens(feats, sample_indices, T, lo, hi, window_sizes=sizes)

    def with_audio(self, X, T, N, audio):
        """a20: frames become [visual N | audio_proj(audio) Na] rows (tdc/cambrian_arch.py:1611-1614)."""
        c, dt, dev = self.c, self.dtype, self.dev
        if audio is None:
            return X, N
        Hp = X.shape[1]
        Na = audio.shape[1]
        A16 = torch.zeros(T * Na, pad64(audio.shape[2]), device=dev, dtype=dt)
        A16[:, : audio.shape[2]] = audio.reshape(T * Na, -1).to(dt)
        Xf = torch.empty(T * (N + Na), Hp, device=dev, dtype=dt)
        Xf.view(T, N + Na, Hp)[:, :N].copy_(X.view(T, N, Hp))          # visual rows (device copy)
        ops.gemm(A16, c.audio_proj.w, c.audio_proj.b, out=Xf, c_map=(Na, N + Na, N, 1))   # audio rows
        return Xf, N + Na

    def make_queries(self, Xf, N, Nf, key_rows):
        """a12: adaptive-avg-pool the (visual part of the) key frames N -> K tokens and apply query_proj.
        key_rows: frame indices into Xf.  Returns [len(key_rows)*K, Dq_pad]."""
        keys = torch.tensor(key_rows, dtype=torch.int32, device=self.dev)
        pooled = ops.adaptive_pool_tokens(Xf, N, self.K, len(key_rows), keys, frame_rows=Nf)
        return ops.gemm(pooled, self.c.query_proj.w, self.c.query_proj.b)

    def learned_queries(self):
        """query_type == 'learned' (tdc/cambrian_arch.py:1639-1640): `query_tokens` [1, K, Dq] -> [K, Dq_pad] 16-bit."""
        qt = self.c.query_tokens
        if qt is None:
            raise RuntimeError("query_type='learned' needs `query_tokens` in the state dict")
        return qt

    def _tile_cross_weights(self):
        """fragment-major copies of the cross-attention query / output weights for tdc_qformer_xattn (made once, on the device)"""
        qf = self.c.qformer
        if qf.cross_k is None:
            return
        for Lr in qf.layers:
            if Lr.cross is not None and Lr.cross.q_tiled is None:
                Lr.cross.q_tiled = ops.xattn_tile_weight(Lr.cross.q.w)
                Lr.cross.out_tiled = ops.xattn_tile_weight(Lr.cross.out.w)

    def _qformer_struct(self):
        """ctypes mirror of tdc_qformer_model (cached)."""
        fused_on = int(getattr(self, "xattn_mode", 1))
        if getattr(self, "_qf_struct", None) is not None and self._qf_struct[2] == fused_on:
            return self._qf_struct
        c, qf = self.c, self.c.qformer
        self._tile_cross_weights()

        def lin(l):
            return L.Lin(l.w.data_ptr(), l.b.data_ptr() if l.b is not None else None, l.w.shape[0], l.w.shape[1])
        zero = L.Lin(None, None, 0, 0)
        layers = (L.QformerLayer * len(qf.layers))()
        for i, Lr in enumerate(qf.layers):
            x = L.QformerLayer()
            x.qkv, x.attn_out = lin(Lr.qkv), lin(Lr.attn_out)
            x.attn_ln_g, x.attn_ln_b = Lr.attn_ln[0].data_ptr(), Lr.attn_ln[1].data_ptr()
            if Lr.cross is not None:
                x.has_cross, x.cross_idx = 1, Lr.cross.idx
                x.cross_q, x.cross_out = lin(Lr.cross.q), lin(Lr.cross.out)
                x.cross_ln_g, x.cross_ln_b = Lr.cross.ln[0].data_ptr(), Lr.cross.ln[1].data_ptr()
                if Lr.cross.q_tiled is not None:
                    x.cross_q_tiled, x.cross_out_tiled = Lr.cross.q_tiled.data_ptr(), Lr.cross.out_tiled.data_ptr()
            else:
                x.has_cross, x.cross_idx, x.cross_q, x.cross_out = 0, 0, zero, zero
            x.fq1, x.fq2 = lin(Lr.ffn_q.fc1), lin(Lr.ffn_q.fc2)
            x.fq_ln_g, x.fq_ln_b = Lr.ffn_q.ln[0].data_ptr(), Lr.ffn_q.ln[1].data_ptr()
            x.ft1, x.ft2 = lin(Lr.ffn_t.fc1), lin(Lr.ffn_t.fc2)
            x.ft_ln_g, x.ft_ln_b = Lr.ffn_t.ln[0].data_ptr(), Lr.ffn_t.ln[1].data_ptr()
            layers[i] = x
        m = L.QformerModel()
        m.dtype, m.dim, m.heads, m.n_layers, m.H, m.eps = ops._dtcode(self.dtype), qf.dim, self.qheads, len(qf.layers), \
            c.H, 1e-12
        m.word, m.pos, m.ldw = qf.word.data_ptr(), qf.pos.data_ptr(), qf.word.stride(0)
        m.emb_ln_g, m.emb_ln_b = qf.emb_ln[0].data_ptr(), qf.emb_ln[1].data_ptr()
        m.cross_kv, m.vision_proj = lin(qf.cross_kv), lin(c.vision_proj)
        if qf.cross_k is not None:
            m.cross_k, m.cross_v, m.cross_bv = lin(qf.cross_k), lin(qf.cross_v), qf.cross_bv.data_ptr()
            m.xattn_mode = fused_on
        m.layers_host = layers
        self._qf_struct = (m, layers, fused_on)
        return self._qf_struct

    def _prompt_tensor(self, prompt_ids):
        """BERT prompt ids as a device int32 tensor, validated on the host (an out-of-range id would fault in the kernel)."""
        if prompt_ids is None or len(prompt_ids) == 0:
            return None
        key = ("ids", tuple(int(i) for i in prompt_ids))
        hit = self._tables.get(key)
        if hit is None:
            qf = self.c.qformer
            assert max(key[1]) < qf.word.shape[0] and min(key[1]) >= 0 and len(key[1]) <= qf.pos.shape[0]
            hit = self._tables[key] = torch.tensor(key[1], dtype=torch.int32, device=self.dev)
        return hit

    def _compress_native(self, enc, F, Nf, qtable, qs, prompt_ids):
        import ctypes as C
        m = self._qformer_struct()[0]
        ids = self._prompt_tensor(prompt_ids)
        Lt = 0 if ids is None else ids.numel()
        K = self.K
        assert qs.dtype == torch.int32 and qs.numel() >= F        # value range checked on the host list (compress_frames)
        assert enc.shape[0] >= F * Nf and enc.shape[1] >= m.cross_kv.k and qtable.shape[1] >= pad64(m.dim)
        lib = L.load()
        need = lib.tdc_qformer_workspace_bytes(C.byref(m), F, K, Lt, Nf)
        ws = getattr(self, "_qf_ws", None)
        if ws is None or ws.numel() < need:
            ws = self._qf_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        out = torch.empty(F * K, pad64(self.c.H), device=self.dev, dtype=self.dtype)
        L.check(lib.tdc_qformer_fwd(C.byref(m), ops._ptr(enc), enc.stride(0), F, Nf, ops._ptr(qtable), qtable.stride(0),
                                    ops._ptr(qs), ops._ptr(ids), Lt, K, ops._ptr(out), out.stride(0), ops._ptr(ws),
                                    ws.numel(), ops._stream()), "tdc_qformer_fwd")
        return out

    def compress_frames(self, Xf, Nf, frame_rows, qtable, qsrc, prompt_ids, keep=None):
        """a13-a18 for the frames `frame_rows` of Xf: Q-Former against queries qtable[qsrc[f]] -> [F*K, >=H] unit rows."""
        c, dev = self.c, self.dev
        K, H = self.K, c.H
        F = len(frame_rows)
        Hp = Xf.shape[1]
        assert F == len(qsrc) and (max(qsrc) + 1) * K <= qtable.shape[0] and min(qsrc) >= 0
        assert min(frame_rows) >= 0 and (max(frame_rows) + 1) * Nf <= Xf.shape[0]
        fr = torch.tensor(frame_rows, dtype=torch.int32, device=dev)
        enc_idx = torch.zeros(F * Nf, 2, dtype=torch.int32, device=dev)
        enc_idx[:, 1] = (fr[:, None] * Nf + torch.arange(Nf, dtype=torch.int32, device=dev)[None, :]).reshape(-1)
        enc = ops.gather_rows([Xf], enc_idx, F * Nf, Hp, validated=True)      # frame_rows range-checked above
        qs = torch.tensor(qsrc, dtype=torch.int32, device=dev)
        if getattr(self, "native_qformer", True):
            comp = self._compress_native(enc, F, Nf, qtable, qs, prompt_ids)
            if keep is not None:
                keep.update(compressed=comp)
            return comp
        h16, S = self.qformer(enc, F, Nf, qtable, qs, prompt_ids)
        comp = ops.gemm(h16, c.vision_proj.w, c.vision_proj.b, M=F * K, a_map=(K, S, 0, 1))
        ops.l2_normalize(comp, F * K, H)
        if keep is not None:
            keep.update(compressed=comp, last_hidden=h16, S=S)
        return comp

    def query_width(self):
        return pad64(self.c.qformer.dim)

    def emit_into(self, Xf, comp, pairs, out):
        """emit() into the first len(pairs) rows of a caller-owned [>= len(pairs), H] buffer (the sharded path's send buffer)."""
        return self.emit(Xf, comp, pairs, out=out)

    def compact_rows(self, src, idx, cols):
        """rows src[idx[:, 1]] -> a new [len(idx), cols] tensor (one tdc_gather_rows; idx validated by its builder)."""
        return ops.gather_rows([src], idx, idx.shape[0], cols, validated=True)

    def emit(self, Xf, comp, pairs, splice=None, out=None):
        """a19: one gather over (frame tokens | context tokens | frame_seg) -> [len(pairs), H].
        a21 hand-off (cambrian_arch.py:1712-1790 / cambrian_qwen.py:457-462): with
        splice = {"table": embed_tokens.weight [V, H] 16-bit on this device, "before": ids, "after": ids} the same
        launch also gathers the text embeddings, so the result is the LLM's inputs_embeds row block
        [len(before) + len(pairs) + len(after), H] with no intermediate visual tensor."""
        import numpy as np
        tables = [Xf, comp if comp is not None else self.c.frame_seg, self.c.frame_seg]
        p_np = pairs.cpu().numpy() if torch.is_tensor(pairs) else np.asarray(pairs, dtype=np.int32)
        p_np = np.ascontiguousarray(p_np.reshape(-1, 2).astype(np.int32, copy=False))
        if splice is not None:
            tab = splice["table"]
            assert tab.is_cuda and tab.device == Xf.device and tab.dtype == Xf.dtype and tab.shape[1] == self.c.H, \
                "prefill hand-off needs embed_tokens on the engine device in the engine dtype"

            def text(ids_):
                t = np.zeros((len(ids_), 2), dtype=np.int32)
                t[:, 0] = 3
                t[:, 1] = np.asarray([int(v) for v in ids_], dtype=np.int32)
                return t
            p_np = np.concatenate([text(splice["before"]), p_np, text(splice["after"])], 0)
            tables.append(tab)
        ops.check_pairs_host(p_np, [t.shape[0] if t.dim() == 2 else 1 for t in tables])
        idx = torch.from_numpy(p_np).to(self.dev)
        return ops.gather_rows(tables, idx, idx.shape[0], self.c.H, out=out, validated=True)

    def compress(self, X, T, N, seg_indices, prompt_ids, max_visual_len, audio=None, keep=None, splice=None):
        """X [T*N, Hp] -> emitted visual tokens [n, H] (tdc/cambrian_arch.py:1520-1709)."""
        return compress_with(self, X, T, N, seg_indices, prompt_ids, max_visual_len, audio, keep, splice)

    # ------------------------------------------------------------------------------------------------ top level
    def encode_video(self, px_siglip, px_dino, image_size, budget_text_len, n_text_tokens, prompt_ids, audio=None,
                     frame_cap=224, keep=None, splice=None, video_index=None, info=None):
        """One video: pixels -> emitted visual tokens [n, H] (S0-S10).  `budget_text_len` is the text length used by
        get_max_num_frames (cambrian_arch.py:753-759), `n_text_tokens` the non-image token count (:1499-1505).
        frame_cap: the reference's "in case of OOM" constant 224 of BOTH caps (cambrian_arch.py:907-916 before the towers,
        :813-822 inside adapt_segment) as a parameter.
        keep: dict that receives every stage tensor (and makes the connector run its per-kernel form, which produces them);
        info: dict that receives the host-side facts only (frame_indices, selected, seg_indices, final_size, n_visual) - free."""
        return encode_video_with(self, px_siglip, px_dino, image_size, budget_text_len, n_text_tokens, prompt_ids,
                                 audio, frame_cap, keep, splice, video_index, info)


# ---------------------------------------------------------------------------------------------------------------------
# Orchestration over an "engine" (VideoEncoder on GPUs; the gloo tests of dist.py plug in a CPU test double that
# implements tower / sims_tensor / connector / with_audio / make_queries / compress_frames / emit / query_width).
def compress_with(e, X, T, N, seg_indices, prompt_ids, max_visual_len, audio=None, keep=None, splice=None, info=None):
    K = e.K
    cfg = getattr(e, "cfg", {})
    Xf, Nf = e.with_audio(X, T, N, audio)
    plan = seg.emit_plan(T, Nf, K, seg_indices, max_visual_len, cfg.get("add_static", True))
    comp = None
    if plan["comp_frames"]:
        if cfg.get("query_type", "Avg_pool") == "learned":      # cambrian_arch.py:1639-1640: one shared query block
            qtable, qsrc = e.learned_queries(), [0] * len(plan["comp_frames"])
        else:
            qtable, qsrc = e.make_queries(Xf, N, Nf, plan["key_frames"]), plan["comp_chunk"]
        comp = e.compress_frames(Xf, Nf, plan["comp_frames"], qtable, qsrc, prompt_ids, keep)
    pairs = seg.emit_pairs(plan, Nf, K)
    if keep is not None:
        keep["plan"] = plan
        keep["n_visual"] = len(pairs)
    if info is not None:
        info["n_visual"] = len(pairs)
    if splice is not None:
        return e.emit(Xf, comp, pairs, splice)
    return e.emit(Xf, comp, pairs)


def sample_indicator(T0, idx, video_index=None):
    """`sample_indices` of tdc/cambrian_arch.py:916-930: one 0/1 entry per second of the video's audio, 1 where a frame
    that survives the a1 cap was sampled.  video_index = the caller's `video_indices[i]` (0/1 per second, 1 = a frame was
    decoded there; None: input frame t is second t)."""
    if video_index is None:
        samp = [0] * T0
        for i in idx:
            samp[i] = 1
        return samp
    vi = [int(v) for v in (video_index.tolist() if hasattr(video_index, "tolist") else video_index)]
    if len(idx) == T0:
        return vi
    pos = [i for i, v in enumerate(vi) if v == 1]
    samp = [0] * len(vi)
    for i in idx:
        samp[pos[i]] = 1
    return samp


def encode_video_with(e, px_siglip, px_dino, image_size, budget_text_len, n_text_tokens, prompt_ids, audio=None,
                      frame_cap=224, keep=None, splice=None, video_index=None, info=None):
    cfg = e.cfg
    T0 = px_siglip.shape[0]
    idx = seg.uniform_indices(T0, min(seg.get_max_num_frames(budget_text_len, cfg), frame_cap))     # a1
    if len(idx) != T0:
        sel = torch.tensor(idx, device=px_siglip.device)
        px_siglip, px_dino = px_siglip[sel], px_dino[sel]
    T = len(idx)
    # The two towers are independent (a5's second frame cap is a no-op once a1 capped at <= frame_cap frames): with
    # `two_streams` they are enqueued on two HIP streams so one tower's kernel tails / memory-bound phases are
    # filled by the other tower's workgroups.
    side = None
    if getattr(e, "two_streams", False) and T <= frame_cap and px_siglip.is_cuda:
        side = e.tower_stream() if hasattr(e, "tower_stream") else torch.cuda.Stream(device=px_siglip.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            sig_early = e.tower("siglip", px_siglip)
    dino = e.tower("dino", px_dino)                                                                 # a4
    mns = cfg.get("max_num_segments", 24)
    sig = None
    if T <= mns + 1:                                                                                # a5
        sel2, seg_idx = list(range(T)), list(range(T))
    else:
        sel2 = seg.uniform_indices(T, frame_cap)
        if len(sel2) != T:
            s2 = torch.tensor(sel2, device=px_siglip.device)
            P = dino.shape[0] // T
            dino = dino.view(T, P, -1)[s2].reshape(len(sel2) * P, -1)
            px_siglip = px_siglip[s2]
            T = len(sel2)
        sims_dev = e.sims_tensor(dino, T)
        ev = e.mark() if hasattr(e, "mark") and sims_dev.is_cuda else None
        if side is None:
            # the SigLIP tower does not depend on the selection: enqueue it BEFORE the one host read of the similarities,
            # so the device keeps working while the host ranks them (a3) - and read them on a side stream (fetch), so the
            # host does not wait for the tower either
            sig = e.tower("siglip", px_siglip)
        sims = e.fetch(sims_dev, ev) if ev is not None else sims_dev.tolist()
        seg_idx = seg.select_segments(sims, mns)
        # a5 at the reference's precision (engine.selection_eps: bf16 DINOv2 operands with the fp16 copy of the tower at hand): when
        # the ranks that decide the selection are closer than the operand type's error, the pairs in that band are re-encoded
        eps = getattr(e, "selection_eps", None)
        band = seg.selection_band(sims, mns, eps) if eps else []
        if band and not seg.band_allowed(band, T, getattr(e, "selection_max_fraction", 0.125)):
            if info is not None:
                info["refine_skipped_pairs"] = len(band)      # a plateau at the decisive rank: the fast tower's ranking stands
            band = []
        if band:
            frames = seg.band_frames(band)
            pxd = px_dino if len(sel2) == px_dino.shape[0] else px_dino[torch.tensor(sel2, device=px_dino.device)]
            fp = e.precise_dino(pxd[torch.tensor(frames, device=pxd.device)])
            Pp = fp.shape[0] // len(frames)
            feats = {f: fp[j * Pp:(j + 1) * Pp] for j, f in enumerate(frames)}
            refined = e.pair_sims(feats, [(i, i + 1) for i in band]).tolist()
            seg_idx = seg.select_refined(sims, mns, eps, band, refined)
        if info is not None:
            info["refined_pairs"] = list(band)
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)
        sig = sig_early
        sig.record_stream(torch.cuda.current_stream())
    elif sig is None:
        sig = e.tower("siglip", px_siglip)                                                          # a3
    sizes = [tuple(image_size)] * T
    X, final_size = e.connector(sig, dino, T, sizes, keep)                                          # a6-a10
    N = X.shape[0] // T
    max_visual_len = cfg["tokenizer_model_max_length"] - cfg.get("inference_max_length", 16) - n_text_tokens
    pid = prompt_ids if cfg.get("text_input", True) else None
    if audio is not None:                                                                           # a20
        audio = e.local_audio(audio, sample_indicator(T0, idx, video_index), T, 0, T)
    vis = compress_with(e, X, T, N, seg_idx, pid, max_visual_len, audio, keep, splice, info)        # a11-a19 (+a21)
    if keep is not None:
        keep.update(frame_indices=idx, selected=sel2, seg_indices=seg_idx, siglip_feat=sig, dino_feat=dino,
                    final_size=final_size, X=X)
    if info is not None:
        info.update(frame_indices=idx, selected=sel2, seg_indices=seg_idx, final_size=final_size)
    return vis
