"""Host-side mirror of the reference's operator interface for the video-encoding path (the drop-in boundary):

    CambrianMetaModel            tdc/cambrian_arch.py:47-484   (owns the parameters, same names / state-dict keys)
    CambrianMetaForCausalLM      tdc/cambrian_arch.py:546-1898 (encode_images, adapt_segment,
                                                                prepare_inputs_labels_for_multimodal -> 10-tuple)

Same class / method / argument names, config keys and error behaviour, so `CambrianQwenForCausalLM.forward/.generate`
(tdc/language_model/cambrian_qwen.py:260-284, :415-438) and the eval drivers call it unchanged.  All tensor math runs
in libtdc_hip.so through `pipeline.VideoEncoder`; this file only holds parameters (nn.Parameter containers whose
names reproduce the reference state dict), does the host integer logic and the text/visual splice (a21).

Deliberate differences (documented in DESIGN.md): the dead `Qformer.cls` LM head is not allocated; a sample with
several <image> tokens raises IndexError as the reference does (it indexes past its one-entry-per-sample feature list,
tdc/cambrian_arch.py:1716), it is not given a meaning of its own; the three aux
return values used only by in-LLM samplers (`connector_only=False`) are returned as None; `video_indices=[None]`
(what generate() passes) is treated as "not given" instead of raising TypeError when the frame cap triggers
(tdc/cambrian_arch.py:919); training-only entry points raise NotImplementedError.
"""
from abc import ABC, abstractmethod

import torch
import torch.nn as nn

from . import lib as L
from . import segment as seg
from .pipeline import VideoEncoder

IGNORE_INDEX = -100        # tdc/constants.py
IMAGE_TOKEN_INDEX = -200


class ParamTree(nn.Module):
    """Container whose nested attribute names reproduce dotted state-dict keys ('0.weight', 'bert.encoder.layer.3...')."""

    def __init__(self, spec=None, init=None):
        super().__init__()
        for name, shape in (spec or {}).items():
            self.add(name, shape, init)

    def add(self, name, shape, init=None):
        head, _, rest = name.partition(".")
        if rest:
            if head not in self._modules:
                self.add_module(head, ParamTree())
            self._modules[head].add(rest, shape, init)
        else:
            p = nn.Parameter(torch.empty(*shape), requires_grad=False)
            (init or _default_init)(name, p)
            self.register_parameter(head, p)


def _default_init(name, p):
    with torch.no_grad():
        if p.ndim >= 2:
            p.normal_(0.0, 0.02)
        elif "weight" in name or "lambda1" in name:
            p.fill_(1.0)
        else:
            p.zero_()


def _randn_init(name, p):
    with torch.no_grad():
        p.normal_(0.0, 1.0)


# ------------------------------------------------------------------------------------------------ parameter specs
def _linear(spec, name, n, k, bias=True):
    spec[name + ".weight"] = (n, k)
    if bias:
        spec[name + ".bias"] = (n,)


def _ln(spec, name, n):
    spec[name + ".weight"] = (n,)
    spec[name + ".bias"] = (n,)


def qformer_spec(dq, layers, ffn, enc_width, vocab, max_pos, cross_freq=2):
    s = {}
    p = "bert."
    s[p + "embeddings.word_embeddings.weight"] = (vocab, dq)
    s[p + "embeddings.position_embeddings.weight"] = (max_pos, dq)
    _ln(s, p + "embeddings.LayerNorm", dq)
    for i in range(layers):
        l = p + "encoder.layer.%d." % i
        for n in ("query", "key", "value"):
            _linear(s, l + "attention.self." + n, dq, dq)
        _linear(s, l + "attention.output.dense", dq, dq)
        _ln(s, l + "attention.output.LayerNorm", dq)
        if i % cross_freq == 0:
            _linear(s, l + "crossattention.self.query", dq, dq)
            _linear(s, l + "crossattention.self.key", dq, enc_width)
            _linear(s, l + "crossattention.self.value", dq, enc_width)
            _linear(s, l + "crossattention.output.dense", dq, dq)
            _ln(s, l + "crossattention.output.LayerNorm", dq)
        for a, b in (("intermediate", "output"), ("intermediate_query", "output_query")):
            _linear(s, l + a + ".dense", ffn, dq)
            _linear(s, l + b + ".dense", dq, ffn)
            _ln(s, l + b + ".LayerNorm", dq)
    return s


def sva_spec(C, depth, n_towers=2, r=2):
    s = {}
    for i in range(depth):
        l = "layers.%d." % i
        for t in range(n_towers):
            s[l + "pos_embed_%d" % t] = (r * r, C)
        _linear(s, l + "proj_context", C, C, False)
        _linear(s, l + "proj_in", C, 2 * C, False)
        _linear(s, l + "proj_out.linear_1", C, C, False)
        _linear(s, l + "proj_out.linear_2", C, C, False)
        _ln(s, l + "norm", C)
        _ln(s, l + "cross_attn.q_proj.0", C)
        _linear(s, l + "cross_attn.q_proj.1", C, C, False)
        for t in range(n_towers):
            for kv in "kv":
                _ln(s, l + "cross_attn.%s_proj_%d.0" % (kv, t), C)
                _linear(s, l + "cross_attn.%s_proj_%d.1" % (kv, t), C, C, False)
        _linear(s, l + "cross_attn.o_proj", C, C, False)
    return s


def siglip_spec(D=1152, layers=27, mlp=4304, n_pos=729, patch=14):
    s = {"embeddings.patch_embedding.weight": (D, 3, patch, patch), "embeddings.patch_embedding.bias": (D,),
         "embeddings.position_embedding.weight": (n_pos, D)}
    for i in range(layers):
        l = "encoder.layers.%d." % i
        _ln(s, l + "layer_norm1", D)
        _ln(s, l + "layer_norm2", D)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _linear(s, l + "self_attn." + n, D, D)
        _linear(s, l + "mlp.fc1", mlp, D)
        _linear(s, l + "mlp.fc2", D, mlp)
    return s


def dino_spec(D=1536, layers=40, hidden=4096, n_pos=37 * 37, patch=14):
    s = {"embeddings.cls_token": (1, 1, D), "embeddings.position_embeddings": (1, 1 + n_pos, D),
         "embeddings.patch_embeddings.projection.weight": (D, 3, patch, patch),
         "embeddings.patch_embeddings.projection.bias": (D,)}
    for i in range(layers):
        l = "encoder.layer.%d." % i
        _ln(s, l + "norm1", D)
        _ln(s, l + "norm2", D)
        for n in ("query", "key", "value"):
            _linear(s, l + "attention.attention." + n, D, D)
        _linear(s, l + "attention.output.dense", D, D)
        s[l + "layer_scale1.lambda1"] = (D,)
        s[l + "layer_scale2.lambda1"] = (D,)
        _linear(s, l + "mlp.weights_in", 2 * hidden, D)
        _linear(s, l + "mlp.weights_out", D, hidden)
    _ln(s, "layernorm", D)
    return s


class VisionTowerHandle:
    """Stands where the reference keeps SiglipVisionTower / DinoVisionTower objects (a plain python list,
    tdc/cambrian_arch.py:62; tdc/multimodal_encoder/builder.py:8-37).  `vision_tower` holds the HF-named parameters."""

    def __init__(self, kind, name, arch):
        self.kind, self.vision_tower_name = kind, name
        self.arch = dict(arch)
        self.hidden_size = arch["D"]
        self.heads = arch["heads"]
        self.is_loaded = False
        self.vision_tower = None
        self._interp_size = 576
        self.image_size = 384 if kind == "siglip" else 378

    def load_model(self, device_map=None, state_dict=None):
        """Allocates the tower parameters (random init); pass `state_dict` (HF names) to load released weights - the
        reference pulls them from the hub (siglip_encoder.py:28, dino_encoder.py:28), which this image cannot reach."""
        a = self.arch
        if self.kind == "siglip":
            spec = siglip_spec(a["D"], a["layers"], a["mlp"], a["n_pos"])
        else:
            spec = dino_spec(a["D"], a["layers"], a["mlp"], a["n_pos"])
        self.vision_tower = ParamTree(spec)
        if state_dict is not None:
            sd = {k.replace("vision_model.", ""): v for k, v in state_dict.items()}
            self.vision_tower.load_state_dict(sd, strict=False)
        self.is_loaded = True


SIGLIP_SO400M = dict(D=1152, layers=27, mlp=4304, n_pos=729, heads=16)
DINOV2_GIANT = dict(D=1536, layers=40, mlp=4096, n_pos=37 * 37, heads=24)


def build_vision_tower_aux_list(config, tower_archs=None, **kwargs):
    """tdc/multimodal_encoder/builder.py:8-37: tower picked by name ('siglip' / 'dinov2')."""
    names = getattr(config, "mm_vision_tower_aux_list", getattr(config, "vision_tower_aux_list", None))
    out = []
    for i, n in enumerate(names):
        if "siglip" in n.lower():
            out.append(VisionTowerHandle("siglip", n, (tower_archs or {}).get("siglip", SIGLIP_SO400M)))
        elif "dinov2" in n.lower():
            out.append(VisionTowerHandle("dino", n, (tower_archs or {}).get("dino", DINOV2_GIANT)))
        else:
            raise ValueError(f"Unknown vision tower: {n}")
    return out


class BeatsHandle:
    """stands where the reference keeps `audio_encoder.beats` (a BEATs nn.Module): checkpoint cfg + state dict; the
    compute lives in beats.BeatsEncoder."""

    def __init__(self, cfg, state):
        self.cfg = dict(cfg) if not hasattr(cfg, "__dict__") or isinstance(cfg, dict) else dict(cfg.__dict__)
        self.state = state
        self._enc = None

    def state_dict(self):
        return self.state

    def extract_features(self, source, padding_mask=None, feature_only=True, device=None, dtype=torch.float16):
        """BEATs.extract_features(..., feature_only=True) (BEATs.py:131-178) -> (features, padding_mask)."""
        if not feature_only:
            raise NotImplementedError("the AS2M classifier head is not on the path")
        if self._enc is None:
            from .beats import BeatsEncoder
            self._enc = BeatsEncoder(self.state, self.cfg, dtype=dtype,
                                     device=device or ("cuda:%d" % torch.cuda.current_device()))
        feats = self._enc.extract_features(source, padding_mask)
        if padding_mask is not None:         # the reference hands back the mask reduced to one flag per token (BEATs.py:142-153)
            pm = torch.as_tensor(padding_mask).bool()
            frames = int(L.load().tdc_fbank_frames(int(pm.shape[1])))
            padding_mask = self._enc.forward_padding_mask(feats.shape[1], self._enc.forward_padding_mask(frames, pm))
        return feats, padding_mask


class AudioEncoderHandle:
    """tdc/audio_models/audio_encoder.py:20-70 reduced to what the path uses: `.beats_path`, `.beats`."""

    def __init__(self, beats_path="", beats_ckpt=None):
        self.beats_path = beats_path
        self.beats = None
        if beats_ckpt is None and beats_path:
            import os
            if os.path.exists(beats_path):
                beats_ckpt = torch.load(beats_path, map_location="cpu")
        if beats_ckpt is not None:
            self.beats = BeatsHandle(beats_ckpt["cfg"], beats_ckpt["model"])


class CambrianMetaModel:
    """Mixin placed before the HF base model in the MRO (tdc/cambrian_arch.py:47-181)."""

    def __init__(self, config):
        super(CambrianMetaModel, self).__init__(config)
        if hasattr(config, "mm_vision_tower_aux_list"):
            projector_type = getattr(config, "mm_projector_type", "linear")
            if projector_type != "sva":
                raise NotImplementedError("only mm_projector_type='sva' is on the accelerated path "
                                          "(tdc/cambrian_arch.py:55-161)")
            C = config.vision_hidden_size
            H = config.hidden_size
            self.vision_tower_aux_list = build_vision_tower_aux_list(
                config, tower_archs=getattr(config, "tdc_tower_archs", None), delay_load=True)
            self.mm_projector = ParamTree({"0.weight": (H, C * config.num_query_group), "0.bias": (H,),
                                           "2.weight": (H, H), "2.bias": (H,)})
            for i, t in enumerate(self.vision_tower_aux_list):
                setattr(self, "mm_projector_aux_%d" % i, ParamTree({
                    "0.weight": (C, t.hidden_size), "0.bias": (C,), "2.weight": (C, C), "2.bias": (C,),
                    "3.weight": (C,), "3.bias": (C,)}))
            tok = config.mm_vision_tower_aux_token_len_list
            for g in range(config.num_query_group):
                r = int(tok[0] ** 0.5) // int(config.query_num_list[g] ** 0.5)
                setattr(self, "vision_sampler_%d" % g, ParamTree(
                    sva_spec(C, config.connector_depth, len(self.vision_tower_aux_list), r),
                    init=lambda n, p: (_randn_init if n.startswith("pos_embed") else _default_init)(n, p)))
            if not getattr(config, "connector_only", True):
                raise NotImplementedError("connector_only=False (in-LLM vision samplers) is outside the hot path")
            self.vision_query = nn.Parameter(torch.randn(config.num_query_group, C), requires_grad=False)
            self.image_newline = nn.Parameter(torch.randn(H) * 0.02, requires_grad=False)
            self.frame_seg = nn.Parameter(torch.randn(H), requires_grad=False)
        self.initialize_compressor(config=config, pretrained_qformer=None,
                                   context_token_num=getattr(config, "context_token_num", 16))
        if getattr(config, "audio_input", False):
            self.audio_proj = ParamTree({"weight": (config.hidden_size, 768), "bias": (config.hidden_size,)})
        self._tdc_encoder = None

    def get_vision_tower_aux_list(self):
        return getattr(self, "vision_tower_aux_list", None)

    def get_frame_pos(self, time_range):
        raise NotImplementedError("frame_pos=True is not used by the released configs (tdc/cambrian_arch.py:183-190)")

    def initialize_vision_modules(self, model_args, fsdp=None):
        raise NotImplementedError("training-time module initialisation (tdc/cambrian_arch.py:206-401) is out of scope")

    def initialize_audio(self, model_args=None, beats_path="./checkpoints/audio_encoder/BEATs/"
                         "BEATs_iter3_plus_AS2M_finetuned_on_AS2M_cpt2.pt", beats_ckpt=None):
        """tdc/cambrian_arch.py:451-467: `audio_proj` + the BEATs half of AudioEncoder (audio_encoder.py:60-70; the
        Whisper / speech-Q-Former half is commented out in the reference).  `beats_ckpt` = {'cfg': ..., 'model': state
        dict} (what torch.load(beats_path) returns) may be passed directly."""
        if not hasattr(self, "audio_proj"):
            self.audio_proj = ParamTree({"weight": (self.config.hidden_size, 768), "bias": (self.config.hidden_size,)})
        self.audio_encoder = AudioEncoderHandle(beats_path, beats_ckpt)
        self._tdc_beats = None
        return self.audio_encoder

    def tdc_beats(self, device=None, dtype=None):
        """BeatsEncoder built (once) from audio_encoder.beats' checkpoint; None when no audio encoder is attached."""
        ae = getattr(self, "audio_encoder", None)
        if ae is None or ae.beats is None:
            return None
        if getattr(self, "_tdc_beats", None) is None:
            from .beats import BeatsEncoder
            eng = self.tdc_engine()
            self._tdc_beats = BeatsEncoder(ae.beats.state, ae.beats.cfg, dtype=dtype or eng.dtype, device=device or eng.dev)
        return self._tdc_beats

    def initialize_compressor(self, config, pretrained_qformer=None, context_token_num=16):
        """tdc/cambrian_arch.py:469-484 (+ init_Qformer :403-424): bert-base Q-Former with cross-attention to the LLM
        embedding width every other layer; query_proj / vision_proj."""
        H = config.hidden_size
        qc = getattr(config, "tdc_qformer_arch", None) or dict(hidden=768, layers=12, heads=12, ffn=3072, vocab=30522,
                                                               max_pos=512)
        self._qformer_arch = dict(qc)
        self.Qformer = ParamTree(qformer_spec(qc["hidden"], qc["layers"], qc["ffn"], H, qc["vocab"], qc["max_pos"]))
        self.query_tokens = nn.Parameter(torch.zeros(1, context_token_num, qc["hidden"]).normal_(0, 0.02),
                                         requires_grad=False)
        self.vision_proj = ParamTree({"weight": (H, qc["hidden"]), "bias": (H,)})
        self.query_proj = ParamTree({"weight": (qc["hidden"], H), "bias": (qc["hidden"],)})
        self.bert_tokenizer = None
        try:  # the reference loads ./checkpoints/bert-base-uncased (tdc/cambrian_arch.py:405)
            from transformers import BertTokenizer
            self.bert_tokenizer = BertTokenizer.from_pretrained("./checkpoints/bert-base-uncased",
                                                                truncation_side="right")
        except Exception:
            self.bert_tokenizer = None

    # ---- HIP engine -------------------------------------------------------------------------------------------
    def tdc_state_dict(self):
        """reference-named state dict (without 'model.') of everything on the path, towers included."""
        sd = {}
        for k, v in self.state_dict().items():
            if k.startswith(("mm_projector", "vision_sampler_", "vision_query", "image_newline", "frame_seg",
                             "Qformer.", "vision_proj", "query_proj", "audio_proj", "query_tokens")):
                sd[k] = v
        for i, t in enumerate(self.vision_tower_aux_list):
            if not t.is_loaded:
                t.load_model()
            for k, v in t.vision_tower.state_dict().items():
                sd["vision_tower_aux_list.%d.vision_tower.%s" % (i, k)] = v
        return sd

    def tdc_engine(self, device=None, dtype=None, refresh=False):
        """Build (once) the VideoEncoder from the current parameters: pads / fuses / uploads the weights.
        Keys that are not reference keys (all optional; INTEGRATION.md lists them):
        `config.tdc_fp8_towers = True` / 2 / 3 selects e4m3 operands for the towers' qkv / fc1 GEMMs (2: out-proj / fc2 as
        well; 3: fc1 also writes the e4m3 MLP hidden itself) - a throughput mode, not a parity mode;
        `config.tdc_tower_dtype = "bfloat16" | "float16"`: GEMM operand type of the two ViT towers under a connector / Q-Former
        in `dtype`; `config.tdc_dino_dtype`: the same for the DINOv2 tower alone (it alone drives the a5 segment selection,
        tdc/cambrian_arch.py:832-849: "float16" there keeps the similarities at the reference's own precision under a bf16
        SigLIP tower);
        `config.tdc_selection_refine` (default: automatic - on when the DINOv2 operands are bf16) / `config.tdc_selection_eps`
        (default 1e-3): the a5 segment selection at the reference's precision under bf16 DINOv2 operands - the pairs whose
        similarities decide the selection and lie closer than the operand type's error are re-encoded by an fp16-operand copy of
        the DINOv2 tower, as long as their frames number at most max(8, `config.tdc_selection_max_fraction` (default 1/8) x frames)
        (VideoEncoder.selection_refine);
        `config.tdc_tower_res_dtype = "float16" | "bfloat16" | "float32"`: the towers' residual stream in HBM.  Default:
        "float16" when the towers' operands are fp16 - the reference's own arithmetic, its HF towers run under
        torch_dtype=float16 (tdc/builder.py:69) -, "float32" otherwise (bf16 operands reach 3e38, an fp16 stream ends at 65504:
        that combination - what bench.py measures - is opt-in; fp8 towers follow the same rule);
        `config.tdc_tower_batch`: frames per tower batch; default: every frame of the call up to 512, bounded by the free HBM
        (VideoEncoder.auto_tower_batch; the result does not depend on it bit for bit, the reference's own chunk is 64,
        tdc/cambrian_arch.py:698-745);
        `config.tdc_frame_cap` (default 224 = the reference's "in case of OOM" constant, tdc/cambrian_arch.py:907-916,813-822):
        the cap of both frame sub-samplings; `config.tdc_shard_frames`: see prepare_inputs_labels_for_multimodal.
        bench.py's line is reproduced by dtype=float16, tdc_tower_dtype="bfloat16", tdc_tower_res_dtype="float16",
        tdc_frame_cap=T (its `product_setting` field says so; `bench.py --via-mixin` runs exactly that)."""
        if self._tdc_encoder is None or refresh:
            cfg = {k: getattr(self.config, k) for k in dir(self.config)
                   if not k.startswith("_") and (isinstance(getattr(self.config, k, None), (int, float, str, bool, list)) or
                                                 (k.startswith("tdc_") and isinstance(getattr(self.config, k, None), torch.dtype)))}
            device = device or ("cuda:%d" % torch.cuda.current_device())
            dtype = dtype or (self.dtype if self.dtype in (torch.float16, torch.bfloat16) else torch.float16)
            towers = self.vision_tower_aux_list
            fp8 = int(cfg.get("tdc_fp8_towers", 0) or 0)
            tower_dtype = _dtype_key(cfg, "tdc_tower_dtype", allow32=False)
            dino_dtype = _dtype_key(cfg, "tdc_dino_dtype", allow32=False)
            if "tdc_tower_res_dtype" in cfg:
                res = _dtype_key(cfg, "tdc_tower_res_dtype")
            else:
                res = torch.float16 if (tower_dtype or dtype) == torch.float16 and (dino_dtype or tower_dtype or dtype) == \
                    torch.float16 else None
            sel_refine = cfg.get("tdc_selection_refine")            # None = automatic (on under bf16 DINOv2 operands)
            if sel_refine is not None and not isinstance(sel_refine, bool):
                raise ValueError("config.tdc_selection_refine must be True / False (or absent for the automatic choice), got %r"
                                 % (sel_refine,))
            sel_eps = cfg.get("tdc_selection_eps", 1e-3)
            if isinstance(sel_eps, bool) or not isinstance(sel_eps, (int, float)) or not (0 < sel_eps < 1):
                raise ValueError("config.tdc_selection_eps must be a similarity error bound in (0, 1), got %r" % (sel_eps,))
            sel_frac = cfg.get("tdc_selection_max_fraction", 0.125)
            if isinstance(sel_frac, bool) or not isinstance(sel_frac, (int, float)) or not (0 <= sel_frac <= 1):
                raise ValueError("config.tdc_selection_max_fraction must be a fraction of the frames in [0, 1], got %r" % (sel_frac,))
            tb = cfg.get("tdc_tower_batch")
            if tb is not None and (isinstance(tb, bool) or not isinstance(tb, int) or tb < 0):
                raise ValueError("config.tdc_tower_batch must be a positive frame count (or 0 / absent for the automatic "
                                 "choice), got %r" % (tb,))
            self._tdc_encoder = VideoEncoder(self.tdc_state_dict(), cfg, dtype=dtype, device=device,
                                             siglip_heads=towers[0].heads, dino_heads=towers[1].heads,
                                             qformer_heads=self._qformer_arch["heads"], fp8_towers=fp8,
                                             tower_batch=tb or None, tower_dtype=tower_dtype, dino_dtype=dino_dtype,
                                             ln_fuse=bool(cfg.get("tdc_ln_fuse", False)),
                                             selection_refine=sel_refine, selection_eps=sel_eps,
                                             selection_max_fraction=sel_frac,
                                             tower_res_dtype=res)
        return self._tdc_encoder

    def tdc_frame_cap(self):
        """`config.tdc_frame_cap`: the reference's constant 224 of both frame caps (tdc/cambrian_arch.py:907-916,813-822; SURVEY
        D3) as a config key.  224 (the default) is the parity setting."""
        cap = getattr(self.config, "tdc_frame_cap", 224)
        if cap is None:
            return 224
        if isinstance(cap, bool) or not isinstance(cap, int) or cap < 1:
            raise ValueError("config.tdc_frame_cap must be a positive frame count, got %r" % (cap,))
        return cap

    def tdc_sharded_engine(self):
        """dist.ShardedVideoEncoder over the default process group when `config.tdc_shard_frames` is set and
        torch.distributed runs with more than one rank (one process per GPU, RCCL); None otherwise."""
        if not getattr(self.config, "tdc_shard_frames", False):
            return None
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
            return None
        sh = getattr(self, "_tdc_sharded", None)
        if sh is None or sh.e is not self.tdc_engine():
            from .dist import ShardedVideoEncoder
            sh = self._tdc_sharded = ShardedVideoEncoder(self.tdc_engine(), dist.get_rank(), dist.get_world_size())
        return sh


_DT16 = {"bfloat16": torch.bfloat16, "bf16": torch.bfloat16, "float16": torch.float16, "fp16": torch.float16,
         "half": torch.float16, "float32": None, "fp32": None, "float": None, None: None}


def _dtype_key(cfg, key, allow32=True):
    """config string -> torch dtype (None = fp32 / not given), with an error that names the key and the accepted values."""
    v = cfg.get(key)
    if isinstance(v, torch.dtype):
        v = str(v).replace("torch.", "")
    if v not in _DT16 or (not allow32 and v is not None and _DT16[v] is None):
        ok = sorted(k for k in _DT16 if k is not None and (allow32 or _DT16[k] is not None))
        raise ValueError("config.%s = %r is not one of %s" % (key, v, ok))
    return _DT16[v]


class CambrianMetaForCausalLM(ABC):
    """tdc/cambrian_arch.py:546-1898."""

    @abstractmethod
    def get_model(self):
        pass

    def get_vision_tower_aux_list(self):
        return self.get_model().get_vision_tower_aux_list()

    # ---- a2 -----------------------------------------------------------------------------------------------------
    def encode_images(self, image_aux_list, encode_type=None):
        """tdc/cambrian_arch.py:698-745: run a tower ('dino' -> last entry, 'siglip' -> first, None -> all) in 64-frame
        batches; returns [B, 576, D] tensors (list for encode_type None)."""
        eng = self.get_model().tdc_engine()
        towers = self.get_model().get_vision_tower_aux_list()

        def run(kind, px, D):
            out = eng.tower(kind, px.to(eng.dev))
            return out[:, :D].reshape(px.shape[0], -1, D)
        if encode_type == "dino":
            return run("dino", image_aux_list[-1], towers[-1].hidden_size)
        if encode_type == "siglip":
            return run("siglip", image_aux_list[0], towers[0].hidden_size)
        return [run(t.kind, px, t.hidden_size) for px, t in zip(image_aux_list, towers)]

    # ---- a1 -----------------------------------------------------------------------------------------------------
    def _budget_text_len(self, cur_input_ids):
        pad_id = 128002 if "llama" in getattr(self.get_model().config, "model_type", "") else 151643
        pos = torch.where(cur_input_ids == pad_id)[-1]
        return int(pos[0]) if len(pos) > 0 else len(cur_input_ids)

    def _cfg(self):
        c = self.get_model().config
        keys = ("context_token_num", "audio_input", "add_static", "tokenizer_model_max_length", "inference_max_length",
                "max_num_segments", "text_input", "query_type")
        d = {k: getattr(c, k) for k in keys if hasattr(c, k)}
        return d

    def get_max_num_frames(self, cur_input_ids):
        """tdc/cambrian_arch.py:748-780."""
        return seg.get_max_num_frames(self._budget_text_len(cur_input_ids), self._cfg())

    def adapt_segment(self, feature_list, split_sizes, new_image_aux_list, window_size=64, threshold=0.9,
                      max_num_segments=24):
        """tdc/cambrian_arch.py:783-861 (same 5-tuple).  feature_list: DINO features [sum T, 576, D]."""
        eng = self.get_model().tdc_engine()
        feats = torch.split(feature_list, split_sizes, dim=0)
        a0 = torch.split(new_image_aux_list[0], split_sizes, dim=0)
        a1 = torch.split(new_image_aux_list[1], split_sizes, dim=0)
        out_f, out_0, out_1, new_sizes, sel_all, seg_all = [], [], [], [], [], []
        for i, f in enumerate(feats):
            T = len(f)
            if T <= max_num_segments + 1:
                idx, segi = list(range(T)), list(range(T))
            else:
                idx = seg.uniform_indices(T, self.get_model().tdc_frame_cap())     # :813-822 (config.tdc_frame_cap)
                ff = f[idx] if len(idx) != T else f
                P, D = ff.shape[1], ff.shape[2]
                flat = ff.reshape(len(idx) * P, D).contiguous()
                if D % 8:
                    raise ValueError("feature width must be a multiple of 8")
                segi = seg.select_segments(eng.frame_sims(flat.to(eng.dtype), len(idx)), max_num_segments)
            sel = torch.tensor(idx)
            out_f.append(f[idx] if len(idx) != T else f)
            out_0.append(a0[i][idx] if len(idx) != T else a0[i])
            out_1.append(a1[i][idx] if len(idx) != T else a1[i])
            new_sizes.append(len(idx))
            sel_all.append(sel)
            seg_all.append(torch.tensor(segi))
        return torch.cat(out_f, 0), new_sizes, [torch.cat(out_0, 0), torch.cat(out_1, 0)], sel_all, seg_all

    # ---- the hot path ---------------------------------------------------------------------------------------------
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images, image_aux_attention_masks_list=None, image_sizes=None,
                                             video_indices=None, prompts=None, audios=None):
        """tdc/cambrian_arch.py:864-1844.  Returns the reference's 10-tuple."""
        model = self.get_model()
        towers = model.get_vision_tower_aux_list()
        if towers is None or images is None or input_ids.shape[1] == 1:
            return (input_ids, position_ids, attention_mask, past_key_values, None, labels, None, None, None, None)
        eng = model.tdc_engine()
        cfgd = self._cfg()
        K = cfgd.get("context_token_num", 16)
        H = model.config.hidden_size
        is_video = type(images[0]) is list or images[0].ndim == 5
        bsz = input_ids.shape[0]
        # every integer decision below (text lengths, <image> positions, which rows survive the mask) is taken on ONE host copy of
        # the ids / mask: a device-side torch.where(...).tolist() waits for everything queued on the stream - the previous
        # video's encode when calls follow each other
        ids_host = input_ids.detach().cpu()
        mask_host = None if attention_mask is None else attention_mask.detach().cpu().bool()
        visual = []          # per sample: [n_tokens, H] tensor on the engine device
        spliced = []         # per sample: `visual[i]` already holds the text rows around the visual tokens
        final_size = []
        for i in range(bsz):
            if is_video:
                vid_s, vid_d = images[0][i], images[1][i]
                if vid_s.ndim == 3:
                    vid_s, vid_d = vid_s.unsqueeze(0), vid_d.unsqueeze(0)
            else:
                vid_s, vid_d = images[0][i:i + 1], images[1][i:i + 1]
            cur_ids = ids_host[i]
            if mask_host is not None:
                cur_ids = cur_ids[mask_host[i] | (cur_ids == IMAGE_TOKEN_INDEX)]
            n_text = int((cur_ids != IMAGE_TOKEN_INDEX).sum())
            prompt_ids = None
            if is_video and cfgd.get("text_input", True):
                prompt = prompts[i] if prompts is not None else None
                if isinstance(prompt, str):
                    if model.bert_tokenizer is None:
                        raise RuntimeError("model.bert_tokenizer is not set (the reference loads "
                                           "./checkpoints/bert-base-uncased, tdc/cambrian_arch.py:405)")
                    prompt_ids = model.bert_tokenizer(prompt, padding="longest", truncation=True, max_length=256,
                                                      return_tensors="pt").input_ids[0].tolist()
                elif prompt is not None:
                    prompt_ids = [int(x) for x in prompt]      # pre-tokenised BERT ids
            audio = None
            if audios is not None and audios[i] is not None:
                # {"beats_windows": [BEATs features of the consecutive 10-s windows, [1, n, 768] each]} or
                # {"audio_tokens": [T, 50, 768]} (already interleaved); the BEATs encoder itself is not part of the path
                audio = audios[i]
                if not isinstance(audio, dict):
                    audio = {"audio_tokens": audio.to(eng.dev)}
                elif audio.get("audio_wav") is not None and audio.get("beats_windows") is None \
                        and audio.get("audio_tokens") is None:
                    # the reference's own audio dict (cambrian_arch.py:1547): raw waveform -> BEATs on the device
                    eng.beats = model.tdc_beats()
            keep = {}
            if is_video:
                # a21 hand-off: when embed_tokens lives on the engine device in the engine dtype the
                # emission gather also fetches the text embeddings and writes the inputs_embeds rows directly
                splice = None
                emb_w = getattr(getattr(model, "embed_tokens", None), "weight", None)
                pos_img = torch.where(cur_ids == IMAGE_TOKEN_INDEX)[0].tolist()
                if (getattr(self, "tdc_prefill_handoff", True) and emb_w is not None and emb_w.is_cuda
                        and emb_w.device == eng.dev and emb_w.dtype == eng.dtype and len(pos_img) == 1):
                    splice = {"table": emb_w, "before": cur_ids[:pos_img[0]].tolist(),
                              "after": cur_ids[pos_img[0] + 1:].tolist()}
                # video_indices[i]: 0/1 per second of audio, 1 where a frame was decoded (cambrian_arch.py:916-926); [None]
                # (what generate() passes) / None: input frame t is second t
                vindex = video_indices[i] if video_indices is not None and i < len(video_indices) else None
                cap = model.tdc_frame_cap()
                btl = self._budget_text_len(ids_host[i])
                sharded = model.tdc_sharded_engine()
                if sharded is not None:
                    # config.tdc_shard_frames under an initialised torch.distributed (one process per GPU, every rank called
                    # with the same video - how the reference's eval drivers start their workers, eval/eval_mlvu.py:129-157 -
                    # but here the ranks split the FRAMES of the one video): this rank encodes frames [lo, hi) of the a1
                    # selection, dist.ShardedVideoEncoder exchanges what crosses the rank boundaries and all-gathers the
                    # emitted tokens, so every rank returns the same 10-tuple as the serial path, bit for bit
                    fp = sharded.frame_plan(vid_s.shape[0], btl, cap, vindex)
                    ts = getattr(model.config, "tdc_two_streams", None)
                    eng.two_streams = bool(ts) if ts is not None else (fp["hi"] - fp["lo"]) <= 128
                    sel_s = torch.as_tensor(fp["siglip_frames"], dtype=torch.long, device=vid_s.device)
                    sel_d = torch.as_tensor(fp["dino_frames"], dtype=torch.long, device=vid_d.device)
                    vis = sharded.encode_video(vid_s[sel_s].to(eng.dev), vid_d[sel_d].to(eng.dev), fp["T"],
                                               tuple(image_sizes[i]), n_text, prompt_ids, audio=audio,
                                               sample_indices=fp["sample_indices"])
                    keep["final_size"] = [seg.unpad_newline_map(eng.side, tuple(image_sizes[i]), 0)[1]] * fp["T"]
                    splice = None
                else:
                    ts = getattr(model.config, "tdc_two_streams", None)
                    eng.two_streams = bool(ts) if ts is not None else min(vid_s.shape[0], cap) <= 128
                    vis = eng.encode_video(vid_s.to(eng.dev), vid_d.to(eng.dev), tuple(image_sizes[i]),
                                           budget_text_len=btl, n_text_tokens=n_text, prompt_ids=prompt_ids, audio=audio,
                                           frame_cap=cap, info=keep, splice=splice, video_index=vindex)
                spliced.append(splice is not None)
            else:
                # single images: every image is a static frame, no segmentation / Q-Former (cambrian_arch.py:980-983)
                sig = eng.tower("siglip", vid_s.to(eng.dev))
                dino = eng.tower("dino", vid_d.to(eng.dev))
                X, sizes = eng.connector(sig, dino, 1, [tuple(image_sizes[i])])
                vis = X[:, :H]
                keep["final_size"] = sizes
            if not is_video:
                spliced.append(False)
            visual.append(vis)
            final_size.extend(keep["final_size"])
        # ---- a21: splice text embeddings and visual tokens, truncate, pad (cambrian_arch.py:1425-1495, :1712-1844)
        _labels, _position_ids, _attention_mask = labels, position_ids, attention_mask
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
        else:
            attention_mask = attention_mask.bool()
        if position_ids is None:
            position_ids = torch.arange(0, input_ids.shape[1], dtype=torch.long, device=input_ids.device)
        if labels is None:
            labels = torch.full_like(input_ids, IGNORE_INDEX)
        attention_mask = attention_mask | (input_ids == IMAGE_TOKEN_INDEX)
        keep_host = torch.ones_like(ids_host, dtype=torch.bool) if mask_host is None else \
            (mask_host | (ids_host == IMAGE_TOKEN_INDEX))
        embed = model.embed_tokens
        new_embeds, new_labels = [], []
        for i in range(bsz):
            if bool(keep_host[i].all()):
                ids, lab, ids_h = input_ids[i], labels[i], ids_host[i]
            else:           # integer row indices made on the host: a boolean mask on the device would wait for its own count
                rows = torch.nonzero(keep_host[i])[:, 0]
                ids, lab, ids_h = input_ids[i][rows.to(input_ids.device)], labels[i][rows.to(labels.device)], ids_host[i][rows]
            img_pos = torch.where(ids_h == IMAGE_TOKEN_INDEX)[0].tolist()
            vis = visual[i]
            if not img_pos:
                new_embeds.append(torch.cat([embed(ids), vis[0:0].to(embed.weight.dtype)], 0))
                new_labels.append(lab)
                continue
            if len(img_pos) != 1:
                # The reference takes one entry of its per-video feature list per <image> token (cur_image_idx,
                # tdc/cambrian_arch.py:1457-1495, :1712-1727) while that list holds one entry per sample: a sample with
                # several <image> tokens runs off its end.  Same exception type here.
                raise IndexError("list index out of range: %d <image> tokens in sample %d, one video / image per sample "
                                 "(tdc/cambrian_arch.py:1716)" % (len(img_pos), i))
            p = img_pos[0]
            if spliced[i]:
                n_vis = vis.shape[0] - (ids.shape[0] - 1)
                new_embeds.append(vis)
                new_labels.append(torch.cat([lab[:p], torch.full((n_vis,), IGNORE_INDEX, device=lab.device,
                                                                 dtype=lab.dtype), lab[p + 1:]]))
                continue
            text = embed(torch.cat([ids[:p], ids[p + 1:]]))
            vis = vis.to(text.dtype).to(text.device)
            new_embeds.append(torch.cat([text[:p], vis, text[p:]], 0))
            new_labels.append(torch.cat([lab[:p], torch.full((vis.shape[0],), IGNORE_INDEX, device=lab.device,
                                                             dtype=lab.dtype), lab[p + 1:]]))
        mx = getattr(self.config, "tokenizer_model_max_length", None)
        if mx is not None:
            new_embeds = [x[:mx] for x in new_embeds]
            new_labels = [x[:mx] for x in new_labels]
        max_len = max(x.shape[0] for x in new_embeds)
        left = getattr(self.config, "tokenizer_padding_side", "right") == "left"
        emb_pad = []
        lab_pad = torch.full((bsz, max_len), IGNORE_INDEX, dtype=new_labels[0].dtype, device=new_labels[0].device)
        att = torch.zeros((bsz, max_len), dtype=attention_mask.dtype, device=attention_mask.device)
        pos = torch.zeros((bsz, max_len), dtype=position_ids.dtype, device=position_ids.device)
        for i, (e, l) in enumerate(zip(new_embeds, new_labels)):
            n = e.shape[0]
            if n == max_len:            # nothing to pad (always so for one sample): no 2 x 650-MB copy of a T = 512 stream
                emb_pad.append(e)
            else:
                z = torch.zeros((max_len - n, e.shape[1]), dtype=e.dtype, device=e.device)
                emb_pad.append(torch.cat((z, e), 0) if left else torch.cat((e, z), 0))
            if n > 0:
                sl = slice(max_len - n, max_len) if left else slice(0, n)
                lab_pad[i, sl] = l
                att[i, sl] = True
                pos[i, sl] = torch.arange(0, n, dtype=pos.dtype, device=pos.device)
        new_input_embeds = emb_pad[0].unsqueeze(0) if bsz == 1 else torch.stack(emb_pad, 0)
        new_labels_out = None if _labels is None else lab_pad
        att_out = None if _attention_mask is None else att.to(dtype=_attention_mask.dtype)
        pos_out = None if _position_ids is None else pos
        return (None, pos_out, att_out, past_key_values, new_input_embeds, new_labels_out, None, None, final_size, None)

    def initialize_vision_tokenizer(self, model_args, tokenizer):
        raise NotImplementedError("tokenizer surgery for training (tdc/cambrian_arch.py:1846-1898) is out of scope")
