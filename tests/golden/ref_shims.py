"""Import harness for the *reference* (read-only, /root/reference) used ONLY to generate golden fixtures.

This file never travels the reference anywhere: it imports the reference's own Python modules inside the build
container (where /root/reference is mounted), with the minimum set of shims needed because the container has
transformers 5.x (reference pins 4.46.0, requirements.txt:167) and lacks IPython/torchaudio/hub access.
Shim list follows SURVEY.md section 8(c).  Nothing here is imported by the product path, the tests or bench.py:
only `make_golden.py` (run by hand in the build container) uses it.
"""
import sys
import types

REF_ROOT = "/root/reference"


class FakeTokenizerOutput:
    def __init__(self, ids):
        import torch
        self.input_ids = torch.tensor([ids], dtype=torch.long)

    def to(self, device):
        return self


class FakeBertTokenizer:
    """Stands in for BertTokenizer.from_pretrained (no vocab on disk).  Deterministic ids from the prompt."""

    def __init__(self, vocab=30522):
        self.vocab = vocab

    def __len__(self):
        return self.vocab

    def __call__(self, prompt, padding=None, truncation=None, max_length=256, return_tensors=None):
        return FakeTokenizerOutput(prompt_to_ids(prompt, self.vocab, max_length))


def prompt_to_ids(prompt, vocab=30522, max_length=256):
    """Deterministic stand-in tokenisation: [CLS]=101, one id per whitespace word, [SEP]=102, truncated."""
    cls, sep, lo = (101, 102, 1000) if vocab > 2000 else (vocab - 2, vocab - 1, 3)
    ids = [cls]
    for w in prompt.split():
        h = 0
        for ch in w.lower():
            h = (h * 131 + ord(ch)) % 1000003
        ids.append(lo + h % (vocab - lo - 2))
    ids = ids[: max_length - 1] + [sep]
    return ids


def install(bert_cfg_kwargs=None, dino_cfg=None):
    """Install shims and return the imported reference modules (arch, Qformer, vision_sampler)."""
    import torch
    import transformers

    # 1. bare `tdc` package so tdc/__init__.py (which imports the HF LLM subclasses) is bypassed
    if "tdc" not in sys.modules:
        pkg = types.ModuleType("tdc")
        pkg.__path__ = [REF_ROOT + "/tdc"]
        sys.modules["tdc"] = pkg
    # 2. names removed from transformers.modeling_utils in 5.x (Qformer.py:39-44)
    import transformers.modeling_utils as mu
    from transformers import pytorch_utils as pu
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), None)
    # 3. stub modules
    ipy = types.ModuleType("IPython")
    ipy.embed = lambda *a, **k: None
    sys.modules.setdefault("IPython", ipy)
    ta = types.ModuleType("torchaudio")
    tac = types.ModuleType("torchaudio.compliance")
    tak = types.ModuleType("torchaudio.compliance.kaldi")
    ta.compliance = tac
    tac.kaldi = tak
    sys.modules.setdefault("torchaudio", ta)
    sys.modules.setdefault("torchaudio.compliance", tac)
    sys.modules.setdefault("torchaudio.compliance.kaldi", tak)

    # 5. no hub: tokenizer / configs
    from transformers import BertTokenizer, BertConfig, Dinov2Config
    BertTokenizer.from_pretrained = classmethod(lambda cls, *a, **k: FakeBertTokenizer())
    _bk = dict(bert_cfg_kwargs or {})
    BertConfig.from_pretrained = classmethod(lambda cls, *a, **k: BertConfig(**_bk))
    if dino_cfg is not None:
        Dinov2Config.from_pretrained = classmethod(lambda cls, *a, **k: dino_cfg)

    import importlib
    Q = importlib.import_module("tdc.Qformer")
    # 4. transformers-5 API drift
    Q.BertPreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    Q.BertModel.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n
    Q.BertLMHeadModel.resize_token_embeddings = lambda self, *a, **k: None
    arch = importlib.import_module("tdc.cambrian_arch")
    vs = importlib.import_module("tdc.vision_sampler")
    return arch, Q, vs
