"""Tiny deterministic stand-ins for the HF models / processors the hot path sits behind, so that the
drop-in entry points (feature caches, steering, attribution patching, the launch scripts) can be
EXECUTED in tests -- here against the HIP path, and in tests/golden/make_golden.py against the
reference itself -- without a checkpoint or network.  Weights come from synth.normalish (counter-based,
bit-identical on both sides).  Shapes follow what the reference code touches:

  TinyLlava            .language_model.get_submodule("layers.N"), .vision_tower, .device, .dtype,
                       forward(input_ids, pixel_values=, image_sizes=, attention_mask=) -> {"logits": ...},
                       generate(**inputs, max_new_tokens=) (greedy, uses a one-token forward per step)
  FakeProcessor        __call__(text=, images=, return_tensors="pt") -> input_ids (BOS first),
                       pixel_values, image_sizes, attention_mask; apply_chat_template; batch_decode
  FakeImage            .convert("RGB"), .size
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

import synth

BOS = 1


class _Block(nn.Module):
    """x + tanh(x W) computed in f32, handed on in fp16 (what an fp16 LLM layer hands to the hook).
    Returns a TUPLE like an HF decoder layer (the reference's attribution hook only handles tuples:
    patching/utils.py:35-38 does list(outputs) in both branches)."""

    def __init__(self, d: int, seed: int):
        super().__init__()
        w = synth.normalish(seed, d * d).reshape(d, d) / np.float32(np.sqrt(d))
        self.w = nn.Parameter(torch.from_numpy(w))

    def forward(self, x):
        h = x.float()
        # causal mixing (a stand-in for attention): every position also sees the mean of its prefix,
        # so a metric at the last position has gradients at all positions
        steps = torch.arange(1, h.shape[1] + 1, device=h.device, dtype=h.dtype)[None, :, None]
        h = h + 0.5 * torch.cumsum(h, dim=1) / steps
        return ((h + torch.tanh(h @ self.w)).to(torch.float16),)


class _LanguageModel(nn.Module):
    def __init__(self, vocab: int, d: int, n_layers: int, seed: int):
        super().__init__()
        emb = synth.normalish(seed, vocab * d).reshape(vocab, d)
        emb[:, 3] *= np.float32(6.0)                       # one massive dim, like a residual stream
        self.embed = nn.Embedding.from_pretrained(torch.from_numpy(emb), freeze=True)
        self.layers = nn.ModuleList([_Block(d, seed + 1 + i) for i in range(n_layers)])
        head = synth.normalish(seed + 50, d * vocab).reshape(d, vocab) / np.float32(np.sqrt(d))
        self.head = nn.Parameter(torch.from_numpy(head))

    @property
    def device(self):
        return self.head.device

    def forward(self, input_ids, inputs_embeds=None):
        h = self.embed(input_ids).to(torch.float16) if inputs_embeds is None else inputs_embeds
        for layer in self.layers:
            h = layer(h)[0]
        return {"logits": h.float() @ self.head, "last_hidden_state": h}


class TinyLlava(nn.Module):
    """Looks like LlavaNextForConditionalGeneration to the code under test (it has `.language_model`
    and `.vision_tower`); an image adds a per-image offset to the embeddings of its sequence."""

    def __init__(self, vocab: int = 40, d: int = 64, n_layers: int = 2, seed: int = 300, max_gen: int = 8):
        super().__init__()
        self.language_model = _LanguageModel(vocab, d, n_layers, seed)
        self.vision_tower = nn.Identity()
        self.d, self.max_gen = d, max_gen

    @property
    def device(self):
        return self.language_model.device

    @property
    def dtype(self):
        return torch.float16

    def forward(self, input_ids=None, pixel_values=None, image_sizes=None, attention_mask=None, **_):
        emb = self.language_model.embed(input_ids)
        if pixel_values is not None:
            emb = emb + pixel_values.float().reshape(pixel_values.shape[0], -1).mean(1)[:, None, None]
        return self.language_model(input_ids, inputs_embeds=emb.to(torch.float16))

    @torch.no_grad()
    def generate(self, input_ids=None, max_new_tokens: int = 4, **kw):
        """Greedy decoding; like HF generate with a KV cache, each new token is ONE single-token forward
        (S = 1), which is what the steering hook distinguishes (features/steering.py:111)."""
        max_new_tokens = min(max_new_tokens, self.max_gen)   # the callers hard-code 512 (steering.py:86)
        kw.pop("synced_gpus", None)                          # (HF generate's lockstep flag: greedy fakes are in lockstep)
        out = input_ids
        nxt = self.forward(input_ids=out, **kw)["logits"][:, -1].argmax(-1, keepdim=True)
        for _ in range(max_new_tokens):
            out = torch.cat([out, nxt], dim=1)
            nxt = self.forward(input_ids=nxt)["logits"][:, -1].argmax(-1, keepdim=True)
        return out


class FakeImage:
    def __init__(self, idx: int, size=(8, 6)):
        self.idx, self.size = idx, size

    def convert(self, mode):
        return self


class FakeProcessor:
    """`<image>` prompt -> BOS + 4 ids derived from the image; 3x4x4 "pixels"."""

    def __init__(self, vocab: int = 40):
        self.vocab = vocab
        self.image_processor = self

    def __call__(self, *args, text=None, images=None, return_tensors="pt", **_):
        if args and images is None and not isinstance(args[0], str):
            images = args[0]                                  # image_processor(images, do_pad=True, ...)
        if images is None:
            images = [None] * (len(text) if isinstance(text, list) else 1)
        if not isinstance(images, (list, tuple)):
            images = [images]
        ids, pix, sizes = [], [], []
        for n, im in enumerate(images):
            i = getattr(im, "idx", n)
            ids.append([BOS] + [(7 * i + 3 * j + 2) % self.vocab for j in range(4)])
            pix.append(synth.normalish(900 + i, 48).reshape(3, 4, 4) * np.float32(0.1))
            sizes.append(list(im.size) if im is not None else [4, 4])
        out = {"input_ids": torch.tensor(ids, dtype=torch.long),
               "pixel_values": torch.from_numpy(np.stack(pix)),
               "image_sizes": torch.tensor(sizes, dtype=torch.long)}
        out["attention_mask"] = torch.ones_like(out["input_ids"])
        return _Batch(out)

    def apply_chat_template(self, conversation, add_generation_prompt=True):
        return " ".join(c.get("text", "<image>") for m in conversation for c in m["content"])

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(str(int(t)) for t in row) for row in ids]


class _Batch(dict):
    def to(self, device):
        return _Batch({k: v.to(device) for k, v in self.items()})


class FakeTokenizer:
    """Whitespace tokenizer over a tiny vocabulary (ids 2..vocab-1 by word hash), BOS = 1, EOS = PAD = 0."""
    eos_token_id, pad_token_id, bos_token_id, eos_token = 0, 0, BOS, "</s>"
    model_max_length = 1 << 20

    def __init__(self, vocab: int = 40):
        self.vocab = vocab

    def convert_tokens_to_ids(self, token: str) -> int:
        return 2 + sum(ord(c) * (i + 1) for i, c in enumerate(token)) % (self.vocab - 2)

    def __call__(self, text, add_special_tokens=True, return_tensors=None, **_):
        single = isinstance(text, str)
        rows = [[self.convert_tokens_to_ids(w) for w in t.split()] for t in ([text] if single else text)]
        if add_special_tokens:
            rows = [[BOS] + r for r in rows]
        if return_tensors == "pt":
            return {"input_ids": torch.tensor(rows, dtype=torch.long)}
        return {"input_ids": rows[0] if single else rows}


class FakeSlowTokenizer(FakeTokenizer):
    """FakeTokenizer that also splits on the EOS *string* and implements `truncation` / `return_overflowing_tokens`
    the way transformers' slow (Python) tokenizers do: one flat `input_ids` (BOS + max_length - 1 ids) and the removed
    tail as a flat `overflowing_tokens` list -- the branch of the reference chunker that re-chunks the overflow
    (sae_auto_interp/sae/data.py:60-70)."""

    def _ids(self, text: str):
        out = []
        for j, seg in enumerate(text.split(self.eos_token)):
            if j:
                out.append(self.eos_token_id)
            out += [self.convert_tokens_to_ids(w) for w in seg.split()]
        return out

    def __call__(self, text, add_special_tokens=True, max_length=None, truncation=False,
                 return_overflowing_tokens=False, return_attention_mask=False, **_):
        if not isinstance(text, str) or not truncation:      # batched / plain calls: the whitespace tokenizer above
            return super().__call__(text, add_special_tokens=add_special_tokens, **_)
        ids = self._ids(text)
        n_special = 1 if add_special_tokens else 0
        out, overflow = ids, []
        if truncation and max_length is not None and len(ids) + n_special > max_length:
            out, overflow = ids[: max_length - n_special], ids[max_length - n_special:]
        import transformers

        enc = transformers.BatchEncoding({"input_ids": ([BOS] if add_special_tokens else []) + out})
        if return_overflowing_tokens:
            enc["overflowing_tokens"] = overflow
        return enc


def make_fast_tokenizer(n_words: int = 60):
    """A real `PreTrainedTokenizerFast` (Rust `tokenizers` WordLevel model, BOS template) built in memory: the fast
    branch of the reference chunker (`return_overflowing_tokens` gives one row per chunk, BOS re-added on each)."""
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast

    vocab = {"<s>": 0, "</s>": 1, "<unk>": 2}
    for i in range(n_words):
        vocab[f"w{i}"] = 3 + i
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split("</s>", "isolated"), pre_tokenizers.Whitespace()])
    tok.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 0)])
    return PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>",
                                   model_max_length=1 << 20)


def chunker_documents(n_docs: int = 2500, seed: int = 5):
    """Documents of 1-40 words w0..w59 (more than one 2048-document batch of the reference chunker)."""
    import random

    rng = random.Random(seed)
    return [" ".join(f"w{rng.randrange(60)}" for _ in range(rng.randint(1, 40))) for _ in range(n_docs)]


class FakeImageDataset:
    """What `load_dataset` hands the image cache: rows {"image": ...}, `.shard(n, i, contiguous=True)`."""

    def __init__(self, n: int, first: int = 0):
        self.items = [{"image": FakeImage(first + i)} for i in range(n)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def shard(self, num_shards, index, contiguous=True):
        per = (len(self.items) + num_shards - 1) // num_shards
        out = FakeImageDataset(0)
        out.items = self.items[index * per:(index + 1) * per]
        return out
