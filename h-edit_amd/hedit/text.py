"""Tokenizer / text-encoder stand-ins for offline runs.

The reference takes both from the SD pipeline (transformers CLIPTokenizer / CLIPTextModel,
text-guided/inversion/inversion_utils.py:25-33) and calls them twice per image -- they are not on
the accelerated path and no vocabulary / weights exist offline.  These classes
give the same call surface with synthetic weights; real transformers objects can be passed to
HEditPipeline instead.
"""
import math
import types

import torch
import torch.nn as nn


class WordTokenizer:
    """Whitespace word-level tokenizer with the CLIPTokenizer surface used by the path
    (``__call__(...).input_ids``, ``encode``, ``decode``, ``model_max_length``)."""
    model_max_length = 77
    bos_token_id, eos_token_id = 49406, 49407

    def __init__(self, vocab_size=49408, stable_ids=False):
        self.vocab_size = vocab_size
        self.stable_ids = stable_ids
        self._ids = {}
        self._words = {}

    @staticmethod
    def _hash_id(w, attempt, n_ids):
        import hashlib
        key = w if attempt == 0 else f"{w}\x00{attempt}"
        return 1 + int.from_bytes(hashlib.sha256(key.encode("utf-8")).digest()[:8], "big") % n_ids

    def _id(self, w):
        # default: ids in order of first sight (what the committed golden vectors were generated with).  stable_ids: a
        # word's id is a function of the word ALONE (64 bits of SHA-256 into the id range), so a synthetic-weights run
        # gives a prompt the same embedding whether its image is edited alone or inside a lock-step batch (the drivers'
        # --random_init pipelines).  The id range has ~49 k slots, so a real prompt set (a few hundred distinct words)
        # does see collisions: the later word is then re-hashed with a counter until it finds a free id -- deterministic
        # for a given order of first sight, and reported once per word, because for THAT word the id now does depend
        # on what was tokenised before it.  `prescan` assigns the ids of a known vocabulary in sorted order up front,
        # which removes that dependence altogether (the drivers call it with every prompt of the run).
        if w not in self._ids:
            if len(self._ids) >= self.bos_token_id - 1:
                raise RuntimeError("WordTokenizer vocabulary exhausted")
            i = 1 + len(self._ids)
            if self.stable_ids:
                n_ids = self.bos_token_id - 1
                attempt = 0
                i = self._hash_id(w, 0, n_ids)
                while i in self._words:
                    attempt += 1
                    i = self._hash_id(w, attempt, n_ids)
                if attempt:
                    import warnings
                    warnings.warn(f"WordTokenizer(stable_ids): {w!r} collides with {self._words[self._hash_id(w, 0, n_ids)]!r}; "
                                  f"re-hashed to id {i} (attempt {attempt}) -- for this word the id depends on the order of "
                                  "first sight; call prescan() with the run's prompts to make it order-independent", stacklevel=3)
            self._ids[w] = i
            self._words[i] = w
        return self._ids[w]

    def prescan(self, prompts):
        """Assign the ids of every word of `prompts` now, in sorted word order: with stable_ids a collision is then
        resolved the same way whatever order the prompts are tokenised in later (batch invariance of a whole run)."""
        words = sorted({w for p in prompts for w in p.split(" ") if w != ""})
        for w in words:
            self._id(w)
        return len(words)

    def encode(self, text):
        return [self.bos_token_id] + [self._id(w) for w in text.split(" ") if w != ""] + [self.eos_token_id]

    def decode(self, ids):
        sp = {self.bos_token_id: "<|startoftext|>", self.eos_token_id: "<|endoftext|>"}
        return "".join(sp.get(int(i), self._words.get(int(i), "?")) for i in ids)

    def __call__(self, prompts, padding="max_length", max_length=None, truncation=True, return_tensors="pt"):
        if isinstance(prompts, str):
            prompts = [prompts]
        max_length = max_length or self.model_max_length
        rows = []
        for p in prompts:
            ids = self.encode(p)[:max_length]
            rows.append(ids + [self.eos_token_id] * (max_length - len(ids)))
        return types.SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.int64))


def prescan_prompts(tokenizer, records):
    """Give a stand-in tokenizer the whole vocabulary of a run before any work starts (a no-op for real tokenizers,
    which have no `prescan`).  `records`: dataset entries with ``original_prompt`` / ``editing_prompt`` as the drivers
    read them (the demo file calls them ``source_prompt`` / ``target_prompt``; square brackets around the edited words are stripped there, so they are stripped here).  Every rank
    of a sharded run passes the FULL dataset, so the ids do not depend on the shard either."""
    if not hasattr(tokenizer, "prescan"):
        return 0
    prompts = []
    for r in records:
        for k in ("original_prompt", "editing_prompt", "source_prompt", "target_prompt"):
            if isinstance(r, dict) and r.get(k):
                prompts.append(r[k].replace("[", "").replace("]", ""))
    return tokenizer.prescan(prompts)


class ClipTextEncoder(nn.Module):
    """CLIP-text-shaped transformer (pre-LN, causal, quick-GELU) with seeded random weights.
    forward(ids) -> (last_hidden_state,) like transformers' CLIPTextModel."""

    def __init__(self, dim=768, layers=12, heads=12, vocab=49408, max_len=77, seed=7):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.dim, self.heads = dim, heads
        self.tok = nn.Parameter(torch.randn(vocab, dim, generator=g) * 0.02)
        self.pos = nn.Parameter(torch.randn(max_len, dim, generator=g) * 0.01)
        blk = []
        for _ in range(layers):
            blk.append(nn.ParameterDict({
                "ln1_w": nn.Parameter(torch.ones(dim)), "ln1_b": nn.Parameter(torch.zeros(dim)),
                "qkv": nn.Parameter(torch.randn(3 * dim, dim, generator=g) / math.sqrt(dim)),
                "qkv_b": nn.Parameter(torch.zeros(3 * dim)),
                "out": nn.Parameter(torch.randn(dim, dim, generator=g) / math.sqrt(dim)),
                "out_b": nn.Parameter(torch.zeros(dim)),
                "ln2_w": nn.Parameter(torch.ones(dim)), "ln2_b": nn.Parameter(torch.zeros(dim)),
                "fc1": nn.Parameter(torch.randn(4 * dim, dim, generator=g) / math.sqrt(dim)),
                "fc1_b": nn.Parameter(torch.zeros(4 * dim)),
                "fc2": nn.Parameter(torch.randn(dim, 4 * dim, generator=g) / math.sqrt(4 * dim)),
                "fc2_b": nn.Parameter(torch.zeros(dim)),
            }))
        self.blocks = nn.ModuleList(blk)
        self.lnf_w = nn.Parameter(torch.ones(dim))
        self.lnf_b = nn.Parameter(torch.zeros(dim))
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, ids):
        F = torch.nn.functional
        b, n = ids.shape
        h = self.tok[ids] + self.pos[:n][None]
        mask = torch.full((n, n), float("-inf"), device=h.device).triu(1)
        hd = self.dim // self.heads
        for p in self.blocks:
            x = F.layer_norm(h, (self.dim,), p["ln1_w"], p["ln1_b"])
            q, k, v = F.linear(x, p["qkv"], p["qkv_b"]).chunk(3, dim=-1)
            q, k, v = (t.reshape(b, n, self.heads, hd).transpose(1, 2) for t in (q, k, v))
            a = (q @ k.transpose(-1, -2)) * hd ** -0.5 + mask
            o = (a.softmax(-1) @ v).transpose(1, 2).reshape(b, n, self.dim)
            h = h + F.linear(o, p["out"], p["out_b"])
            x = F.layer_norm(h, (self.dim,), p["ln2_w"], p["ln2_b"])
            x = F.linear(x, p["fc1"], p["fc1_b"])
            h = h + F.linear(x * torch.sigmoid(1.702 * x), p["fc2"], p["fc2_b"])
        return (F.layer_norm(h, (self.dim,), self.lnf_w, self.lnf_b),)
