"""NumPy fp32 oracle of BertForRetriever.get_embed — TEST INFRASTRUCTURE ONLY.

Restates /root/reference/retrieval/retriever.py:33-43:

    cls = self.bert_{q|c}(input_ids, attention_mask)[1]     # pooled [CLS]
    embed = self.proj_{q|c}(cls)                             # Linear(hidden, 128)

The BertModel arithmetic lives in `transformers` (pinned 2.5.1 in requirements.txt:12; 5.15.0 is
installed here — same architecture).  Published algorithm restated, with the modeling_bert.py
line ranges of the installed copy:
  embeddings  :68-108   LN(word[ids] + type[0] + pos[0..S)), eps 1e-12
  attention   :111-136, :164-202   softmax(QK^T / sqrt(64) + (1-mask) * finfo.min) V
  self-output :289-292  LN(dense(ctx) + residual)
  FFN         :334-350  LN(dense(gelu_erf(dense(h))) + residual)
  pooler      :457-463  tanh(dense(h[:, 0]))

Pinned by tests/golden/encoder_golden.npz: outputs of the reference's own BertForRetriever
(imported from /root/reference in the build container, CPU fp32) on a small random model; see
tests/golden/make_golden.py and tests/test_oracle_bert.py.
"""
import math

import numpy as np

try:  # vectorised erf
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf, otypes=[np.float32])


def _f32(x):
    return np.asarray(x, dtype=np.float32)


def layer_norm(x, g, b, eps):
    mu = x.mean(axis=-1, keepdims=True, dtype=np.float32)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True, dtype=np.float32)
    return ((x - mu) / np.sqrt(var + np.float32(eps)) * g + b).astype(np.float32)


def linear(x, w, b):
    # one [tokens, in] x [in, out] sgemm (a batched matmul over the leading axes threads poorly)
    x = _f32(x)
    y = x.reshape(-1, x.shape[-1]) @ _f32(w).T + _f32(b)
    return y.reshape(x.shape[:-1] + (y.shape[-1],)).astype(np.float32, copy=False)


def gelu_erf(x):
    return (0.5 * x * (1.0 + _erf(x / np.float32(math.sqrt(2.0))))).astype(np.float32)


def _round_f16(x):
    """float32 -> nearest fp16 -> float32: what storing a tensor in fp16 does to it"""
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def _note(probe, key, x):
    if probe is not None:
        probe[key] = max(probe.get(key, 0.0), float(np.abs(x).max()))


def bert_tower(sd, prefix, input_ids, input_mask, n_layers, n_heads, eps=1e-12, return_hidden=False, storage="fp32", probe=None):
    """Pooled output [B, H] of one BertModel tower; sd maps HF key names to arrays.

    storage="fp16": the same float32 arithmetic, but every tensor a mixed-precision run (apex O1: get_embed.py:122-129)
    keeps in fp16 between two operations is rounded to fp16 there -- LayerNorm outputs, the Q / K / V projections, the
    softmax weights entering the second attention product, the context rows, every dense layer's output, the GELU output.
    Comparing a fp16 implementation with THIS variant separates the storage format's error (shared) from an
    implementation's own.  probe (a dict): receives the largest |pre-softmax logit|, |FFN activation| (after GELU) and
    |pre-LayerNorm dense output + residual| met on the way."""
    if storage not in ("fp32", "fp16"):
        raise ValueError(storage)
    r = _round_f16 if storage == "fp16" else (lambda t: t)
    ids = np.asarray(input_ids)
    mask = np.asarray(input_mask).astype(bool)
    B, S = ids.shape
    e = prefix + ".embeddings."
    x = (_f32(sd[e + "word_embeddings.weight"])[ids] + _f32(sd[e + "token_type_embeddings.weight"])[0]
         + _f32(sd[e + "position_embeddings.weight"])[:S][None])
    h = r(layer_norm(x, _f32(sd[e + "LayerNorm.weight"]), _f32(sd[e + "LayerNorm.bias"]), eps))
    H = h.shape[-1]
    dh = H // n_heads
    add_mask = np.where(mask, np.float32(0), np.finfo(np.float32).min).astype(np.float32)[:, None, None, :]
    hidden = [h]
    zero = np.float32(0)
    for i in range(n_layers):
        p = f"{prefix}.encoder.layer.{i}."

        def heads(name):
            # (fp16 storage: the projection is stored without its bias, the bias joins it in one more fp16 rounding)
            y = r(r(linear(h, sd[p + f"attention.self.{name}.weight"], zero)) + _f32(sd[p + f"attention.self.{name}.bias"]))
            return y.reshape(B, S, n_heads, dh).transpose(0, 2, 1, 3)

        q, k, v = heads("query"), heads("key"), heads("value")
        scores = (q @ k.transpose(0, 1, 3, 2)) * np.float32(1.0 / math.sqrt(dh))
        _note(probe, "max_abs_logit", np.where(mask[:, None, None, :], scores, zero))
        scores = scores + add_mask
        scores = scores - scores.max(axis=-1, keepdims=True)
        probs = np.exp(scores, dtype=np.float32)
        denom = probs.sum(axis=-1, keepdims=True, dtype=np.float32)
        # (fp16 storage: the un-normalised weights are the fp16 operand of the second product, the sum stays float32)
        ctx = r(((r(probs) @ v) / denom).transpose(0, 2, 1, 3).reshape(B, S, H))
        a = r(linear(ctx, sd[p + "attention.output.dense.weight"], zero)) + _f32(sd[p + "attention.output.dense.bias"])
        _note(probe, "max_abs_pre_layernorm", a + h)
        h1 = r(layer_norm(a + h, _f32(sd[p + "attention.output.LayerNorm.weight"]),
                          _f32(sd[p + "attention.output.LayerNorm.bias"]), eps))
        f = r(gelu_erf(linear(h1, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"])))
        _note(probe, "max_abs_ffn_activation", f)
        o = r(linear(f, sd[p + "output.dense.weight"], zero)) + _f32(sd[p + "output.dense.bias"])
        _note(probe, "max_abs_pre_layernorm", o + h1)
        h = r(layer_norm(o + h1, _f32(sd[p + "output.LayerNorm.weight"]), _f32(sd[p + "output.LayerNorm.bias"]), eps))
        hidden.append(h)
    pooled = np.tanh(linear(h[:, 0], sd[prefix + ".pooler.dense.weight"], sd[prefix + ".pooler.dense.bias"]))
    if return_hidden:
        return pooled.astype(np.float32), hidden
    return pooled.astype(np.float32)


def get_embed(sd, input_ids, input_mask, is_query_embed, n_layers, n_heads, eps=1e-12, storage="fp32", probe=None):
    """{'embed': [B,128]} of retriever.py:33-43, as a bare array (storage / probe: see bert_tower)."""
    tower, proj = ("bert_q", "proj_q") if is_query_embed else ("bert_c", "proj_c")
    pooled = bert_tower(sd, tower, input_ids, input_mask, n_layers, n_heads, eps, storage=storage, probe=probe)
    out = linear(pooled, sd[proj + ".weight"], sd[proj + ".bias"])
    return _round_f16(out) if storage == "fp16" else out
