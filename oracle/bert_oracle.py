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


def bert_tower(sd, prefix, input_ids, input_mask, n_layers, n_heads, eps=1e-12, return_hidden=False):
    """Pooled output [B, H] of one BertModel tower; sd maps HF key names to arrays."""
    ids = np.asarray(input_ids)
    mask = np.asarray(input_mask).astype(bool)
    B, S = ids.shape
    e = prefix + ".embeddings."
    x = (_f32(sd[e + "word_embeddings.weight"])[ids] + _f32(sd[e + "token_type_embeddings.weight"])[0]
         + _f32(sd[e + "position_embeddings.weight"])[:S][None])
    h = layer_norm(x, _f32(sd[e + "LayerNorm.weight"]), _f32(sd[e + "LayerNorm.bias"]), eps)
    H = h.shape[-1]
    dh = H // n_heads
    add_mask = np.where(mask, np.float32(0), np.finfo(np.float32).min).astype(np.float32)[:, None, None, :]
    hidden = [h]
    for i in range(n_layers):
        p = f"{prefix}.encoder.layer.{i}."

        def heads(name):
            y = linear(h, sd[p + f"attention.self.{name}.weight"], sd[p + f"attention.self.{name}.bias"])
            return y.reshape(B, S, n_heads, dh).transpose(0, 2, 1, 3)

        q, k, v = heads("query"), heads("key"), heads("value")
        scores = (q @ k.transpose(0, 1, 3, 2)) * np.float32(1.0 / math.sqrt(dh)) + add_mask
        scores = scores - scores.max(axis=-1, keepdims=True)
        probs = np.exp(scores, dtype=np.float32)
        probs /= probs.sum(axis=-1, keepdims=True, dtype=np.float32)
        ctx = (probs @ v).transpose(0, 2, 1, 3).reshape(B, S, H)
        a = linear(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        h1 = layer_norm(a + h, _f32(sd[p + "attention.output.LayerNorm.weight"]),
                        _f32(sd[p + "attention.output.LayerNorm.bias"]), eps)
        f = gelu_erf(linear(h1, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        o = linear(f, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        h = layer_norm(o + h1, _f32(sd[p + "output.LayerNorm.weight"]), _f32(sd[p + "output.LayerNorm.bias"]), eps)
        hidden.append(h)
    pooled = np.tanh(linear(h[:, 0], sd[prefix + ".pooler.dense.weight"], sd[prefix + ".pooler.dense.bias"]))
    if return_hidden:
        return pooled.astype(np.float32), hidden
    return pooled.astype(np.float32)


def get_embed(sd, input_ids, input_mask, is_query_embed, n_layers, n_heads, eps=1e-12):
    """{'embed': [B,128]} of retriever.py:33-43, as a bare array."""
    tower, proj = ("bert_q", "proj_q") if is_query_embed else ("bert_c", "proj_c")
    pooled = bert_tower(sd, tower, input_ids, input_mask, n_layers, n_heads, eps)
    return linear(pooled, sd[proj + ".weight"], sd[proj + ".bias"])
