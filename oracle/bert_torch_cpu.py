"""torch-CPU fp32 restatement of BertForRetriever.get_embed — TEST INFRASTRUCTURE ONLY.

The same arithmetic as oracle/bert_oracle.py (the pinned NumPy oracle; see its header for the
reference line ranges: /root/reference/retrieval/retriever.py:33-43 + transformers BertModel), on
torch CPU tensors so that the element-wise work is threaded like the reference's own CPU execution
(the reference IS torch code).  Used by bench.py as the timed encoder `cpu_baseline` (SURVEY 8d:
"the build's fp32 torch restatement, B=32, S=128, all cores") and checked against the golden
fixtures / the NumPy oracle in tests/test_oracle_bert.py.  Never imported by the product.
"""
import math

import torch
import torch.nn.functional as F


@torch.no_grad()
def get_embed(sd, input_ids, input_mask, is_query_embed, n_layers, n_heads, eps=1e-12):
    """sd: HF-key -> CPU float32 tensor.  Returns the [B,128] embedding (CPU float32 tensor)."""
    tower, proj = ("bert_q", "proj_q") if is_query_embed else ("bert_c", "proj_c")
    ids = torch.as_tensor(input_ids, dtype=torch.int64)
    mask = torch.as_tensor(input_mask, dtype=torch.bool)
    B, S = ids.shape
    e = tower + ".embeddings."
    x = sd[e + "word_embeddings.weight"][ids] + sd[e + "token_type_embeddings.weight"][0] \
        + sd[e + "position_embeddings.weight"][:S][None]
    H = x.shape[-1]
    h = F.layer_norm(x, (H,), sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], eps)
    dh = H // n_heads
    add_mask = torch.where(mask, 0.0, torch.finfo(torch.float32).min)[:, None, None, :]
    for i in range(n_layers):
        p = f"{tower}.encoder.layer.{i}."

        def heads(name):
            y = F.linear(h, sd[p + f"attention.self.{name}.weight"], sd[p + f"attention.self.{name}.bias"])
            return y.view(B, S, n_heads, dh).transpose(1, 2)

        q, k, v = heads("query"), heads("key"), heads("value")
        probs = torch.softmax(q @ k.transpose(-1, -2) * (1.0 / math.sqrt(dh)) + add_mask, dim=-1)
        ctx = (probs @ v).transpose(1, 2).reshape(B, S, H)
        a = F.linear(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        h1 = F.layer_norm(a + h, (H,), sd[p + "attention.output.LayerNorm.weight"],
                          sd[p + "attention.output.LayerNorm.bias"], eps)
        f = F.gelu(F.linear(h1, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))  # erf form
        o = F.linear(f, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        h = F.layer_norm(o + h1, (H,), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    pooled = torch.tanh(F.linear(h[:, 0], sd[tower + ".pooler.dense.weight"], sd[tower + ".pooler.dense.bias"]))
    return F.linear(pooled, sd[proj + ".weight"], sd[proj + ".bias"])
