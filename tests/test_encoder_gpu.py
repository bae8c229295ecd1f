"""Encoder parity on the GPU: each HIP kernel against the NumPy oracle's arithmetic, and the whole
BertForRetriever.get_embed against outputs of the reference's own model (encoder_golden.npz).

Tolerances (stated per north_star "within the tolerance ... written in the test"): activations
and weights are fp16 with fp32 accumulation/statistics, the oracle is fp32.  One fp16 rounding
is 2^-11 relative (4.9e-4); per-kernel checks allow 2e-3 + 2e-3*|ref|; the end-to-end embedding
(|values| ~ 0.1-1 after tanh + projection) allows 1.5e-3 absolute on the small golden model (measured 3.9e-4) and
4e-3 absolute / cosine >= 0.9999 at bert-base depth (measured 1.3e-3): a regression that triples the error fails.
"""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import bert_oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def dev16(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=torch.float16)


def close(got, ref, rtol=2e-3, atol=2e-3):
    got = got.float().cpu().numpy() if torch.is_tensor(got) else got
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol)


@pytest.fixture(scope="module")
def lib(gpu_device):
    from proqa_amd import _lib
    return _lib.load()


def stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("hidden,S,B", [(128, 48, 5), (768, 128, 3), (768, 30, 2), (1024, 17, 2)])
def test_embed_layernorm(lib, gpu_device, hidden, S, B):
    from proqa_amd import _lib
    rng = np.random.default_rng(hidden + S)
    vocab = 1000
    word = rng.standard_normal((vocab, hidden)).astype(np.float16)
    pos = rng.standard_normal((S, hidden)).astype(np.float16)
    typ = rng.standard_normal((hidden,)).astype(np.float16)
    g = (1 + 0.1 * rng.standard_normal(hidden)).astype(np.float16)
    b = (0.1 * rng.standard_normal(hidden)).astype(np.float16)
    ids = rng.integers(0, vocab, (B, S))
    ref = bert_oracle.layer_norm(word.astype(np.float32)[ids] + pos.astype(np.float32)[None] + typ.astype(np.float32),
                                 g.astype(np.float32), b.astype(np.float32), 1e-12)
    t = {k: dev16(v, gpu_device) for k, v in dict(word=word, pos=pos, typ=typ, g=g, b=b).items()}
    tid = torch.from_numpy(ids).to(gpu_device)
    out = torch.empty((B * S, hidden), dtype=torch.float16, device=gpu_device)
    _lib.check(lib.proqa_embed_layernorm_f16(tid.data_ptr(), B * S, S, hidden, t["word"].data_ptr(), vocab,
                                             t["pos"].data_ptr(), t["typ"].data_ptr(), t["g"].data_ptr(),
                                             t["b"].data_ptr(), 1e-12, out.data_ptr(), stream()))
    close(out.reshape(B, S, hidden), ref)


@pytest.mark.parametrize("rows,cols", [(7, 128), (1000, 768), (5, 1024)])
def test_bias_residual_layernorm(lib, gpu_device, rows, cols):
    from proqa_amd import _lib
    rng = np.random.default_rng(rows)
    x = rng.standard_normal((rows, cols)).astype(np.float16)
    r = rng.standard_normal((rows, cols)).astype(np.float16)
    bias = rng.standard_normal(cols).astype(np.float16)
    g = (1 + 0.1 * rng.standard_normal(cols)).astype(np.float16)
    b = (0.1 * rng.standard_normal(cols)).astype(np.float16)
    f = lambda a: a.astype(np.float32)
    ref = bert_oracle.layer_norm(f(x) + f(bias) + f(r), f(g), f(b), 1e-12)
    tx, tr, tb, tg, tbb = (dev16(a, gpu_device) for a in (x, r, bias, g, b))
    out = torch.empty_like(tx)
    _lib.check(lib.proqa_bias_residual_layernorm_f16(tx.data_ptr(), tb.data_ptr(), tr.data_ptr(), tg.data_ptr(),
                                                     tbb.data_ptr(), 1e-12, rows, cols, out.data_ptr(), stream()))
    close(out, ref)


@pytest.mark.parametrize("rows,cols", [(3, 512), (4097, 3072)])
def test_bias_gelu(lib, gpu_device, rows, cols):
    from proqa_amd import _lib
    rng = np.random.default_rng(cols)
    x = (3 * rng.standard_normal((rows, cols))).astype(np.float16)
    bias = rng.standard_normal(cols).astype(np.float16)
    ref = bert_oracle.gelu_erf(x.astype(np.float32) + bias.astype(np.float32))
    tx, tb = dev16(x, gpu_device), dev16(bias, gpu_device)
    _lib.check(lib.proqa_bias_gelu_f16(tx.data_ptr(), tb.data_ptr(), rows, cols, stream()))
    close(tx, ref)


def attention_ref(qkv, lens, B, S, NH):
    H = NH * 64
    x = qkv.astype(np.float32).reshape(B, S, 3, NH, 64)
    q, k, v = (x[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    mask = np.arange(S)[None, :] < np.asarray(lens)[:, None]
    add = np.where(mask, 0.0, np.finfo(np.float32).min).astype(np.float32)[:, None, None, :]
    s = q @ k.transpose(0, 1, 3, 2) * np.float32(0.125) + add
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return (p @ v).transpose(0, 2, 1, 3).reshape(B, S, H)


@pytest.mark.parametrize("B,S,NH", [(3, 48, 2), (4, 128, 12), (2, 30, 12), (2, 512, 4), (3, 77, 3), (3, 300, 3), (2, 129, 2), (2, 640, 9)])
def test_attention_with_key_padding(lib, gpu_device, B, S, NH):
    from proqa_amd import _lib
    rng = np.random.default_rng(S * 3 + NH)
    qkv = rng.standard_normal((B * S, 3 * NH * 64)).astype(np.float16)
    lens = rng.integers(1, S + 1, B).astype(np.int32)
    lens[0] = S
    if B > 1:
        lens[1] = 1
    ref = attention_ref(qkv, lens, B, S, NH)
    tq = dev16(qkv, gpu_device)
    tl = torch.from_numpy(lens).to(gpu_device)
    out = torch.empty((B * S, NH * 64), dtype=torch.float16, device=gpu_device)
    _lib.check(lib.proqa_attention_f16(tq.data_ptr(), tl.data_ptr(), B, S, NH, out.data_ptr(), stream()))
    close(out.reshape(B, S, NH * 64), ref, rtol=3e-3, atol=3e-3)
    # softmax spike: one key dominates a row by a wide margin (forces the online-softmax rescale)
    qkv2 = qkv.copy().reshape(B, S, 3, NH, 64)
    qkv2[0, 0, 0, 0] = 6
    qkv2[0, S - 1, 1, 0] = 6          # last key tile carries the spike for query 0 / head 0
    qkv2 = qkv2.reshape(B * S, -1)
    ref2 = attention_ref(qkv2, lens, B, S, NH)
    tq2 = dev16(qkv2, gpu_device)
    _lib.check(lib.proqa_attention_f16(tq2.data_ptr(), tl.data_ptr(), B, S, NH, out.data_ptr(), stream()))
    close(out.reshape(B, S, NH * 64), ref2, rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize("B,S,H", [(5, 48, 128), (9, 128, 768), (70, 3, 768), (512, 128, 768)])
def test_pool_project(lib, gpu_device, B, S, H):
    from proqa_amd import _lib
    rng = np.random.default_rng(B)
    h = rng.standard_normal((B, S, H)).astype(np.float16)
    wp = (rng.standard_normal((H, H)) / np.sqrt(H)).astype(np.float16)
    bp = (0.1 * rng.standard_normal(H)).astype(np.float16)
    wj = (rng.standard_normal((128, H)) / np.sqrt(H)).astype(np.float16)
    bj = (0.1 * rng.standard_normal(128)).astype(np.float16)
    f = lambda a: a.astype(np.float32)
    ref = np.tanh(f(h[:, 0]) @ f(wp).T + f(bp)) @ f(wj).T + f(bj)
    t = [dev16(a, gpu_device) for a in (h, wp, bp, wj, bj)]
    for dtype, code in [(torch.float16, 0), (torch.float32, 1)]:
        out = torch.empty((B, 128), dtype=dtype, device=gpu_device)
        ws = torch.empty((B, H), dtype=torch.float16, device=gpu_device)
        _lib.check(lib.proqa_pool_project_f16(t[0].data_ptr(), B, S, H, t[1].data_ptr(), t[2].data_ptr(),
                                              t[3].data_ptr(), t[4].data_ptr(), ws.data_ptr(), out.data_ptr(), code,
                                              stream()))
        close(out, ref)


def load_golden():
    z = np.load(os.path.join(GOLDEN, "encoder_golden.npz"))
    sd = {k[3:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("w::")}
    with open(os.path.join(GOLDEN, "encoder_config.json")) as f:
        cfg = json.load(f)
    return z, sd, cfg


@pytest.mark.parametrize("B,S,NH", [(3, 48, 2), (5, 128, 12), (2, 512, 4), (3, 77, 3), (1, 1, 12)])
def test_attention_cls_rows(lib, gpu_device, B, S, NH):
    """The [CLS]-only attention of the last layer equals row 0 of the full attention."""
    from proqa_amd import _lib
    rng = np.random.default_rng(S * 5 + NH)
    qkv = rng.standard_normal((B * S, 3 * NH * 64)).astype(np.float16)
    lens = rng.integers(1, S + 1, B).astype(np.int32)
    lens[0] = S
    if B > 1:
        lens[1] = 1
    ref = attention_ref(qkv, lens, B, S, NH)[:, 0]
    tq = dev16(qkv, gpu_device)
    tl = torch.from_numpy(lens).to(gpu_device)
    out = torch.empty((B, NH * 64), dtype=torch.float16, device=gpu_device)
    _lib.check(lib.proqa_attention_cls_f16(tq.data_ptr(), tl.data_ptr(), B, S, NH, out.data_ptr(), stream()))
    close(out, ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("B,S,NH", [(3, 48, 2), (6, 128, 12), (2, 512, 4), (4, 77, 3), (5, 300, 3), (9, 200, 1)])
def test_attention_packed_layout(lib, gpu_device, B, S, NH):
    """Packed (varlen) token layout: same numbers as the padded evaluation, no padding rows."""
    from proqa_amd import _lib
    rng = np.random.default_rng(S * 7 + NH)
    lens = rng.integers(1, S + 1, B).astype(np.int32)
    lens[0] = S
    lens[-1] = 1
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    T = int(cu[-1])
    H = NH * 64
    qkv_packed = rng.standard_normal((T, 3 * H)).astype(np.float16)
    padded = np.zeros((B, S, 3 * H), np.float16)
    for b in range(B):
        padded[b, :lens[b]] = qkv_packed[cu[b]:cu[b + 1]]
    ref = attention_ref(padded.reshape(B * S, -1), lens, B, S, NH)
    tq = dev16(qkv_packed, gpu_device)
    tcu = torch.from_numpy(cu).to(gpu_device)
    out = torch.full((T + 1, H), 7.0, dtype=torch.float16, device=gpu_device)     # row T is a canary
    _lib.check(lib.proqa_attention_varlen_f16(tq.data_ptr(), tcu.data_ptr(), B, S, NH, out.data_ptr(), stream()))
    for b in range(B):
        close(out[cu[b]:cu[b + 1]], ref[b, :lens[b]], rtol=3e-3, atol=3e-3)
    assert (out[T] == 7).all()
    cls = torch.empty((B, H), dtype=torch.float16, device=gpu_device)
    _lib.check(lib.proqa_attention_cls_varlen_f16(tq.data_ptr(), tcu.data_ptr(), B, S, NH, cls.data_ptr(), stream()))
    close(cls, ref[:, 0], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("packed", [False, True])
def test_attention_with_projection_bias(lib, gpu_device, packed):
    """qkv_bias folded into the kernel (query bias added, key bias dropped, value bias on the output)
    equals attention on qkv + bias, for the full and the [CLS]-only kernels, both layouts."""
    from proqa_amd import _lib
    rng = np.random.default_rng(77)
    B, S, NH = 5, 296, 3                  # (more than one 128-key chunk and more than one 128-query workgroup per head)
    H = NH * 64
    lens = np.array([296, 1, 140, 77, 13], np.int32)
    bias = (0.5 * rng.standard_normal(3 * H)).astype(np.float16)
    padded = np.zeros((B, S, 3 * H), np.float16)
    for b in range(B):
        padded[b, :lens[b]] = rng.standard_normal((lens[b], 3 * H)).astype(np.float16)
    ref = attention_ref((padded.astype(np.float32) + bias.astype(np.float32)).astype(np.float16).reshape(B * S, -1),
                        lens, B, S, NH)
    tb = dev16(bias, gpu_device)
    if packed:
        cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        tq = dev16(np.concatenate([padded[b, :lens[b]] for b in range(B)]), gpu_device)
        tl, tcu, T = None, torch.from_numpy(cu).to(gpu_device), int(cu[-1])
    else:
        tq, tl, tcu, T = dev16(padded.reshape(B * S, -1), gpu_device), torch.from_numpy(lens).to(gpu_device), None, B * S
    out = torch.empty((T, H), dtype=torch.float16, device=gpu_device)
    cls = torch.empty((B, H), dtype=torch.float16, device=gpu_device)
    for cls_only, dst in ((0, out), (1, cls)):
        _lib.check(lib.proqa_attention_ex_f16(tq.data_ptr(), tb.data_ptr(), tl.data_ptr() if tl is not None else None,
                                              tcu.data_ptr() if tcu is not None else None, B, S, NH, cls_only,
                                              dst.data_ptr(), stream()))
    for b in range(B):
        rows = out[cu[b]:cu[b + 1]] if packed else out.view(B, S, H)[b, :lens[b]]
        close(rows, ref[b, :lens[b]], rtol=4e-3, atol=4e-3)
    close(cls, ref[:, 0], rtol=4e-3, atol=4e-3)


def test_embed_layernorm_packed_layout(lib, gpu_device):
    from proqa_amd import _lib
    rng = np.random.default_rng(4)
    B, S, H, V = 5, 40, 768, 300
    lens = np.array([40, 1, 17, 33, 8], np.int32)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    T = int(cu[-1])
    ids = rng.integers(0, V, (B, S)).astype(np.int64)
    word, pos, typ = (rng.standard_normal(sh).astype(np.float16) for sh in [(V, H), (S, H), (H,)])
    g, bt = rng.standard_normal(H).astype(np.float16), rng.standard_normal(H).astype(np.float16)
    args = [dev16(a, gpu_device) for a in (word, pos, typ, g, bt)]
    tid, tcu = torch.from_numpy(ids).to(gpu_device), torch.from_numpy(cu).to(gpu_device)
    padded = torch.empty((B * S, H), dtype=torch.float16, device=gpu_device)
    _lib.check(lib.proqa_embed_layernorm_f16(tid.data_ptr(), B * S, S, H, args[0].data_ptr(), V, args[1].data_ptr(),
                                             args[2].data_ptr(), args[3].data_ptr(), args[4].data_ptr(), 1e-12,
                                             padded.data_ptr(), stream()))
    packed = torch.full((T + 1, H), 7.0, dtype=torch.float16, device=gpu_device)
    _lib.check(lib.proqa_embed_layernorm_varlen_f16(tid.data_ptr(), tcu.data_ptr(), B, S, H, args[0].data_ptr(), V,
                                                    args[1].data_ptr(), args[2].data_ptr(), args[3].data_ptr(),
                                                    args[4].data_ptr(), 1e-12, packed.data_ptr(), stream()))
    padded = padded.view(B, S, H)
    for b in range(B):
        assert torch.equal(packed[cu[b]:cu[b + 1]], padded[b, :lens[b]])
    assert (packed[T] == 7).all()


def cosine(a, b):
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


# end-to-end embedding bounds (fp16 weights / activations against the fp32 oracle): ~3x what is measured
TOL_GOLDEN = 1.5e-3      # 2-layer golden model: measured 3.9e-4
TOL_BERT_BASE = 4e-3     # bert-base width, up to 12 layers: measured 1.3e-3
COS_MIN = 0.9999


def test_get_embed_matches_reference_outputs(gpu_device):
    """The reference call shape on the reference's own golden outputs (both towers)."""
    from proqa_amd.retriever import BertForRetriever
    z, sd, cfg = load_golden()
    model = BertForRetriever(cfg, device=gpu_device)
    model.load_state_dict({"module." [:0] + k: v for k, v in sd.items()})
    model.eval()
    batch = {"input_ids": torch.from_numpy(z["input_ids"]).to(gpu_device),
             "input_mask": torch.from_numpy(z["input_mask"]).to(gpu_device)}
    for is_q, key in [(True, "embed_q"), (False, "embed_c")]:
        out = model.get_embed(batch, is_q)["embed"]
        assert out.shape == (32, 128) and out.dtype == torch.float16 and out.is_cuda
        got = out.float().cpu().numpy()
        assert np.abs(got - z[key]).max() < TOL_GOLDEN
        assert cosine(got, z[key]).min() > COS_MIN
    # float() switches the emitted dtype like the reference's non-fp16 run
    out32 = model.float().get_embed(batch, False)["embed"]
    assert out32.dtype == torch.float32
    assert np.abs(out32.cpu().numpy() - z["embed_c"]).max() < TOL_GOLDEN
    # padding invariance: a row alone equals the same row inside a padded batch
    single = {"input_ids": batch["input_ids"][1:2, :3], "input_mask": batch["input_mask"][1:2, :3]}
    alone = model.get_embed(single, False)["embed"].cpu().numpy()
    assert np.abs(alone - z["embed_c_row1_unpadded"]).max() < TOL_GOLDEN
    assert np.abs(alone[0] - out32[1].cpu().numpy()).max() < 2e-3


def test_bert_base_shape_against_oracle(gpu_device):
    """bert-base geometry, random N(0,0.02) weights, variable lengths, vs the NumPy oracle."""
    from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
    sd = {k: v.half().float() for k, v in random_state_dict(BERT_BASE, seed=0).items()}
    model = BertForRetriever(BERT_BASE, device=gpu_device)
    model.load_state_dict(sd)
    rng = np.random.default_rng(0)
    B, S = 6, 128
    lens = np.array([128, 5, 64, 100, 33, 128])
    ids = np.zeros((B, S), np.int64)
    mask = np.zeros((B, S), bool)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(1000, 30522, n)
        ids[b, 0], ids[b, n - 1] = 101, 102
        mask[b, :n] = True
    batch = {"input_ids": torch.from_numpy(ids).to(gpu_device), "input_mask": torch.from_numpy(mask).to(gpu_device)}
    ref = bert_oracle.get_embed({k: v.numpy() for k, v in sd.items()}, ids, mask, False, 12, 12)
    got = {}
    # last layer on the [CLS] rows only / on every token  x  valid tokens packed / padded layout
    for cls_only in (True, False):
        for packed in (True, False):
            model.cls_only_last_layer, model.pack_tokens = cls_only, packed
            got[cls_only, packed] = model.get_embed(batch, False)["embed"].float().cpu().numpy()
            assert np.abs(got[cls_only, packed] - ref).max() < TOL_BERT_BASE
            assert cosine(got[cls_only, packed], ref).min() > COS_MIN
    for key, val in got.items():
        assert np.abs(val - got[True, True]).max() < 3e-3, key
    # lengths handed over from the host (predict()) instead of read back from the device
    model.cls_only_last_layer = model.pack_tokens = True
    alt = model.get_embed(batch, False, check_mask=False, seq_lens_host=lens.tolist())["embed"].float().cpu().numpy()
    np.testing.assert_array_equal(alt, got[True, True])
    with pytest.raises(ValueError, match="seq_lens_host"):
        model.get_embed(batch, False, seq_lens_host=[128] * B)


def trained_like_state_dict(seed=0):
    """bert-base-shaped weights with the statistics a TRAINED checkpoint has and N(0, 0.02) lacks (the authors' checkpoint,
    /root/reference/README.md:14-22, is a download): six outlier hidden channels (embedding / LayerNorm gains x 20-40, a large
    LayerNorm bias on two of them), query / key weights scaled until pre-softmax logits reach +-60, FFN weights scaled until
    intermediate activations reach several hundred and the fp16 GEMM outputs in front of the LayerNorms exceed 1e3."""
    from proqa_amd.retriever import random_state_dict, BERT_BASE
    sd = random_state_dict(BERT_BASE, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    outliers = torch.tensor([7, 130, 308, 381, 588, 701])
    gains = torch.tensor([20.0, 25.0, 30.0, 35.0, 40.0, 28.0])
    for t in ("bert_q", "bert_c"):
        for k in list(sd):
            if not k.startswith(t):
                continue
            if k.endswith("LayerNorm.weight"):
                sd[k][outliers] = sd[k][outliers] * gains * (1.0 + 0.1 * torch.randn(6, generator=g))
            elif k.endswith("LayerNorm.bias"):
                sd[k][outliers[:2]] += torch.tensor([6.0, -4.0])
            elif "attention.self.query.weight" in k or "attention.self.key.weight" in k:
                sd[k] *= 1.6
            elif "intermediate.dense.weight" in k:
                sd[k] *= 4.0
            elif re.search(r"layer\.\d+\.output\.dense\.weight", k):
                sd[k] *= 6.0
    return {k: v.half().float() for k, v in sd.items()}


def test_trained_model_statistics_do_not_break_fp16(gpu_device):
    """Encoder parity where fp16 storage could bite (every other encoder test runs N(0, 0.02) weights, activations O(1)):
    outlier channels, attention logits of +-60, FFN activations in the hundreds, pre-LayerNorm GEMM outputs beyond 1e3.
    A full 512 x 128 batch and a packed variable-length batch against oracle/bert_oracle.py (float32) on their first rows:
    cosine >= 0.9999 AND max relative error <= 1e-2 (an absolute tolerance says nothing at these magnitudes), all finite.
    Whose error is it?  bert_oracle storage="fp16" is the SAME float32 arithmetic with every stored activation rounded to
    fp16 (apex O1's storage format).  Measured over four weight seeds (scripts/dev_trained_stats_err.py, round 6): that
    oracle differs from the fp32 one by 1.4e-3 .. 4.8e-3, the kernels by 1.5e-3 .. 4.9e-3 -- the same size, seed by seed
    (4.17 / 4.14, 1.43 / 1.50-2.00, 3.04 / 2.21, 4.81 / 4.58-4.88 e-3) -- and the kernels differ from the fp16-storage
    oracle by 1.5e-3 .. 7.5e-3: two fp16 evaluations with different summation orders round differently and their errors
    add like independent noise, so a bound BELOW the format's own error against the fp16-storage oracle (the round-5 review
    proposed 2e-3) cannot hold for any implementation.  What is asserted instead: the kernels' error against fp32 is at most
    twice the format's own (+1e-3), i.e. it is the storage format's error, not the kernels'.  The probe asserts that the
    regime is reached inside the oracle: logits beyond +-60, GELU outputs beyond 200, pre-LayerNorm sums beyond 1e3.
    The relative error of this regime is a noisy quantity: 1.5e-3 .. 4.9e-3 over four weight seeds (scripts/dev_trained_stats_err.py;
    1.8e-3 .. 8.2e-3 with the all-keys-in-LDS attention kernel of rounds 1-4, worst cosine 0.99995), so the bound sits above
    the range, not on one seed's value.
    Tolerance: fp16 storage of activations costs 2^-11 relative per stored tensor; through 12 layers of residual + LayerNorm
    the measured error is 1-2e-3 of the output's largest entry -- the same as with toy weights, i.e. nothing is lost to
    range.  Mirrors /root/reference/retrieval/retriever.py:33-43 under apex O1 (get_embed.py:122-129: fp16 GEMMs with fp32
    accumulation, fp32 softmax / LayerNorm)."""
    from proqa_amd.retriever import BertForRetriever, BERT_BASE
    sd = trained_like_state_dict()
    model = BertForRetriever(BERT_BASE, device=gpu_device)
    model.load_state_dict(sd)
    sd_np = {k: v.numpy() for k, v in sd.items()}
    rng = np.random.default_rng(2)
    n_ref = 24
    # what the statistics do inside the oracle: the regime the docstring promises is really reached
    ids_probe = rng.integers(1000, 30522, (2, 128))
    ids_probe[:, 0], ids_probe[:, -1] = 101, 102
    probe = {}
    hidden = bert_oracle.bert_tower(sd_np, "bert_c", ids_probe, np.ones((2, 128), bool), 12, 12, return_hidden=True, probe=probe)[1]
    assert max(float(np.abs(h).max()) for h in hidden) > 20.0                       # outlier channels after LayerNorm
    assert probe["max_abs_logit"] > 60.0, probe                                      # pre-softmax attention logits
    assert probe["max_abs_ffn_activation"] > 200.0, probe                            # GELU outputs
    assert probe["max_abs_pre_layernorm"] > 1000.0, probe                            # dense output + residual entering a LayerNorm
    for B, S, lens in ((512, 128, None), (64, 128, rng.integers(9, 129, 64))):
        ids = rng.integers(1000, 30522, (B, S))
        mask = np.ones((B, S), bool)
        if lens is not None:
            for b, n in enumerate(lens):
                ids[b, n:] = 0
                mask[b, n:] = False
                ids[b, n - 1] = 102
        ids[:, 0] = 101
        if lens is None:
            ids[:, -1] = 102
        batch = {"input_ids": torch.from_numpy(ids).to(gpu_device), "input_mask": torch.from_numpy(mask).to(gpu_device)}
        ref = bert_oracle.get_embed(sd_np, ids[:n_ref], mask[:n_ref], False, 12, 12)
        # the same float32 arithmetic with every stored activation rounded to fp16 (apex O1's storage): the error the FORMAT
        # causes is shared with it, what is left against it is the kernels' own
        ref16 = bert_oracle.get_embed(sd_np, ids[:n_ref], mask[:n_ref], False, 12, 12, storage="fp16")
        fmt = np.abs(ref16 - ref).max() / np.abs(ref).max()
        for cls_only, packed in ((True, True), (False, False)):
            model.cls_only_last_layer, model.pack_tokens = cls_only, packed
            got = model.get_embed(batch, False)["embed"].float().cpu().numpy()
            assert np.isfinite(got).all()
            rel = np.abs(got[:n_ref] - ref).max() / np.abs(ref).max()
            rel16 = np.abs(got[:n_ref] - ref16).max() / np.abs(ref).max()
            print(f"trained statistics B={B} cls_only={cls_only} packed={packed}: vs fp32 oracle {rel:.2e}, vs fp16-storage oracle "
                  f"{rel16:.2e}, fp16-storage vs fp32 oracle {fmt:.2e}")
            assert cosine(got[:n_ref], ref).min() >= 0.9999, (B, cls_only, packed, cosine(got[:n_ref], ref).min())
            assert rel <= 1e-2, (B, cls_only, packed, rel)
            # the kernels' error is of the size of the FORMAT's: at most twice what fp16 storage alone does to the fp32 oracle
            assert rel <= 2.0 * fmt + 1e-3, (B, cls_only, packed, rel, fmt)
            # and two fp16 evaluations differ from each other by no more than the sum of their distances from fp32
            assert rel16 <= rel + fmt + 1e-3, (B, cls_only, packed, rel16, rel, fmt)


@pytest.mark.parametrize("lens", [[9], [30], [5, 17, 30, 12], [20] * 6 + [8], [33, 31, 32, 32]])
def test_question_sized_batches_against_oracle(gpu_device, lens):
    """A few short questions (<= 128 token rows): every dense layer runs on small_dense_mfma
    (encoder_kernels.hip) instead of the library GEMM; bert-base geometry, 2 layers, both towers' flags."""
    from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE, config_from_dict
    cfg = config_from_dict(dict(BERT_BASE, num_hidden_layers=2))
    sd = {k: v.half().float() for k, v in random_state_dict(cfg, seed=9).items()}
    model = BertForRetriever(cfg, device=gpu_device)
    model.load_state_dict(sd)
    rng = np.random.default_rng(len(lens))
    B, S = len(lens), max(lens)
    ids = np.zeros((B, S), np.int64)
    mask = np.zeros((B, S), bool)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(1000, 30522, n)
        mask[b, :n] = True
    batch = {"input_ids": torch.from_numpy(ids).to(gpu_device), "input_mask": torch.from_numpy(mask).to(gpu_device)}
    for is_query in (True, False):
        ref = bert_oracle.get_embed({k: v.numpy() for k, v in sd.items()}, ids, mask, is_query, 2, 12)
        for cls_only in (True, False):
            for packed in (True, False):
                model.cls_only_last_layer, model.pack_tokens = cls_only, packed
                got = model.get_embed(batch, is_query)["embed"].float().cpu().numpy()
                assert np.abs(got - ref).max() < TOL_BERT_BASE, (is_query, cls_only, packed)
                assert cosine(got, ref).min() > COS_MIN


def test_graphed_question_encoder_replays_the_plain_forward(gpu_device):
    """GraphedQuestionEncoder (online_retriever.py): one captured HIP graph per question length.  The replay is the plain
    call bit for bit -- for a second question of the same length, for other lengths, and after a larger batch through the
    same tower has replaced the encoder's workspace (the graphs are then captured again, not replayed onto freed memory)."""
    from proqa_amd.online_retriever import GraphedQuestionEncoder
    from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE, config_from_dict
    cfg = config_from_dict(dict(BERT_BASE, num_hidden_layers=3))
    model = BertForRetriever(cfg, device=gpu_device)
    model.load_state_dict(random_state_dict(cfg, seed=4))
    rng = np.random.default_rng(0)

    def plain(ids):
        t = torch.tensor([ids], dtype=torch.int64, device=gpu_device)
        return model.get_embed({"input_ids": t, "input_mask": torch.ones_like(t, dtype=torch.bool)}, True)["embed"].cpu().numpy()

    enc = GraphedQuestionEncoder(model)
    questions = [rng.integers(1000, 30522, n).tolist() for n in (16, 16, 7, 30, 16, 1, 7)]
    for ids in questions:
        got = enc(ids)
        assert got.shape == (1, 128)
        np.testing.assert_array_equal(got.cpu().numpy(), plain(ids))
    assert enc.captures == 4                                         # lengths 16, 7, 30, 1: captured once each
    big = torch.from_numpy(rng.integers(1000, 30522, (64, 128))).to(gpu_device)
    model.get_embed({"input_ids": big, "input_mask": torch.ones_like(big, dtype=torch.bool)}, True)   # replaces the workspace
    for ids in questions[:3]:
        np.testing.assert_array_equal(enc(ids).cpu().numpy(), plain(ids))
    assert enc.captures == 6                                         # 16 and 7 again
    # results of earlier calls are tensors of their own, not views of the graph's output
    a = enc(questions[0])
    b = enc(questions[1])
    assert not torch.equal(a, b)


def test_rejects_bad_inputs(gpu_device):
    from proqa_amd.retriever import BertForRetriever
    z, sd, cfg = load_golden()
    model = BertForRetriever(cfg, device=gpu_device)
    with pytest.raises(RuntimeError):
        model.get_embed({"input_ids": torch.zeros((1, 4), dtype=torch.long, device=gpu_device),
                         "input_mask": torch.ones((1, 4), dtype=torch.bool, device=gpu_device)}, True)
    with pytest.raises(RuntimeError):
        model.load_state_dict({k: v for k, v in list(sd.items())[:-1]})
    model.load_state_dict(sd)
    with pytest.raises(RuntimeError):      # CPU tensors: there is no CPU path
        model.get_embed({"input_ids": torch.zeros((1, 4), dtype=torch.long),
                         "input_mask": torch.ones((1, 4), dtype=torch.bool)}, True)
    holes = torch.tensor([[True, False, True, True]], device=gpu_device)
    with pytest.raises(ValueError):
        model.get_embed({"input_ids": torch.ones((1, 4), dtype=torch.long, device=gpu_device), "input_mask": holes}, True)
    with pytest.raises(ValueError):
        model.get_embed({"input_ids": torch.ones((1, 300), dtype=torch.long, device=gpu_device),
                         "input_mask": torch.ones((1, 300), dtype=torch.bool, device=gpu_device)}, True)
    empty = model.get_embed({"input_ids": torch.zeros((0, 4), dtype=torch.long, device=gpu_device),
                             "input_mask": torch.zeros((0, 4), dtype=torch.bool, device=gpu_device)}, True)["embed"]
    assert empty.shape == (0, 128)


def test_ragged_full_length_batch(gpu_device):
    """A last batch whose B*S is no multiple of the GEMM row tile, all sequences at full length (padded
    layout): the [CLS] gather must skip the tile-padding rows of the workspace."""
    from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE, config_from_dict
    cfg = config_from_dict(dict(BERT_BASE, num_hidden_layers=1))
    sd = {k: v.half().float() for k, v in random_state_dict(cfg, seed=2).items()}
    model = BertForRetriever(cfg, device=gpu_device)
    model.load_state_dict(sd)
    rng = np.random.default_rng(1)
    B, S = 39, 128                                   # 4992 rows -> 5120 with tile padding
    ids = rng.integers(1000, 30522, (B, S))
    mask = np.ones((B, S), bool)
    batch = {"input_ids": torch.from_numpy(ids).to(gpu_device), "input_mask": torch.from_numpy(mask).to(gpu_device)}
    ref = bert_oracle.get_embed({k: v.numpy() for k, v in sd.items()}, ids, mask, False, 1, 12)
    for cls_only in (True, False):
        model.cls_only_last_layer = cls_only
        got = model.get_embed(batch, False)["embed"].float().cpu().numpy()
        assert np.abs(got - ref).max() < TOL_BERT_BASE


def test_configs1_batch_512_through_predict_and_npy(gpu_device, tmp_path):
    """BASELINE.json configs[1] at its own shape: bert-base x 12 layers, batches of 512 x 128 through the product's loop
    (get_embed.predict -> torch.cat -> D2H -> npy.save).  A 512-passage batch is 65 536 token rows: the fused dense + bias +
    GELU kernel (proqa_gemm_tn_f16) and the [CLS]-only last layer really run, which the 6-passage test above does not
    reach.  Three batches: full length, lengths ~U[32,128] (valid tokens packed), and the ragged 64-passage tail of a
    1M-passage corpus (1M = 1953 x 512 + 64).  Rows are independent of their batch, so the NumPy oracle is run on 14 rows
    spread over the batches only."""
    from types import SimpleNamespace
    from proqa_amd import npy
    from proqa_amd.get_embed import predict
    from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
    sd = {k: v.half().float() for k, v in random_state_dict(BERT_BASE, seed=0).items()}
    model = BertForRetriever(BERT_BASE, device=gpu_device)
    model.load_state_dict(sd)
    model.half()
    B, S = 512, 128
    g = torch.Generator().manual_seed(4)
    full = torch.randint(1000, 30522, (B, S), generator=g, dtype=torch.int64)
    full[:, 0], full[:, -1] = 101, 102
    lens = torch.randint(32, S + 1, (B,), generator=g)
    vmask = torch.arange(S)[None, :] < lens[:, None]
    vids = torch.where(vmask, torch.randint(1000, 30522, (B, S), generator=g, dtype=torch.int64), torch.zeros((), dtype=torch.int64))
    vids[:, 0] = 101
    vids[torch.arange(B), lens - 1] = 102
    tail = torch.randint(1000, 30522, (64, S), generator=g, dtype=torch.int64)
    tail[:, 0], tail[:, -1] = 101, 102
    batches = [{"input_ids": full, "input_mask": torch.ones((B, S), dtype=torch.bool)},
               {"input_ids": vids, "input_mask": vmask},
               {"input_ids": tail, "input_mask": torch.ones((64, S), dtype=torch.bool)}]
    embeds = predict(SimpleNamespace(), model, iter(batches), gpu_device, is_query_embed=False)
    assert embeds.shape == (2 * B + 64, 128) and embeds.dtype == torch.float16
    path = str(tmp_path / "para_embed.npy")
    npy.save(path, embeds.cpu().numpy())
    back = np.load(path)                                   # numpy reads what the product wrote
    assert back.shape == (2 * B + 64, 128) and back.dtype == np.float16
    rows = [0, 1, 63, 200, 255, 256, 400, 511,             # the full-length batch: both halves of every 256-row GEMM tile
            B + 3, B + 77, B + 300, B + 511,               # the packed variable-length batch
            2 * B, 2 * B + 63]                             # the ragged tail
    all_ids = torch.cat([b["input_ids"] for b in batches]).numpy()
    all_mask = torch.cat([b["input_mask"] for b in batches]).numpy()
    sd_np = {k: v.numpy() for k, v in sd.items()}
    ref = bert_oracle.get_embed(sd_np, all_ids[rows], all_mask[rows], False, 12, 12)
    got = back[rows].astype(np.float32)
    assert np.abs(got - ref).max() < TOL_BERT_BASE, np.abs(got - ref).max(axis=1)
    assert cosine(got, ref).min() > COS_MIN


def test_randomised_encoder_fuzz(gpu_device):
    """20 s of scripts/dev_fuzz_encoder.py: random model widths / depths / batch shapes / lengths, both
    towers, packed and padded layouts, [CLS]-only and every-token last layer, against the NumPy oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "dev_fuzz_encoder.py"), "20", "11"],
                         capture_output=True, text=True, cwd=root, timeout=600)
    assert out.returncode == 0 and "encoder fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_gemm_tuning_keeps_the_numbers(gpu_device):
    """Opt-in rocBLAS solution tuning changes at most the summation order of the dense layers."""
    from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE, config_from_dict
    cfg = config_from_dict(dict(BERT_BASE, num_hidden_layers=2))
    sd = {k: v.half().float() for k, v in random_state_dict(cfg, seed=5).items()}
    model = BertForRetriever(cfg, device=gpu_device)
    model.load_state_dict(sd)
    rng = np.random.default_rng(2)
    ids = torch.from_numpy(rng.integers(1000, 30522, (64, 128))).to(gpu_device)        # 8192 rows: tuned shapes
    batch = {"input_ids": ids, "input_mask": torch.ones_like(ids, dtype=torch.bool)}
    base = model.get_embed(batch, False)["embed"].float().cpu().numpy()
    model.tune_gemms(True)
    tuned = model.get_embed(batch, False)["embed"].float().cpu().numpy()
    again = model.get_embed(batch, False)["embed"].float().cpu().numpy()
    assert np.abs(tuned - base).max() < 3e-3
    np.testing.assert_array_equal(tuned, again)


def test_row_without_valid_tokens_does_not_disturb_the_batch(gpu_device):
    """An all-padding row (mask all False) is evaluated as one token in both layouts; the other rows of the
    batch are unaffected and everything stays finite."""
    from proqa_amd.retriever import BertForRetriever
    z, sd, cfg = load_golden()
    model = BertForRetriever(cfg, device=gpu_device)
    model.load_state_dict(sd)
    ids = torch.from_numpy(z["input_ids"]).to(gpu_device).clone()
    mask = torch.from_numpy(z["input_mask"]).to(gpu_device).clone()
    ref = model.get_embed({"input_ids": ids, "input_mask": mask}, False)["embed"].float().cpu().numpy()
    mask[5] = False
    ids[5] = 0
    for packed in (True, False):
        model.pack_tokens = packed
        out = model.get_embed({"input_ids": ids, "input_mask": mask}, False)["embed"].float().cpu().numpy()
        assert np.isfinite(out).all()
        keep = [i for i in range(len(out)) if i != 5]
        assert np.abs(out[keep] - ref[keep]).max() < 2e-3


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (768, 512, 768), (2048, 3072, 768), (1024, 768, 3072)])
def test_hand_written_dense_layer_matches_fp32_reference(gpu_device, M, N, K):
    """proqa_gemm_tn_f16 (256x256x64 MFMA tiles, LDS-DMA ring, fused epilogues) against a plain fp32 product of the same
    fp16 operands: every epilogue, a feature count that is not one tile, a K that is many steps.  Tolerance = fp16
    rounding of the result (2^-11 relative) plus the fp32 accumulation order."""
    import ctypes
    from proqa_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=gpu_device).manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g, device=gpu_device).half()
    w = (torch.randn((N, K), generator=g, device=gpu_device) * 0.05).half()
    b = torch.randn(N, generator=g, device=gpu_device).half()
    ref = x.float() @ w.float().t()
    for epi in (0, 1, 2):
        want = ref if epi == 0 else ref + b.float()
        if epi == 2:
            want = torch.nn.functional.gelu(want)          # erf form, like hidden_act = 'gelu'
        y = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu_device)
        _lib.check(lib.proqa_gemm_tn_f16(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, epi,
                                         _lib.current_stream_ptr()))
        torch.cuda.synchronize()
        err = (y.float() - want).abs()
        tol = 2.0 ** -10 * want.abs() + 2e-3
        assert bool((err <= tol).all()), (epi, float(err.max()))
    # shapes the tiling cannot take are refused, not mis-computed
    with pytest.raises(_lib.ProqaError):
        _lib.check(lib.proqa_gemm_tn_f16(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M - 1, N, K, 0,
                                         _lib.current_stream_ptr()))


def test_hand_written_dense_layer_is_race_free_under_load(gpu_device):
    """Race screen of the anti-phase K loop (LDS-DMA ring refilled region by region behind counted vmcnt waits): every
    output is accumulated in a fixed order, so repeated launches must be BIT-identical -- with another stream keeping the
    GPU unevenly busy, over shapes with one and many K-steps, one and many tiles per workgroup, with and without epilogue."""
    import hashlib
    from proqa_amd import _lib
    lib = _lib.load()
    side = torch.cuda.Stream(device=gpu_device)
    junk = torch.randn((4096, 4096), device=gpu_device, dtype=torch.float16)
    for (M, N, K) in [(65536, 3072, 768), (16384, 768, 3072), (4096, 256, 64), (7936, 768, 768), (256, 3072, 128)]:
        g = torch.Generator(device=gpu_device).manual_seed(M + N + K)
        x = torch.randn((M, K), generator=g, device=gpu_device).half()
        w = (torch.randn((N, K), generator=g, device=gpu_device) * 0.05).half()
        b = torch.randn(N, generator=g, device=gpu_device).half()
        digests = {0: set(), 2: set()}
        for it in range(16):
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    junk @ junk
            epi = 2 if it % 2 else 0
            y = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu_device)
            _lib.check(lib.proqa_gemm_tn_f16(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, epi,
                                             _lib.current_stream_ptr()))
            torch.cuda.synchronize()
            digests[epi].add(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest())
        assert len(digests[0]) == 1 and len(digests[2]) == 1, (M, N, K)
        # ... and the one result is the right one (full-size check of the plain product on a row sample)
        rows = torch.randint(0, M, (64,), device=gpu_device)
        y = torch.empty((M, N), dtype=torch.float16, device=gpu_device)
        _lib.check(lib.proqa_gemm_tn_f16(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, 0,
                                         _lib.current_stream_ptr()))
        want = x[rows].float() @ w.float().t()
        assert bool(((y[rows].float() - want).abs() <= 2.0 ** -10 * want.abs() + 2e-3).all())


def test_pinned_library_kernel_equals_the_rocblas_path(gpu_device, monkeypatch):
    """The large dense layers on the hipBLASLt kernel pinned by name (csrc/lt_gemm.cpp; the default wherever the library
    holds one of the preferred names) against rocblas_gemm_ex (PROQA_LT_GEMM=0): same embeddings to fp16 round-off on a
    bert-base-width stack, full-length and packed batches; the handle reports which kernel it pinned, and ragged products
    (a feature count that is not a multiple of 256) stay on rocBLAS."""
    from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
    cfg = dict(BERT_BASE, num_hidden_layers=2)
    sd = random_state_dict(cfg, seed=6)
    B, S = 160, 128
    ids = torch.randint(1000, 30000, (B, S), device=gpu_device)
    lens = torch.randint(40, S + 1, (B,), device=gpu_device)
    names = {}
    for mask in (torch.ones((B, S), dtype=torch.bool, device=gpu_device), torch.arange(S, device=gpu_device)[None, :] < lens[:, None]):
        outs = []
        for mode in ("1", "0"):
            monkeypatch.setenv("PROQA_LT_GEMM", mode)
            model = BertForRetriever(cfg, device=gpu_device)
            model.load_state_dict(sd)
            outs.append(model.get_embed({"input_ids": ids, "input_mask": mask}, False)["embed"].float())
            names[mode] = model.gemm_kernels()[False]
        assert float((outs[0] - outs[1]).abs().max()) < 5e-3
    assert names["0"] == ""                                   # switched off: rocblas_gemm_ex
    if names["1"]:                                            # (a library build without the preferred names pins nothing)
        assert "MT256x256x64" in names["1"]


def test_library_dense_layer_grid(gpu_device):
    """proqa_encoder_dense (the handle's library dense layer: the pinned hipBLASLt kernel for whole-tile products with
    >= 4 K steps and >= 4096 rows, rocblas_gemm_ex otherwise) against an fp32 product, on both sides of every edge of
    that class -- K = 64 and 128 are shapes the pinned kernel claims to support and computes wrongly."""
    from proqa_amd import _lib
    from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
    cfg = dict(BERT_BASE, num_hidden_layers=1)
    model = BertForRetriever(cfg, device=gpu_device)
    model.load_state_dict(random_state_dict(cfg, seed=0))
    lib, h = _lib.load(), model.towers[False]._handle
    g = torch.Generator(device=gpu_device).manual_seed(0)
    for M in (300, 4096, 4100, 6400, 33024):
        for N in (64, 256, 768, 2304):
            for K in (64, 128, 192, 256, 320, 768, 3072):
                x = torch.randn((M, K), generator=g, device=gpu_device).half()
                w = (torch.randn((N, K), generator=g, device=gpu_device) * 0.05).half()
                out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu_device)
                _lib.check(lib.proqa_encoder_dense(h, x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, _lib.current_stream_ptr()))
                ref = x.float() @ w.float().t()
                assert torch.isfinite(out).all(), (M, N, K)
                err = float((out.float() - ref).abs().max())
                assert err <= 2e-3 * max(1.0, float(ref.abs().max())), (M, N, K, err)


def test_fused_ffn1_path_equals_library_path(gpu_device, monkeypatch):
    """The encoder with BertIntermediate on the hand-written GEMM (default for >= 64 row tiles) and on the library GEMM +
    bias_gelu (PROQA_FFN1=lib) agree to fp16 round-off on a bert-base-width layer stack."""
    from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
    cfg = dict(BERT_BASE, num_hidden_layers=2)
    sd = random_state_dict(cfg, seed=3)
    B, S = 160, 128                                         # 20480 token rows = 80 tiles: the fused path is taken
    ids = torch.randint(1000, 30000, (B, S), device=gpu_device)
    mask = torch.ones((B, S), dtype=torch.bool, device=gpu_device)
    outs = []
    for mode in ("own", "lib"):
        monkeypatch.setenv("PROQA_FFN1", mode)
        model = BertForRetriever(cfg, device=gpu_device)
        model.load_state_dict(sd)
        outs.append(model.get_embed({"input_ids": ids, "input_mask": mask}, False)["embed"].float())
    assert float((outs[0] - outs[1]).abs().max()) < 5e-3
