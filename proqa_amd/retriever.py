"""Dual-tower BERT retriever on MI355X.

Mirrors /root/reference/retrieval/retriever.py (BertForRetriever :10-20, get_embed :33-43):

    model.get_embed({'input_ids': LongTensor[B,L], 'input_mask': BoolTensor[B,L]}, is_query_embed)
        -> {'embed': Tensor[B,128]}

Each tower is BertModel (embeddings, N x BertLayer, pooler) followed by Linear(hidden, 128) on
the pooled [CLS] vector.  The whole tower runs inside libproqa_hip.so (`proqa_encoder_forward`):
the dense projections as rocBLAS fp16 GEMMs with fp32 accumulation, everything between them as
hand-written HIP kernels -- embedding gather+LayerNorm, MFMA attention with the key-padding mask,
bias+GELU(erf), bias+residual+LayerNorm, and the fused pooler(tanh)+projection head.  PyTorch only
owns the weight / activation tensors whose device pointers cross the C ABI.
Weights are held in fp16 (the reference runs apex AMP O1 / .half() for --fp16); accumulation,
softmax and LayerNorm statistics are fp32.  There is no CPU path.
"""
import ctypes
import re
from types import SimpleNamespace

import torch

from . import _lib
from ._lib import PROQA_F16, PROQA_F32, EMBED_DIM


def config_from_dict(d):
    """BertConfig-like namespace from a plain dict (config.json of a HF model directory)."""
    return SimpleNamespace(
        vocab_size=d["vocab_size"], hidden_size=d["hidden_size"],
        num_hidden_layers=d["num_hidden_layers"], num_attention_heads=d["num_attention_heads"],
        intermediate_size=d["intermediate_size"], max_position_embeddings=d["max_position_embeddings"],
        type_vocab_size=d.get("type_vocab_size", 2), layer_norm_eps=d.get("layer_norm_eps", 1e-12),
        hidden_act=d.get("hidden_act", "gelu"))


BERT_BASE = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2,
                 layer_norm_eps=1e-12, hidden_act="gelu")


def tower_keys(prefix, n_layers):
    """state_dict keys of one BertModel tower, in the reference's (HF) naming."""
    keys = [f"{prefix}.embeddings.{n}" for n in (
        "word_embeddings.weight", "position_embeddings.weight", "token_type_embeddings.weight",
        "LayerNorm.weight", "LayerNorm.bias")]
    for i in range(n_layers):
        p = f"{prefix}.encoder.layer.{i}"
        for m in ("attention.self.query", "attention.self.key", "attention.self.value",
                  "attention.output.dense", "attention.output.LayerNorm", "intermediate.dense",
                  "output.dense", "output.LayerNorm"):
            keys += [f"{p}.{m}.weight", f"{p}.{m}.bias"]
    keys += [f"{prefix}.pooler.dense.weight", f"{prefix}.pooler.dense.bias"]
    return keys


class _Tower:
    """fp16 device weights of one BERT tower, laid out for the kernels (fused QKV, [out,in] GEMM operands)."""

    def __init__(self, sd, prefix, proj_prefix, cfg, device):
        def w(name):
            return sd[name].detach().to(device=device, dtype=torch.float16).contiguous()

        e = f"{prefix}.embeddings"
        self.word = w(f"{e}.word_embeddings.weight")
        self.pos = w(f"{e}.position_embeddings.weight")
        self.type0 = w(f"{e}.token_type_embeddings.weight")[0].contiguous()
        self.emb_g = w(f"{e}.LayerNorm.weight")
        self.emb_b = w(f"{e}.LayerNorm.bias")
        self.layers = []
        for i in range(cfg.num_hidden_layers):
            p = f"{prefix}.encoder.layer.{i}"
            qkv_w = torch.cat([w(f"{p}.attention.self.{n}.weight") for n in ("query", "key", "value")], 0)
            qkv_b = torch.cat([w(f"{p}.attention.self.{n}.bias") for n in ("query", "key", "value")], 0)
            # weights stay in nn.Linear's [out, in] layout: x @ W.t() is the BLAS "TN" GEMM, measured
            # 12-19 % faster at these shapes than the pre-transposed "NN" form (scripts/dev_gemm_timing.py)
            self.layers.append(SimpleNamespace(
                qkv_w=qkv_w.contiguous(), qkv_b=qkv_b.contiguous(),                 # [3H, H]
                ao_w=w(f"{p}.attention.output.dense.weight"),                        # [H, H]
                ao_b=w(f"{p}.attention.output.dense.bias"),
                ln1_g=w(f"{p}.attention.output.LayerNorm.weight"), ln1_b=w(f"{p}.attention.output.LayerNorm.bias"),
                ff1_w=w(f"{p}.intermediate.dense.weight"),                           # [I, H]
                ff1_b=w(f"{p}.intermediate.dense.bias"),
                ff2_w=w(f"{p}.output.dense.weight"),                                 # [H, I]
                ff2_b=w(f"{p}.output.dense.bias"),
                ln2_g=w(f"{p}.output.LayerNorm.weight"), ln2_b=w(f"{p}.output.LayerNorm.bias")))
        self.pool_w = w(f"{prefix}.pooler.dense.weight")   # [H, H] (out, in): read row-wise by the kernel
        self.pool_b = w(f"{prefix}.pooler.dense.bias")
        self.proj_w = w(f"{proj_prefix}.weight")           # [128, H]
        self.proj_b = w(f"{proj_prefix}.bias")
        self._handle = None
        self._create_encoder(cfg, device)

    def _create_encoder(self, cfg, device):
        """proqa_encoder_create over these tensors' device pointers (the tensors stay owned here)."""
        lib = _lib.load()
        layers = (_lib.BertLayer * len(self.layers))()
        for dst, L in zip(layers, self.layers):
            for name, _ in _lib.BertLayer._fields_:
                setattr(dst, name, getattr(L, name).data_ptr())
        bw = _lib.BertWeights(hidden=cfg.hidden_size, n_layers=len(self.layers), n_heads=cfg.num_attention_heads,
                              intermediate=cfg.intermediate_size, max_position=cfg.max_position_embeddings,
                              vocab=self.word.shape[0], layer_norm_eps=float(cfg.layer_norm_eps),
                              word_emb=self.word.data_ptr(), pos_emb=self.pos.data_ptr(), type_emb=self.type0.data_ptr(),
                              emb_ln_g=self.emb_g.data_ptr(), emb_ln_b=self.emb_b.data_ptr(), layers=layers,
                              pool_w=self.pool_w.data_ptr(), pool_b=self.pool_b.data_ptr(),
                              proj_w=self.proj_w.data_ptr(), proj_b=self.proj_b.data_ptr())
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.proqa_encoder_create(ctypes.byref(bw), ctypes.byref(handle)))
        self._handle = handle

    def close(self):
        if getattr(self, "_handle", None):
            _lib.load().proqa_encoder_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BertForRetriever:
    """Inference-only dual-tower retriever with the reference's get_embed call shape."""

    def __init__(self, config, args=None, device=None):
        self.config = config if not isinstance(config, dict) else config_from_dict(config)
        if self.config.hidden_size != self.config.num_attention_heads * 64:
            raise ValueError("the attention kernel is built for head_dim 64 (bert-base/large geometry)")
        if getattr(self.config, "hidden_act", "gelu") != "gelu":
            raise ValueError("only hidden_act='gelu' (erf) is implemented, as in bert-base-uncased")
        self.args = args
        self._lib = _lib.load()
        _lib.require_gpu()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.towers = {}
        self.out_dtype = torch.float16
        # last layer: attention output, dense blocks and LayerNorms for the [CLS] rows only (same result:
        # nothing else of that layer reaches the pooler); False runs every token through it
        self.cls_only_last_layer = True
        # evaluate valid tokens only (packed layout) whenever the lengths are known on the host
        self.pack_tokens = True

    # -- reference-compatible surface -----------------------------------------------------
    def state_dict_keys(self):
        n = self.config.num_hidden_layers
        return (tower_keys("bert_q", n) + tower_keys("bert_c", n)
                + ["proj_q.weight", "proj_q.bias", "proj_c.weight", "proj_c.bias"])

    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference checkpoint layout (get_embed.py:22-27 strips 'module.' first).

        torch-1.4-era checkpoints have no `position_ids` buffers; newer ones may — ignored.
        """
        sd = {k: v for k, v in state_dict.items() if not k.endswith("position_ids")}
        want = self.state_dict_keys()
        missing = [k for k in want if k not in sd]
        unexpected = [k for k in sd if k not in set(want)]
        if missing or (strict and unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for BertForRetriever: "
                               f"missing keys {missing[:8]}{'...' if len(missing) > 8 else ''}; "
                               f"unexpected keys {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
        self.towers = {
            True: _Tower(sd, "bert_q", "proj_q", self.config, self.device),
            False: _Tower(sd, "bert_c", "proj_c", self.config, self.device),
        }
        return self

    def tune_gemms(self, enable=True):
        """Opt-in: let each tower pick, per large GEMM shape, the fastest rocBLAS solution (timed at the first
        forward that meets the shape; proqa_encoder_set_gemm_tuning).  ~2 % on bert-base batches of 512 x 128."""
        for tw in self.towers.values():
            _lib.check(self._lib.proqa_encoder_set_gemm_tuning(tw._handle, 1 if enable else 0))
        return self

    def gemm_kernels(self):
        """{tower: name of the hipBLASLt kernel pinned for its large dense layers, "" = rocblas_gemm_ex} (proqa_encoder_gemm_kernel)."""
        import ctypes
        out = {}
        for key, tw in self.towers.items():
            buf = ctypes.create_string_buffer(640)
            _lib.check(self._lib.proqa_encoder_gemm_kernel(tw._handle, buf, len(buf)))
            out[key] = buf.value.decode()
        return out

    def workspace(self, is_query_embed):
        """(base address, bytes) of a tower's activation workspace (proqa_encoder_workspace): changes when a larger batch
        replaces it -- what a captured HIP graph of the forward has to watch."""
        import ctypes
        base, size = ctypes.c_void_p(), ctypes.c_size_t()
        _lib.check(self._lib.proqa_encoder_workspace(self.towers[bool(is_query_embed)]._handle, ctypes.byref(base), ctypes.byref(size)))
        return (base.value or 0, size.value)

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("proqa_amd.BertForRetriever runs on MI355X only; there is no CPU path")
        if self.towers and device != self.device:
            raise RuntimeError("move the model before load_state_dict")
        self.device = device
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def half(self):
        self.out_dtype = torch.float16
        return self

    def float(self):
        self.out_dtype = torch.float32
        return self

    def __call__(self, batch):
        raise NotImplementedError("training forward (retriever.py:22-31) is outside the encode/search hot path")

    @torch.no_grad()
    def get_embed(self, batch, is_query_embed, check_mask=True, seq_lens_host=None):
        """check_mask=False skips the device-side validation of the right-padding (it costs a
        host sync per batch); callers that validated the mask on the host pass False -- and the
        per-sequence lengths as seq_lens_host -- so that batches can be pipelined on several
        streams (proqa_amd.get_embed.predict)."""
        ids, mask = batch["input_ids"], batch["input_mask"]
        emb = self.encode(ids, mask, bool(is_query_embed), check_mask=check_mask, seq_lens_host=seq_lens_host)
        return {"embed": emb}

    # -- implementation -------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, input_ids, input_mask, is_query_embed, check_mask=True, seq_lens_host=None):
        """[B,S] ids + right-padded bool mask -> [B,128] embeddings (proqa_encoder_forward).

        Padding is not computed when the sequence lengths are known on the host: tokens are packed
        back to back ([T, hidden], T = sum of lengths) for every per-token operator, the embedding
        gather and the attention take the offsets.  The lengths come from `seq_lens_host` (predict()
        passes them from the collated CPU batch) or from the one host round trip `check_mask` makes
        anyway; with neither, the padded [B*S] layout is evaluated."""
        if not self.towers:
            raise RuntimeError("load_state_dict must be called before get_embed")
        if not input_ids.is_cuda:
            raise RuntimeError("get_embed expects CUDA tensors (the reference feeds move_to_cuda(batch))")
        tw = self.towers[bool(is_query_embed)]
        cfg = self.config
        B, S = input_ids.shape
        if S > cfg.max_position_embeddings:
            raise ValueError(f"sequence length {S} exceeds max_position_embeddings {cfg.max_position_embeddings}")
        out = torch.empty((B, EMBED_DIM), dtype=self.out_dtype, device=self.device)
        if B == 0:
            return out
        ids = input_ids.contiguous().to(torch.int64)
        mask = input_mask.to(torch.bool)
        # a row without any valid token is evaluated as its first (padding) token, like the kernels' own clamp
        lens = mask.sum(dim=1).clamp_(min=1).to(torch.int32).contiguous()
        n_valid = -1
        if seq_lens_host is not None:
            n_valid = int(sum(max(int(v), 1) for v in seq_lens_host))
        if check_mask:
            # em_collate pads on the right: the mask of every row is a prefix of ones
            bad = (mask[:, 1:] & ~mask[:, :-1]).any() if S > 1 else torch.zeros((), dtype=torch.bool, device=mask.device)
            probe = torch.stack([bad.to(torch.int64), lens.sum(dtype=torch.int64)]).cpu()   # one host round trip
            if bool(probe[0]):
                raise ValueError("input_mask must be right-padded (a prefix of True per row), as em_collate produces")
            if n_valid >= 0 and n_valid != int(probe[1]):
                raise ValueError("seq_lens_host does not match input_mask")
            n_valid = int(probe[1])
        flags = (_lib.ENC_CLS_ONLY_LAST if self.cls_only_last_layer else 0) | (_lib.ENC_PACKED if self.pack_tokens else 0)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.proqa_encoder_forward(
                tw._handle, ids.data_ptr(), lens.data_ptr(), B, S, n_valid, flags, out.data_ptr(),
                PROQA_F16 if self.out_dtype == torch.float16 else PROQA_F32, _lib.current_stream_ptr()))
        return out


def random_state_dict(config, seed=0, std=0.02):
    """N(0, std) weights in the reference checkpoint layout (synthetic benchmark / tests)."""
    cfg = config if not isinstance(config, dict) else config_from_dict(config)
    g = torch.Generator().manual_seed(seed)
    H, I = cfg.hidden_size, cfg.intermediate_size
    shapes = {}
    for t in ("bert_q", "bert_c"):
        shapes[f"{t}.embeddings.word_embeddings.weight"] = (cfg.vocab_size, H)
        shapes[f"{t}.embeddings.position_embeddings.weight"] = (cfg.max_position_embeddings, H)
        shapes[f"{t}.embeddings.token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
        for k in tower_keys(t, cfg.num_hidden_layers):
            if k in shapes:
                continue
            if k.endswith("LayerNorm.weight") or k.endswith("LayerNorm.bias") or k.endswith(".bias"):
                width = I if "intermediate.dense" in k else H
                shapes[k] = (width,)
            elif "intermediate.dense.weight" in k:
                shapes[k] = (I, H)
            elif re.search(r"layer\.\d+\.output\.dense\.weight", k):
                shapes[k] = (H, I)
            else:
                shapes[k] = (H, H)
    for p in ("proj_q", "proj_c"):
        shapes[f"{p}.weight"] = (EMBED_DIM, H)
        shapes[f"{p}.bias"] = (EMBED_DIM,)
    sd = {}
    for k, shp in shapes.items():
        if k.endswith("LayerNorm.weight"):
            sd[k] = 1.0 + std * torch.randn(shp, generator=g)
        else:
            sd[k] = std * torch.randn(shp, generator=g)
    return sd
