"""Randomised parity fuzz of the HIP encoder against the NumPy BERT oracle (dev; run on the MI355X).
usage: python scripts/dev_fuzz_encoder.py [seconds] [seed]"""
import sys, time
from types import SimpleNamespace
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import bert_oracle
from proqa_amd.retriever import BertForRetriever, random_state_dict, config_from_dict

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
t_end = time.time() + budget
n_cases, worst = 0, 0.0
while time.time() < t_end:
    heads = int(rng.choice([1, 2, 3, 12]))
    layers = int(rng.choice([1, 2, 3])) if heads < 12 else 1
    inter = int(rng.choice([64, 256, 520]))
    cfg = config_from_dict({"vocab_size": 300, "hidden_size": heads * 64, "num_hidden_layers": layers,
                            "num_attention_heads": heads, "intermediate_size": inter, "max_position_embeddings": 520,
                            "layer_norm_eps": 1e-12, "hidden_act": "gelu"})
    sd = {k: v.half().float() for k, v in random_state_dict(cfg, seed=int(rng.integers(1 << 30)), std=0.05).items()}
    model = BertForRetriever(cfg, device=dev)
    model.load_state_dict(sd)
    sd_np = {k: v.numpy() for k, v in sd.items()}
    for _ in range(4):
        B, S = int(rng.integers(1, 25)), int(rng.choice([1, 2, 17, 31, 32, 33, 64, 100, 128, 129, 160, 200, 257, 300, 512, 520]))
        lens = rng.integers(1, S + 1, B)
        lens[rng.integers(0, B)] = S
        ids = np.zeros((B, S), np.int64); mask = np.zeros((B, S), bool)
        for b, n in enumerate(lens):
            ids[b, :n] = rng.integers(1, 300, n); mask[b, :n] = True
        is_q = bool(rng.integers(0, 2))
        ref = bert_oracle.get_embed(sd_np, ids, mask, is_q, layers, heads)
        batch = {"input_ids": torch.from_numpy(ids).to(dev), "input_mask": torch.from_numpy(mask).to(dev)}
        for cls_only in (True, False):
            for packed in (True, False):
                model.cls_only_last_layer, model.pack_tokens = cls_only, packed
                got = model.get_embed(batch, is_q)["embed"].float().cpu().numpy()
                err = float(np.abs(got - ref).max())
                worst = max(worst, err)
                if not np.isfinite(got).all() or err > 4e-3:
                    print(f"MISMATCH heads={heads} layers={layers} inter={inter} B={B} S={S} cls_only={cls_only} "
                          f"packed={packed} err={err} lens={lens.tolist()[:10]}")
                    sys.exit(1)
        n_cases += 1
print(f"encoder fuzz ok: {n_cases} batches x 4 modes in {budget:.0f} s, worst max-abs err {worst:.2e} (seed {seed})")
