"""Error of the encoder under trained-model statistics (tests/test_encoder_gpu.py::test_trained_model_statistics_do_not_break_fp16)
for several weight seeds: max relative error and worst cosine against oracle/bert_oracle.py (dev; run on the MI355X)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import bert_oracle  # noqa: E402
from test_encoder_gpu import trained_like_state_dict, cosine  # noqa: E402
from proqa_amd.retriever import BertForRetriever, BERT_BASE  # noqa: E402

dev = torch.device("cuda:0")
for seed in [int(v) for v in sys.argv[1:]] or [0, 1, 2]:
    sd = trained_like_state_dict(seed)
    model = BertForRetriever(BERT_BASE, device=dev)
    model.load_state_dict(sd)
    sd_np = {k: v.numpy() for k, v in sd.items()}
    rng = np.random.default_rng(2 + seed)
    n_ref = 24
    ids = rng.integers(1000, 30522, (512, 128))
    ids[:, 0], ids[:, -1] = 101, 102
    mask = np.ones((512, 128), bool)
    batch = {"input_ids": torch.from_numpy(ids).to(dev), "input_mask": torch.from_numpy(mask).to(dev)}
    ref = bert_oracle.get_embed(sd_np, ids[:n_ref], mask[:n_ref], False, 12, 12)
    ref16 = bert_oracle.get_embed(sd_np, ids[:n_ref], mask[:n_ref], False, 12, 12, storage="fp16")
    print(f"seed {seed}: fp16-storage oracle vs fp32 oracle {np.abs(ref16 - ref).max() / np.abs(ref).max():.2e}")
    for cls_only, packed in ((True, True), (False, False)):
        model.cls_only_last_layer, model.pack_tokens = cls_only, packed
        got = model.get_embed(batch, False)["embed"].float().cpu().numpy()
        rel = np.abs(got[:n_ref] - ref).max() / np.abs(ref).max()
        rel16 = np.abs(got[:n_ref] - ref16).max() / np.abs(ref).max()
        print(f"seed {seed} cls_only {cls_only} packed {packed}: max rel err {rel:.2e} (vs the fp16-storage oracle {rel16:.2e})  "
              f"worst cosine {cosine(got[:n_ref], ref).min():.6f}")
