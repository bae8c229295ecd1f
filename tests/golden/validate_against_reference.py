"""Randomised validation of the restatements against the REFERENCE's own code (build container only:
imports /root/reference; not collected by pytest, nothing here travels to the GPU box).

    python tests/golden/validate_against_reference.py

 1. host scoring: proqa_amd SimpleTokenizer / normalize / para_has_answer vs retrieval/basic_tokenizer.py,
    utils.py, eval_retrieval.py:27-45 on 20000 random strings (unicode, punctuation, combining marks);
 2. data: proqa_amd EmDataset / em_collate vs retrieval/datasets.py:29-45,257-305 on random word salads,
    three (max_length, max_query_length) settings, both is_query_embed values;
 3. encoder oracles: oracle/bert_oracle.py and oracle/bert_torch_cpu.py vs retrieval/retriever.py:33-43
    (BertForRetriever.get_embed, CPU fp32) on four model shapes beyond the committed golden one.
Last run: all three pass (max encoder error 1.5e-6)."""
import json
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference/retrieval"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

sys.modules.setdefault("faiss", types.ModuleType("faiss"))      # absent here; only the scorer is used
sys.path.insert(0, REF)
import basic_tokenizer as ref_tok          # noqa: E402
import datasets as ref_ds                  # noqa: E402
import eval_retrieval as ref_eval          # noqa: E402
import retriever as ref_retriever          # noqa: E402
import utils as ref_utils                  # noqa: E402
for m in (ref_tok, ref_ds, ref_eval, ref_retriever, ref_utils):
    assert m.__file__.startswith("/root/reference"), m.__file__
sys.path.insert(0, ROOT)
from oracle import bert_oracle, bert_torch_cpu                                                    # noqa: E402
from proqa_amd import basic_tokenizer as my_tok, datasets as my_ds, eval_retrieval as my_eval, utils as my_utils  # noqa: E402


def host_scoring(n_cases=20000):
    random.seed(0)
    alphabet = list("abcdefghij ABC 0123  .,;:'\"-()[]!?\u00e9\u00e8\u00fc\u00f1\u00e7\u00c5\u00df\u00f8"
                    "\u65e5\u672c\u8a9e\u03b1\u03b2\u03b3\u0301\u0308\t\n") + \
        ["New York", "U.S.", "3.14", "don't", "co-op", "\uff11\uff12", "\u2167"]

    def rand_text(n):
        return "".join(random.choice(alphabet) for _ in range(n))

    ref_t, my_t = ref_tok.SimpleTokenizer(), my_tok.SimpleTokenizer()
    ref_eval.PROCESS_TOK, my_eval.PROCESS_TOK = ref_t, my_t
    for _ in range(n_cases):
        text = rand_text(random.randint(0, 80))
        a = ref_t.tokenize(ref_utils.normalize(text)).words(uncased=True)
        b = my_t.tokenize(my_utils.normalize(text)).words(uncased=True)
        assert a == b, (text, a, b)
        ans = [rand_text(random.randint(1, 6)) for _ in range(random.randint(1, 3))]
        if random.random() < 0.5 and len(text) > 10:
            s = random.randint(0, len(text) - 5)
            ans.append(text[s:s + random.randint(1, 8)])
        out = []
        for fn in (ref_eval.para_has_answer, my_eval.para_has_answer):
            try:
                out.append(fn(ans, text, True))
            except Exception as e:          # same exception type counts as the same behaviour
                out.append(("exc", type(e).__name__))
        assert out[0] == out[1], (text, ans, out)
    print(f"host scoring: {n_cases} random cases identical")


def data_pipeline():
    from transformers import BertTokenizer
    d = tempfile.mkdtemp()
    shutil.copy(os.path.join(ROOT, "tests", "golden", "vocab_small.txt"), os.path.join(d, "vocab.txt"))
    tok = BertTokenizer.from_pretrained(d)
    vocab = [line.strip() for line in open(os.path.join(d, "vocab.txt"))]
    random.seed(1)
    words = [w for w in vocab if not w.startswith("[")] + ["zzzunknown", "Paris,", "don't", "U.S.", "\u00e9", "\u65e5\u672c"]
    p = os.path.join(d, "data.txt")
    for is_q in (True, False):
        with open(p, "w") as f:
            for i in range(300):
                t = " ".join(random.choice(words) for _ in range(random.randint(0, 60)))
                f.write(json.dumps({"question": t, "text": t, "id": i, "answer": ["x"]}) + "\n")
        for max_len, max_q in [(30, 10), (512, 30), (8, 4)]:
            a = ref_ds.EmDataset(tok, p, max_q, max_len, is_q)
            b = my_ds.EmDataset(tok, p, max_q, max_len, is_q)
            assert len(a) == len(b)
            ia, ib = [a[i] for i in range(len(a))], [b[i] for i in range(len(b))]
            for x, y in zip(ia, ib):
                assert x.keys() == y.keys() and all(torch.equal(x[k], y[k]) for k in x)
            for lo in range(0, 300, 37):
                ca, cb = ref_ds.em_collate(ia[lo:lo + 37]), my_ds.em_collate(ib[lo:lo + 37])
                assert ca.keys() == cb.keys()
                assert all(torch.equal(ca[k], cb[k]) and ca[k].dtype == cb[k].dtype for k in ca)
    print("EmDataset / em_collate: identical on 2 x 3 x 300 random items")


def encoder_oracles():
    from transformers import BertConfig, BertModel
    rng = np.random.default_rng(0)
    worst = 0.0
    for case, (H, heads, layers, inter) in enumerate([(64, 1, 1, 100), (192, 3, 3, 300), (256, 4, 2, 1024), (128, 2, 4, 64)]):
        cfg = BertConfig(vocab_size=200, hidden_size=H, num_hidden_layers=layers, num_attention_heads=heads,
                         intermediate_size=inter, max_position_embeddings=80, type_vocab_size=2,
                         layer_norm_eps=1e-12, hidden_act="gelu")
        tmp = tempfile.mkdtemp()
        torch.manual_seed(case)
        BertModel(cfg).save_pretrained(tmp)
        model = ref_retriever.BertForRetriever(cfg, types.SimpleNamespace(bert_model_name=tmp))
        with torch.no_grad():
            for _, prm in model.named_parameters():
                prm.add_(0.05 * torch.randn_like(prm))
        model.eval()
        sd = {k: v.detach().numpy().astype(np.float32) for k, v in model.state_dict().items() if v.dtype.is_floating_point}
        B, S = 7, 33
        lens = rng.integers(1, S + 1, B)
        lens[0] = S
        ids, mask = np.zeros((B, S), np.int64), np.zeros((B, S), bool)
        for b, n in enumerate(lens):
            ids[b, :n] = rng.integers(1, 200, n)
            mask[b, :n] = True
        batch = {"input_ids": torch.from_numpy(ids), "input_mask": torch.from_numpy(mask)}
        for is_q in (True, False):
            with torch.no_grad():
                ref = model.get_embed(batch, is_q)["embed"].numpy()
            e1 = np.abs(bert_oracle.get_embed(sd, ids, mask, is_q, layers, heads) - ref).max()
            e2 = np.abs(bert_torch_cpu.get_embed({k: torch.from_numpy(v) for k, v in sd.items()}, ids, mask, is_q,
                                                 layers, heads).numpy() - ref).max()
            worst = max(worst, float(e1), float(e2))
            assert e1 < 5e-5 and e2 < 5e-5, (case, is_q, e1, e2)
    print(f"encoder oracles: 4 shapes x 2 towers, worst max-abs error {worst:.2e}")


if __name__ == "__main__":
    host_scoring()
    data_pipeline()
    encoder_oracles()
