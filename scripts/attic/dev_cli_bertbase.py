"""One-off: get_embed.py CLI at bert-base shape with ragged batches (random weights) vs single-row encodes."""
import json, os, shutil, sys, tempfile
import numpy as np, torch
sys.path.insert(0, ".")
from proqa_amd import get_embed
from proqa_amd.retriever import BertForRetriever, random_state_dict, BERT_BASE
d = tempfile.mkdtemp()
md = os.path.join(d, "bert"); os.makedirs(md)
shutil.copy("tests/golden/vocab_small.txt", os.path.join(md, "vocab.txt"))
json.dump(dict(BERT_BASE, model_type="bert"), open(os.path.join(md, "config.json"), "w"))
sd = random_state_dict(BERT_BASE, seed=3)
torch.save({"module." + k: v for k, v in sd.items()}, os.path.join(d, "ckpt.pt"))
vocab = [l.strip() for l in open(os.path.join(md, "vocab.txt")) if l.strip() and not l.startswith("[")]
rng = np.random.default_rng(0)
lines = []
with open(os.path.join(d, "paras.txt"), "w") as f:
    for i in range(700):
        n = int(rng.integers(1, 200)) if i % 7 else 400          # some longer than max_seq_length
        t = " ".join(rng.choice(vocab, n))
        lines.append(t)
        f.write(json.dumps({"id": i, "text": t}) + "\n")
out = get_embed.main(["--do_predict", "--bert_model_name", md, "--fp16", "--init_checkpoint", os.path.join(d, "ckpt.pt"),
                      "--eval-workers", "2", "--predict_batch_size", "300", "--max_seq_length", "128",
                      "--predict_file", os.path.join(d, "paras.txt"), "--embed_save_path", os.path.join(d, "emb.npy")])
emb = np.load(out)
assert emb.shape == (700, 128) and emb.dtype == np.float16 and np.isfinite(emb.astype(np.float32)).all()
from transformers import BertTokenizer
tok = BertTokenizer.from_pretrained(md)
model = BertForRetriever(BERT_BASE, device=torch.device("cuda", 0)); model.load_state_dict(sd)
worst = 0.0
for i in [0, 1, 6, 7, 299, 300, 301, 599, 600, 650, 699]:
    ids = torch.tensor([tok.encode(lines[i], max_length=128, truncation=True)], device="cuda")
    one = model.get_embed({"input_ids": ids, "input_mask": torch.ones_like(ids, dtype=torch.bool)}, False)["embed"]
    worst = max(worst, float(np.abs(one.float().cpu().numpy()[0] - emb[i].astype(np.float32)).max()))
print("cli bert-base ragged ok, worst row diff vs single encode", worst)
assert worst < 4e-3
shutil.rmtree(d)
