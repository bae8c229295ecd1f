"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own code.

Run in the build container only (needs /root/reference; nothing here travels to the GPU box
except the small data files it writes):

    python tests/golden/make_golden.py

G1 encoder_golden.npz   reference BertForRetriever.get_embed (CPU fp32) on a small random model
G2 tokenize_golden.json reference EmDataset + em_collate on ~20 strings, max_length 30 / 512 / 8
G3 recall_golden.json   reference get_score / para_has_answer on a tiny sqlite DB (+ the DB itself)
G4 search_golden.json   SHA-256 of the NumPy oracle's top-k on seeded inputs (regenerable)
G5 npy_f2.npy/npy_f4.npy byte-exact np.save outputs
"""
import hashlib
import json
import os
import sqlite3
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/retrieval"
sys.path.insert(0, ROOT)

SMALL_CFG = dict(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                 intermediate_size=512, max_position_embeddings=128, type_vocab_size=2,
                 layer_norm_eps=1e-12, hidden_act="gelu")

WORDS = ["the", "of", "and", "in", "to", "was", "is", "for", "as", "on", "with", "by", "he", "at", "from",
         "his", "it", "an", "are", "which", "paris", "france", "capital", "city", "river", "who", "what",
         "when", "where", "born", "president", "first", "world", "war", "king", "queen", "new", "york",
         "film", "music", "album", "band", "team", "season", "game", "university", "school", "state",
         "united", "states", "##s", "##ed", "##ing", "##ly", "##er", "##est", "##ion", "##al", "a", "b",
         "c", "d", "e", "f", "g", "h", "i", "j", "k", "l", "m", "n", "o", "p", "q", "r", "s", "t", "u",
         "v", "w", "x", "y", "z", "0", "1", "2", "3", "4", "5", "6", "7", "8", "9", ".", ",", "?", "!",
         "'", "-", "(", ")", "##a", "##b", "##c", "##d", "##e", "##i", "##n", "##o", "##r", "##t", "##u"]


def write_vocab(path):
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS
    toks += [f"tok{i}" for i in range(SMALL_CFG["vocab_size"] - len(toks))]
    assert len(toks) == SMALL_CFG["vocab_size"], len(toks)
    with open(path, "w") as f:
        f.write("\n".join(toks) + "\n")
    return toks


def make_model_dir(tmp):
    from transformers import BertConfig, BertModel
    torch.manual_seed(1234)
    cfg = BertConfig(**SMALL_CFG)
    BertModel(cfg).save_pretrained(tmp)
    write_vocab(os.path.join(tmp, "vocab.txt"))
    return cfg


def ref_import(name):
    sys.path.insert(0, REF)
    try:
        if name == "eval_retrieval" and "faiss" not in sys.modules:
            sys.modules["faiss"] = types.ModuleType("faiss")  # absent here; only the scorer is used
        return __import__(name)
    finally:
        sys.path.remove(REF)


def g1_encoder(tmp, cfg):
    retriever = ref_import("retriever")
    torch.manual_seed(99)
    model = retriever.BertForRetriever(cfg, types.SimpleNamespace(bert_model_name=tmp))
    # make both towers differ and round every weight to fp16 so the GPU path holds the same values
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.startswith("bert_c") or name.startswith("proj"):
                p.add_(0.02 * torch.randn_like(p))
            if "LayerNorm.weight" in name:
                p.add_(0.1 * torch.randn_like(p))
            if name.endswith("bias"):
                p.add_(0.05 * torch.randn_like(p))
            p.copy_(p.half().float())
    model.eval()
    rng = np.random.default_rng(7)
    B, S = 32, 48
    lens = rng.integers(3, S + 1, B)
    lens[0], lens[1] = S, 3
    ids = np.zeros((B, S), np.int64)
    mask = np.zeros((B, S), bool)
    for b, n in enumerate(lens):
        ids[b, 0], ids[b, n - 1] = 101, 102
        ids[b, 1:n - 1] = rng.integers(104, SMALL_CFG["vocab_size"], n - 2)
        mask[b, :n] = True
    batch = {"input_ids": torch.from_numpy(ids), "input_mask": torch.from_numpy(mask)}
    out = {}
    with torch.no_grad():
        out["embed_q"] = model.get_embed(batch, True)["embed"].numpy()
        out["embed_c"] = model.get_embed(batch, False)["embed"].numpy()
        hs = model.bert_c(batch["input_ids"], batch["input_mask"], output_hidden_states=True).hidden_states
        # per-layer checksums over VALID positions (padding rows are not part of the contract)
        m = torch.from_numpy(mask)[..., None].float()
        out["hidden_abs_mean_c"] = np.array([float((h.abs() * m).sum() / (m.sum() * h.shape[-1])) for h in hs],
                                            np.float64)
        out["hidden_cls_c"] = np.stack([h[:, 0].numpy() for h in hs])
        # pad-invariance probe: row 1 alone (no padding) must equal row 1 in the padded batch
        single = {"input_ids": batch["input_ids"][1:2, :3], "input_mask": batch["input_mask"][1:2, :3]}
        out["embed_c_row1_unpadded"] = model.get_embed(single, False)["embed"].numpy()
    sd = {k: v.numpy().astype(np.float16) for k, v in model.state_dict().items() if not k.endswith("position_ids")}
    np.savez_compressed(os.path.join(HERE, "encoder_golden.npz"), input_ids=ids, input_mask=mask,
                        **{"w::" + k: v for k, v in sd.items()}, **out)
    with open(os.path.join(HERE, "encoder_config.json"), "w") as f:
        json.dump(SMALL_CFG, f, indent=1)
    print("G1", out["embed_q"].shape, len(sd), "tensors")


TEXTS = [
    "Paris is the capital of France.", "who was the first president of the united states?",
    "The river runs by the city and the university", "new york state", "a", "",
    "what is the capital city of france", "born in 1984, he was king",
    "The band's first album was music for a film!", "x y z 0 1 2 3 4 5 6 7 8 9",
    "unknownword anotherone", "Team season game (school)", "the " * 40, "When, where? who - what",
    "QUEEN of the world war", "states' united", "i j k l m n o p q r s t u v w", "of and in to was is",
    "h e l l o", "Capital  of   the\tworld\nwar",
]


def g2_tokenize(tmp):
    datasets = ref_import("datasets")
    from transformers import BertTokenizer
    tok = BertTokenizer.from_pretrained(tmp)
    out = {"texts": TEXTS, "cases": []}
    for is_query, max_q, max_len in [(True, 30, 512), (False, 30, 512), (False, 30, 8), (True, 6, 512)]:
        key = "question" if is_query else "text"
        path = os.path.join(tmp, "in.jsonl")
        with open(path, "w") as f:
            for t in TEXTS:
                f.write(json.dumps({key: t, "id": 0}) + "\n")
        ds = datasets.EmDataset(tok, path, max_q, max_len, is_query)
        samples = [ds[i] for i in range(len(ds))]
        batch = datasets.em_collate(samples)
        out["cases"].append({"is_query": is_query, "max_query_length": max_q, "max_length": max_len,
                             "item_lengths": [int(s["input_ids"].numel()) for s in samples],
                             "input_ids": batch["input_ids"].tolist(),
                             "input_mask": batch["input_mask"].int().tolist()})
    assert datasets.em_collate([]) == {}
    with open(os.path.join(HERE, "tokenize_golden.json"), "w") as f:
        json.dump(out, f)
    with open(os.path.join(tmp, "vocab.txt")) as src, open(os.path.join(HERE, "vocab_small.txt"), "w") as dst:
        dst.write(src.read())
    print("G2", len(out["cases"]), "cases")


DOCS = [
    ("d0", "Paris is the capital and most populous city of France."),
    ("d1", "The Seine is a river in northern France."),
    ("d2", "George Washington was the first President of the United States."),
    ("d3", "Beyoncé Knowles released the album in 2003."),          # NFC e-acute
    ("d4", "Beyoncé is a singer."),                                  # NFD e + combining acute
    ("d5", "New York City (NYC) is the most populous city in the U.S."),
    ("d6", "It's a 3.5-star film, isn't it?"),
    ("d7", "東京 is the capital of Japan; Tōkyō in romaji."),
    ("d8", "The quick brown fox jumps over the lazy dog"),
    ("d9", ""),
    ("d10", "washington washington washington"),
    ("d11", "Naïve café owners in Zürich"),
    ("d12", "C++ and C# are programming languages; so is F#."),
    ("d13", "The year 1984 was written by George Orwell."),
    ("d14", "e = mc^2 ... said Einstein"),
    ("d15", "Straße means street"),
    ("d16", "PARIS, TEXAS is a 1984 film"),
    ("d17", "Mr. O'Neil's dog"),
    ("d18", "tab\tseparated\nnewline text"),
    ("d19", "the the the"),
]

QAS = [
    {"question": "capital of france?", "answer": ["Paris"]},
    {"question": "first us president", "answer": ["George Washington", "Washington"]},
    {"question": "who sang", "answer": ["Beyoncé"]},
    {"question": "biggest us city", "answer": ["new york city", "NYC"]},
    {"question": "star rating", "answer": ["3.5-star"]},
    {"question": "capital of japan", "answer": ["東京"]},
    {"question": "no match", "answer": ["zebra crossing"]},
    {"question": "multi token", "answer": ["lazy dog", "quick red"]},
    {"question": "languages", "answer": ["C#"]},
    {"question": "umlaut", "answer": ["zurich", "Zürich"]},
    {"question": "apostrophe", "answer": ["O'Neil"]},
    {"question": "empty alias", "answer": ["", "fox"]},
]


def g3_recall(tmp):
    ev = ref_import("eval_retrieval")
    db_path = os.path.join(HERE, "recall_docs.db")
    if os.path.exists(db_path):
        os.remove(db_path)
    conn = sqlite3.connect(db_path)
    conn.execute("CREATE TABLE documents (id PRIMARY KEY, text)")
    utils = ref_import("utils")
    conn.executemany("INSERT INTO documents VALUES (?,?)", [(utils.normalize(i), t) for i, t in DOCS])
    conn.commit()
    conn.close()
    idx_id = {str(i): DOCS[i][0] for i in range(len(DOCS))}
    rng = np.random.default_rng(3)
    topk = 8
    I = np.stack([rng.permutation(len(DOCS))[:topk] for _ in QAS]).astype(np.int64)
    I[0, 0], I[1, 6], I[3, 5], I[7, 7] = 0, 2, 5, 8      # plant hits at chosen ranks
    cwd = os.getcwd()
    os.makedirs(os.path.join(tmp, "pretrained_models"), exist_ok=True)
    os.makedirs(os.path.join(tmp, "run"), exist_ok=True)
    with open(os.path.join(tmp, "pretrained_models", "idx_id.json"), "w") as f:
        json.dump(idx_id, f)
    os.chdir(os.path.join(tmp, "run"))
    try:
        doc_ids = ev.convert_idx2id(I)         # reads ../pretrained_models/idx_id.json
    finally:
        os.chdir(cwd)
    ev.init(db_path)
    scores = {}
    for topk_arg in (8, 5, 3):
        scores[str(topk_arg)] = [ev.get_score((qa["question"], qa["answer"], ids), topk=topk_arg)
                                 for qa, ids in zip(QAS, doc_ids)]
    matched = [[list(ev.para_has_answer(qa["answer"], text, True)) for _, text in DOCS] for qa in QAS]
    # the printed lines of __main__ (:115-123), for --topk 8
    from collections import defaultdict
    agg = defaultdict(list)
    for r in scores["8"]:
        for k, v in r.items():
            agg[k].append(v)
    lines = ["Top {} Recall for {} QA pairs: {} ...".format(k, len(v), np.mean(v)) for k, v in agg.items()]
    tok = ev.PROCESS_TOK
    tokenized = [[list(t[:2]) + [list(t[2])] for t in tok.tokenize(utils.normalize(text)).data] for _, text in DOCS]
    with open(os.path.join(HERE, "recall_golden.json"), "w") as f:
        json.dump({"docs": DOCS, "qas": QAS, "idx_id": idx_id, "I": I.tolist(), "doc_ids": doc_ids,
                   "scores": scores, "matched": matched, "lines_topk8": lines, "tokenized": tokenized}, f)
    print("G3", lines)


def g4_search():
    from oracle import search_oracle
    out = {}
    rng = np.random.default_rng(0)
    xb = rng.standard_normal((4096, 128)).astype(np.float16)
    xq = rng.standard_normal((64, 128)).astype(np.float16)
    D, I = search_oracle.topk_ip(xq, xb, 80)
    out["normal_4096x64_k80"] = {"seed": 0, "I_sha256": hashlib.sha256(I.tobytes()).hexdigest(),
                                 "D_sha256": hashlib.sha256(D.tobytes()).hexdigest(), "I_row0": I[0].tolist()}
    rng = np.random.default_rng(1)
    xb = rng.integers(-4, 5, (4096, 128)).astype(np.float16)
    xq = rng.integers(-4, 5, (64, 128)).astype(np.float16)
    D, I = search_oracle.topk_ip(xq, xb, 80)
    out["int_4096x64_k80"] = {"seed": 1, "I_sha256": hashlib.sha256(I.tobytes()).hexdigest(),
                              "D_sha256": hashlib.sha256(D.tobytes()).hexdigest(), "I_row0": I[0].tolist()}
    with open(os.path.join(HERE, "search_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("G4 ok")


def g5_npy():
    rng = np.random.default_rng(5)
    a = rng.standard_normal((10, 128)).astype(np.float16)
    np.save(os.path.join(HERE, "npy_f2.npy"), a)
    np.save(os.path.join(HERE, "npy_f4.npy"), a.astype(np.float32)[:3])
    print("G5 ok")


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as tmp:
        cfg = make_model_dir(tmp)
        g1_encoder(tmp, cfg)
        g2_tokenize(tmp)
        g3_recall(tmp)
    g4_search()
    g5_npy()
