"""End to end through the reference's two command lines: get_embed.py -> .npy -> eval_retrieval.py.

A small BERT (golden weights written as a torch checkpoint with DataParallel 'module.' prefixes),
the golden documents as the corpus, checked against the NumPy oracles on the same inputs.
"""
import json
import os
import shutil
import sqlite3

import numpy as np
import pytest
import torch

from oracle import bert_oracle, search_oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    d = tmp_path_factory.mktemp("e2e")
    model_dir = d / "small-bert"
    model_dir.mkdir()
    shutil.copy(os.path.join(GOLDEN, "vocab_small.txt"), model_dir / "vocab.txt")
    cfg = json.load(open(os.path.join(GOLDEN, "encoder_config.json")))
    cfg["model_type"] = "bert"
    (model_dir / "config.json").write_text(json.dumps(cfg))
    z = np.load(os.path.join(GOLDEN, "encoder_golden.npz"))
    sd = {k[3:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("w::")}
    torch.save({"module." + k: v for k, v in sd.items()}, d / "checkpoint_best.pt")
    gold = json.load(open(os.path.join(GOLDEN, "recall_golden.json")))
    with open(d / "paras.txt", "w") as f:
        for doc_id, text in gold["docs"]:
            f.write(json.dumps({"id": doc_id, "text": text}) + "\n")
    with open(d / "qa.txt", "w") as f:
        for qa in gold["qas"]:
            f.write(json.dumps(qa) + "\n")
    shutil.copy(os.path.join(GOLDEN, "recall_docs.db"), d / "paras.db")
    return d, sd, cfg, gold


def oracle_embed(sd, cfg, tokenizer, texts, max_len, is_query):
    rows = [tokenizer.encode(t, max_length=max_len, truncation=True) for t in texts]
    S = max(len(r) for r in rows)
    ids = np.zeros((len(rows), S), np.int64)
    mask = np.zeros((len(rows), S), bool)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = r
        mask[i, :len(r)] = True
    return bert_oracle.get_embed({k: v.numpy() for k, v in sd.items()}, ids, mask, is_query,
                                 cfg["num_hidden_layers"], cfg["num_attention_heads"])


def test_get_embed_then_eval_retrieval(gpu_device, workdir, capsys):
    from transformers import BertTokenizer
    from proqa_amd import get_embed, eval_retrieval, gen_index_id_map
    d, sd, cfg, gold = workdir
    common = ["--do_predict", "--bert_model_name", str(d / "small-bert"), "--fp16",
              "--init_checkpoint", str(d / "checkpoint_best.pt"), "--eval-workers", "0", "--prefix", "eval-para"]
    para_out = get_embed.main(common + ["--predict_batch_size", "7", "--predict_file", str(d / "paras.txt"),
                                        "--embed_save_path", str(d / "para_embed")])     # '.npy' appended
    q_out = get_embed.main(common + ["--predict_batch_size", "512", "--predict_file", str(d / "qa.txt"),
                                     "--is_query_embed", "--embed_save_path", str(d / "q_embed.npy")])
    xb, xq = np.load(para_out), np.load(q_out)
    assert xb.dtype == np.float16 and xb.shape == (len(gold["docs"]), 128) and para_out.endswith("para_embed.npy")
    assert xq.dtype == np.float16 and xq.shape == (len(gold["qas"]), 128)
    tok = BertTokenizer.from_pretrained(str(d / "small-bert"))
    ref_b = oracle_embed(sd, cfg, tok, [t for _, t in gold["docs"]], 512, False)
    ref_q = oracle_embed(sd, cfg, tok, [qa["question"] for qa in gold["qas"]], 30, True)
    assert np.abs(xb.astype(np.float32) - ref_b).max() < 1.5e-3
    assert np.abs(xq.astype(np.float32) - ref_q).max() < 1.5e-3

    n = gen_index_id_map.build(str(d / "paras.txt"), str(d / "idx_id.json"))
    assert n == len(gold["docs"])
    capsys.readouterr()
    lines = eval_retrieval.main([str(d / "qa.txt"), para_out, q_out, str(d / "paras.db"), "--topk", "8",
                                 "--num-workers", "2", "--idx-id-map", str(d / "idx_id.json")])
    printed = capsys.readouterr().out.strip().splitlines()
    assert printed == lines and len(lines) == 5
    # expected lines: NumPy oracle search on the SAME embeddings + the (reference-pinned) host scorer
    D, I = search_oracle.topk_ip(xq, xb, 8)
    eval_retrieval.init(str(d / "paras.db"))
    doc_ids = eval_retrieval.convert_idx2id(I, str(d / "idx_id.json"))
    res = [eval_retrieval.get_score((qa["question"], qa["answer"], ids), topk=8) for qa, ids in zip(gold["qas"], doc_ids)]
    for line, key in zip(lines, ["8", "5", "10", "20", "50"]):
        want = "Top {} Recall for {} QA pairs: {} ...".format(key, len(res), np.mean([r[key] for r in res]))
        assert line == want


def test_get_embed_error_behaviour(workdir):
    from proqa_amd import get_embed
    d, *_ = workdir
    with pytest.raises(ValueError):
        get_embed.main(["--predict_file", "x"])                              # neither do_train nor do_predict
    with pytest.raises(ValueError):
        get_embed.main(["--do_predict"])                                     # no predict_file
    with pytest.raises(AssertionError):
        get_embed.main(["--do_predict", "--predict_file", str(d / "qa.txt"), "--is_query_embed",
                        "--bert_model_name", str(d / "small-bert"), "--eval-workers", "0"])   # no checkpoint
    with pytest.raises(SystemExit):
        get_embed.main(["--no_such_flag"])


def test_get_embed_loader_variants_write_the_same_file(gpu_device, workdir):
    """The command line with the producer-thread loader (--eval-workers > 0: native WordPiece for ASCII sentences, the
    tokenizer for the rest) and with --eval-workers 0 (tokenised in the consumer thread) writes the same embeddings; an
    empty input file gives an empty [0, 128] index; a malformed line surfaces as an error, not a hang."""
    from proqa_amd import get_embed
    d, sd, cfg, gold = workdir
    mixed = d / "mixed.txt"
    texts = [t for _, t in gold["docs"]] + ["plain ascii sentence with punctuation, numbers 123 and a verylongwordthatissplit",
                                             "café naïve 中文 text", "[SEP] literal special token", "", "x" * 300]
    mixed.write_text("".join(json.dumps({"id": i, "text": t}) + "\n" for i, t in enumerate(texts)))
    common = ["--do_predict", "--bert_model_name", str(d / "small-bert"), "--fp16", "--init_checkpoint",
              str(d / "checkpoint_best.pt"), "--predict_batch_size", "5", "--predict_file", str(mixed)]
    a = np.load(get_embed.main(common + ["--eval-workers", "6", "--embed_save_path", str(d / "mixed_a.npy")]))
    b = np.load(get_embed.main(common + ["--eval-workers", "0", "--embed_save_path", str(d / "mixed_b.npy")]))
    assert a.shape == (len(texts), 128) and a.dtype == np.float16
    np.testing.assert_array_equal(a, b)
    empty = d / "empty.txt"
    empty.write_text("")
    e = np.load(get_embed.main(common[:-1] + [str(empty), "--eval-workers", "4", "--embed_save_path", str(d / "empty.npy")]))
    assert e.shape == (0, 128)
    bad = d / "bad.txt"
    bad.write_text(json.dumps({"text": "fine"}) + "\nnot json at all\n")
    with pytest.raises(ValueError):
        get_embed.main(common[:-1] + [str(bad), "--eval-workers", "4", "--embed_save_path", str(d / "bad.npy")])
