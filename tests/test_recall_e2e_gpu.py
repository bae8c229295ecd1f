"""Synthetic end-to-end run (BASELINE.json configs[4] / SURVEY section 8d config 5 stand-in): a
100k-passage corpus with planted answers is encoded by get_embed.py on the GPU, searched by
eval_retrieval.py on the GPU, and the printed Recall@k lines are compared with the CPU NumPy search
on the SAME embeddings + the (reference-pinned) host scorer: |delta| <= 1e-4 (north_star)."""
import json
import os
import re
import shutil
import socket
import sqlite3
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import search_oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_DOCS, N_QA, TOPK = 100000, 150, 80


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    d = tmp_path_factory.mktemp("e2e_big")
    rng = np.random.default_rng(11)
    vocab = [w.strip() for w in open(os.path.join(GOLDEN, "vocab_small.txt")) if w.strip().isalpha() and len(w.strip()) > 1]
    model_dir = d / "small-bert"
    model_dir.mkdir()
    shutil.copy(os.path.join(GOLDEN, "vocab_small.txt"), model_dir / "vocab.txt")
    cfg = json.load(open(os.path.join(GOLDEN, "encoder_config.json")))
    cfg["model_type"] = "bert"
    (model_dir / "config.json").write_text(json.dumps(cfg))
    z = np.load(os.path.join(GOLDEN, "encoder_golden.npz"))
    torch.save({k[3:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("w::")}, d / "ckpt.pt")
    # answers are 4-word phrases of in-vocabulary words (an out-of-vocabulary marker would push every
    # planted passage towards [UNK] and out of every top-80 of the random-weight model)
    answers = [" ".join(rng.choice(vocab, size=4)) for j in range(N_QA)]
    assert len(set(answers)) == N_QA
    docs = []
    for i in range(N_DOCS):
        words = list(rng.choice(vocab, size=int(rng.integers(5, 40))))
        if i % 7 == 0:                                  # plant an answer alias in every 7th passage
            words.insert(int(rng.integers(0, len(words))), answers[(i // 7) % N_QA])
        docs.append((f"doc{i}", " ".join(words)))
    with open(d / "paras.txt", "w") as f:
        for doc_id, text in docs:
            f.write(json.dumps({"id": doc_id, "text": text}) + "\n")
    with open(d / "qa.txt", "w") as f:
        for j in range(N_QA):
            q = " ".join(rng.choice(vocab, size=6))
            f.write(json.dumps({"question": q, "answer": [answers[j], "never matches xyz"]}) + "\n")
    conn = sqlite3.connect(d / "paras.db")
    conn.execute("CREATE TABLE documents (id PRIMARY KEY, text)")
    conn.executemany("INSERT INTO documents VALUES (?,?)", docs)
    conn.commit()
    conn.close()
    return d


def test_recall_lines_match_cpu_path(gpu_device, corpus, capsys):
    from proqa_amd import eval_retrieval, gen_index_id_map, get_embed
    d = corpus
    common = ["--do_predict", "--bert_model_name", str(d / "small-bert"), "--fp16", "--init_checkpoint",
              str(d / "ckpt.pt"), "--eval-workers", "8", "--predict_batch_size", "512"]
    para = get_embed.main(common + ["--predict_file", str(d / "paras.txt"), "--embed_save_path", str(d / "para_embed.npy")])
    qry = get_embed.main(common + ["--predict_file", str(d / "qa.txt"), "--is_query_embed", "--embed_save_path",
                                   str(d / "q_embed.npy")])
    gen_index_id_map.build(str(d / "paras.txt"), str(d / "idx_id.json"), sidecar=True)
    capsys.readouterr()
    lines = eval_retrieval.main([str(d / "qa.txt"), para, qry, str(d / "paras.db"), "--topk", str(TOPK),
                                 "--num-workers", "4", "--idx-id-map", str(d / "idx_id.json")])
    got = {int(re.match(r"Top (\d+) Recall", ln).group(1)): float(ln.split(": ")[1].split(" ")[0]) for ln in lines}
    assert list(got) == [80, 5, 10, 20, 50]

    xb, xq = np.load(para), np.load(qry)
    assert xb.shape == (N_DOCS, 128) and xq.shape == (N_QA, 128)
    D, I = search_oracle.topk_ip(xq, xb, TOPK)            # the CPU path on the same embeddings
    qas = [json.loads(ln) for ln in open(d / "qa.txt")]
    eval_retrieval.init(str(d / "paras.db"))
    doc_ids = eval_retrieval.convert_idx2id(I, str(d / "idx_id.ids"))     # via the binary sidecar
    res = [eval_retrieval.get_score((qa["question"], qa["answer"], ids), topk=TOPK) for qa, ids in zip(qas, doc_ids)]
    want = {int(k): float(np.mean([r[k] for r in res])) for k in res[0]}
    for k in (5, 20, 80, 10, 50):                                         # Recall@{5,20,80} of BASELINE.json (+10, 50)
        assert abs(got[k] - want[k]) <= 1e-4, (k, got[k], want[k])
    assert 0.0 < want[80] <= 1.0                                          # planted answers are being found


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_cli_under_two_ranks_prints_the_same_lines(gpu_device, corpus, tmp_path):
    """The drop-in command line under a torchrun-style launch (WORLD_SIZE=2; both ranks share the one GPU of the test
    box, gloo carries the exchange because RCCL refuses two ranks on one device): every rank loads only its half of
    para_embed.npy, rank 0 prints.  stdout byte-identical to the single-process run, D and I identical."""
    d = corpus
    para, qry = str(d / "para_embed.npy"), str(d / "q_embed.npy")
    if not (os.path.exists(para) and os.path.exists(qry)):
        pytest.skip("needs the files of test_recall_lines_match_cpu_path")
    cmd = [sys.executable, os.path.join(ROOT, "eval_retrieval.py"), str(d / "qa.txt"), para, qry, str(d / "paras.db"),
           "--topk", str(TOPK), "--num-workers", "4", "--idx-id-map", str(d / "idx_id.json")]
    single = subprocess.run(cmd + ["--dump-results", str(tmp_path / "single.npz")], env=os.environ.copy(),
                            capture_output=True, text=True, timeout=900)
    assert single.returncode == 0, single.stderr[-2000:]
    env = dict(os.environ, PROQA_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen(cmd + ["--dump-results", str(tmp_path / "sharded.npz")],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    # (the gloo transport of this test announces its connections on stdout from C++; RCCL does not)
    printed = ["".join(ln for ln in o[0].splitlines(True) if not ln.startswith("[Gloo]")) for o in outs]
    assert printed[0] == single.stdout and len(single.stdout.splitlines()) == 5
    assert printed[1] == ""                                   # only rank 0 prints
    a, b = np.load(tmp_path / "single.npz"), np.load(tmp_path / "sharded.npz")
    np.testing.assert_array_equal(a["I"], b["I"])
    np.testing.assert_array_equal(a["D"], b["D"])
    # --shard queries: every rank loads the whole file and searches its half of the questions; the same lines, the same result
    env["MASTER_PORT"] = str(_free_port())
    procs = [subprocess.Popen(cmd + ["--shard", "queries", "--dump-results", str(tmp_path / "qsharded.npz")],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    printed = ["".join(ln for ln in o[0].splitlines(True) if not ln.startswith("[Gloo]")) for o in outs]
    assert printed[0] == single.stdout and printed[1] == ""
    c = np.load(tmp_path / "qsharded.npz")
    np.testing.assert_array_equal(a["I"], c["I"])
    np.testing.assert_array_equal(a["D"], c["D"])
