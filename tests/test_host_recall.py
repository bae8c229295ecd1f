"""Recall scorer (para_has_answer / get_score / convert_idx2id / SimpleTokenizer / DocDB) against
outputs of the reference's own functions (recall_golden.json + recall_docs.db)."""
import json
import os

import numpy as np
import pytest

from proqa_amd import eval_retrieval as ev
from proqa_amd.basic_tokenizer import SimpleTokenizer
from proqa_amd.utils import DocDB, normalize

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLDEN, "recall_golden.json")) as f:
        g = json.load(f)
    ev.init(os.path.join(GOLDEN, "recall_docs.db"))
    return g


def test_tokenizer_matches_reference(gold):
    tok = SimpleTokenizer()
    for (doc_id, text), want in zip(gold["docs"], gold["tokenized"]):
        got = [[t[0], t[1], list(t[2])] for t in tok.tokenize(normalize(text)).data]
        assert got == want, doc_id


def test_para_has_answer_matrix(gold):
    for qa, row in zip(gold["qas"], gold["matched"]):
        for (doc_id, text), (hit, span) in zip(gold["docs"], row):
            assert list(ev.para_has_answer(qa["answer"], text, True)) == [hit, span], (qa, doc_id)
            assert ev.para_has_answer(qa["answer"], text) == hit


def test_docdb_lookup(gold):
    db = DocDB(os.path.join(GOLDEN, "recall_docs.db"))
    for doc_id, text in gold["docs"]:
        assert db.get_doc_text(doc_id) == text
    assert db.get_doc_text("missing") is None
    assert sorted(db.get_doc_ids()) == sorted(normalize(d[0]) for d in gold["docs"])
    db.close()


def test_convert_idx2id_and_scores(gold, tmp_path):
    p = tmp_path / "idx_id.json"
    p.write_text(json.dumps(gold["idx_id"]))
    doc_ids = ev.convert_idx2id(np.array(gold["I"], dtype=np.int64), str(p))
    assert doc_ids == gold["doc_ids"]
    for topk, want in gold["scores"].items():
        got = [ev.get_score((qa["question"], qa["answer"], ids), topk=int(topk))
               for qa, ids in zip(gold["qas"], doc_ids)]
        assert got == want
        assert list(got[0].keys()) == list(dict.fromkeys([topk, "5", "10", "20", "50"]))


def test_default_idx_id_path_is_the_references(gold, tmp_path, monkeypatch):
    (tmp_path / "pretrained_models").mkdir()
    (tmp_path / "run").mkdir()
    (tmp_path / "pretrained_models" / "idx_id.json").write_text(json.dumps(gold["idx_id"]))
    monkeypatch.chdir(tmp_path / "run")
    assert ev.convert_idx2id(gold["I"]) == gold["doc_ids"]


def test_cli_parser_defaults():
    a = ev.build_parser().parse_args(["qa.txt", "idx.npy", "q.npy", "paras.db"])
    assert (a.topk, a.num_workers, a.idx_id_map) == (80, 10, "../pretrained_models/idx_id.json")
    a = ev.build_parser().parse_args(["qa.txt", "idx.npy", "q.npy", "paras.db", "--topk", "8", "--num-workers", "2"])
    assert (a.topk, a.num_workers) == (8, 2)


def test_sidecar_map_equals_json_route(gold, tmp_path):
    """the binary sidecar returns exactly what idx_id.json returns."""
    from proqa_amd import gen_index_id_map as gm
    corpus = tmp_path / "paras.txt"
    docs = gold["docs"] + [["unié-\u00e9", "x"], [12345, "int id"], ["with \"quote\"", "y"]]
    corpus.write_text("".join(json.dumps({"id": d[0], "text": d[1]}) + "\n" for d in docs))
    out = tmp_path / "idx_id.json"
    n = gm.build(str(corpus), str(out), sidecar=True)
    assert n == len(docs)
    assert json.load(open(out)) == {str(i): d[0] for i, d in enumerate(docs)}
    rows = np.array([[0, 5, n - 1], [n - 2, 3, n - 3]])
    via_json = ev.convert_idx2id(rows, str(out))
    via_sidecar = ev.convert_idx2id(rows, str(tmp_path / "idx_id.ids"))
    assert via_json == via_sidecar
    with pytest.raises(KeyError):
        ev.convert_idx2id(np.array([[n]]), str(tmp_path / "idx_id.ids"))
    with pytest.raises(KeyError):
        ev.convert_idx2id(np.array([[n]]), str(out))


def test_eval_retrieval_command_line_does_not_import_pytorch():
    """The single-process eval_retrieval.py path needs numpy, sqlite and the C library only: importing PyTorch costs a second
    or two of a ~2 s command and makes the scorer pool's fork heavier.  (The row-sharded path under torchrun imports it.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import proqa_amd.eval_retrieval as e, proqa_amd._lib as l; l.load(); "
            "e.finish_distributed(); assert 'torch' not in sys.modules, 'torch was imported'" % root)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
