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


def _reference_para_has_answer(answer, para):
    """the reference's loop, verbatim in behaviour (eval_retrieval.py:27-45): tokenise everything, compare runs of words"""
    text = ev.PROCESS_TOK.tokenize(normalize(para)).words(uncased=True)
    for alias in answer:
        needle = ev.PROCESS_TOK.tokenize(normalize(alias)).words(uncased=True)
        for i in range(len(text) - len(needle) + 1):
            if text[i:i + len(needle)] == needle:
                return True
    return False


def test_the_substring_prefilter_never_changes_a_verdict(gold):
    """para_has_answer tokenises a paragraph only if some alias has all its tokens in it as substrings of the folded text.
    10 000 random (answer, paragraph) pairs over an alphabet built to hurt: Greek capitals with both lower-case sigmas (the
    one context-sensitive lower-case rule), the dotted capital I (lower-cases to two characters), combining marks (NFD),
    apostrophes inside words, empty and punctuation-only aliases."""
    rng = np.random.default_rng(5)
    alphabet = list("abAB \u03a3\u03c3\u03c2\u039f\u0394\u0130\u00e9e\u0301'.-\u4e2d") + ["  "]
    def rand_text(n):
        return "".join(rng.choice(alphabet, size=n))
    same = 0
    for _ in range(10000):
        para = rand_text(int(rng.integers(0, 40)))
        answer = [rand_text(int(rng.integers(0, 5))) for _ in range(int(rng.integers(1, 4)))]
        if rng.random() < 0.3 and len(para) > 4:        # plant an alias cut out of the paragraph (often a real match)
            a = int(rng.integers(0, len(para) - 2))
            answer.append(para[a:a + int(rng.integers(1, 6))].swapcase())
        want = _reference_para_has_answer(answer, para)
        assert ev.para_has_answer(answer, para) == want, (answer, para)
        assert ev.para_has_answer(answer, para, True)[0] == want
        same += want
    assert 500 < same < 9500     # both verdicts are exercised


def test_text_sidecar_scores_like_the_sqlite_route(gold, tmp_path):
    """SURVEY section 8f row 3, second half: the passage texts by ROW from the memory-mapped sidecar give the score dicts the
    reference's route (row -> doc id -> sqlite) gives, and SidecarMap.take is the per-id lookup, vectorised."""
    from proqa_amd import gen_index_id_map as gm
    corpus = tmp_path / "paras.txt"
    corpus.write_text("".join(json.dumps({"id": d[0], "text": d[1]}) + "\n" for d in gold["docs"]))
    out = tmp_path / "idx_id.json"
    gm.build(str(corpus), str(out), texts=True)
    txt = gm.text_sidecar_of(str(out))
    assert txt == str(tmp_path / "idx_id.txt") and gm.text_sidecar_of(str(tmp_path / "idx_id.ids")) == txt
    texts = gm.TextSidecar(txt)
    assert [texts[i] for i in range(len(texts))] == [d[1] for d in gold["docs"]]
    with pytest.raises(KeyError):
        texts[len(texts)]
    # rows of the golden I matrix index gold["idx_id"]; map them to rows of THIS corpus (= position of the doc in gold["docs"])
    pos = {normalize(d[0]): i for i, d in enumerate(gold["docs"])}
    ev.init(os.path.join(GOLDEN, "recall_docs.db"), txt)
    try:
        for topk, want in gold["scores"].items():
            for qa, ids, w in zip(gold["qas"], gold["doc_ids"], want):
                rows = [pos[normalize(i)] for i in ids]
                assert ev.get_score_rows((qa["question"], qa["answer"], rows), topk=int(topk)) == w
    finally:
        ev.init(os.path.join(GOLDEN, "recall_docs.db"))
    m = gm.SidecarMap(str(tmp_path / "idx_id.ids"))
    rows = np.random.default_rng(1).integers(0, len(m), (7, 5))
    assert m.take(rows) == [[m[int(i)] for i in r] for r in rows]
    assert m.take(np.zeros((0, 3), np.int64)) == []


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
