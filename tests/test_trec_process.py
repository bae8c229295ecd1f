"""Host side of proqa_amd/trec_process.py against golden outputs of the reference's trec_process.py
(tests/golden/trec_golden.json, made by tests/golden/make_trec_golden.py)."""
import json
import os

import numpy as np
import pytest

from proqa_amd import trec_process

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "trec_golden.json")))


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def test_prepare_corpus_matches_reference_output(tmp_path):
    src = _write(tmp_path, "collection.tsv", GOLDEN["collection_tsv"])
    dst = str(tmp_path / "paras.txt")
    trec_process.prepare_corpus(src, dst)
    assert open(dst).read() == GOLDEN["prepare_corpus"]


def test_prepare_corpus_rejects_a_malformed_line_like_the_reference(tmp_path):
    src = _write(tmp_path, "collection.tsv", "0\ttext\textra\n")
    with pytest.raises(ValueError):
        trec_process.prepare_corpus(src, str(tmp_path / "out.txt"))


def test_extract_labels_matches_reference_output(tmp_path, capsys):
    qrels = _write(tmp_path, "qrels.tsv", GOLDEN["qrels_tsv"])
    queries = _write(tmp_path, "queries.tsv", GOLDEN["queries_tsv"])
    dst = str(tmp_path / "train.txt")
    trec_process.extract_labels(input=qrels, output=dst, queries=queries)
    assert open(dst).read() == GOLDEN["extract_labels"]
    assert capsys.readouterr().out == GOLDEN["extract_labels_stdout"]


def test_extract_labels_unknown_query_id_is_a_key_error(tmp_path):
    qrels = _write(tmp_path, "qrels.tsv", "99\t0\t1\t1\n")
    queries = _write(tmp_path, "queries.tsv", "1\ta query\n")
    with pytest.raises(KeyError):
        trec_process.extract_labels(input=qrels, output=str(tmp_path / "o.txt"), queries=queries)


def test_label_rows():
    I = np.array([5, 9, -1, 2, 9], dtype=np.int64)
    assert trec_process.label_rows(I, [9, 2]).tolist() == [0, 1, 0, 1, 1]
    assert trec_process.label_rows(I, []).tolist() == [0, 0, 0, 0, 0]
    assert trec_process.label_rows(I, [-1]).tolist() == [0, 0, 1, 0, 0]   # the reference's `in` would say the same
