"""trec_process.retrieve_topk (the k = 10000 caller of the search) on the GPU against the golden made by the reference's
own retrieve_topk (tests/golden/make_trec_golden.py): byte-identical output file, identical printed line."""
import hashlib
import json
import os

import pytest

from trec_inputs import trec_inputs

pytestmark = pytest.mark.gpu
GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "trec_golden.json")))["retrieve_topk"]


def test_retrieve_topk_output_is_byte_identical_to_the_reference(gpu_device, tmp_path, capsys):
    from proqa_amd import trec_process
    paras, queries, qfile = trec_inputs(str(tmp_path), n=GOLDEN["n"], nq=GOLDEN["nq"], seed=GOLDEN["seed"])
    out = str(tmp_path / "processed.txt")
    recall = trec_process.retrieve_topk(index_path=paras, query_embeds=queries, query_input=qfile, output=out)
    assert capsys.readouterr().out == GOLDEN["stdout"]
    blob = open(out, "rb").read()
    first = json.loads(blob.split(b"\n")[0])
    assert list(first.keys()) == GOLDEN["first_keys"]
    assert first["para_embed_idx"][:16] == GOLDEN["first_rows"]
    assert sum(first["para_labels"]) == GOLDEN["first_label_sum"]
    assert len(blob) == GOLDEN["output_bytes"]
    assert hashlib.sha256(blob).hexdigest() == GOLDEN["output_sha256"]
    assert f"Avg recall: {recall}\n" == GOLDEN["stdout"]


def test_retrieve_topk_with_fewer_rows_than_k(gpu_device, tmp_path):
    """12 queries over 300 rows: every row is returned once, the tail is -1 and carries no label."""
    from proqa_amd import trec_process
    paras, queries, qfile = trec_inputs(str(tmp_path), n=300, nq=12, seed=5)
    out = str(tmp_path / "processed.txt")
    trec_process.retrieve_topk(index_path=paras, query_embeds=queries, query_input=qfile, output=out)
    for line in open(out):
        s = json.loads(line)
        rows = s["para_embed_idx"]
        assert len(rows) == 10000 and sorted(rows[:300]) == list(range(300)) and set(rows[300:]) == {-1}
        assert sum(s["para_labels"]) == len(set(s["labels"])) and not any(s["para_labels"][300:])
