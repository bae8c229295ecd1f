"""trec_process.retrieve_topk (the k = 10000 caller of the search) on the GPU against the golden made by the reference's
own retrieve_topk (tests/golden/make_trec_golden.py): byte-identical output file, identical printed line."""
import hashlib
import json
import os
import socket
import subprocess
import sys

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


def test_retrieve_topk_under_two_ranks_writes_the_same_file(gpu_device, tmp_path):
    """retrieve_topk under a torchrun-style launch (2 ranks on the one GPU, gloo): the corpus is row-sharded, the
    k = 10000 lists of both shards are merged, rank 0 writes -- the reference's golden bytes again."""
    paras, queries, qfile = trec_inputs(str(tmp_path), n=GOLDEN["n"], nq=GOLDEN["nq"], seed=GOLDEN["seed"])
    out = str(tmp_path / "processed.txt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from proqa_amd import trec_process; "
            "trec_process.retrieve_topk(index_path=%r, query_embeds=%r, query_input=%r, output=%r)" % (root, paras, queries, qfile, out))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PROQA_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    printed = ["".join(ln for ln in o[0].splitlines(True) if not ln.startswith("[Gloo]")) for o in outs]   # gloo's own chatter
    assert printed[0] == GOLDEN["stdout"] and printed[1] == ""
    blob = open(out, "rb").read()
    assert len(blob) == GOLDEN["output_bytes"]
    assert hashlib.sha256(blob).hexdigest() == GOLDEN["output_sha256"]
