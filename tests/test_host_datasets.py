"""EmDataset / em_collate / collate_tokens against the reference's outputs (tokenize_golden.json)."""
import json
import os

import pytest
import torch

from proqa_amd import datasets

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def tokenizer(tmp_path_factory):
    # the same construction path as get_embed.py: BertTokenizer.from_pretrained(<model dir>)
    import shutil
    from transformers import BertTokenizer
    d = tmp_path_factory.mktemp("model")
    shutil.copy(os.path.join(GOLDEN, "vocab_small.txt"), d / "vocab.txt")
    return BertTokenizer.from_pretrained(str(d))


def test_emdataset_and_collate_match_reference(tokenizer, tmp_path):
    with open(os.path.join(GOLDEN, "tokenize_golden.json")) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        key = "question" if case["is_query"] else "text"
        path = tmp_path / "in.jsonl"
        path.write_text("".join(json.dumps({key: t, "id": 0}) + "\n" for t in gold["texts"]))
        ds = datasets.EmDataset(tokenizer, str(path), case["max_query_length"], case["max_length"], case["is_query"])
        assert len(ds) == len(gold["texts"])
        samples = [ds[i] for i in range(len(ds))]
        assert [int(s["input_ids"].numel()) for s in samples] == case["item_lengths"]
        assert all(s["input_mask"].dtype == torch.bool and bool(s["input_mask"].all()) for s in samples)
        batch = datasets.em_collate(samples)
        assert batch["input_ids"].dtype == torch.int64
        assert batch["input_ids"].tolist() == case["input_ids"]
        assert batch["input_mask"].int().tolist() == case["input_mask"]


def test_collate_edge_cases():
    assert datasets.em_collate([]) == {}
    a, b = torch.tensor([1, 2, 3]), torch.tensor([4])
    assert datasets.collate_tokens([a, b], 0).tolist() == [[1, 2, 3], [4, 0, 0]]
    assert datasets.collate_tokens([a, b], 9, left_pad=True).tolist() == [[1, 2, 3], [9, 9, 4]]
    assert datasets.collate_tokens([a], 0, eos_idx=3, move_eos_to_beginning=True).tolist() == [[3, 1, 2]]
