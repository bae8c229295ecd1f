"""Inference dataset + collate of the encode path.

Mirrors /root/reference/retrieval/datasets.py: collate_tokens (:29-45), EmDataset (:257-295),
em_collate (:298-305).  Tokenisation is transformers' BertTokenizer, exactly as in the reference.
"""
import json

import torch
from torch.utils.data import Dataset


def collate_tokens(values, pad_idx, eos_idx=None, left_pad=False, move_eos_to_beginning=False):
    """Stack 1-D tensors into a [len(values), max_len] tensor padded with pad_idx."""
    width = max(v.size(0) for v in values)
    out = values[0].new_full((len(values), width), pad_idx)
    for row, v in zip(out, values):
        n = v.size(0)
        dst = row[width - n:] if left_pad else row[:n]
        if move_eos_to_beginning:
            assert v[-1] == eos_idx
            dst[0] = eos_idx
            dst[1:] = v[:-1]
        else:
            dst.copy_(v)
    return out


class EmDataset(Dataset):
    """JSON-lines file -> {'input_ids': LongTensor[len], 'input_mask': BoolTensor[len]}.

    Queries read key 'question' and are truncated to max_query_length; passages read key 'text'
    and are truncated to max_length (both counts include [CLS] and [SEP]).
    """

    def __init__(self, tokenizer, data_path, max_query_length, max_length, is_query_embed):
        super().__init__()
        self.tokenizer = tokenizer
        self.is_query_embed = is_query_embed
        print(f"Loading data from {data_path}")
        with open(data_path) as f:
            self.data = [json.loads(line.strip()) for line in f.readlines()]
        self.max_length = max_query_length if is_query_embed else max_length
        print(f"Max sequence length: {self.max_length}")

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        sample = self.data[index]
        sent = sample["question"] if self.is_query_embed else sample["text"]
        ids = self.tokenizer.encode(sent, max_length=self.max_length, truncation=True)
        sent_ids = torch.LongTensor(ids)
        return {"input_ids": sent_ids, "input_mask": torch.ones(sent_ids.shape).bool()}


def em_collate(samples):
    if len(samples) == 0:
        return {}
    return {
        "input_ids": collate_tokens([s["input_ids"] for s in samples], 0),
        "input_mask": collate_tokens([s["input_mask"] for s in samples], 0),
    }
