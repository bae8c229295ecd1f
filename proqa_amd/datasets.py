"""Inference dataset + collate of the encode path.

Mirrors /root/reference/retrieval/datasets.py: collate_tokens (:29-45), EmDataset (:257-295),
em_collate (:298-305).  Tokenisation is transformers' BertTokenizer, exactly as in the reference.

`EmDataset` + `em_collate` are the reference's per-item shapes.  `get_embed.py` feeds the GPU through `EmTextView` +
`TokenizeCollate` instead: the same ids and masks (checked against the per-item path and the reference's golden batches in
tests/test_host_datasets.py), produced one BATCH at a time inside the DataLoader workers -- one call into the tokenizer's
batch entry point, ids written straight into the padded [B, L] array, the valid lengths handed over as a host list -- so
that a worker ships three arrays per batch instead of 2 x 512 small tensors and the consumer thread does no reduction.
"""
import json
import os

import numpy as np
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


class EmTextView(Dataset):
    """The sentences of an EmDataset, untokenised: item i is the string EmDataset.__getitem__(i) would encode."""

    def __init__(self, dataset):
        self.data = dataset.data
        self.key = "question" if dataset.is_query_embed else "text"

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        return self.data[index][self.key]


class TokenizeCollate:
    """collate_fn over strings: tokenizer.encode(sent, max_length=L, truncation=True) for every sentence of the batch
    (datasets.py:285-286) and the right-padding of em_collate (:298-305) in one step.

    Returns {'input_ids': int64 [B, Lmax] (pad 0), 'input_mask': bool [B, Lmax], 'seq_lens': list[int]} -- the first two
    are exactly em_collate([EmDataset[i] ...]); 'seq_lens' spares the consumer the mask reduction.
    With a tokenizer that is backed by the `tokenizers` library the batch goes through its encode_batch on a private
    copy (own truncation setting, one thread per DataLoader worker: the workers are the parallelism); any other tokenizer
    is called sentence by sentence, as the reference does."""

    def __init__(self, tokenizer, max_length):
        self.tokenizer = tokenizer
        self.max_length = int(max_length)
        self._backend_json = None
        backend = getattr(tokenizer, "backend_tokenizer", None) or getattr(tokenizer, "_tokenizer", None)
        if backend is not None and hasattr(backend, "encode_batch") and hasattr(backend, "to_str"):
            self._backend_json = backend.to_str()
        self._backend = None

    def __getstate__(self):      # the private backend is rebuilt in every worker
        state = dict(self.__dict__)
        state["_backend"] = None
        return state

    def _encode(self, texts):
        if self._backend_json is None:
            return [self.tokenizer.encode(t, max_length=self.max_length, truncation=True) for t in texts]
        if self._backend is None:
            from tokenizers import Tokenizer
            os.environ.setdefault("TOKENIZERS_PARALLELISM", "false")
            self._backend = Tokenizer.from_str(self._backend_json)
            self._backend.enable_truncation(max_length=self.max_length)
            self._backend.no_padding()
        return [e.ids for e in self._backend.encode_batch(list(texts), add_special_tokens=True)]

    def __call__(self, texts):
        if len(texts) == 0:
            return {}
        ids = self._encode(texts)
        lens = [len(x) for x in ids]
        width = max(lens)
        out = np.zeros((len(ids), width), dtype=np.int64)
        for row, x in zip(out, ids):
            row[:len(x)] = x
        mask = np.arange(width)[None, :] < np.asarray(lens)[:, None]
        return {"input_ids": torch.from_numpy(out), "input_mask": torch.from_numpy(mask), "seq_lens": lens}
