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


class JsonlTexts(Dataset):
    """The sentences of a JSON-lines file, parsed on demand: item i is what EmTextView(EmDataset(...))[i] is.

    EmDataset parses every line before the first batch can be encoded (datasets.py:271-272; 3.5 us per passage: a minute
    for an 18M-passage corpus, with the GPU idle).  Here the file is read once as raw lines -- its length is known at once,
    which the row sharding and the pre-sized output need -- and a line is parsed when the loader's producer thread asks
    for it, i.e. beside the encoding of the batches before it.  A malformed line raises there, as it does in EmDataset."""

    def __init__(self, data_path, max_query_length, max_length, is_query_embed):
        self.is_query_embed = is_query_embed
        self.key = "question" if is_query_embed else "text"
        print(f"Loading data from {data_path}")
        # (raw bytes: json.loads takes them as they are, and the native tokenizer reads the member out of the line itself)
        with open(data_path, "rb") as f:
            self.lines = f.readlines()
        self.max_length = max_query_length if is_query_embed else max_length
        print(f"Max sequence length: {self.max_length}")

    def __len__(self):
        return len(self.lines)

    def __getitem__(self, index):
        return json.loads(self.lines[index].strip())[self.key]


class TokenizeCollate:
    """collate_fn over strings: tokenizer.encode(sent, max_length=L, truncation=True) for every sentence of the batch
    (datasets.py:285-286) and the right-padding of em_collate (:298-305) in one step.

    Returns {'input_ids': int64 [B, Lmax] (pad 0), 'input_mask': bool [B, Lmax], 'seq_lens': list[int]} -- the first two
    are exactly em_collate([EmDataset[i] ...]); 'seq_lens' spares the consumer the mask reduction.
    With a tokenizer that is backed by the `tokenizers` library the batch goes through its encode_batch on a private
    copy (own truncation setting, one thread per DataLoader worker: the workers are the parallelism); any other tokenizer
    is called sentence by sentence, as the reference does."""

    def __init__(self, tokenizer, max_length, parallel=False, native_threads=0):
        """parallel: let the tokenizer library spread a batch over its own thread pool (TextBatchLoader: one producer
        thread in the process that feeds the GPU); off inside DataLoader workers, which are the parallelism there.
        native_threads > 0: sentences go through libproqa_hip.so's own WordPiece (proqa_wordpiece_*, that many threads: ids
        written straight into the batch array, no Python object per token); the sentences it declines (a '[', characters
        beyond the Basic Multilingual Plane) -- and every sentence if the tokenizer is not a plain BERT WordPiece one --
        go through the tokenizer itself, so the batch is the reference's either way."""
        self.tokenizer = tokenizer
        self.max_length = int(max_length)
        self.parallel = bool(parallel)
        self.native_threads = int(native_threads)
        self._native = None
        self._native_spec = self._native_vocab(tokenizer) if self.native_threads > 0 else None
        self._backend_json = None
        backend = getattr(tokenizer, "backend_tokenizer", None) or getattr(tokenizer, "_tokenizer", None)
        if backend is not None and hasattr(backend, "encode_batch") and hasattr(backend, "to_str"):
            self._backend_json = backend.to_str()
        self._backend = None

    def __getstate__(self):      # the private backend / native handle is rebuilt in every worker
        state = dict(self.__dict__)
        state["_backend"] = None
        state["_native"] = None
        return state

    @staticmethod
    def _native_vocab(tokenizer):
        """(vocab bytes in id order, do_lower_case) if `tokenizer` is a BERT WordPiece tokenizer the native code restates
        (contiguous ids, [UNK]/[CLS]/[SEP] present, '##' continuation prefix, default clean-up), else None."""
        try:
            vocab = tokenizer.get_vocab()
            toks = sorted(vocab, key=vocab.get)
            if [vocab[t] for t in toks] != list(range(len(toks))) or any("\n" in t for t in toks):
                return None
            if not {"[UNK]", "[CLS]", "[SEP]"} <= set(vocab):
                return None
            if (tokenizer.unk_token, tokenizer.cls_token, tokenizer.sep_token) != ("[UNK]", "[CLS]", "[SEP]"):
                return None
            backend = getattr(tokenizer, "backend_tokenizer", None) or getattr(tokenizer, "_tokenizer", None)
            if backend is not None:
                import json as _json
                spec = _json.loads(backend.to_str())
                norm, model = spec.get("normalizer") or {}, spec.get("model") or {}
                if norm.get("type") != "BertNormalizer" or not norm.get("clean_text", True) or \
                        not norm.get("handle_chinese_chars", True):
                    return None
                # accents are stripped iff the model is uncased (strip_accents None follows lowercase): the only
                # combination the generated tables (scripts/gen_wordpiece_tables.py) describe
                if norm.get("strip_accents") not in (None, bool(norm.get("lowercase", True))):
                    return None
                # tokens added on top of vocab.txt are matched in the raw text before anything else: only the five
                # bracketed special tokens are expected (a text with a '[' goes to the tokenizer itself anyway)
                if any("[" not in (a.get("content") or "") for a in spec.get("added_tokens") or []):
                    return None
                if model.get("type") != "WordPiece" or model.get("continuing_subword_prefix") != "##" or \
                        model.get("max_input_chars_per_word", 100) != 100 or model.get("unk_token") != "[UNK]":
                    return None
                if (spec.get("pre_tokenizer") or {}).get("type") != "BertPreTokenizer":
                    return None
                lower = bool(norm.get("lowercase", True))
            else:
                # a pure-Python BertTokenizer: the defaults of its BasicTokenizer only, and no added tokens
                if not getattr(tokenizer, "do_basic_tokenize", True) or getattr(tokenizer, "added_tokens_encoder", None):
                    return None
                basic = getattr(tokenizer, "basic_tokenizer", None)
                if basic is not None and (getattr(basic, "never_split", None) or not getattr(basic, "tokenize_chinese_chars", True)
                                          or getattr(basic, "strip_accents", None) not in (None, bool(getattr(basic, "do_lower_case", True)))):
                    return None
                # the generated tables restate the tokenizers library's BertNormalizer; the pure-Python BasicTokenizer differs from
                # it beyond ASCII (str.lower() applies the final-sigma rule, NFC before anything else, another Unicode data
                # version): the native path takes ASCII texts only, everything else goes to the tokenizer itself (bit 1)
                lower = int(bool(getattr(tokenizer, "do_lower_case", True))) | 2
            return ("\n".join(toks).encode("utf-8"), int(lower))
        except Exception:
            return None

    def _native_handle(self):
        if self._native is None and self._native_spec is not None:
            import ctypes
            from . import _lib
            lib = _lib.load()
            h = ctypes.c_void_p()
            blob, lower = self._native_spec
            _lib.check(lib.proqa_wordpiece_create(blob, len(blob), int(lower), ctypes.byref(h)))
            self._native = (lib, h)
        return self._native

    def _encode(self, texts):
        if self._backend_json is None:
            return [self.tokenizer.encode(t, max_length=self.max_length, truncation=True) for t in texts]
        if self._backend is None:
            from tokenizers import Tokenizer
            os.environ["TOKENIZERS_PARALLELISM"] = "true" if self.parallel else "false"
            self._backend = Tokenizer.from_str(self._backend_json)
            self._backend.enable_truncation(max_length=self.max_length)
            self._backend.no_padding()
        if os.environ.get("PROQA_LOADER_DEBUG"):
            import time
            t0 = time.perf_counter()
            enc = self._backend.encode_batch(list(texts), add_special_tokens=True)
            t1 = time.perf_counter()
            out = [e.ids for e in enc]
            t2 = time.perf_counter()
            self._dbg = getattr(self, "_dbg", [0.0, 0.0, 0])
            self._dbg[0] += t1 - t0
            self._dbg[1] += t2 - t1
            self._dbg[2] += 1
            if self._dbg[2] % 100 == 0:
                import sys
                print(f"[loader] {self._dbg[2]} batches: encode_batch {self._dbg[0]:.2f} s, ids lists {self._dbg[1]:.2f} s", file=sys.stderr)
            return out
        return [e.ids for e in self._backend.encode_batch(list(texts), add_special_tokens=True)]

    def _call_native(self, texts):
        """The batch through proqa_wordpiece_encode_batch; sentences it declines (a '[', non-BMP characters) through _encode."""
        import ctypes
        lib, h = self._native_handle()
        n, L = len(texts), self.max_length
        raw = [t.encode("utf-8") for t in texts]
        ptrs = (ctypes.c_char_p * n)(*raw)
        sizes = np.fromiter(map(len, raw), dtype=np.int64, count=n)
        ids = np.empty((n, L), dtype=np.int64)
        lens = np.empty(n, dtype=np.int32)
        from . import _lib
        _lib.check(lib.proqa_wordpiece_encode_batch(h, ptrs, sizes.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n, L,
                                                    ids.ctypes.data, lens.ctypes.data, self.native_threads))
        return self._finish_native(ids, lens, lambda i: texts[i])

    def call_lines(self, lines, key):
        """The batch straight from JSON-lines records (bytes): == self([json.loads(l)[key] for l in lines]).  With the
        native tokenizer the member is read out of every record by the library (proqa_wordpiece_encode_jsonl_batch: no
        Python object per passage); records it does not take (-2) are parsed here, texts it declines (-1) are tokenised by
        the tokenizer itself."""
        if self._native_spec is None or len(lines) == 0:
            return self([json.loads(line.strip())[key] for line in lines])
        import ctypes
        from . import _lib
        lib, h = self._native_handle()
        n, L = len(lines), self.max_length
        ptrs = (ctypes.c_char_p * n)(*lines)
        sizes = np.fromiter(map(len, lines), dtype=np.int64, count=n)
        ids = np.empty((n, L), dtype=np.int64)
        lens = np.empty(n, dtype=np.int32)
        _lib.check(lib.proqa_wordpiece_encode_jsonl_batch(h, ptrs, sizes.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n,
                                                          key.encode("utf-8"), L, ids.ctypes.data, lens.ctypes.data,
                                                          self.native_threads))
        return self._finish_native(ids, lens, lambda i: json.loads(lines[i].strip())[key])

    def _finish_native(self, ids, lens, text_of):
        rest = np.nonzero(lens < 0)[0]
        if len(rest):
            for i, x in zip(rest.tolist(), self._encode([text_of(i) for i in rest.tolist()])):
                ids[i, :len(x)] = x
                ids[i, len(x):] = 0
                lens[i] = len(x)
        width = int(lens.max())
        out = np.ascontiguousarray(ids[:, :width])
        mask = np.arange(width)[None, :] < lens[:, None]
        return {"input_ids": torch.from_numpy(out), "input_mask": torch.from_numpy(mask), "seq_lens": lens.tolist()}

    def __call__(self, texts):
        if len(texts) == 0:
            return {}
        if self._native_spec is not None:
            return self._call_native(texts)
        ids = self._encode(texts)
        lens = [len(x) for x in ids]
        width = max(lens)
        out = np.zeros((len(ids), width), dtype=np.int64)
        for row, x in zip(out, ids):
            row[:len(x)] = x
        mask = np.arange(width)[None, :] < np.asarray(lens)[:, None]
        return {"input_ids": torch.from_numpy(out), "input_mask": torch.from_numpy(mask), "seq_lens": lens}


class TextBatchLoader:
    """Tokenised batches of a sequence of sentences, produced by a background thread of THIS process.

    The reference feeds its GPU from `DataLoader(num_workers=32)` processes (retrieval/get_embed.py:93-96).  Worker
    processes cost a fork of a process that holds the model and a pickle + shared-memory hop per batch.  The WordPiece
    tokenizer is native code that releases the GIL and spreads a
    batch over its own threads, so one producer thread here keeps `prefetch` collated batches ahead of the consumer: no
    processes, no copies between them.  The producer (and the tokenizer pool it starts) runs at a lower priority than the
    thread feeding the GPU.  Yields what `collate` returns, in order; an exception in the producer is re-raised here."""

    def __init__(self, texts, batch_size, collate, prefetch=8, lo=0, hi=None, producers=1):
        """producers: threads that take turns on the batches (batch i belongs to producer i % producers; the consumer
        reads their queues round-robin, so the order is kept).  One is the default: two concurrent batch calls share the
        tokenizer's thread pool and measured slower (28 k vs 38 k passages/s on 14 threads)."""
        self.texts, self.batch_size, self.collate, self.prefetch = texts, int(batch_size), collate, int(prefetch)
        self.lo, self.hi = int(lo), len(texts) if hi is None else int(hi)
        self.producers = max(1, int(producers))

    def __len__(self):
        return max(0, -(-(self.hi - self.lo) // self.batch_size))

    def __iter__(self):
        import copy
        import queue
        import threading
        n_prod = self.producers
        queues = [queue.Queue(maxsize=max(1, self.prefetch // n_prod)) for _ in range(n_prod)]
        stop = threading.Event()
        starts = list(range(self.lo, self.hi, self.batch_size))
        by_line = isinstance(self.texts, JsonlTexts) and hasattr(self.collate, "call_lines")

        def put_last(q, item):
            """The end marker / the exception: like a batch, never blocks past a consumer that has gone away."""
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return
                except queue.Full:
                    pass

        def produce(p, collate):
            try:
                os.nice(10)          # Linux: per thread; the tokenizer's pool is started from here and inherits it
            except OSError:
                pass
            q = queues[p]
            try:
                for b0 in starts[p::n_prod]:
                    if stop.is_set():
                        return
                    b1 = min(b0 + self.batch_size, self.hi)
                    if by_line:          # JsonlTexts: the records go to the collate as they are
                        item = collate.call_lines(self.texts.lines[b0:b1], self.texts.key)
                    else:
                        item = collate([self.texts[i] for i in range(b0, b1)])
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            pass
                put_last(q, None)
            except BaseException as e:      # handed to the consumer
                put_last(q, e)

        for p in range(n_prod):
            # (every producer its own collate object: TokenizeCollate keeps a private tokenizer handle)
            collate = self.collate if p == 0 else copy.copy(self.collate)
            threading.Thread(target=produce, args=(p, collate), name=f"proqa-tokenize-{p}", daemon=True).start()
        try:
            for b in range(len(starts)):
                item = queues[b % n_prod].get()
                if isinstance(item, BaseException):
                    raise item
                if item is None:            # cannot happen before the last batch of that producer
                    raise RuntimeError("tokenizer thread ended early")
                yield item
        finally:
            stop.set()
