"""EmDataset / em_collate / collate_tokens against the reference's outputs (tokenize_golden.json)."""
import json
import os

import numpy as np
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


def _random_strings(n, seed):
    import random
    rng = random.Random(seed)
    vocab = [w.strip() for w in open(os.path.join(GOLDEN, "vocab_small.txt"))]
    words = [w for w in vocab if w.isalpha()]
    odd = [0x20, 0x9, 0xa, 0xd, 0xa0, 0x2003, 0x200b, 0x3000, 0xad, 0x0, 0xfffd, 0x7f, 0x85]

    def one():
        out = []
        for _ in range(rng.randrange(0, 70)):
            r = rng.random()
            if r < 0.5:
                out.append(rng.choice(words))
            elif r < 0.6:
                out.append(rng.choice(words).upper())
            elif r < 0.7:
                out.append(chr(rng.choice(odd)))
            elif r < 0.8:
                out.append(chr(rng.randrange(0x20, 0x7f)))
            elif r < 0.9:
                out.append(chr(rng.randrange(0xa0, 0x3000)))
            elif r < 0.95:
                out.append(chr(rng.randrange(0x4e00, 0x9fff)))
            else:
                out.append("".join(chr(rng.randrange(0x300, 0x36f)) for _ in range(2)))
        return " ".join(out) if rng.random() < 0.7 else "".join(out)

    return [one() for _ in range(n)] + ["", " ", "a" * 300, "é" * 50, "İstanbul ǅ ß ﬁ", "[SEP] x [CLS]"]


def test_batch_tokenising_collate_matches_reference_batches(tokenizer, tmp_path):
    """get_embed.py's loader (EmTextView + TokenizeCollate: one tokenizer call per batch inside the workers) produces the
    reference's batches: the golden ones, bit for bit, plus the valid lengths as a host list."""
    with open(os.path.join(GOLDEN, "tokenize_golden.json")) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        key = "question" if case["is_query"] else "text"
        path = tmp_path / "in.jsonl"
        path.write_text("".join(json.dumps({key: t, "id": 0}) + "\n" for t in gold["texts"]))
        ds = datasets.EmDataset(tokenizer, str(path), case["max_query_length"], case["max_length"], case["is_query"])
        view = datasets.EmTextView(ds)
        assert len(view) == len(gold["texts"]) and view[3] == gold["texts"][3]
        batch = datasets.TokenizeCollate(tokenizer, ds.max_length)([view[i] for i in range(len(view))])
        assert batch["input_ids"].dtype == torch.int64 and batch["input_mask"].dtype == torch.bool
        assert batch["input_ids"].tolist() == case["input_ids"]
        assert batch["input_mask"].int().tolist() == case["input_mask"]
        assert batch["seq_lens"] == case["item_lengths"]
    assert datasets.TokenizeCollate(tokenizer, 30)([]) == {}


def test_batch_tokenising_collate_equals_the_per_item_path_on_random_strings(tokenizer, tmp_path):
    """5000 random strings (words of the vocabulary, upper case, ASCII and non-ASCII symbols, CJK, combining marks,
    control and odd space characters, literal special tokens): per batch of 64, ids / masks identical to EmDataset +
    em_collate, for the query and the passage length limits; also through DataLoader workers (the collate is pickled)."""
    from torch.utils.data import DataLoader
    texts = _random_strings(5000, 17)
    path = tmp_path / "rand.jsonl"
    path.write_text("".join(json.dumps({"text": t, "question": t}) + "\n" for t in texts))
    for is_query, limit in ((True, 30), (False, 512), (False, 16)):
        ds = datasets.EmDataset(tokenizer, str(path), limit if is_query else 30, limit, is_query)
        view = datasets.EmTextView(ds)
        collate = datasets.TokenizeCollate(tokenizer, ds.max_length)
        for b0 in range(0, len(ds), 64):
            idx = range(b0, min(b0 + 64, len(ds)))
            want = datasets.em_collate([ds[i] for i in idx])
            got = collate([view[i] for i in idx])
            assert torch.equal(got["input_ids"], want["input_ids"]), (is_query, limit, b0)
            assert torch.equal(got["input_mask"], want["input_mask"]), (is_query, limit, b0)
            assert got["seq_lens"] == want["input_mask"].sum(1).tolist()
    loader = DataLoader(view, batch_size=100, collate_fn=collate, num_workers=2)
    n = 0
    for b, batch in enumerate(loader):
        want = datasets.em_collate([ds[i] for i in range(b * 100, min(b * 100 + 100, len(ds)))])
        assert torch.equal(batch["input_ids"], want["input_ids"])
        n += batch["input_ids"].shape[0]
        if b == 5:
            break
    assert n == 600


def test_text_batch_loader_yields_the_batches_in_order(tokenizer, tmp_path):
    """get_embed.py's loader: a producer thread of the same process, batches in file order, a row range for sharded runs,
    producer exceptions re-raised in the consumer."""
    texts = _random_strings(700, 3)
    path = tmp_path / "rand.jsonl"
    path.write_text("".join(json.dumps({"text": t}) + "\n" for t in texts))
    ds = datasets.EmDataset(tokenizer, str(path), 30, 64, False)
    view = datasets.EmTextView(ds)
    collate = datasets.TokenizeCollate(tokenizer, ds.max_length, parallel=True)
    loader = datasets.TextBatchLoader(view, 100, collate, prefetch=2, lo=50, hi=len(view))
    assert len(loader) == (len(view) - 50 + 99) // 100
    n = 50
    for batch in loader:
        m = batch["input_ids"].shape[0]
        want = datasets.em_collate([ds[i] for i in range(n, n + m)])
        assert torch.equal(batch["input_ids"], want["input_ids"]) and torch.equal(batch["input_mask"], want["input_mask"])
        n += m
    assert n == len(view)
    assert list(datasets.TextBatchLoader(view, 100, collate, lo=10, hi=10)) == []

    def broken(_texts):
        raise RuntimeError("tokenizer blew up")

    with pytest.raises(RuntimeError, match="blew up"):
        list(datasets.TextBatchLoader(view, 100, broken))
    # a consumer that stops early does not leave the producer blocked on a full queue
    it = iter(datasets.TextBatchLoader(view, 10, collate, prefetch=1))
    next(it)
    it.close()


def _random_ascii(n, seed):
    import random
    rng = random.Random(seed)
    vocab = [w.strip() for w in open(os.path.join(GOLDEN, "vocab_small.txt"))]
    words = [w for w in vocab if w.isalpha()]

    def one():
        out = []
        for _ in range(rng.randrange(0, 50)):
            r = rng.random()
            if r < 0.45:
                out.append(rng.choice(words))
            elif r < 0.55:
                out.append(rng.choice(words).upper())
            elif r < 0.65:
                out.append(rng.choice(words) + rng.choice(words))                    # splits into word pieces
            elif r < 0.8:
                out.append(chr(rng.randrange(0, 128)))                               # every ASCII code, controls included
            elif r < 0.9:
                out.append("".join(chr(rng.randrange(33, 127)) for _ in range(rng.randrange(1, 6))))
            else:
                out.append("x" * rng.randrange(95, 110))                             # around the 100-character word limit
        return (" " if rng.random() < 0.7 else "").join(out)

    return [one() for _ in range(n)] + ["", " ", "\t\n", "a\x00b", "a\x0bb c\x1fd", "it's", "U.S.A.", "x" * 100, "x" * 101]


@pytest.mark.parametrize("lower", [True, False])
def test_native_wordpiece_equals_the_reference_tokenizer(tmp_path, lower):
    """libproqa_hip.so's WordPiece (proqa_wordpiece_*, the loader's fast path for plain-ASCII sentences) against
    transformers' BertTokenizer: ids, masks and lengths identical on 6000 random ASCII strings (all 128 codes, word-piece
    splits, words around the 100-character limit, truncation at three limits), on sentences it must decline (non-ASCII,
    a literal special token: those rows come from the tokenizer itself) and on the reference's golden batches; uncased and
    cased models."""
    import shutil
    from transformers import BertTokenizer
    d = tmp_path / "model"
    d.mkdir()
    shutil.copy(os.path.join(GOLDEN, "vocab_small.txt"), d / "vocab.txt")
    tok = BertTokenizer.from_pretrained(str(d), do_lower_case=lower)
    texts = _random_ascii(6000, 23 + lower) + _random_strings(600, 5) + ["[CLS] x [SEP]", "café au lait", "[unused1]"]
    for limit in (16, 64, 512):
        ref = datasets.TokenizeCollate(tok, limit)
        nat = datasets.TokenizeCollate(tok, limit, native_threads=3)
        assert nat._native_spec is not None and nat._native_spec[1] == lower
        for b0 in range(0, len(texts), 97):
            a, b = ref(texts[b0:b0 + 97]), nat(texts[b0:b0 + 97])
            assert torch.equal(a["input_ids"], b["input_ids"]), (limit, b0)
            assert torch.equal(a["input_mask"], b["input_mask"]) and a["seq_lens"] == b["seq_lens"]
    if lower:
        with open(os.path.join(GOLDEN, "tokenize_golden.json")) as f:
            gold = json.load(f)
        for case in gold["cases"]:
            L = case["max_query_length"] if case["is_query"] else case["max_length"]
            batch = datasets.TokenizeCollate(tok, L, native_threads=2)(gold["texts"])
            assert batch["input_ids"].tolist() == case["input_ids"] and batch["seq_lens"] == case["item_lengths"]


def _unicode_vocab_and_texts(n_texts, seed):
    """A synthetic vocabulary with tokens of many scripts (whole words and ## pieces of the NORMALISED forms the
    tokenizer produces) and random strings over Latin-1 / Latin Extended / Greek / Cyrillic / Hebrew / Arabic / Devanagari /
    Thai / Hangul / kana / CJK / punctuation / symbol / combining-mark / white-space / control blocks."""
    import random
    import unicodedata
    rng = random.Random(seed)
    blocks = [(0x00A0, 0x024F), (0x0370, 0x03FF), (0x0400, 0x04FF), (0x0590, 0x05FF), (0x0600, 0x06FF), (0x0900, 0x097F),
              (0x0E00, 0x0E7F), (0x1100, 0x11FF), (0x1E00, 0x1FFF), (0x2000, 0x206F), (0x2070, 0x21FF), (0x2200, 0x22FF),
              (0x3000, 0x30FF), (0x3400, 0x3500), (0x4E00, 0x4F00), (0xAC00, 0xAD00), (0xF900, 0xFA6A), (0xFE30, 0xFE6F),
              (0xFF00, 0xFFEF), (0x0300, 0x036F), (0x0080, 0x009F), (0xE000, 0xE010), (0xFFF0, 0xFFFF)]
    pool = [chr(c) for a, b in blocks for c in range(a, b + 1) if not 0xD800 <= c <= 0xDFFF]
    ascii_words = ["the", "of", "and", "cafe", "naive", "resume", "uber", "strasse", "hello", "world", "tokyo", "a", "i", "x"]

    def one():
        out = []
        for _ in range(rng.randrange(0, 30)):
            r = rng.random()
            if r < 0.25:
                out.append(rng.choice(ascii_words))
            elif r < 0.35:
                out.append(rng.choice(ascii_words).upper())
            elif r < 0.75:
                out.append("".join(rng.choice(pool) for _ in range(rng.randrange(1, 6))))
            elif r < 0.85:
                out.append(rng.choice(ascii_words) + rng.choice(pool) + rng.choice(ascii_words))
            elif r < 0.9:
                out.append("".join(chr(rng.randrange(0, 128)) for _ in range(3)).replace("[", "("))
            else:
                out.append(unicodedata.normalize("NFC", rng.choice("aeiounc") + rng.choice("\u0301\u0308\u0303\u0327\u030c")))
        return (" " if rng.random() < 0.7 else "").join(out)

    texts = [one() for _ in range(n_texts)] + ["Ελληνικά ΚΑΙ Σ", "İstanbul ǅ ß ẞ", "한국어 텍스트", "日本語のテキスト。", "a\u00adb\u200bc\ufeffd",
                                               "x\u2028y\u3000z", "≠ ≤ ≥ ∑", "ﬁne ﬂow", "Ⅷ ½ ²", "e\u0301\u0327 o\u0327\u0301", "ॐ नमः", "ไทย", "\ufffd\x00"]
    return pool, ascii_words, texts


@pytest.mark.parametrize("lower", [True, False])
def test_native_wordpiece_handles_non_ascii_text_itself(tmp_path, lower):
    """Integer parity of the table-driven Unicode path of proqa_wordpiece_encode_batch: 20 000 random strings over
    two dozen blocks of the Basic Multilingual Plane, ids / lengths identical to transformers' BertTokenizer at two
    truncation limits, uncased and cased -- and the native code must tokenise them ITSELF (only a '[', a character beyond
    the BMP or one of the 15 reordering-sensitive marks may be handed back)."""
    import ctypes
    from tokenizers import normalizers
    from transformers import BertTokenizer
    from proqa_amd import _lib
    pool, ascii_words, texts = _unicode_vocab_and_texts(20000, 41 + lower)
    # vocabulary: specials, ASCII characters, every NORMALISED character of a sample of the pool (whole and ##), words
    norm = normalizers.BertNormalizer(lowercase=lower)
    import random
    rng = random.Random(3)
    chars = set()
    for c in rng.sample(pool, len(pool) // 2):
        chars.update(ch for ch in norm.normalize_str(c) if not ch.isspace())
    chars.update(chr(c) for c in range(33, 127))
    chars.discard("[")
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + sorted(chars) + ["##" + c for c in sorted(chars) if c.isalnum() or ord(c) > 0x2FF]
    vocab += ascii_words + [w.upper() for w in ascii_words if not lower] + ["##" + w for w in ascii_words]
    pairs = sorted({a + b for a in rng.sample(sorted(chars), 200) for b in rng.sample(sorted(chars), 5)})
    vocab += [p_ for p_ in pairs if p_ not in chars] + ["##" + p_ for p_ in pairs[::3]]
    vocab = list(dict.fromkeys(vocab))
    d = tmp_path / "model"
    d.mkdir()
    (d / "vocab.txt").write_text("\n".join(vocab) + "\n", encoding="utf-8")
    tok = BertTokenizer.from_pretrained(str(d), do_lower_case=lower)
    spec = datasets.TokenizeCollate._native_vocab(tok)
    assert spec is not None and spec[1] == lower
    lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(lib.proqa_wordpiece_create(spec[0], len(spec[0]), 1 if lower else 0, ctypes.byref(h)))
    try:
        raw = [t.encode("utf-8") for t in texts]
        n = len(raw)
        ptrs = (ctypes.c_char_p * n)(*raw)
        sizes = np.fromiter(map(len, raw), dtype=np.int64, count=n)
        for L in (24, 512):
            ids = np.empty((n, L), dtype=np.int64)
            lens = np.empty(n, dtype=np.int32)
            _lib.check(lib.proqa_wordpiece_encode_batch(h, ptrs, sizes.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n, L,
                                                        ids.ctypes.data, lens.ctypes.data, 4))
            declined = int((lens < 0).sum())
            assert declined <= n // 100, declined            # (the pool holds two of the reordering-sensitive marks: ~0.6 % of the texts)
            unk = vocab.index("[UNK]")
            known = 0
            for i, text in enumerate(texts):
                if lens[i] < 0:
                    continue
                want = tok.encode(text, max_length=L, truncation=True)
                assert ids[i, :lens[i]].tolist() == want and not ids[i, lens[i]:].any(), (L, i, text)
                known += sum(1 for v in want if v != unk)
            assert known > 5 * n                             # the comparison is not [UNK] == [UNK]
    finally:
        lib.proqa_wordpiece_free(h)
    # texts the native code hands back
    for text in ("a [SEP] b", "emoji \U0001F600", "tone \u302e mark"):
        b = text.encode("utf-8")
        one = np.empty((1, 16), dtype=np.int64)
        ln = np.empty(1, dtype=np.int32)
        h2 = ctypes.c_void_p()
        _lib.check(lib.proqa_wordpiece_create(spec[0], len(spec[0]), 1 if lower else 0, ctypes.byref(h2)))
        _lib.check(lib.proqa_wordpiece_encode_batch(h2, (ctypes.c_char_p * 1)(b), (ctypes.c_int64 * 1)(len(b)), 1, 16,
                                                    one.ctypes.data, ln.ctypes.data, 1))
        lib.proqa_wordpiece_free(h2)
        assert ln[0] == -1, text


def test_native_jsonl_records_equal_json_loads(tokenizer, tmp_path):
    """proqa_wordpiece_encode_jsonl_batch reads sample[key] out of the JSON-lines record itself: on 5000 random records
    (key order, escapes of every kind, \\uXXXX incl. surrogate pairs, numbers, literals, white space, repeated keys) and on
    malformed / unusual ones the batch equals the one made from json.loads(line)[key]; whatever the native parser does
    not take is parsed by Python, and a line Python rejects still raises."""
    import random
    rng = random.Random(17)
    texts = _random_strings(2500, 31) + _random_ascii(2500, 32)
    lines = []
    for i, t in enumerate(texts):
        rec = {"id": rng.choice([i, f"doc{i}", -i, 1.5e3, None, True]), "text": t, "title": rng.choice(["T", "q\"uote", "tab\there", ""])}
        if rng.random() < 0.3:
            rec = dict(reversed(list(rec.items())))
        line = json.dumps(rec, ensure_ascii=rng.random() < 0.5, separators=rng.choice([(",", ":"), (", ", ": "), (" ,\t", " : ")]))
        if rng.random() < 0.1:
            line = "  " + line + " \r"
        lines.append((line + "\n").encode("utf-8"))
    special = [b'{"text": "first", "text": "second value"}\n',                      # a repeated key: the last one counts
               b'{"text": "nested next", "meta": {"a": [1, 2]}}\n',                 # nested container: Python's business
               b'{"meta": ["x"], "text": "list first"}\n',
               b'{"text": "pair \\ud83d\\ude00 end"}\n',                              # surrogate escape pair
               b'{"text": "sol\\/idus \\b\\f"}\n', b'{"text":"","id":0}\n', b'{"id": -0.5e-3, "text": "num"}\n',
               '{"text": "caf\u00e9 na\u00efve"}\n'.encode("utf-8")]
    lines += special
    key = "text"
    for limit in (12, 64):
        ref = datasets.TokenizeCollate(tokenizer, limit)
        nat = datasets.TokenizeCollate(tokenizer, limit, native_threads=3)
        for b0 in range(0, len(lines), 211):
            chunk = lines[b0:b0 + 211]
            a = ref([json.loads(ln.strip())[key] for ln in chunk])
            b = nat.call_lines(chunk, key)
            assert torch.equal(a["input_ids"], b["input_ids"]) and a["seq_lens"] == b["seq_lens"], (limit, b0)
    # what the native parser takes and what it leaves (lens -2) -- and it must take the plain records
    import ctypes
    from proqa_amd import _lib
    lib, h = datasets.TokenizeCollate(tokenizer, 32, native_threads=1)._native_handle()

    def native_len(line):
        ids = np.empty((1, 32), dtype=np.int64)
        ln = np.empty(1, dtype=np.int32)
        _lib.check(lib.proqa_wordpiece_encode_jsonl_batch(h, (ctypes.c_char_p * 1)(line), (ctypes.c_int64 * 1)(len(line)), 1, b"text", 32,
                                                          ids.ctypes.data, ln.ctypes.data, 1))
        return int(ln[0])
    assert native_len(b'{"id": 3, "text": "plain record"}\n') > 0
    assert native_len(special[0]) > 0 and native_len(special[4]) > 0 and native_len(special[6]) > 0
    for odd in (special[1], special[2], special[3], b'{"text": 5}', b'{"id": 1}', b'{"text": "x"} trailing', b'{"text": "x",}',
                b'{"text": "bad \\x escape"}', b'{"text": "ctrl \x01"}', b'{"id": 01, "text": "x"}', b'{"id": NaN, "text": "x"}',
                b'["text"]', b'', b'{"text": "unterminated'):
        assert native_len(odd) == -2, odd
    # a record Python rejects raises from the collate, as it does in EmDataset
    with pytest.raises(ValueError):
        datasets.TokenizeCollate(tokenizer, 32, native_threads=2).call_lines([b'{"text": "ok"}', b'{"text": "x",}'], "text")
    # the loader takes this path for a JsonlTexts view and yields the same batches as the per-item path
    path = tmp_path / "paras.txt"
    path.write_bytes(b"".join(lines[:700]))
    view = datasets.JsonlTexts(str(path), 30, 48, False)
    fast = list(datasets.TextBatchLoader(view, 64, datasets.TokenizeCollate(tokenizer, 48, native_threads=2)))
    slow = [datasets.TokenizeCollate(tokenizer, 48)([view[i] for i in range(b0, min(b0 + 64, 700))]) for b0 in range(0, 700, 64)]
    assert len(fast) == len(slow) and all(torch.equal(x["input_ids"], y["input_ids"]) for x, y in zip(fast, slow))


def test_native_wordpiece_is_declined_for_other_tokenizers(tokenizer):
    """A tokenizer the native code does not restate (here: a vocabulary with a gap in its ids) keeps the library path."""
    class Odd:
        unk_token, cls_token, sep_token = "[UNK]", "[CLS]", "[SEP]"

        def get_vocab(self):
            return {"[UNK]": 0, "[CLS]": 1, "[SEP]": 3}
    assert datasets.TokenizeCollate._native_vocab(Odd()) is None
    assert datasets.TokenizeCollate._native_vocab(tokenizer) is not None


def test_a_pure_python_tokenizer_gets_the_native_path_for_ascii_only(tokenizer):
    """The generated tables restate the tokenizers library's BertNormalizer.  A pure-Python BertTokenizer (what the reference's
    transformers 2.5.1 BertTokenizer is) lower-cases whole strings -- final sigma: "ΟΔΟΣ" -> "οδος", per character "οδοσ" --
    NFC-normalises first and carries another Unicode data version, so for such a tokenizer the native code takes ASCII texts
    only (bit 1 of proqa_wordpiece_create's flag) and flags everything else for the tokenizer itself."""
    import ctypes
    from proqa_amd import _lib

    class PurePython:   # no backend_tokenizer / _tokenizer: the slow-tokenizer branch of _native_vocab
        unk_token, cls_token, sep_token = "[UNK]", "[CLS]", "[SEP]"
        do_lower_case = True
        do_basic_tokenize = True
        added_tokens_encoder = {}
        basic_tokenizer = None

        def get_vocab(self):
            return {"[PAD]": 0, "[UNK]": 1, "[CLS]": 2, "[SEP]": 3, "street": 4, "οδος": 5, "οδοσ": 6}
    spec = datasets.TokenizeCollate._native_vocab(PurePython())
    assert spec is not None and spec[1] == 3                      # lower-case | ASCII only
    assert datasets.TokenizeCollate._native_vocab(tokenizer)[1] in (0, 1)   # the tokenizers-backed one: whole BMP
    lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(lib.proqa_wordpiece_create(spec[0], len(spec[0]), spec[1], ctypes.byref(h)))
    try:
        for text, want in (("Street", [2, 4, 3]), ("ΟΔΟΣ", None), ("street ΟΔΟΣ", None), ("caf\u00e9", None)):
            b = text.encode("utf-8")
            row = np.zeros((1, 8), dtype=np.int64)
            ln = np.empty(1, dtype=np.int32)
            _lib.check(lib.proqa_wordpiece_encode_batch(h, (ctypes.c_char_p * 1)(b), (ctypes.c_int64 * 1)(len(b)), 1, 8,
                                                        row.ctypes.data, ln.ctypes.data, 1))
            if want is None:
                assert ln[0] == -1, text
            else:
                assert ln[0] == len(want) and row[0, : ln[0]].tolist() == want
    finally:
        lib.proqa_wordpiece_free(h)


def test_jsonl_texts_is_the_lazy_form_of_emdataset(tokenizer, tmp_path):
    texts = _random_strings(300, 9)
    path = tmp_path / "in.jsonl"
    path.write_text("".join(json.dumps({"text": t, "question": t[::-1], "id": i}) + "\n" for i, t in enumerate(texts)))
    for is_query in (True, False):
        ds = datasets.EmDataset(tokenizer, str(path), 30, 64, is_query)
        lazy = datasets.JsonlTexts(str(path), 30, 64, is_query)
        view = datasets.EmTextView(ds)
        assert len(lazy) == len(ds) and lazy.max_length == ds.max_length
        assert all(lazy[i] == view[i] for i in range(len(ds)))
    bad = tmp_path / "bad.jsonl"
    bad.write_text('{"text": "a"}\nnot json\n')
    lazy = datasets.JsonlTexts(str(bad), 30, 64, False)
    assert len(lazy) == 2 and lazy[0] == "a"
    with pytest.raises(ValueError):
        lazy[1]
