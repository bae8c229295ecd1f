"""Word tokenizer used for answer string-matching in the recall evaluation.

Restates the behaviour of SimpleTokenizer / Tokens in
/root/reference/retrieval/basic_tokenizer.py:13-50,233-272 (DrQA-derived): tokens are maximal
runs of letters/digits/marks, or any single character that is neither a separator nor a control
character; matching is done on lower-cased token text.
"""
import regex

_TOKEN = regex.compile(r"([\p{L}\p{N}\p{M}]+)|([^\p{Z}\p{C}])",
                       flags=regex.IGNORECASE | regex.UNICODE | regex.MULTILINE)


class Tokens:
    """A tokenised text: a list of (token, token_with_trailing_whitespace, (start, end))."""

    TEXT, TEXT_WS, SPAN = 0, 1, 2

    def __init__(self, data, annotators=None, opts=None):
        self.data = data
        self.annotators = annotators or set()
        self.opts = opts or {}

    def __len__(self):
        return len(self.data)

    def slice(self, i=None, j=None):
        return Tokens(self.data[i:j], self.annotators, self.opts)

    def untokenize(self):
        return "".join(t[self.TEXT_WS] for t in self.data).strip()

    def words(self, uncased=False):
        if uncased:
            return [t[self.TEXT].lower() for t in self.data]
        return [t[self.TEXT] for t in self.data]

    def offsets(self):
        return [t[self.SPAN] for t in self.data]


class SimpleTokenizer:
    def __init__(self, **kwargs):
        self.annotators = set()

    def tokenize(self, text):
        spans = [m.span() for m in _TOKEN.finditer(text)]
        data = []
        for i, (start, end) in enumerate(spans):
            ws_end = spans[i + 1][0] if i + 1 < len(spans) else end
            data.append((text[start:end], text[start:ws_end], (start, end)))
        return Tokens(data, self.annotators)

    def shutdown(self):
        pass
