"""Build the row-index -> document-id map of the index.

Drop-in for /root/reference/retrieval/gen_index_id_map.py:3-9 (which hard-codes its paths): writes
`idx_id.json` = {"<row>": doc_id}, the file eval_retrieval.py:68-76 json.load()s.  At 18M rows
that JSON costs tens of seconds and gigabytes to parse, so a compact binary sidecar can be
written next to it: `<stem>.ids` (the JSON-encoded ids, newline
separated) + `<stem>.off.npy` (int64 byte offsets); `proqa_amd.eval_retrieval.convert_idx2id`
memory-maps it when given the `.ids` path and returns exactly what the JSON route returns.

The passage TEXTS can travel the same way (`texts=True` / `--texts`): `<stem>.txt` (the UTF-8 texts of the corpus file, each
distinct text once) + `<stem>.txtoff.npy` (int64 [rows, 2]: first byte and end of the row's text -- rows with the same text
share it).  eval_retrieval.py then scores a hit from the memory-mapped text of its ROW -- no
row -> doc id -> sqlite round trip per hit (/root/reference/retrieval/eval_retrieval.py:47-52, utils.py:96-105: 162 560
single-row queries per 2032 questions x top-80).  The texts must be the ones the DB holds for those ids (both come from the same
passage file in the reference's pipeline, README.md:25-37); the sqlite route stays the default.
"""
import json
import sys

import numpy as np


def build(corpus_path, out_path, sidecar=False, texts=False):
    mapping = {}
    text_list = [] if texts else None
    with open(corpus_path) as f:
        for idx, line in enumerate(f):
            item = json.loads(line.strip())
            mapping[idx] = item["id"]
            if texts:
                text_list.append(item["text"])
    with open(out_path, "w") as f:
        json.dump(mapping, f)
    if sidecar or texts:
        write_sidecar([mapping[i] for i in range(len(mapping))], sidecar_stem(out_path))
    if texts:
        write_text_sidecar(text_list, sidecar_stem(out_path))
    return len(mapping)


def sidecar_stem(json_path):
    return json_path[:-5] if json_path.endswith(".json") else json_path


def write_sidecar(ids, stem):
    """ids in row order -> <stem>.ids + <stem>.off.npy"""
    offsets = np.zeros(len(ids) + 1, dtype=np.int64)
    with open(stem + ".ids", "wb") as f:
        pos = 0
        for i, doc_id in enumerate(ids):
            b = json.dumps(doc_id).encode("utf-8") + b"\n"
            f.write(b)
            pos += len(b)
            offsets[i + 1] = pos
    np.save(stem + ".off.npy", offsets)
    return stem + ".ids"


def write_text_sidecar(texts, stem):
    """texts in row order -> <stem>.txt (UTF-8; every distinct text once) + <stem>.txtoff.npy (int64 [rows, 2] byte spans)"""
    spans = np.zeros((len(texts), 2), dtype=np.int64)
    seen = {}
    with open(stem + ".txt", "wb") as f:
        pos = 0
        for i, text in enumerate(texts):
            span = seen.get(text)
            if span is None:
                b = text.encode("utf-8")
                f.write(b)
                span = seen[text] = (pos, pos + len(b))
                pos += len(b)
            spans[i] = span
    np.save(stem + ".txtoff.npy", spans)
    return stem + ".txt"


def text_sidecar_of(mapping_path):
    """path of the text sidecar that belongs to an --idx-id-map argument (idx_id.json or idx_id.ids), or None"""
    import os
    stem = mapping_path[:-4] if mapping_path.endswith(".ids") else sidecar_stem(mapping_path)
    return stem + ".txt" if os.path.exists(stem + ".txt") and os.path.exists(stem + ".txtoff.npy") else None


class TextSidecar:
    """row -> passage text over the memory-mapped text sidecar"""

    def __init__(self, txt_path):
        self.spans = np.load(txt_path[:-4] + ".txtoff.npy", mmap_mode="r")
        import os
        self.blob = np.memmap(txt_path, dtype=np.uint8, mode="r") if os.path.getsize(txt_path) > 0 else np.zeros(0, np.uint8)

    def __len__(self):
        return len(self.spans)

    def __getitem__(self, row):
        row = int(row)
        if row < 0 or row >= len(self):
            raise KeyError(str(row))
        lo, hi = self.spans[row]
        return bytes(self.blob[int(lo):int(hi)]).decode("utf-8")


class SidecarMap:
    """O(1) row -> doc id lookups over the memory-mapped sidecar (no parse of the whole map)."""

    def __init__(self, ids_path):
        stem = ids_path[:-4] if ids_path.endswith(".ids") else ids_path
        self.offsets = np.load(stem + ".off.npy", mmap_mode="r")
        self.blob = np.memmap(stem + ".ids", dtype=np.uint8, mode="r")

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, row):
        row = int(row)
        if row < 0 or row >= len(self):
            raise KeyError(str(row))   # the JSON route raises KeyError for unknown rows
        lo, hi = int(self.offsets[row]), int(self.offsets[row + 1])
        return json.loads(bytes(self.blob[lo:hi - 1]).decode("utf-8"))

    def take(self, rows):
        """ids of an integer array of rows (any shape) as nested lists of that shape: ONE gather of the bytes of all ids and
        ONE json parse, instead of a Python-level lookup per id (2.3 us each: 0.38 s of a 2.0 s command line at 2032 x 80)."""
        rows = np.asarray(rows)
        flat = rows.reshape(-1).astype(np.int64)
        if flat.size == 0:
            return rows.tolist()
        if flat.min() < 0 or flat.max() >= len(self):
            bad = flat[(flat < 0) | (flat >= len(self))][0]
            raise KeyError(str(int(bad)))
        lo = np.take(self.offsets, flat)
        hi = np.take(self.offsets, flat + 1)
        lens = (hi - lo).astype(np.int64)                      # each id with its trailing newline
        starts = np.cumsum(lens) - lens
        src = np.repeat(lo - starts, lens) + np.arange(int(lens.sum()), dtype=np.int64)
        buf = np.asarray(self.blob[src]) if isinstance(self.blob, np.memmap) else self.blob[src]
        buf[starts + lens - 1] = ord(",")                      # the separators (a JSON-encoded id holds no raw newline)
        ids = json.loads(b"[" + buf[:-1].tobytes() + b"]")
        out = np.empty(flat.size, dtype=object)
        out[:] = ids
        return out.reshape(rows.shape).tolist()


if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 else "../data/para_doc.db"
    dst = sys.argv[2] if len(sys.argv) > 2 else "index_data/idx_id.json"
    build(src, dst, sidecar="--sidecar" in sys.argv, texts="--texts" in sys.argv)
