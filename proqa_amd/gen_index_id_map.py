"""Build the row-index -> document-id map of the index.

Drop-in for /root/reference/retrieval/gen_index_id_map.py:3-9 (which hard-codes its paths): writes
`idx_id.json` = {"<row>": doc_id}, the file eval_retrieval.py:68-76 json.load()s.  At 18M rows
that JSON costs tens of seconds and gigabytes to parse, so a compact binary sidecar can be
written next to it: `<stem>.ids` (the JSON-encoded ids, newline
separated) + `<stem>.off.npy` (int64 byte offsets); `proqa_amd.eval_retrieval.convert_idx2id`
memory-maps it when given the `.ids` path and returns exactly what the JSON route returns.
"""
import json
import sys

import numpy as np


def build(corpus_path, out_path, sidecar=False):
    mapping = {}
    with open(corpus_path) as f:
        for idx, line in enumerate(f):
            mapping[idx] = json.loads(line.strip())["id"]
    with open(out_path, "w") as f:
        json.dump(mapping, f)
    if sidecar:
        write_sidecar([mapping[i] for i in range(len(mapping))], sidecar_stem(out_path))
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


if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 else "../data/para_doc.db"
    dst = sys.argv[2] if len(sys.argv) > 2 else "index_data/idx_id.json"
    build(src, dst, sidecar="--sidecar" in sys.argv)
