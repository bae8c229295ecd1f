"""Build idx_id.json ({row index: document id}) from the JSON-lines passage file.

Drop-in for /root/reference/retrieval/gen_index_id_map.py:3-9 (which hard-codes its paths).
"""
import json
import sys


def build(corpus_path, out_path):
    mapping = {}
    with open(corpus_path) as f:
        for idx, line in enumerate(f):
            mapping[idx] = json.loads(line.strip())["id"]
    with open(out_path, "w") as f:
        json.dump(mapping, f)
    return len(mapping)


if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 else "../data/para_doc.db"
    dst = sys.argv[2] if len(sys.argv) > 2 else "index_data/idx_id.json"
    build(src, dst)
