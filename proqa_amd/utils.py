"""Small host utilities of the retrieval path.

Mirrors the pieces of /root/reference/retrieval/utils.py that the encode/search path uses:
move_to_cuda (:5-22), normalize (:63-65) and DocDB (:68-105).
"""
import sqlite3
import unicodedata


def move_to_cuda(sample, device=None):
    """Recursively move tensors in dicts/lists to the GPU (reference: unconditional .cuda())."""
    import torch   # (not at module level: the eval_retrieval.py command line needs DocDB / normalize only, and runs
    #                without PyTorch in the process -- a second of start-up and a lighter fork for its scorer pool)
    if len(sample) == 0:
        return {}
    if torch.is_tensor(sample):
        return sample.cuda(device, non_blocking=True)
    if isinstance(sample, dict):
        return {k: move_to_cuda(v, device) if _container_or_tensor(v) else v for k, v in sample.items()}
    if isinstance(sample, list):
        return [move_to_cuda(v, device) if _container_or_tensor(v) else v for v in sample]
    return sample


def _container_or_tensor(v):
    import torch
    return torch.is_tensor(v) or isinstance(v, (dict, list))


def normalize(text):
    """NFD-normalise, as the reference does before tokenising and before DB lookups."""
    return unicodedata.normalize("NFD", text)


class DocDB:
    """sqlite table documents(id, text); get_doc_text(doc_id) -> text or None."""

    def __init__(self, db_path=None):
        self.path = db_path
        self.connection = sqlite3.connect(self.path, check_same_thread=False)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        self.connection.close()

    def get_doc_ids(self):
        cur = self.connection.cursor()
        try:
            cur.execute("SELECT id FROM documents")
            return [row[0] for row in cur.fetchall()]
        finally:
            cur.close()

    def get_doc_text(self, doc_id):
        cur = self.connection.cursor()
        try:
            cur.execute("SELECT text FROM documents WHERE id = ?", (normalize(doc_id),))
            row = cur.fetchone()
            return None if row is None else row[0]
        finally:
            cur.close()
