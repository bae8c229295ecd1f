"""Per-question retrieval of the reference's OnlineSampler, on the exact GPU index.

/root/reference/qa/online_sampler.py builds `faiss.IndexIVFFlat(quantizer, 128, 100)` with
`nprobe = 20` over `para_embed` (:75-79) and, for one question at a time, encodes the question
with the query tower, searches k = 5000 (training, :113) or k = eval_k (:274), maps rows to
paragraph ids through `index2paraid` and gathers `para_embed[I]` (:116-117, :277).  Its own
commented-out alternative is the exact `IndexFlatIP` (:80-82).

`OnlineRetriever` is that retrieval step on `proqa_amd.index.IndexFlatIP`: exact instead of
approximate (a superset in quality of the IVF probe; over 18M rows 0.87 ms per question at k <= 80, 1.1 ms
at k = 5000), same outputs.  The sampler's span matching / batching is training code and is not rebuilt here.
No CPU path.
"""
import numpy as np
import torch

from .index import IndexFlatIP


class OnlineRetriever:
    def __init__(self, para_embed, index2paraid=None, device=None, index=None):
        """para_embed: [N,128] float16/float32 array (the np.load'ed index, qa/train_retrieve_qa.py:115);
        index2paraid: the idx_id.json mapping {"<row>": paragraph id} (the reference's format), a row-ordered
        sequence of paragraph ids (ten times cheaper per lookup: at k = 5000 the dict route costs ~3 ms per question,
        more than the encode and the search together), or None.
        index: an IndexFlatIP that already holds the rows in HBM (then para_embed may be None or just a dtype: the
        rows the sampler gathers come from the index's copy either way)."""
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if index is not None:
            self.index = index
            if para_embed is None or isinstance(para_embed, (type, np.dtype, str)):
                self.dtype = np.dtype(para_embed or np.float16)      # just the dtype of the rows to hand back
                para_embed = None
            else:
                self.dtype = np.asarray(para_embed).dtype
        else:
            para_embed = np.ascontiguousarray(para_embed)
            with torch.cuda.device(self.device):      # proqa_index_create binds the index to the current device
                self.index = IndexFlatIP(128, capacity=para_embed.shape[0])
                step = 1 << 21
                for r0 in range(0, para_embed.shape[0], step):
                    self.index.add(para_embed[r0:r0 + step])
            self.dtype = para_embed.dtype
        self.para_embed = para_embed
        self.index2paraid = index2paraid

    @torch.no_grad()
    def embed_question(self, retriever, tokenizer, question, max_query_length):
        """q_embed of online_sampler.py:106-111 with proqa_amd's BertForRetriever (query tower)."""
        ids = torch.tensor([tokenizer.encode(question, max_length=max_query_length, truncation=True)],
                           dtype=torch.int64, device=self.device)
        mask = torch.ones_like(ids, dtype=torch.bool)
        return retriever.get_embed({"input_ids": ids, "input_mask": mask}, True)["embed"]

    def retrieve(self, q_embed, k=5000):
        """q_embed [1,128] (numpy or CUDA tensor) -> (para_embed_idx int64 [k], para ids or None, para_embeds [k,128]),
        the three values the sampler derives from `self.index.search(q_embed, k)`.  The rows are gathered from the
        index's copy in HBM (reconstruct_batch_device; 5000 random 256-byte rows of a multi-GB host array cost ~1 ms)
        and come back in para_embed's dtype with para_embed's values."""
        if isinstance(q_embed, torch.Tensor):
            q = q_embed.reshape(1, -1).to(self.device)
        else:
            q = torch.from_numpy(np.ascontiguousarray(np.asarray(q_embed).reshape(1, -1))).to(self.device)
        _, I = self.index.search_device(q, k)
        I = I.reshape(-1)
        want = torch.float16 if self.dtype == np.float16 else torch.float32
        rows = self.index.reconstruct_batch_device(I, want)
        para_embed_idx = I.cpu().numpy()
        live = para_embed_idx >= 0                                   # fewer than k rows in the index
        para_embed_idx = para_embed_idx[live]
        para_embeds = rows.cpu().numpy()[live].astype(self.dtype, copy=False)
        para_idx = None
        if self.index2paraid is not None:
            # (tolist() first: str() of a Python int is four times cheaper than of a numpy scalar, and at k = 5000 this
            # loop is otherwise the slowest part of the call)
            ids = para_embed_idx.tolist()
            if isinstance(self.index2paraid, dict):
                para_idx = list(map(self.index2paraid.__getitem__, map(str, ids)))
            else:
                para_idx = list(map(self.index2paraid.__getitem__, ids))
        return para_embed_idx, para_idx, para_embeds
