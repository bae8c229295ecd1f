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


class GraphedQuestionEncoder:
    """The query tower over ONE question as a captured HIP graph per question length.

    A 16-token question is ~90 dependent launches of microsecond kernels (0.54 ms).  Opt-in, because it measured NO gain on
    MI355X / ROCm 7.2: the replay takes 0.537 ms -- the time is the GPU's dispatch of dependent kernels, not the host's
    launch calls (ABLATIONS R5.9); kept for hosts whose launch path is slower than this one's.  The first question of a
    length is encoded twice the plain way (which also sizes the encoder's workspace), then captured (`torch.cuda.CUDAGraph` around `BertForRetriever.get_embed`; the library's
    forward allocates nothing and never waits for the host once its workspace fits); later questions of that length
    copy their ids into the graph's input and replay it: one launch call.  Same kernels, same bits as the plain call.
    The graphs hold the address of the encoder's workspace: a larger batch through the same tower replaces it
    (proqa_encoder_workspace), which is checked before every replay -- the graphs are then dropped and captured again."""

    def __init__(self, retriever, max_lengths=64):
        self.retriever = retriever
        self.device = retriever.device
        self.max_lengths = max_lengths
        self._graphs = {}          # question length -> (graph, ids tensor, mask tensor, output tensor)
        self._workspace = None
        self._stream = torch.cuda.Stream(device=self.device)
        self.captures = 0

    def _capture(self, length):
        ids = torch.zeros((1, length), dtype=torch.int64, device=self.device)
        mask = torch.ones((1, length), dtype=torch.bool, device=self.device)
        batch = {"input_ids": ids, "input_mask": mask}
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._stream):
            for _ in range(2):
                self.retriever.get_embed(batch, True, check_mask=False, seq_lens_host=[length])
        self._stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=self._stream):
            out = self.retriever.get_embed(batch, True, check_mask=False, seq_lens_host=[length])["embed"]
        self.captures += 1
        return graph, ids, mask, out

    @torch.no_grad()
    def __call__(self, token_ids):
        """token_ids: the ids of one question (list or 1-D / [1, L] tensor) -> [1, 128] embedding (a fresh tensor)."""
        with torch.cuda.device(self.device):
            ids = torch.as_tensor(token_ids, dtype=torch.int64).reshape(1, -1)
            length = ids.shape[1]
            workspace = self.retriever.workspace(True)
            if workspace != self._workspace:
                self._graphs.clear()
            entry = self._graphs.get(length)
            if entry is None:
                if len(self._graphs) >= self.max_lengths:
                    self._graphs.pop(next(iter(self._graphs)))
                entry = self._graphs[length] = self._capture(length)
                self._workspace = self.retriever.workspace(True)
            graph, ids_in, _, out = entry
            ids_in.copy_(ids, non_blocking=False)
            graph.replay()
            return out.clone()


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
        self._graphed = {}

    @torch.no_grad()
    def embed_question(self, retriever, tokenizer, question, max_query_length, graphed=False):
        """q_embed of online_sampler.py:106-111 with proqa_amd's BertForRetriever (query tower).  graphed: replay the
        forward as one captured HIP graph per question length (GraphedQuestionEncoder; the same kernels and bits, and on this
        platform the same time)."""
        token_ids = tokenizer.encode(question, max_length=max_query_length, truncation=True)
        if graphed:
            enc = self._graphed.get(id(retriever))
            if enc is None:
                enc = self._graphed[id(retriever)] = GraphedQuestionEncoder(retriever)
            return enc(token_ids)
        ids = torch.tensor([token_ids], dtype=torch.int64, device=self.device)
        mask = torch.ones_like(ids, dtype=torch.bool)
        return retriever.get_embed({"input_ids": ids, "input_mask": mask}, True, check_mask=False,
                                   seq_lens_host=[len(token_ids)])["embed"]

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
