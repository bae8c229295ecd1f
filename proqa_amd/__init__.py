"""proqa_amd — MI355X-native implementation of ProQA's encode + exact top-k retrieval path.

Host-side modules mirror the reference's own (retrieval/get_embed.py, eval_retrieval.py,
retriever.py, datasets.py, ...) over the C ABI of csrc/libproqa_hip.so (include/proqa_hip.h).
"""
__all__ = ["index", "_lib"]
