"""Flag surface of get_embed.py: every flag of the reference's config.py is accepted."""
import pytest

from proqa_amd import config, get_embed


def test_reference_invocations_parse():
    # README.md:29-37 (queries) and get_para_embed.sh:2-11 (passages)
    a = config.get_args(["--do_predict", "--predict_batch_size", "512", "--bert_model_name", "bert-base-uncased",
                         "--fp16", "--predict_file", "../data/WebQuestions-test.txt", "--init_checkpoint", "x.pt",
                         "--is_query_embed", "--embed_save_path", "../data/wq_test_query_embed.npy"])
    assert a.do_predict and a.fp16 and a.is_query_embed and a.predict_batch_size == 512
    a = config.get_args(["--do_predict", "--prefix", "eval-para", "--predict_batch_size", "300", "--bert_model_name",
                         "bert-base-uncased", "--fp16", "--predict_file", "../data/wiki_splits.txt",
                         "--init_checkpoint", "ck.pt", "--embed_save_path", "encodings/para_embed.npy",
                         "--eval-workers", "32"])
    assert a.eval_workers == 32 and a.prefix == "eval-para" and not a.is_query_embed


def test_defaults_match_reference_config():
    a = config.get_args([])
    assert a.bert_model_name == "bert-large-cased-whole-word-masking"
    assert (a.max_seq_length, a.max_query_length, a.predict_batch_size, a.eval_workers) == (512, 30, 100, 32)
    assert (a.seed, a.local_rank, a.fp16_opt_level, a.prefix) == (3, -1, "O1", "eval")
    assert a.eval_period == 2500 and a.accumulate_gradients == 1 and a.num_train_epochs == 5000
    # training-only flags are accepted (and ignored by the encode path)
    config.get_args(["--do_train", "--train_file", "t", "--learning_rate", "1e-5", "--max-pool", "--shared-norm",
                     "--joint-train", "--use-whole-model", "--retriever-path", "p", "--qa-drop", "0.1",
                     "--filter", "--debug", "--verbose", "--load", "--efficient_eval", "--no_cuda"])


def test_main_argument_errors_need_no_gpu():
    with pytest.raises(ValueError, match="At least one of"):
        get_embed.main([])
    with pytest.raises(ValueError, match="predict_file"):
        get_embed.main(["--do_predict"])
    with pytest.raises(ValueError, match="accumulate_gradients"):
        get_embed.main(["--do_predict", "--accumulate_gradients", "0"])
    with pytest.raises(SystemExit):
        get_embed.main(["--bogus"])


def test_product_never_touches_the_oracle_or_the_reference():
    """oracle/ is test infrastructure: nothing under proqa_amd/ (or the root CLI shims) may import it, and
    nothing shipped may read /root/reference at run time."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, f) for f in ("get_embed.py", "eval_retrieval.py", "group_paras.py")]
    for d, _, names in os.walk(os.path.join(root, "proqa_amd")):
        files += [os.path.join(d, n) for n in names if n.endswith((".py", ".cpp", ".hip", ".h"))]
    for path in files:
        text = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), path
        code = "\n".join(line.split("#")[0] for line in text.splitlines()) if path.endswith(".py") else ""
        assert "open('/root/reference" not in code and 'open("/root/reference' not in code, path
        assert "sys.path.insert(0, '/root/reference" not in code and 'sys.path.insert(0, "/root/reference' not in code, path
    for path in (os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")):
        assert "/root/reference" not in open(path).read(), path
