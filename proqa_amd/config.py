"""Command-line flags of the retrieval scripts.

Drop-in for /root/reference/retrieval/config.py:4-93: every flag the reference declares is
accepted with the same spelling (mixed '-' / '_'), type and default, so existing shell scripts
(get_para_embed.sh, README.md:29-37) run unchanged.  Training-only flags are parsed and ignored
by the encode path.  Extra flags of this implementation are listed in EXTRA_FLAGS.
"""
import argparse

# (flag, kind, default) — kind is a type for valued flags or "flag" for store_true switches
_REFERENCE_FLAGS = (
    ("--bert_model_name", str, "bert-large-cased-whole-word-masking"),
    ("--output_dir", str, "logs"),
    ("--weight_decay", float, 0.0),
    ("--load", "flag", False),
    ("--num_workers", int, 5),
    ("--train_file", str, ""),
    ("--predict_file", str, ""),
    ("--init_checkpoint", str, ""),
    ("--max_seq_length", int, 512),
    ("--max_query_length", int, 30),
    ("--do_train", "flag", False),
    ("--do_predict", "flag", False),
    ("--train_batch_size", int, 8),
    ("--predict_batch_size", int, 100),
    ("--learning_rate", float, 5e-5),
    ("--adam_epsilon", float, 1e-8),
    ("--num_train_epochs", float, 5000),
    ("--wait_step", int, 100),
    ("--save_checkpoints_steps", int, 20000),
    ("--iterations_per_loop", int, 1000),
    ("--no_cuda", "flag", False),
    ("--local_rank", int, -1),
    ("--accumulate_gradients", int, 1),
    ("--seed", int, 3),
    ("--gradient_accumulation_steps", int, 1),
    ("--eval-period", int, 2500),
    ("--verbose", "flag", False),
    ("--efficient_eval", "flag", False),
    ("--max_grad_norm", float, 5.0),
    ("--fp16", "flag", False),
    ("--fp16_opt_level", str, "O1"),
    ("--filter", "flag", False),
    ("--prefix", str, "eval"),
    ("--debug", "flag", False),
    ("--eval-workers", int, 32),
    ("--use-whole-model", "flag", False),
    ("--joint-train", "flag", False),
    ("--max-pool", "flag", False),
    ("--shared-norm", "flag", False),
    ("--retriever-path", str, ""),
    ("--qa-drop", float, 0),
    ("--embed_save_path", str, ""),
    ("--is_query_embed", "flag", False),
)

EXTRA_FLAGS = (
    # fp32 .npy output (the reference writes whatever dtype the model emitted: fp16 under --fp16)
    ("--embed_dtype", str, "auto"),
)


def build_parser():
    parser = argparse.ArgumentParser()
    for flag, kind, default in _REFERENCE_FLAGS + EXTRA_FLAGS:
        if kind == "flag":
            parser.add_argument(flag, action="store_true", default=default)
        else:
            parser.add_argument(flag, type=kind, default=default)
    return parser


def get_args(argv=None):
    return build_parser().parse_args(argv)
