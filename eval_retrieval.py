"""`python eval_retrieval.py ...` — same command line as the reference's retrieval/eval_retrieval.py."""
from proqa_amd.eval_retrieval import main

if __name__ == "__main__":
    main()
