"""`python group_paras.py ...` — same command line as the reference's retrieval/group_paras.py."""
from proqa_amd.group_paras import main

if __name__ == "__main__":
    main()
