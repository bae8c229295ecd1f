"""`python get_embed.py ...` — same command line as the reference's retrieval/get_embed.py."""
from proqa_amd.get_embed import main

if __name__ == "__main__":
    main()
