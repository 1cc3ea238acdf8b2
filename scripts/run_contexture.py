"""Alias of scripts.run_texture under the reference's actual file name (scripts/run_contexture.py)."""
from scripts.run_texture import main

if __name__ == '__main__':
    main()
