#!/usr/bin/env python
"""Entry point with the reference's name: ``python main_both.py --late_fusion=MM_RCA --reverse ...``."""
from garbage_classification_rca_amd.main_both import main

if __name__ == "__main__":
    main()
