#!/usr/bin/env python
"""Entry point with the reference's name: ``python calculate_test_accuracy_both.py --late_fusion=MM_RCA ... --model_path=...``."""
from garbage_classification_rca_amd.calculate_test_accuracy_both import main

if __name__ == "__main__":
    main()
